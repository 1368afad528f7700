"""CPU pins of the oracle for the Gamma-real extension (oracle/gamma_real.py): its pair tables against the library's
host tables, the restricted operator against dense diagonalisation of the complex oracle Hamiltonian, and LOBPCG on
the real unknowns against the reference-restating complex LOBPCG."""
import ctypes as C

import numpy as np
import pytest

import oracle
from oracle import gamma_real as gr
from oracle.lobpcg import PreconditionerTPA, lobpcg_hyper

import dftk_jl_amd as dftk
from dftk_jl_amd._lib import check


def _block(Ecut=4, fft=(15, 15, 15), terms=("Kinetic", "AtomicLocal", "AtomicNonlocal")):
    Si = oracle.ElementPsp("Si", oracle.load_psp_hgh("Si", "lda"))
    lat = 10.26 / 2 * np.array([[0, 1, 1.0], [1, 0, 1.0], [1, 1, 0.0]])
    model = oracle.Model(lat, [Si, Si], [np.ones(3) / 8, -np.ones(3) / 8], terms=terms)
    basis = oracle.PlaneWaveBasis(model, Ecut, oracle.ExplicitKpoints([[0, 0, 0]], [1.0]), fft_size=fft)
    _, ham = oracle.energy_hamiltonian(basis, None, None)
    return basis, ham[0]


def test_pair_tables_equal_the_librarys():
    lib = dftk.load_library()
    basis, H = _block(6, (18, 18, 18))
    m = np.ascontiguousarray(basis.kpoints[0].mapping, dtype=np.int64)
    g, mg = gr.pair_tables(basis.fft_size, m)
    cnt = C.c_int64()
    rows, prt = np.zeros(len(g), dtype=np.int32), np.zeros(len(g), dtype=np.int32)
    nx, ny, nz = basis.fft_size
    check(lib.dftk_mi_gamma_tables_host(nx, ny, nz, len(m), m.ctypes.data, C.byref(cnt), rows.ctypes.data, prt.ctypes.data))
    assert cnt.value == len(g) and np.array_equal(rows, g) and np.array_equal(prt, mg)
    with pytest.raises(ValueError):
        gr.pair_tables((8, 8, 8), np.arange(512))                      # Nyquist points
    b2, H2 = _block(6, (18, 18, 18))
    shifted = oracle.PlaneWaveBasis(b2.model, 6, oracle.ExplicitKpoints([[0.25, 0, 0]], [1.0]), fft_size=(18, 18, 18))
    with pytest.raises(ValueError):
        gr.pair_tables(shifted.fft_size, shifted.kpoints[0].mapping)   # k != 0: no inversion symmetry


def test_format_roundtrip_and_inner_products():
    basis, H = _block()
    blk = gr.RealSymmetricBlock(H, basis.fft_size)
    rng = np.random.default_rng(0)
    r = rng.standard_normal((blk.n_real, 4))
    x = blk.unpack(r)
    assert np.allclose(blk.pack(x), r, atol=1e-15)
    cube = basis.ifft(basis.kpoints[0], x[:, 0], normalize=False)
    assert np.abs(cube.imag).max() < 1e-12 * np.abs(cube).max()        # a real field
    assert np.allclose(r.T @ r, x.conj().T @ x, atol=1e-13)            # real dots = complex inner products
    z = rng.standard_normal((H.n_G, 3)) + 1j * rng.standard_normal((H.n_G, 3))
    zs = blk.unpack(blk.pack(z))                                       # the real-symmetric part: a projection
    assert np.allclose(blk.unpack(blk.pack(zs)), zs, atol=1e-15)


@pytest.mark.parametrize("terms", [("Kinetic",), ("Kinetic", "AtomicLocal", "AtomicNonlocal")])
def test_restricted_operator_is_real_symmetric_with_the_complex_spectrum(terms):
    basis, H = _block(terms=terms)
    blk = gr.RealSymmetricBlock(H, basis.fft_size)
    Hr = blk.to_dense()
    assert np.abs(Hr - Hr.T).max() < 1e-12 * np.abs(Hr).max()
    np.testing.assert_allclose(np.linalg.eigvalsh(Hr), np.linalg.eigvalsh(H.to_dense()), atol=1e-11)


def test_lobpcg_on_the_real_unknowns_equals_the_complex_iteration():
    basis, H = _block(5, (16, 16, 16))
    rng = np.random.default_rng(3)
    M = 8
    X0 = np.linalg.qr(rng.standard_normal((H.n_G, M)) + 1j * rng.standard_normal((H.n_G, M)))[0]
    rr = gr.lobpcg_gamma_real(H, basis.fft_size, X0, tol=1e-9, n_conv_check=6)
    rc = lobpcg_hyper(H.mul, X0, prec=PreconditionerTPA(H.kinetic), tol=1e-9, n_conv_check=6)
    dense = np.linalg.eigvalsh(H.to_dense())
    assert rr["converged"] and rc["converged"]
    np.testing.assert_allclose(rr["λ"][:6], dense[:6], atol=1e-9)
    np.testing.assert_allclose(rr["λ"][:6], rc["λ"][:6], atol=1e-9)
    assert abs(rr["n_iter"] - rc["n_iter"]) <= 10
    X = rr["X"]
    assert np.linalg.norm(X.conj().T @ X - np.eye(M)) < 1e-10
    blk = gr.RealSymmetricBlock(H, basis.fft_size)
    assert np.allclose(blk.unpack(blk.pack(X)), X, atol=1e-14)         # real-symmetric vectors come back
    R = H.mul(X) - X * rr["λ"][None, :]
    assert np.linalg.norm(R[:, :6], axis=0).max() < 1e-9               # eigenvectors of the GENERAL operator
    assert not np.any(np.imag(rr["X_real"]))                           # the iteration never left the real numbers


def test_two_bands_share_one_transform():
    """The packing identities of the device path against the oracle's local operator and density: with a real potential
    and an even kinetic factor, H_loc (a + i b) = H_loc a + i H_loc b with both parts real-symmetric, and
    |IFFT(a + i b)|^2 splits into the two real fields."""
    basis, H = _block(5, (16, 16, 16))
    blk = gr.RealSymmetricBlock(H, basis.fft_size)
    g, mg, n = blk.g, blk.mg, H.n_G
    rng = np.random.default_rng(1)
    ra, rb = rng.standard_normal(blk.n_real), rng.standard_normal(blk.n_real)
    ha, hb = gr.real_to_half(ra), gr.real_to_half(rb)
    z = gr.pack_pair(ha, hb, g, mg, n)
    w = H.apply_local(z[:, None])[:, 0] + H.kinetic * z
    a, b = gr.unpack_pair(w, g, mg)
    xa, xb = gr.from_half(ha, g, mg, n), gr.from_half(hb, g, mg, n)
    ref_a = gr.to_half((H.apply_local(xa[:, None])[:, 0] + H.kinetic * xa)[:, None], g, mg)[:, 0]
    ref_b = gr.to_half((H.apply_local(xb[:, None])[:, 0] + H.kinetic * xb)[:, None], g, mg)[:, 0]
    assert np.allclose(a, ref_a, atol=1e-13) and np.allclose(b, ref_b, atol=1e-13)
    assert a[0].imag == 0 and b[0].imag == 0
    # odd band count: b = None
    a1, b1 = gr.unpack_pair(H.apply_local(gr.pack_pair(ha, None, g, mg, n)[:, None])[:, 0], g, mg)
    assert np.allclose(b1, 0, atol=1e-13)
    # density: Re^2 and Im^2 of the packed transform are the two bands' densities
    kpt = basis.kpoints[0]
    cz = basis.ifft(kpt, z, normalize=False)
    ca, cb = basis.ifft(kpt, xa, normalize=False), basis.ifft(kpt, xb, normalize=False)
    assert np.allclose(cz.real ** 2, np.abs(ca) ** 2, atol=1e-12 * np.abs(ca).max() ** 2)
    assert np.allclose(cz.imag ** 2, np.abs(cb) ** 2, atol=1e-12 * np.abs(cb).max() ** 2)


def test_phase_alignment_takes_over_complex_orbitals_without_loss():
    """Real-symmetric vectors times arbitrary global phases (one exactly i, whose plain symmetric part vanishes): after
    align_phase the real-symmetric part is +-the real field itself."""
    basis, H = _block()
    blk = gr.RealSymmetricBlock(H, basis.fft_size)
    rng = np.random.default_rng(2)
    r = rng.standard_normal((blk.n_real, 6))
    x = blk.unpack(r)
    phases = np.exp(1j * np.array([0.0, np.pi / 2, 1.0, -2.5, np.pi, 0.3]))
    y = x * phases[None, :]
    lost = blk.unpack(blk.pack(y))                                     # without alignment: column 1 is annihilated
    assert np.linalg.norm(lost[:, 1]) < 1e-12 * np.linalg.norm(x[:, 1])
    z = blk.unpack(blk.pack(gr.align_phase(y, blk.g, blk.mg)))
    sign = np.sign(np.real(np.sum(np.conj(x) * z, axis=0)))
    assert np.all(np.abs(sign) == 1) and np.allclose(z, x * sign[None, :], atol=1e-13)
    assert np.allclose(gr.align_phase(x, blk.g, blk.mg), x, atol=0)    # already aligned: untouched, bit for bit
