"""CPU oracle: a NumPy/SciPy fp64 restatement of DFTK.jl's plane-wave SCF hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``dftk.jl_amd/`` may import this package;
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg do,
and there only as the checker / the timed CPU baseline, never as the product path.

Every function cites the reference file:line it restates (paths relative to the
DFTK.jl checkout, v0.7.26).  The oracle is pinned against the reference's own golden
vectors in ``tests/test_oracle_golden.py`` (HGH known-answer values, FFT-size rule,
Ewald energies, LOBPCG eigenvalue pins, per-term energies of a guess density, the
ABINIT-referenced silicon LDA SCF).  Parity status per third-party dependency:

* FFT (FFTW in the reference)          -> scipy.fft (pocketfft); exact definition, pinned.
* BLAS/LAPACK (OpenBLAS)               -> numpy/scipy LAPACK; pinned via eigenvalue tests.
* XC ``lda_x + lda_c_vwn`` (Libxc)     -> closed forms, pinned by the reference's E["Xc"] values.
* XC ``lda_c_pw`` (``LDA()`` default)  -> closed form from PW92; **parity unpinned**
  (the reference holds no numeric pin for it).
* Spglib symmetry reduction            -> not restated; explicit k-lists / symmetries=false only.
"""
from .psp import PspHgh, load_psp_hgh, HGH_TABLE  # noqa: F401
from .basis import (Model, ElementPsp, PlaneWaveBasis, Kpoint, compute_fft_size,  # noqa: F401
                    MonkhorstPack, ExplicitKpoints, model_DFT, model_atomic)
from .terms import energy_hamiltonian, guess_density, HamiltonianBlock  # noqa: F401
from .lobpcg import lobpcg_hyper, LOBPCG, PreconditionerTPA, diagonalize_all_kblocks  # noqa: F401
from .mixing import (SimpleMixing, KerkerMixing, KerkerDosMixing, DielectricMixing, LdosMixing,  # noqa: F401
                     HybridMixing, Chi0Mixing)
from .scf import (self_consistent_field, compute_density, compute_occupation,  # noqa: F401
                  AdaptiveBands, next_density)
