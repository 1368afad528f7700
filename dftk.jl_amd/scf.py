"""SCF driver glue around the device hot path (host mirror of src/scf/*.jl, src/occupation.jl).

``self_consistent_field`` (self_consistent_field.jl:164-289) with ``ScfAndersonDensitySolver``
(scf_solvers.jl:68-102, anderson.jl:36-130), the mixing rules of mixing.py (default ``LdosMixing``, simple
mixing at T = 0, chi0models.jl:32, mixing.jl:264-266), ``AdaptiveDiagtol`` (scf_callbacks.jl:191-212),
``AdaptiveBands`` (nbands_algorithm.jl:52-110), ``next_density`` (:80-129) and the Fermi-level
search (occupation.jl:53-211; None / Fermi-Dirac / Gaussian smearing).  Everything cube- or
block-sized stays in HBM; the host only sees eigenvalues, occupations and scalars.
"""
from __future__ import annotations

import math
import os
import time

import numpy as np
import torch
from scipy.special import erfc

from .densities import compute_density
from .eigen import diagonalize_all_kblocks, lobpcg_hyper
from .mixing import LdosMixing
from .terms import energy_hamiltonian, guess_density

EPS = float(np.finfo(np.float64).eps)


# ---------------------------------------------------------------------------------- occupations
def _smear(kind, x):
    if kind == "none":
        return np.where(x > 0, 0.0, 1.0)
    if kind == "fermi_dirac":      # Smearing.jl:66-76 (overflow-safe form)
        out = np.empty_like(x)
        pos = x > 0
        y = np.exp(-x[pos])
        out[pos] = y / (1 + y)
        out[~pos] = 1 / (1 + np.exp(x[~pos]))
        return out
    if kind == "gaussian":
        return erfc(x) / 2
    raise NotImplementedError(f"smearing {kind}")


def _occupations(model, eigenvalues, eF):
    T = model.temperature
    out = []
    for ek in eigenvalues:
        ek = np.asarray(ek, dtype=float)
        x = np.where(ek > eF, np.inf, -np.inf) if T == 0 else (ek - eF) / T
        out.append(model.filled_occupation * _smear(model.smearing if T > 0 else "none", x))
    return out


def _fermi_bisection(basis, model, kw, eig_all, lo, hi):
    """The bisection loop of FermiBisection as ONE library call (``dftk_mi_fermi_bisection``: host-only C, same rule --
    halve until the midpoint equals an end point); None when the library cannot take it."""
    import ctypes as C
    import os
    kind = {"fermi_dirac": 1, "gaussian": 2}.get(model.smearing)
    if kind is None or os.environ.get("DFTK_MI_TORCH_LOCAL"):
        return None
    lib = getattr(basis, "lib", None)
    if lib is None:              # (a host-only entry point: usable without a device)
        try:
            from . import _lib
            lib = _lib.load()
        except Exception:
            return None
    nb = np.ascontiguousarray([len(e) for e in eig_all], dtype=np.int32)
    flat = np.ascontiguousarray(np.concatenate(eig_all), dtype=np.float64)
    kw_c = np.ascontiguousarray(kw, dtype=np.float64)
    out = C.c_double()
    st = lib.dftk_mi_fermi_bisection(len(eig_all), nb.ctypes.data, flat.ctypes.data, kw_c.ctypes.data, kind,
                                     float(model.temperature), float(model.filled_occupation), float(model.n_electrons),
                                     float(lo), float(hi), C.byref(out))
    return float(out.value) if st == 0 else None


def compute_occupation(basis, eigenvalues, tol_n_elec: float = 1e-6):
    """occupation.jl:53-132,160-211.  ``eigenvalues`` are this rank's k-points.  The reference reduces every
    trial electron count over ``comm_kpts`` (weighted_ksum, PlaneWaveBasis.jl:509-512) -- up to 200 all-reduces
    per bisection; here the (n_k x n_bands) eigenvalues are gathered ONCE and every rank runs the same search
    on the host."""
    model, comm = basis.model, basis.comm_kpts
    local = [(float(w), np.asarray(e, dtype=float).tolist()) for w, e in zip(basis.kweights, eigenvalues)]
    parts = comm.gather_lists(local)
    kw = np.array([w for part in parts for w, _ in part])
    eig_all = [np.asarray(e, dtype=float) for part in parts for _, e in part]

    # one rectangular array when every k-point carries the same number of bands (the usual case): a trial Fermi level
    # is then ONE vectorised evaluation instead of a Python loop over the k-points (72 of them for BASELINE configs[2],
    # ~170 trials per SCF step); the per-k sums are formed first and added in k order, exactly like the loop
    rect = np.stack(eig_all) if len({len(e) for e in eig_all}) == 1 else None

    def excess(eF):
        if rect is not None:
            occ = _occupations(model, [rect], eF)[0]
            return float(np.dot(kw, occ.sum(axis=1))) - model.n_electrons
        occ = _occupations(model, eig_all, eF)
        return float(sum(w * o.sum() for w, o in zip(kw, occ))) - model.n_electrons

    n_fill = -(-model.n_electrons // (model.n_spin_components * model.filled_occupation))
    homo = max(float(e[n_fill - 1]) for e in eig_all)
    lumo = min((float(np.min(e[n_fill:])) if len(e) > n_fill else math.inf) for e in eig_all)
    eF = homo + 1 if lumo == math.inf else (homo + lumo) / 2           # guess_fermi_level_intocc_
    if model.temperature == 0:
        if abs(excess(eF)) > tol_n_elec:
            raise RuntimeError("Unable to find non-fractional occupations that have the correct number of "
                               "electrons. You should add a temperature.")
    elif abs(excess(eF)) >= tol_n_elec / 10:                           # FermiBisection (:99-132)
        if excess(eF) < 0:
            lo, hi = eF, max(float(np.max(e)) for e in eig_all) + 1
        else:
            lo, hi = min(float(np.min(e)) for e in eig_all) - 1, eF
        eF = _fermi_bisection(basis, model, kw, eig_all, lo, hi)
        if eF is None:           # (smearing the library does not know, or DFTK_MI_TORCH_LOCAL=1: the interpreted twin)
            for _ in range(200):
                mid = 0.5 * (lo + hi)
                if mid == lo or mid == hi:
                    break
                if excess(mid) < 0:
                    lo = mid
                else:
                    hi = mid
            eF = 0.5 * (lo + hi)
    return _occupations(model, eigenvalues, eF), eF


# ---------------------------------------------------------------------------------- band counts
def default_n_bands(model, temperature_factor=1.05):
    min_n = -(-model.n_electrons // (model.n_spin_components * model.filled_occupation))
    return int(math.ceil(min_n * (1.0 if model.temperature == 0 else temperature_factor)))


class AdaptiveBands:
    """nbands_algorithm.jl:52-110."""

    def __init__(self, model, n_bands_converge=None, occupation_threshold=1e-6, gap_min=1e-2):
        self.n_bands_converge = default_n_bands(model, 1.05) if n_bands_converge is None else int(n_bands_converge)
        self.n_bands_compute = max(3 + self.n_bands_converge, default_n_bands(model, 1.20))
        self.occupation_threshold = occupation_threshold
        self.gap_min = gap_min

    def determine_n_bands(self, occupation, eigenvalues, psi):
        n_psi = max(p.shape[0] for p in psi) if psi is not None else 0
        if occupation is None:
            return (self.n_bands_converge + self.n_bands_compute) // 2, max(self.n_bands_compute, n_psi)
        n_occ = 0
        for occk in occupation:
            idx = np.nonzero(np.abs(occk) >= self.occupation_threshold)[0]
            n_occ = max(n_occ, int(idx[-1]) + 1 if len(idx) else len(occk) + 1)
        n_conv = max(self.n_bands_converge, n_occ)
        n_eps = 0
        if eigenvalues is not None:
            for ek in eigenvalues:
                if n_conv > len(ek):
                    n_eps = max(n_eps, len(ek) + 1)
                    continue
                idx = np.nonzero(ek <= ek[n_conv - 1] + self.gap_min)[0]
                n_eps = max(n_eps, int(idx[-1]) + 1 if len(idx) else len(ek) + 1)
        return n_conv, max(self.n_bands_compute, n_eps, n_conv + 3, n_psi)


class FixedBands:
    """nbands_algorithm.jl:21-37."""

    def __init__(self, n_bands_converge, n_bands_compute=None, occupation_threshold=1e-6):
        self.n_bands_converge = n_bands_converge
        self.n_bands_compute = n_bands_compute if n_bands_compute is not None else n_bands_converge + 3
        self.occupation_threshold = occupation_threshold

    @classmethod
    def for_model(cls, model, temperature_factor_converge=1.20, occupation_threshold=1e-6):
        """``FixedBands(model; temperature_factor_converge)`` (nbands_algorithm.jl:26-30)."""
        n_conv = default_n_bands(model, temperature_factor_converge)
        return cls(n_conv, n_conv + 3, occupation_threshold)

    def determine_n_bands(self, occupation, eigenvalues, psi):
        return self.n_bands_converge, self.n_bands_compute


def next_density(ham, nbandsalg, eigensolver=lobpcg_hyper, psi=None, eigenvalues=None, occupation=None,
                 tol=1e-6, generator=None, seed=0, timers=None, extra_weights=None, coarse_start=True):
    """self_consistent_field.jl:80-129.  ``extra_weights(basis, eigenvalues, eF, psi) -> (weights, threshold) | None``: a second
    set of band weights accumulated in the density pass (``rho_extra`` of the result; the stepper passes the LDOS weights of
    its mixing)."""
    basis = ham[0].basis
    n_conv, n_comp = nbandsalg.determine_n_bands(occupation, eigenvalues, psi)
    if psi is not None:
        n_comp = max(n_comp, max(p.shape[0] for p in psi))
    n_comp = int(basis.comm_kpts.max_scalar(n_comp))                       # mpi_max(n_bands_compute)
    t0 = time.time()
    eig = diagonalize_all_kblocks(eigensolver, ham, n_comp, psiguess=psi, n_conv_check=n_conv, tol=tol,
                                  miniter=1, generator=generator, seed=seed, coarse_start=coarse_start)
    t1 = time.time()
    occ, eF = compute_occupation(basis, eig["λ"], tol_n_elec=nbandsalg.occupation_threshold)
    # extra_weights (optional callback of the mixing: LdosMixing's local density of states): accumulated in the same pass
    extra = extra_weights(basis, eig["λ"], eF, eig["X"]) if extra_weights is not None else None
    rho_extra = None
    if extra is not None:
        rho, rho_extra = compute_density(basis, eig["X"], occ, nbandsalg.occupation_threshold,
                                         real_symmetric=eig.get("real_symmetric"), extra_weights=extra[0],
                                         extra_threshold=extra[1])
    else:
        rho = compute_density(basis, eig["X"], occ, nbandsalg.occupation_threshold,
                              real_symmetric=eig.get("real_symmetric"))
    if timers is not None:
        timers["diagonalization"] = timers.get("diagonalization", 0.0) + t1 - t0
        timers["occupation+density"] = timers.get("occupation+density", 0.0) + time.time() - t1
    n_matvec = int(basis.comm_kpts.sum_scalar(eig["n_matvec"]))    # (not over comm_pw: those ranks share the blocks)
    return dict(psi=eig["X"], eigenvalues=eig["λ"], occupation=occ, eF=eF, rho=rho, diagonalization=eig,
                n_bands_converge=n_conv, n_matvec=n_matvec, rho_extra=rho_extra,
                n_matvec_coarse=int(eig.get("n_matvec_coarse", 0)))


# ---------------------------------------------------------------------------------- Anderson
class AndersonAcceleration:
    """anderson.jl:36-130.  The history (<= m cube-sized vectors) stays in HBM; the m x m
    least-squares problem min |Pf_n + M beta| is solved on the host from the Gram matrix
    M'M (cond(R)^2 = cond(M'M) gives the reference's conditioning test on the QR factor).
    Only elementwise products / row sums / axpys touch the cube-sized vectors.  The inner products <r_i, r_j> of
    the history are kept from step to step: a step computes only the nh + 1 products with the new residual, in ONE
    stacked reduction and ONE host fetch (a fetch is a device synchronisation; the full Gram matrix was ~13 of them
    and ~55 cube-sized reductions per step)."""

    def __init__(self, m=10, maxcond=1e6, errorfactor=1e5):
        self.iterates, self.residuals, self.errors = [], [], []
        self.gram = np.zeros((0, 0))          # gram[i, j] = <residuals[i], residuals[j]>
        self.m, self.maxcond, self.errorfactor = m, maxcond, errorfactor

    def _push(self, x, r, row, rr_new):
        """Append (x, r); ``row[i]`` = <residuals[i], r>, ``rr_new`` = <r, r>."""
        self.iterates.append(x.reshape(-1).clone())
        self.residuals.append(r.reshape(-1).clone())
        self.errors.append(math.sqrt(max(rr_new, 0.0)))
        n = self.gram.shape[0]
        g = np.zeros((n + 1, n + 1))
        g[:n, :n] = self.gram
        g[:n, n] = g[n, :n] = row
        g[n, n] = rr_new
        self.gram = g
        if len(self.iterates) > self.m:
            self._delete([0])

    def _delete(self, idxs):
        for i in sorted(idxs, reverse=True):
            for lst in (self.iterates, self.residuals, self.errors):
                lst.pop(i)
        keep = [i for i in range(self.gram.shape[0]) if i not in set(idxs)]
        self.gram = self.gram[np.ix_(keep, keep)]

    def __call__(self, x, alpha, Pfx):
        if self.m == 0:
            return x + alpha * Pfx
        pf = Pfx.reshape(-1)
        xf = x.reshape(-1)
        # <r_i, pf> for the whole history and <pf, pf>: one stacked reduction, one fetch
        vals = torch.stack([(ri * pf).sum() for ri in self.residuals] + [(pf * pf).sum()]).cpu().numpy()
        rp_all, pp = vals[:-1].astype(float), float(vals[-1])
        if not self.iterates:
            self._push(x, Pfx, rp_all, pp)
            return x + alpha * Pfx
        err_n = math.sqrt(max(pp, 0.0))
        min_error = min(self.errors + [err_n])
        drop = [i for i, e in enumerate(self.errors[:-1]) if e > self.errorfactor * min_error]
        if drop:
            self._delete(drop)
            rp_all = np.delete(rp_all, drop)
        nh = len(self.residuals)
        rr, rp = self.gram, rp_all
        keep = list(range(nh))
        while True:
            # M[:, j] = r_j - pf  =>  G = M'M, b = M'pf
            G = rr[np.ix_(keep, keep)] - rp[keep][:, None] - rp[keep][None, :] + pp
            bvec = rp[keep] - pp
            ev = np.linalg.eigvalsh(G)
            cond_R = math.sqrt(max(ev[-1], 0.0) / max(ev[0], 1e-300)) if ev[-1] > 0 else 1.0
            if len(keep) > 1 and cond_R > self.maxcond:
                worst = int(np.argmax([self.errors[k] for k in keep[:-1]]))
                keep.pop(worst)
                continue
            break
        if len(keep) < nh:
            gone = [k for k in range(nh) if k not in keep]
            self._delete(gone)
            rp = np.delete(rp, gone)
        betas = -np.linalg.lstsq(G, bvec, rcond=None)[0]
        xn = xf + alpha * pf
        sb = float(np.sum(betas))
        # sum_i beta_i (x_i - x + alpha (r_i - pf))
        xn = xn - sb * (xf + alpha * pf)
        for ib, beta in enumerate(betas):
            xn.add_(self.iterates[ib], alpha=float(beta))
            xn.add_(self.residuals[ib], alpha=float(beta) * alpha)
        self._push(x, Pfx, rp, pp)
        return xn.reshape(x.shape)


class AndersonNative:
    """The same accelerator behind the C ABI (``dftk_mi_anderson_*``, csrc/mix_kernels.hip): history on the device, a step
    is one reduction kernel, one host synchronisation, the small least-squares problem on the host and one fused update
    kernel (the torch formulation above: ~25 launches).  ``AndersonAcceleration`` stays as its parity twin
    (``DFTK_MI_TORCH_MIX=1``)."""

    def __init__(self, basis, m=10, maxcond=1e6, errorfactor=1e5):
        self.basis, self.m, self.maxcond, self.errorfactor = basis, int(m), float(maxcond), float(errorfactor)
        self.handle, self.n = None, None

    def _ensure(self, n):
        import ctypes as C
        from . import _lib
        if self.handle is not None and self.n == n:
            return
        self.close()
        h = C.c_void_p()
        _lib.check(self.basis.lib.dftk_mi_anderson_create(self.basis.handle, n, self.m, self.maxcond, self.errorfactor, C.byref(h)))
        self.handle, self.n = h, n

    def __call__(self, x, alpha, Pfx):
        from . import _lib
        xin = x.to(torch.float64).contiguous()
        pf = Pfx.to(torch.float64).contiguous()
        self._ensure(xin.numel())
        out = torch.empty_like(xin)
        self.basis.pre_call()
        _lib.check(self.basis.lib.dftk_mi_anderson_step(self.handle, xin.data_ptr(), float(alpha), pf.data_ptr(), out.data_ptr(), None))
        return out

    @property
    def n_history(self):
        return 0 if self.handle is None else int(self.basis.lib.dftk_mi_anderson_history(self.handle))

    def close(self):
        if self.handle is not None:
            try:
                self.basis.lib.dftk_mi_anderson_destroy(self.handle)
            except Exception:
                pass
            self.handle = None

    def __del__(self):
        self.close()


def determine_diagtol(n_iter, history_drho, ratio=0.2, diagtol_max=0.005, diagtol_first=None, diagtol_min=None):
    """``determine_diagtol(::AdaptiveDiagtol, info)`` (scf_callbacks.jl:191-212): ``diagtol_first = 6 diagtol_max``
    unless given, ``min(diagtol_first, 5 diagtol_max)`` while ``n_iter <= 1``, then ``ratio * min(history_drho)``
    clamped to ``[diagtol_min = 100 eps, diagtol_max]``."""
    if diagtol_first is None:
        diagtol_first = 6 * diagtol_max
    if n_iter <= 1:
        return min(diagtol_first, 5 * diagtol_max)
    diagtol = min(history_drho) * ratio
    if not math.isfinite(diagtol):
        raise FloatingPointError("AdaptiveDiagtol: non-finite density change")
    return float(np.clip(diagtol, 100 * EPS if diagtol_min is None else diagtol_min, diagtol_max))


NONLINEAR_TERMS = ("Hartree", "Xc", "LocalNonlinearity")   # <: TermNonlinear (hartree.jl:24, xc.jl:75, local_nonlinearity.jl:7)


def has_nonlinear_terms(model):
    """``any(t -> t isa TermNonlinear, basis.terms)``; an ``Xc`` without functionals instantiates as ``TermNoop``
    (xc.jl:33) and does not count."""
    return any(t in NONLINEAR_TERMS and not (t == "Xc" and not model.functionals) for t in model.term_types)


def default_diagtolalg(basis, tol):
    """``default_diagtolalg(basis; tol)`` (scf_callbacks.jl:220-230): exact exchange -> ``ratio_rhodiff = 5e-4``;
    any nonlinear term (Hartree, Xc: every DFT model) -> plain ``AdaptiveDiagtol()``, i.e. 0.025 on the first two
    steps; only LINEAR models (one diagonalisation is the answer) take ``diagtol_first = tol / 5``."""
    terms = basis.model.term_types
    if "ExactExchange" in terms:
        kw = dict(ratio=5e-4)
    elif has_nonlinear_terms(basis.model):
        kw = dict()
    else:
        kw = dict(diagtol_first=tol / 5)

    def determine_tol(n_iter, history_drho):
        return determine_diagtol(n_iter, history_drho, **kw)
    determine_tol.params = kw
    return determine_tol


class ScfStepper:
    """One object = the state of ``self_consistent_field`` between fixed-point iterations
    (``fixpoint_map`` + ``ScfAndersonDensitySolver`` step, self_consistent_field.jl:200-257,
    scf_solvers.jl:85-98).  ``step()`` performs exactly one SCF iteration."""

    def __init__(self, basis, rho=None, psi=None, tol=1e-6, damping=0.8, nbandsalg=None, is_converged=None,
                 eigensolver=lobpcg_hyper, anderson_m=10, seed=0, determine_tol=determine_diagtol, mixing=None,
                 phase_timers=None, coarse_start=True):
        basis._require_gpu()
        self.basis = basis
        # two-level start of the first diagonalisation when the basis carries a companion basis (basis.py; an extension)
        self.coarse_start = bool(coarse_start)
        self.phase_timers = (os.environ.get("DFTK_MI_PHASE_TIMERS") is not None) if phase_timers is None else bool(phase_timers)
        # mixing = LdosMixing() as the reference (self_consistent_field.jl:177): simple mixing at T = 0
        self.mixing = mixing if mixing is not None else LdosMixing()
        # per-step nonlocal energy from the Ritz values (terms.energy_hamiltonian); the energies returned by
        # finalize() / self_consistent_field always come from the projections themselves
        self.ritz_energies = os.environ.get("DFTK_MI_EXACT_STEP_ENERGIES") is None
        self.gen = torch.Generator(device=basis.device)
        self.gen.manual_seed(seed + 7919 * basis.comm_kpts.rank)
        self.seed = seed
        self.rho_in = guess_density(basis) if rho is None else rho
        self.nbandsalg = nbandsalg if nbandsalg is not None else AdaptiveBands(basis.model)
        self.is_converged = is_converged or (lambda info: info["history_drho"][-1] < tol)   # ScfConvergenceDensity
        if determine_tol is None or determine_tol is determine_diagtol:
            determine_tol = default_diagtolalg(basis, tol)      # self_consistent_field.jl:179
        self.eigensolver, self.damping, self.determine_tol = eigensolver, damping, determine_tol
        from .mixing import _torch_mix
        self.accel = AndersonAcceleration(m=anderson_m) if _torch_mix() else AndersonNative(basis, m=anderson_m)
        self.sqrt_dvol = math.sqrt(basis.dvol)
        self.info = dict(psi=psi, occupation=None, eigenvalues=None, eF=None, n_iter=0, n_matvec=0, converged=False,
                         history_Etot=[], history_drho=[], rho=self.rho_in, timings=[])

    def step(self):
        # the whole step runs with the library's stream as torch's current stream: torch kernels and library kernels are
        # ordered by that one stream, no host synchronisation around the library calls (basis.pre_call / post_call)
        with self.basis.on_library_stream():
            out = self._step()
            self.basis.sync()             # what the caller reads next (on ITS stream) is complete
        return out

    def _step(self):
        basis, info = self.basis, self.info
        t_it = time.time()
        timers = {}

        def lap(name, t0):
            # per-phase device synchronisations only when asked for (phase_timers / DFTK_MI_PHASE_TIMERS=1): they are
            # five of a k-point step's host synchronisations; without them the entries are host enqueue times
            if self.phase_timers:
                basis.sync()
            timers[name] = timers.get(name, 0.0) + time.time() - t0
            return time.time()

        t = time.time()
        # only the Hamiltonian of rho_in is needed here; the psi-dependent energy terms of this call
        # would be thrown away (the energies reported for the step are those of (psi_out, rho_out) below)
        _, ham = energy_hamiltonian(basis, None, None, rho=self.rho_in, only_hamiltonian=True)
        t = lap("energy_hamiltonian", t)
        diagtol = self.determine_tol(info["n_iter"], info["history_drho"])
        nxt = next_density(ham, self.nbandsalg, self.eigensolver, psi=info["psi"], eigenvalues=info["eigenvalues"],
                           occupation=info["occupation"], tol=diagtol, generator=self.gen, seed=self.seed,
                           timers=timers, extra_weights=getattr(self.mixing, "extra_density_weights", None),
                           coarse_start=self.coarse_start)
        t = time.time()
        # int V_in rho_out (Ritz-value form of the nonlocal energy) and ||rho_out - rho_in||^2 in one library call / one fetch
        ritz_pot = self._ritz_potential(ham) if self.ritz_energies else None
        sums = self._step_sums(nxt["rho"], ritz_pot, self.rho_in)
        energies, _ = energy_hamiltonian(basis, nxt["psi"], nxt["occupation"], rho=nxt["rho"], only_energies=True,
                                         eigenvalues=nxt["eigenvalues"], eF=nxt["eF"], ritz_potential=ritz_pot,
                                         ritz_occupation_threshold=self.nbandsalg.occupation_threshold,
                                         ritz_potential_dot=None if sums is None or ritz_pot is None else sums[0])
        t = lap("energies", t)
        drho = nxt["rho"] - self.rho_in
        n_matvec_total = info["n_matvec"] + nxt["n_matvec"]
        info = dict(info, **nxt)
        info.update(n_iter=info["n_iter"] + 1, n_matvec=n_matvec_total, n_matvec_step=nxt["n_matvec"],
                    energies=energies, ham=ham, rho_in=self.rho_in, diagtol=diagtol,
                    history_Etot=info["history_Etot"] + [energies.total],
                    history_drho=info["history_drho"] + [(float(torch.linalg.norm(drho).item()) if sums is None
                                                          else math.sqrt(max(sums[1], 0.0))) * self.sqrt_dvol])
        # rank 0 decides for everybody (mpi_bcast(converged, comm_kpts), self_consistent_field.jl:249): the ranks
        # hold bit-identical densities, but a control-flow split on a last-bit difference must never deadlock
        conv = bool(self.is_converged(info))
        for comm in (basis.comm_kpts, basis.comm_pw):
            if comm.size > 1:
                conv = bool(comm.gather_lists(conv)[0])
        info["converged"] = conv
        info["timings"] = info["timings"] + [time.time() - t_it]
        if not info["converged"]:
            # rho_next = Anderson(rho_in, beta, mix_density(mixing, rho_out - rho_in))   (:247, scf_solvers.jl:85-98)
            t = time.time()
            pf = self.mixing.mix_density(basis, drho, eF=nxt["eF"], eigenvalues=nxt["eigenvalues"], psi=nxt["psi"],
                                         occupation=nxt["occupation"], rho_in=self.rho_in, ldos=nxt.get("rho_extra"))
            self.rho_in = self.accel(self.rho_in, self.damping, pf)
            lap("mixing", t)
        info["timers"] = timers
        info["timers_synced"] = self.phase_timers
        self.info = info
        return info

    def _step_sums(self, rho_out, v_in, rho_in):
        """``dftk_mi_step_sums``: (sum V_in rho_out, sum (rho_out - rho_in)^2) with one kernel and one fetch; None with
        ``DFTK_MI_TORCH_LOCAL=1`` (the torch twins take over) or for arrays the call cannot take."""
        import ctypes as C
        import os
        basis = self.basis
        if os.environ.get("DFTK_MI_TORCH_LOCAL") is not None:
            return None
        arrs = [a for a in (rho_out, v_in, rho_in) if a is not None]
        if not all(a.is_cuda and a.dtype == torch.float64 and a.is_contiguous() and a.numel() == rho_out.numel() for a in arrs):
            return None
        out = (C.c_double * 2)()
        basis.pre_call()
        from . import _lib
        _lib.check(basis.lib.dftk_mi_step_sums(basis.handle, rho_out.numel(), rho_out.data_ptr(),
                                               v_in.data_ptr() if v_in is not None else None, rho_in.data_ptr(), out))
        return float(out[0]), float(out[1])

    @staticmethod
    def _ritz_potential(ham):
        """The local potential the step's Hamiltonian was built from; with collinear spin both channels' potentials
        (the spin-up blocks come first, PlaneWaveBasis.jl:50-53)."""
        if ham[0].basis.model.n_spin_components == 2:
            return torch.stack([ham[0].potential, ham[len(ham) // 2].potential])
        return ham[0].potential

    def finalize(self):
        info = self.info
        with self.basis.on_library_stream():
            energies, ham = energy_hamiltonian(self.basis, info["psi"], info["occupation"], rho=info["rho"],
                                               eigenvalues=info["eigenvalues"], eF=info["eF"])
            self.basis.sync()
        info.update(energies=energies, ham=ham)
        return info


def self_consistent_field(basis, rho=None, psi=None, tol=1e-6, maxiter=100, damping=0.8, nbandsalg=None,
                          is_converged=None, callback=None, eigensolver=lobpcg_hyper, anderson_m=10, seed=0,
                          determine_tol=determine_diagtol, mixing=None, coarse_start=True):
    """``self_consistent_field(basis; rho, psi, tol, maxiter, damping, nbandsalg, is_converged, callback,
    eigensolver)`` (self_consistent_field.jl:164-289)."""
    t0 = time.time()
    stepper = ScfStepper(basis, rho=rho, psi=psi, tol=tol, damping=damping, nbandsalg=nbandsalg,
                         is_converged=is_converged, eigensolver=eigensolver, anderson_m=anderson_m, seed=seed,
                         determine_tol=determine_tol, mixing=mixing, coarse_start=coarse_start)
    for _ in range(maxiter):
        info = stepper.step()
        if callback is not None:
            callback(info)
        if info["converged"]:
            break
    info = stepper.finalize()
    info.update(runtime=time.time() - t0, basis=basis, mixing=stepper.mixing, damping=damping,
                occupation_threshold=stepper.nbandsalg.occupation_threshold, algorithm="SCF")
    return info


class ScfDefaultCallback:
    """scf_callbacks.jl:30-124 (table of n, Energy, log10 dE, log10 drho, Diag, dtime)."""

    def __init__(self, stream=None):
        self.prev_E = None
        self.stream = stream

    def __call__(self, info):
        import sys
        out = self.stream or sys.stdout
        if info["n_iter"] == 1:
            print("n     Energy            log10(ΔE)   log10(Δρ)   Diag   Δtime", file=out)
            print("---   ---------------   ---------   ---------   ----   ------", file=out)
        E = info["energies"].total
        dE = "         " if self.prev_E is None else f"{math.log10(abs(E - self.prev_E) + 1e-300):9.2f}"
        self.prev_E = E
        diag = float(np.mean(info["diagonalization"]["n_iter"]))
        print(f"{info['n_iter']:3d}   {E:+15.12f}   {dE}   {math.log10(info['history_drho'][-1] + 1e-300):9.2f}   "
              f"{diag:4.1f}   {info['timings'][-1]:6.2f}s", file=out, flush=True)
