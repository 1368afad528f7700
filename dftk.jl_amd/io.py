"""``scfres`` wire formats: the dictionary / JSON layout of DFTK's ``scfres_to_dict`` and a checkpoint.

Reference: src/input_output.jl:75-110 (``todict(model)``), :181-204 (``todict(basis)``), :236-328
(``band_data_to_dict``), :345-386 (``scfres_to_dict``), ext/DFTKJSON3Ext.jl (``save_scfres(::Val{:json})``),
src/scf/scfres.jl:69-86 (``save_scfres`` front end: format by file extension, master rank writes).

So that results of the device path drop into DFTK-side post-processing, the same keys are written:
model (``lattice``, ``recip_lattice``, ``atomic_positions``, ``element_symbols``, ``temperature``, ``smearing``,
``n_electrons`` ...), basis (``kcoords``, ``kweights``, ``n_kpoints``, ``fft_size``, ``dvol``, ``Ecut``), bands
(``n_bands``, ``eigenvalues`` / ``occupation`` as (n_bands, n_kpoints, n_spin) arrays, ``εF``, ``diagonalization``),
SCF (``energies``, ``converged``, ``norm_Δρ``, ``n_iter``, ``n_matvec``, ``history_Etot``, ``history_Δρ``,
``n_bands_converge``, ``damping_value``, ``mixing``, ``scfres_extra_keys``) and optionally ``ρ``.

Array convention in JSON: nested lists in Julia's (column-major) nesting, i.e. the LAST Julia dimension is the
outermost list: ``eigenvalues[spin][kpoint][band]``, ``ρ[spin][iz][iy][ix]`` -- exactly the memory order of the
device tensors.  ``.npz`` files (``save_scfres("x.npz")``) additionally hold ``ψ`` and restart an SCF through
``load_scfres`` (the reference's jld2 checkpoint; HDF5 is not available in this image).
"""
from __future__ import annotations

import json
import os

import numpy as np


def _tolist(x):
    if hasattr(x, "detach"):
        x = x.detach().cpu().numpy()
    if isinstance(x, np.ndarray):
        return x.tolist()
    if isinstance(x, (np.floating, np.integer)):
        return x.item()
    return x


def _symmetries_to_dict(symmetries) -> dict:
    """``[symop.W for symop in symmetries]`` / ``[symop.w ...]`` (input_output.jl:101-102): each W as Julia writes a
    matrix -- the list of its COLUMNS."""
    return {"symmetries_rotations": [np.asarray(s.W, dtype=int).T.tolist() for s in symmetries],
            "symmetries_translations": [np.asarray(s.w, dtype=float).tolist() for s in symmetries]}


def model_to_dict(model) -> dict:
    """``todict(model)`` (input_output.jl:75-110).  Matrices as Julia writes them: list of COLUMNS."""
    lat = np.asarray(model.lattice)
    return {
        "model_name": "custom", "lattice": lat.T.tolist(), "recip_lattice": np.asarray(model.recip_lattice).T.tolist(),
        "n_dim": 3, "spin_polarization": model.spin_polarization, "n_spin_components": model.n_spin_components,
        "temperature": model.temperature, "smearing": str(model.smearing), "n_atoms": len(model.atoms),
        "element_symbols": [a.symbol for a in model.atoms], "species": [a.symbol for a in model.atoms],
        "atomic_positions": [np.asarray(p).tolist() for p in model.positions],
        "atomic_positions_cart": [(lat @ np.asarray(p)).tolist() for p in model.positions],
        "n_electrons": model.n_electrons, "pseudofamily": "hgh",
        **_symmetries_to_dict(model.symmetries),
        "terms": list(model.term_types), "functionals": list(model.functionals),
    }


def basis_to_dict(basis) -> dict:
    """``todict(basis)`` (input_output.jl:181-204); k-points of ALL ranks (``kcoords_global``)."""
    d = model_to_dict(basis.model)
    recip = np.asarray(basis.model.recip_lattice)
    kc = [np.asarray(k).tolist() for k in basis.kcoords_global]
    d.update({
        "kgrid": getattr(basis, "kgrid_description", f"ExplicitKpoints with {len(kc)} k-points"), "kcoords": kc, "kcoords_cart": [(recip @ np.asarray(k)).tolist() for k in kc],
        "kweights": list(map(float, basis.kweights_global)), "n_kpoints": len(kc), "fft_size": list(basis.fft_size),
        "dvol": basis.dvol, "Ecut": basis.Ecut, "variational": True,
        "symmetries_respect_rgrid": bool(getattr(basis, "symmetries_respect_rgrid", True)),
        "use_symmetries_for_kpoint_reduction": bool(getattr(basis, "use_symmetries_for_kpoint_reduction", False)),
    })
    # the discretisation may have broken some of the model's symmetries: the basis' own list replaces them (:198-202)
    d.update(_symmetries_to_dict(basis.symmetries))
    return d


def _gather_kpts(basis, local):
    """``gather_kpts_block`` (PlaneWaveBasis.jl): per-k host data of all ranks of comm_kpts, in global k order."""
    parts = basis.comm_kpts.gather_lists([_tolist(x) for x in local])
    return [x for part in parts for x in part]


def _by_spin(basis, local):
    """Per-(k, spin) host data -> [spin][kpoint (global order)]: every rank lists its spin-up blocks, then its spin-down
    blocks (PlaneWaveBasis.jl:50-53); the reference's files index [spin][kpoint][...] (gather_kpts_block)."""
    n_spin = basis.model.n_spin_components
    parts = basis.comm_kpts.gather_lists([_tolist(x) for x in local])
    out = [[] for _ in range(n_spin)]
    for part in parts:
        n = len(part) // n_spin
        for s_ in range(n_spin):
            out[s_].extend(part[s_ * n:(s_ + 1) * n])
    return out


def scfres_to_dict(scfres: dict, save_psi: bool = False, save_rho: bool = True) -> dict:
    """``scfres_to_dict`` (input_output.jl:345-386) for the dict returned by ``self_consistent_field``."""
    basis = scfres["basis"] if "basis" in scfres else scfres["ham"][0].basis
    d = basis_to_dict(basis)
    eig = _by_spin(basis, [np.asarray(e, dtype=float) for e in scfres["eigenvalues"]])
    occ = _by_spin(basis, [np.asarray(o, dtype=float) for o in scfres["occupation"]])
    n_bands = min(len(e) for es in eig for e in es)
    d["n_bands"] = n_bands
    d["eigenvalues"] = [[list(e[:n_bands]) for e in es] for es in eig]          # [spin][kpoint][band]
    d["occupation"] = [[list(o[:n_bands]) for o in os_] for os_ in occ]
    d["εF"] = scfres["eF"]
    diag = scfres["diagonalization"]
    d["diagonalization"] = {
        "n_matvec": int(basis.comm_kpts.sum_scalar(diag["n_matvec"])), "converged": bool(diag["converged"]),
        "residual_norms": [[list(np.asarray(r)[:n_bands]) for r in rs] for rs in _by_spin(basis, diag["residual_norms"])],
        "n_iter": _by_spin(basis, [int(n) for n in diag["n_iter"]]),
    }
    if save_rho:
        rho_ = scfres["rho"]
        d["ρ"] = _tolist(rho_) if rho_.dim() == 4 else [_tolist(rho_)]       # [spin][iz][iy][ix]
        d["τ"] = None
    d["energies"] = {k: float(v) for k, v in scfres["energies"].items()}
    d["energies"]["total"] = float(scfres["energies"].total)
    extra = {"converged": bool(scfres["converged"]), "norm_Δρ": float(scfres["history_drho"][-1]),
             "n_iter": int(scfres["n_iter"]), "n_matvec": int(scfres["n_matvec"]),
             "history_Etot": list(map(float, scfres["history_Etot"])), "history_Δρ": list(map(float, scfres["history_drho"])),
             "n_bands_converge": int(scfres["n_bands_converge"]), "damping_value": float(scfres.get("damping", 0.8)),
             "occupation_threshold": float(scfres.get("occupation_threshold", 1e-6)), "algorithm": "SCF",
             "runtime_ns": int(1e9 * scfres.get("runtime", 0.0))}
    d.update(extra)
    d["mixing"] = type(scfres["mixing"]).__name__ if "mixing" in scfres else "LdosMixing"
    d["eigensolver"] = "lobpcg_hyper (dftk_mi355x)"
    d["scfres_extra_keys"] = list(extra)
    if save_psi:
        n_G = [int(p.shape[1]) for p in scfres["psi"]]
        d["kpt_n_G_vectors"] = _by_spin(basis, n_G)
        d["kpt_max_n_G"] = int(basis.comm_kpts.max_scalar(max(n_G)))
    return d


def save_scfres(filename: str, scfres: dict, save_psi=None, save_rho=None, extra_data=None):
    """``save_scfres(filename, scfres; save_ψ, save_ρ, extra_data)`` (scfres.jl:69-86): ``.json`` (metadata, bands,
    energies; ρ only on request, as the reference) or ``.npz`` (checkpoint with ρ and ψ).  Rank 0 of the
    communicators writes; every rank must call (collective gathers)."""
    ext = os.path.splitext(filename)[1].lower()
    if ext not in (".json", ".npz"):
        raise ValueError(f"Extension '{ext}' not supported (json, npz).")
    basis = scfres["basis"] if "basis" in scfres else scfres["ham"][0].basis
    save_psi = (ext == ".npz") if save_psi is None else save_psi
    save_rho = (ext != ".json") if save_rho is None else save_rho
    d = scfres_to_dict(scfres, save_psi=save_psi, save_rho=save_rho and ext == ".json")
    d.update(extra_data or {})
    master = basis.comm_kpts.rank == 0 and basis.comm_pw.rank == 0
    if ext == ".json":
        if master:
            with open(filename + ".new", "w") as fh:
                json.dump(d, fh)
            os.replace(filename + ".new", filename)
        return
    if basis.comm_kpts.size > 1 or basis.comm_pw.size > 1:
        raise NotImplementedError("npz checkpoints are written by single-rank runs")
    arrays = {"meta_json": np.array(json.dumps(d))}
    if save_rho:
        arrays["rho"] = scfres["rho"].detach().cpu().numpy()
    if save_psi:
        for ik, p in enumerate(scfres["psi"]):
            arrays[f"psi_{ik}"] = p.detach().cpu().numpy()
    tmp = filename + ".new.npz"
    np.savez(tmp, **arrays)
    os.replace(tmp, filename)


def load_scfres(filename: str, basis=None) -> dict:
    """``load_scfres`` (scfres.jl:1-35): the metadata dict plus, with a ``basis``, device tensors ``rho`` / ``psi``
    ready to restart ``self_consistent_field(basis, rho=..., psi=...)``.  The stored and the passed basis must agree
    (FFT size, Ecut, k-points), as the reference demands."""
    import torch
    ext = os.path.splitext(filename)[1].lower()
    if ext == ".json":
        with open(filename) as fh:
            return json.load(fh)
    data = np.load(filename, allow_pickle=False)
    out = json.loads(str(data["meta_json"]))
    if basis is not None:
        if list(basis.fft_size) != out["fft_size"] or abs(basis.Ecut - out["Ecut"]) > 1e-12 \
                or len(basis.kcoords_global) != out["n_kpoints"]:
            raise ValueError("stored and passed basis are inconsistent (fft_size / Ecut / k-points)")
        kc = np.asarray([np.asarray(k, dtype=float) for k in basis.kcoords_global]).reshape(-1, 3)
        if (np.abs(kc - np.asarray(out["kcoords"], dtype=float).reshape(-1, 3)).max(initial=0.0) > 1e-10
                or np.abs(np.asarray(basis.kweights_global, dtype=float)
                          - np.asarray(out["kweights"], dtype=float)).max(initial=0.0) > 1e-10):
            raise ValueError("stored and passed basis are inconsistent (k-point coordinates / weights)")
        if basis.comm_kpts.size > 1 or basis.comm_pw.size > 1:
            raise NotImplementedError("npz checkpoints restart single-rank runs")
        # kpt_n_G_vectors is stored [spin][kpoint]; basis.kpoints lists the spin-up blocks, then the spin-down ones
        if "kpt_n_G_vectors" in out and [int(k.n_G) for k in basis.kpoints] != [int(n) for per_spin in out["kpt_n_G_vectors"]
                                                                                 for n in per_spin]:
            raise ValueError("stored and passed basis are inconsistent (plane waves per k-point / spin components)")
        dev = basis.device
        if "rho" in data:
            out["rho"] = torch.from_numpy(data["rho"]).to(dev)
        # one block per (k-point, spin component): psi_0 .. psi_{n_spin n_k - 1} in the order of basis.kpoints
        psi = [torch.from_numpy(data[f"psi_{ik}"]).to(dev) for ik in range(len(basis.kpoints)) if f"psi_{ik}" in data]
        if psi:
            if len(psi) != len(basis.kpoints):
                raise ValueError(f"checkpoint holds {len(psi)} orbital blocks, the basis has {len(basis.kpoints)} k-blocks")
            out["psi"] = psi
    return out
