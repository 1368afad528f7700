"""HGH/GTH separable dual-space Gaussian pseudopotentials (oracle restatement).

Restates ``src/pseudo/PspHgh.jl`` (parser :25-94, local part :110-124, projectors
:140-164, energy correction :173-184) and the projector counting helpers of
``src/pseudo/NormConservingPsp.jl:187-234``.  Test infrastructure only.

The parameter sets in ``HGH_TABLE`` are the published Goedecker-Teter-Hutter /
Hartwigsen-Goedecker-Hutter values (the same numbers the reference vendors as
``data/psp/hgh/{lda,pbe}/*.hgh``), held here as plain data so that nothing needs
``/root/reference`` at run time.
"""
from __future__ import annotations

import math
import re
from dataclasses import dataclass, field

import numpy as np


@dataclass
class PspHgh:
    """Fields as ``struct PspHgh`` (PspHgh.jl:4-13)."""
    Zion: int
    rloc: float
    cloc: np.ndarray            # 4 coefficients (zero padded)
    lmax: int
    rp: list                    # projector radius per l
    h: list                     # h[l] = (nproj_l x nproj_l) coupling matrix
    identifier: str = ""
    description: str = ""

    def count_n_proj_radial(self, l=None):
        if l is None:
            return sum(self.h[ll].shape[0] for ll in range(self.lmax + 1))
        return self.h[l].shape[0]

    def count_n_proj(self, l=None):
        if l is None:
            return sum(self.count_n_proj(ll) for ll in range(self.lmax + 1))
        return self.count_n_proj_radial(l) * (2 * l + 1)


def make_psp(Zion, rloc, cloc, rp, h, identifier="", description=""):
    """``PspHgh(Zion, rloc, cloc, rp, h)`` (PspHgh.jl:96-107)."""
    cloc = list(cloc) + [0.0] * (4 - len(cloc))
    h = [np.array(hl, dtype=float).reshape(len(hl), len(hl)) if len(hl) else np.zeros((0, 0))
         for hl in h]
    return PspHgh(int(Zion), float(rloc), np.array(cloc, dtype=float), len(h) - 1,
                  [float(r) for r in rp], h, identifier, description)


def parse_hgh(text: str, identifier="") -> PspHgh:
    """Parser for the ABINIT/cp2k style ``.hgh`` text format (PspHgh.jl:25-94)."""
    lines = text.splitlines()
    description = lines[0]
    n_elec = [int(p) for p in re.match(r"^ *(([0-9]+ *)+)", lines[1]).group(1).split()]
    Zion = sum(n_elec)
    m = re.match(r"^ *([-.0-9]+) +([0-9]+)( +([-.0-9]+ *)+)? *", lines[2])
    rloc = float(m.group(1))
    nloc = int(m.group(2))
    cloc = [float(p) for p in m.group(3).split()] if m.group(3) else []
    assert len(cloc) == nloc
    lmax = int(re.match(r"^ *([0-9]+)", lines[3]).group(1)) - 1
    rp, h = [], []
    cur = 4
    for _l in range(lmax + 1):
        m = re.match(r"^ *([-.0-9]+) +([0-9]+)( +([-.0-9]+ *)+)? *", lines[cur])
        rp.append(float(m.group(1)))
        nproj = int(m.group(2))
        hl = np.zeros((nproj, nproj))
        if nproj == 0:
            h.append(hl)
            cur += 1
            continue
        hcoeff = [float(p) for p in m.group(3).split()]
        for i in range(nproj):
            for j in range(i, nproj):
                hl[j, i] = hl[i, j] = hcoeff[j - i]
            cur += 1
            if cur >= len(lines):
                break
            m2 = re.match(r"^ *(([-.0-9]+ *)+)", lines[cur])
            if m2 is None:
                break
            hcoeff = [float(p) for p in m2.group(1).split()]
        h.append(hl)
    return make_psp(Zion, rloc, cloc, rp, h, identifier, description)


# Published GTH/HGH parameter sets: key -> (Zion, rloc, cloc, rp, h upper triangles as matrices)
def _sym(*rows):
    n = len(rows)
    m = np.zeros((n, n))
    for i, row in enumerate(rows):
        for k, v in enumerate(row):
            m[i, i + k] = m[i + k, i] = v
    return m


HGH_TABLE = {
    # Si GTH-PADE-q4 (LDA)
    ("Si", "lda"): dict(Zion=4, rloc=0.44, cloc=[-7.33610297], rp=[0.42273813, 0.48427842],
                        h=[_sym([5.90692831, -1.26189397], [3.25819622]), _sym([2.72701346])],
                        description="Si GTH-PADE-q4 GTH-LDA-q4"),
    # Si GTH-PBE-q4
    ("Si", "pbe"): dict(Zion=4, rloc=0.44, cloc=[-6.26928833], rp=[0.43563383, 0.49794218],
                        h=[_sym([8.95174150, -2.70627082], [3.49378060]), _sym([2.43127673])],
                        description="Si GTH-PBE-q4"),
    # Al GTH-PADE-q3 (LDA)
    ("Al", "lda"): dict(Zion=3, rloc=0.45, cloc=[-8.49135116], rp=[0.46010427, 0.53674439],
                        h=[_sym([5.08833953, -1.03784325], [2.67969975]), _sym([2.19343827])],
                        description="Al GTH-PADE-q3 GTH-LDA-q3"),
    # Al GTH-PBE-q3
    ("Al", "pbe"): dict(Zion=3, rloc=0.45, cloc=[-7.55476126], rp=[0.48743529, 0.56218949],
                        h=[_sym([6.95993832, -1.88883584], [2.43847659]), _sym([1.86529857])],
                        description="Al GTH-PBE-q3"),
    # C GTH-PADE-q4 (LDA)
    ("C", "lda"): dict(Zion=4, rloc=0.34883045, cloc=[-8.51377110, 1.22843203],
                       rp=[0.30455321, 0.23267730],
                       h=[_sym([9.52284179]), np.zeros((0, 0))],
                       description="C GTH-PADE-q4 GTH-LDA-q4"),
    # C GTH-PBE-q4
    ("C", "pbe"): dict(Zion=4, rloc=0.33847124, cloc=[-8.80367398, 1.33921085],
                       rp=[0.30257575, 0.29150694],
                       h=[_sym([9.62248665]), np.zeros((0, 0))],
                       description="C GTH-PBE-q4"),
    # Fe GTH-PADE-q8 (LDA): s, p, d channels with 3, 2, 1 radial projectors
    ("Fe", "lda"): dict(Zion=8, rloc=0.61, cloc=[], rp=[0.45448200, 0.63890282, 0.30873177],
                        h=[_sym([3.01664046, -1.00040646, 0.79478164], [2.58303836, -2.05211737], [3.25763534]),
                           _sym([1.49964199, -0.13812935], [0.32687369]), _sym([-9.14535371])],
                        description="Fe GTH-PADE-q8 GTH-LDA-q8"),
}

ATOMIC_NUMBER = {"H": 1, "C": 6, "Al": 13, "Si": 14, "Fe": 26}


def load_psp_hgh(symbol: str, functional: str = "lda") -> PspHgh:
    d = HGH_TABLE[(symbol, functional)]
    return make_psp(d["Zion"], d["rloc"], d["cloc"], d["rp"], d["h"],
                    identifier=f"hgh/{functional}/{symbol.lower()}-q{d['Zion']}",
                    description=d["description"])


def eval_psp_local_fourier(psp: PspHgh, p):
    """V_loc(p) = int V_loc(r) e^{-ip.r} dr, zero at p == 0 (PspHgh.jl:110-124)."""
    p = np.asarray(p, dtype=float)
    out = np.zeros_like(p)
    nz = p != 0
    pp = p[nz]
    rloc, Zion, c = psp.rloc, float(psp.Zion), psp.cloc
    t = pp * rloc
    t2 = t * t
    P = (c[0] + c[1] * (3 - t2) + c[2] * (15 - 10 * t2 + t2 * t2)
         + c[3] * (105 - 105 * t2 + 21 * t2 * t2 - t2 ** 3))
    out[nz] = (4 * math.pi * rloc ** 2 * (-Zion + math.sqrt(math.pi / 2) * rloc * t2 * P)
               * np.exp(-t2 / 2) / t2)
    return out


def eval_psp_projector_fourier(psp: PspHgh, i: int, l: int, p):
    """Radial projector in Fourier space divided by p^l (PspHgh.jl:140-164); i is 1-based."""
    p = np.asarray(p, dtype=float)
    rp = psp.rp[l]
    t = p * rp
    t2 = t * t
    common = 4 * math.pi ** 1.25 * math.sqrt(2 ** (l + 1) * rp ** 3) * np.exp(-t2 / 2)
    if l == 0:
        if i == 1:
            return common
        if i == 2:
            return common * 2 / math.sqrt(15) * (3 - t2)
        if i == 3:
            return common * 4 / (3 * math.sqrt(105)) * (15 - 10 * t2 + t2 * t2)
    if l == 1:
        if i == 1:
            return common * 1 / math.sqrt(3) * rp
        if i == 2:
            return common * 2 / math.sqrt(105) * rp * (5 - t2)
        if i == 3:
            return common * 4 / (3 * math.sqrt(1155)) * rp * (35 - 14 * t2 + t2 * t2)
    if l == 2:
        if i == 1:
            return common * 1 / math.sqrt(15) * rp ** 2
        if i == 2:
            return common * 2 / (3 * math.sqrt(105)) * rp ** 2 * (7 - t2)
    if l == 3 and i == 1:
        return common * 1 / math.sqrt(105) * rp ** 3
    raise NotImplementedError(f"l={l} i={i}")


def eval_psp_energy_correction(psp: PspHgh) -> float:
    """DC part of (Coulomb - V_loc) (PspHgh.jl:173-184)."""
    coeffs = np.array([1.0, 3.0, 15.0, 105.0])
    diff = psp.Zion * psp.rloc ** 2 / 2 + math.sqrt(math.pi / 2) * psp.rloc ** 3 * float(
        np.sum(coeffs * psp.cloc))
    return 4 * math.pi * diff


def solid_harmonic_real(l: int, m: int, r):
    """Real solid harmonics r^l Y_lm (src/common/spherical_harmonics.jl:31-66); r is (n,3)."""
    r = np.asarray(r, dtype=float)
    x, y, z = r[..., 0], r[..., 1], r[..., 2]
    pi = math.pi
    if l == 0:
        return np.full(x.shape, math.sqrt(1 / (4 * pi)))
    if l == 1:
        return math.sqrt(3 / (4 * pi)) * {-1: y, 0: z, 1: x}[m]
    if l == 2:
        if m == -2:
            return math.sqrt(15 / (4 * pi)) * x * y
        if m == -1:
            return math.sqrt(15 / (4 * pi)) * y * z
        if m == 0:
            return math.sqrt(5 / (16 * pi)) * (2 * z * z - x * x - y * y)
        if m == 1:
            return math.sqrt(15 / (4 * pi)) * x * z
        if m == 2:
            return math.sqrt(15 / (16 * pi)) * (x * x - y * y)
    if l == 3:
        if m == -3:
            return math.sqrt(35 / (32 * pi)) * (3 * x * x - y * y) * y
        if m == -2:
            return math.sqrt(105 / (4 * pi)) * x * y * z
        if m == -1:
            return math.sqrt(21 / (32 * pi)) * y * (4 * z * z - x * x - y * y)
        if m == 0:
            return math.sqrt(7 / (16 * pi)) * z * (2 * z * z - 3 * x * x - 3 * y * y)
        if m == 1:
            return math.sqrt(21 / (32 * pi)) * x * (4 * z * z - x * x - y * y)
        if m == 2:
            return math.sqrt(105 / (16 * pi)) * (x * x - y * y) * z
        if m == 3:
            return math.sqrt(35 / (32 * pi)) * (x * x - 3 * y * y) * x
    raise NotImplementedError(f"l={l} m={m}")
