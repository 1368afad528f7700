"""HGH/GTH pseudopotentials for the host mirror (src/pseudo/PspHgh.jl, NormConservingPsp.jl:187-234).

Evaluation functions are written with torch so that the O(n_G) / O(N) set-up arrays
(form factors, local potential) are produced directly in HBM.
"""
from __future__ import annotations

import math
import re
from dataclasses import dataclass

import numpy as np
import torch


@dataclass
class PspHgh:
    """``struct PspHgh`` (PspHgh.jl:4-13)."""
    Zion: int
    rloc: float
    cloc: tuple
    lmax: int
    rp: tuple
    h: tuple            # h[l]: (nproj_l, nproj_l) numpy array
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


def _psp(Zion, rloc, cloc, rp, hupper, identifier="", description=""):
    cloc = tuple(list(cloc) + [0.0] * (4 - len(cloc)))
    h = []
    for rows in hupper:
        n = len(rows)
        m = np.zeros((n, n))
        for i, row in enumerate(rows):
            for k, v in enumerate(row):
                m[i, i + k] = m[i + k, i] = v
        h.append(m)
    return PspHgh(int(Zion), float(rloc), cloc, len(h) - 1, tuple(rp), tuple(h), identifier, description)


# Published GTH / HGH parameters (Goedecker, Teter, Hutter 1996; Hartwigsen, Goedecker, Hutter 1998;
# Krack 2005 for PBE): (Zion, rloc, cloc, rp, upper triangles of h per l)
_TABLE = {
    ("Si", "lda"): (4, 0.44, [-7.33610297], [0.42273813, 0.48427842],
                    [[[5.90692831, -1.26189397], [3.25819622]], [[2.72701346]]], "Si GTH-PADE-q4"),
    ("Si", "pbe"): (4, 0.44, [-6.26928833], [0.43563383, 0.49794218],
                    [[[8.95174150, -2.70627082], [3.49378060]], [[2.43127673]]], "Si GTH-PBE-q4"),
    ("Al", "lda"): (3, 0.45, [-8.49135116], [0.46010427, 0.53674439],
                    [[[5.08833953, -1.03784325], [2.67969975]], [[2.19343827]]], "Al GTH-PADE-q3"),
    ("Al", "pbe"): (3, 0.45, [-7.55476126], [0.48743529, 0.56218949],
                    [[[6.95993832, -1.88883584], [2.43847659]], [[1.86529857]]], "Al GTH-PBE-q3"),
    ("C", "lda"): (4, 0.34883045, [-8.51377110, 1.22843203], [0.30455321, 0.23267730],
                   [[[9.52284179]], []], "C GTH-PADE-q4"),
    ("C", "pbe"): (4, 0.33847124, [-8.80367398, 1.33921085], [0.30257575, 0.29150694],
                   [[[9.62248665]], []], "C GTH-PBE-q4"),
    ("Fe", "lda"): (8, 0.61, [], [0.45448200, 0.63890282, 0.30873177],
                    [[[3.01664046, -1.00040646, 0.79478164], [2.58303836, -2.05211737], [3.25763534]],
                     [[1.49964199, -0.13812935], [0.32687369]], [[-9.14535371]]], "Fe GTH-PADE-q8"),
}
ATOMIC_NUMBER = {"H": 1, "C": 6, "Al": 13, "Si": 14, "Fe": 26}


def load_psp(symbol: str, functional: str = "lda") -> PspHgh:
    """``load_psp`` for the vendored HGH family (data/psp/hgh/<functional>/<symbol>-q<Z>.hgh)."""
    Z, rloc, cloc, rp, h, desc = _TABLE[(symbol, functional)]
    return _psp(Z, rloc, cloc, rp, h, f"hgh/{functional}/{symbol.lower()}-q{Z}.hgh", desc)


def parse_psp_hgh(text: str, identifier: str = "") -> PspHgh:
    """Reader of the ``.hgh`` text format (PspHgh.jl:25-94)."""
    lines = text.splitlines()
    Zion = sum(int(p) for p in re.match(r"^ *(([0-9]+ *)+)", lines[1]).group(1).split())
    m = re.match(r"^ *([-.0-9]+) +([0-9]+)( +([-.0-9]+ *)+)? *", lines[2])
    rloc, nloc = float(m.group(1)), int(m.group(2))
    cloc = [float(p) for p in m.group(3).split()] if m.group(3) else []
    if len(cloc) != nloc:
        raise ValueError("malformed HGH file: local coefficients")
    lmax = int(re.match(r"^ *([0-9]+)", lines[3]).group(1)) - 1
    rp, hup, cur = [], [], 4
    for _ in range(lmax + 1):
        m = re.match(r"^ *([-.0-9]+) +([0-9]+)( +([-.0-9]+ *)+)? *", lines[cur])
        rp.append(float(m.group(1)))
        nproj = int(m.group(2))
        rows = []
        if nproj == 0:
            hup.append(rows)
            cur += 1
            continue
        coeff = [float(p) for p in m.group(3).split()]
        for i in range(nproj):
            rows.append(coeff[:nproj - i])
            cur += 1
            if cur >= len(lines):
                break
            m2 = re.match(r"^ *(([-.0-9]+ *)+)", lines[cur])
            if m2 is None:
                break
            coeff = [float(p) for p in m2.group(1).split()]
        hup.append(rows)
    return _psp(Zion, rloc, cloc, rp, hup, identifier, lines[0])


def eval_psp_local_fourier(psp: PspHgh, p: torch.Tensor) -> torch.Tensor:
    """PspHgh.jl:110-124; zero at p == 0 (compensating background)."""
    t2 = (p * psp.rloc) ** 2
    c = psp.cloc
    P = (c[0] + c[1] * (3 - t2) + c[2] * (15 - 10 * t2 + t2 * t2)
         + c[3] * (105 - 105 * t2 + 21 * t2 * t2 - t2 ** 3))
    safe = torch.where(t2 > 0, t2, torch.ones_like(t2))
    val = (4 * math.pi * psp.rloc ** 2 * (-float(psp.Zion) + math.sqrt(math.pi / 2) * psp.rloc * safe * P)
           * torch.exp(-safe / 2) / safe)
    return torch.where(t2 > 0, val, torch.zeros_like(val))


def eval_psp_projector_fourier(psp: PspHgh, i: int, l: int, p: torch.Tensor) -> torch.Tensor:
    """PspHgh.jl:140-164 (i is 1-based; includes the division by p^l)."""
    rp = psp.rp[l]
    t2 = (p * rp) ** 2
    common = 4 * math.pi ** 1.25 * math.sqrt(2 ** (l + 1) * rp ** 3) * torch.exp(-t2 / 2)
    table = {
        (0, 1): lambda: common,
        (0, 2): lambda: common * (2 / math.sqrt(15)) * (3 - t2),
        (0, 3): lambda: common * (4 / (3 * math.sqrt(105))) * (15 - 10 * t2 + t2 * t2),
        (1, 1): lambda: common * (rp / math.sqrt(3)),
        (1, 2): lambda: common * (2 * rp / math.sqrt(105)) * (5 - t2),
        (1, 3): lambda: common * (4 * rp / (3 * math.sqrt(1155))) * (35 - 14 * t2 + t2 * t2),
        (2, 1): lambda: common * (rp ** 2 / math.sqrt(15)),
        (2, 2): lambda: common * (2 * rp ** 2 / (3 * math.sqrt(105))) * (7 - t2),
        (3, 1): lambda: common * (rp ** 3 / math.sqrt(105)),
    }
    if (l, i) not in table:
        raise NotImplementedError(f"HGH projector l={l}, i={i}")
    return table[(l, i)]()


def eval_psp_energy_correction(psp: PspHgh) -> float:
    """PspHgh.jl:173-184."""
    coeffs = (1.0, 3.0, 15.0, 105.0)
    diff = (psp.Zion * psp.rloc ** 2 / 2
            + math.sqrt(math.pi / 2) * psp.rloc ** 3 * sum(a * b for a, b in zip(coeffs, psp.cloc)))
    return 4 * math.pi * diff


def solid_harmonic_real(l: int, m: int, r: torch.Tensor) -> torch.Tensor:
    """r^l Y_lm, real form (src/common/spherical_harmonics.jl:31-66); r has shape (n, 3)."""
    x, y, z = r[:, 0], r[:, 1], r[:, 2]
    pi = math.pi
    if l == 0:
        return torch.full_like(x, math.sqrt(1 / (4 * pi)))
    if l == 1:
        return math.sqrt(3 / (4 * pi)) * {-1: y, 0: z, 1: x}[m]
    if l == 2:
        return {-2: lambda: math.sqrt(15 / (4 * pi)) * x * y,
                -1: lambda: math.sqrt(15 / (4 * pi)) * y * z,
                0: lambda: math.sqrt(5 / (16 * pi)) * (2 * z * z - x * x - y * y),
                1: lambda: math.sqrt(15 / (4 * pi)) * x * z,
                2: lambda: math.sqrt(15 / (16 * pi)) * (x * x - y * y)}[m]()
    if l == 3:
        return {-3: lambda: math.sqrt(35 / (32 * pi)) * (3 * x * x - y * y) * y,
                -2: lambda: math.sqrt(105 / (4 * pi)) * x * y * z,
                -1: lambda: math.sqrt(21 / (32 * pi)) * y * (4 * z * z - x * x - y * y),
                0: lambda: math.sqrt(7 / (16 * pi)) * z * (2 * z * z - 3 * x * x - 3 * y * y),
                1: lambda: math.sqrt(21 / (32 * pi)) * x * (4 * z * z - x * x - y * y),
                2: lambda: math.sqrt(105 / (16 * pi)) * (x * x - y * y) * z,
                3: lambda: math.sqrt(35 / (32 * pi)) * (x * x - 3 * y * y) * x}[m]()
    raise NotImplementedError(f"l={l}")
