"""dftk.jl_amd -- MI355X-native plane-wave Kohn-Sham SCF hot path behind DFTK.jl's operator API.

The directory name contains a dot, so import it through the repo-root shim::

    import dftk_jl_amd as dftk

Layout: ``csrc/`` hand-written HIP kernels + the C ABI (include/dftk_mi355x.h),
``_lib.py`` the ctypes binding, and the host-side mirror of the reference interface
(``PlaneWaveBasis``, ``Kpoint``, ``DftHamiltonianBlock`` + ``mul_``, ``lobpcg_hyper``,
``compute_density``, ``self_consistent_field``).
"""
from ._lib import load as load_library, DftkMiError, EXPORTED_SYMBOLS  # noqa: F401
from ._build import build as build_library  # noqa: F401
from .psp import PspHgh, load_psp, parse_psp_hgh  # noqa: F401,E402
from .model import (Model, ElementPsp, model_DFT, model_atomic, create_supercell, silicon_cell,  # noqa: F401,E402
                    MonkhorstPack, ExplicitKpoints)
from .comm import KptComm, split_evenly, distribute_kpoints  # noqa: F401,E402
from .basis import PlaneWaveBasis, Kpoint, compute_fft_size  # noqa: F401,E402
from .hamiltonian import DftHamiltonianBlock, mul_  # noqa: F401,E402
from .terms import energy_hamiltonian, guess_density  # noqa: F401,E402
from .eigen import (lobpcg_hyper, diagonalize_all_kblocks, PreconditionerTPA, random_orbitals,  # noqa: F401,E402
                    columnwise_norms, columnwise_dots, ortho_qr, interpolate_kpoint, lobpcg_residual_history)
from .densities import compute_density  # noqa: F401,E402
from .mixing import (SimpleMixing, KerkerMixing, KerkerDosMixing, DielectricMixing, LdosMixing, HybridMixing,  # noqa: F401,E402
                     Chi0Mixing, compute_dos, compute_ldos)
from .scf import (self_consistent_field, next_density, compute_occupation, AdaptiveBands, FixedBands,  # noqa: F401,E402
                  AndersonAcceleration, determine_diagtol, ScfDefaultCallback, ScfStepper)
from .io import scfres_to_dict, save_scfres, load_scfres, basis_to_dict, model_to_dict  # noqa: F401,E402
from .symmetry import (SymOp, symmetry_operations, symmetrize_rho, irreducible_kcoords,  # noqa: F401,E402
                       symmetries_preserving_kgrid, symmetries_preserving_rgrid, check_group)
from .memory_usage import estimate_memory_usage, plan_planewave_sharded, MemoryStatistics  # noqa: F401,E402
