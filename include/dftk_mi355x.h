/* dftk_mi355x.h -- C ABI of the MI355X-native plane-wave Kohn-Sham hot path.
 *
 * Drop-in boundary for DFTK.jl's SCF hot path.  The reference has no
 * FFI; its seams are Julia-level (the `architecture` kwarg, `HamiltonianBlock` + `mul!`, the
 * `eigensolver=` kwarg of `self_consistent_field`, `compute_density`).  Each entry point below
 * names the reference function it replaces (file:line under the DFTK.jl checkout); the Julia
 * `ccall` shim a maintainer would add is in INTEGRATION.md, the Python (ctypes) binding used
 * by this repo's host mirror is dftk.jl_amd/_lib.py.
 *
 * Conventions
 *  - complex = interleaved (re, im) IEEE fp64 (`dftk_mi_cplx`, 16 B); all arithmetic is fp64.
 *  - matrices are column-major with an explicit leading dimension counted in elements;
 *    orbital blocks psi are n_G x n_bands (one column per band), as in DFTK.
 *  - cubes are (nx, ny, nz) with x fastest: linear index i = ix + nx*(iy + ny*iz) -- exactly
 *    Julia's column-major linear index minus one.  `mapping0` is that 0-based index.
 *  - pointers suffixed `_h` are host pointers, `_d` device pointers (same device as the basis).
 *    Device buffers are owned by the caller; the library never frees them and only keeps the
 *    pointers explicitly documented as "borrowed".
 *  - every call returns an int status: 0 ok; <0 invalid argument / runtime (HIP, RCCL) error;
 *    >0 numerical failure (non-finite values, Cholesky breakdown, eigen-solver not converged).
 *    `dftk_mi_last_error()` gives a thread-local message for the last non-zero status.
 *  - calls are asynchronous on the basis' HIP stream unless they return host data
 *    (`dftk_mi_lobpcg`); `dftk_mi_basis_sync` blocks.
 *  - no call falls back to the CPU: a missing GPU is an error (status -100).
 */
#ifndef DFTK_MI355X_H
#define DFTK_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { double re, im; } dftk_mi_cplx;

typedef struct dftk_mi_basis  dftk_mi_basis;   /* FFTGrid + device/stream          */
typedef struct dftk_mi_kblock dftk_mi_kblock;  /* Kpoint + DftHamiltonianBlock     */
typedef struct dftk_mi_comm   dftk_mi_comm;    /* comm_kpts (RCCL communicator)    */

/* ---- status codes -------------------------------------------------------------------------- */
#define DFTK_MI_OK                 0
#define DFTK_MI_EINVAL            (-1)
#define DFTK_MI_EHIP              (-2)
#define DFTK_MI_ERCCL             (-3)
#define DFTK_MI_ENOGPU            (-100)
#define DFTK_MI_NUM_NONFINITE       1   /* NaN/Inf met (reference: @assert !any(isnan, AX))      */
#define DFTK_MI_NUM_CHOLESKY        2   /* ortho!: Cholesky kept failing (reference: SVD fallback)*/
#define DFTK_MI_NUM_NORMALIZATION   3   /* "LOBPCG is badly failing to keep the vectors normalized"*/
#define DFTK_MI_NUM_EIGEN           4   /* dense Hermitian eigensolver did not converge          */
#define DFTK_MI_NUM_TOO_SMALL       5   /* N > 3M violated (lobpcg_hyper_impl.jl:363)            */

const char* dftk_mi_last_error(void);
/* Library / build identification: "dftk_mi355x <version> gfx950 ..." */
const char* dftk_mi_version(void);

/* ---- basis: replaces FFTGrid(fft_size, unit_cell_volume, arch)  (src/fft.jl:76-98) -----------
 * Holds the 3 one-dimensional mixed-radix plans (radices 2,3,4,5 + generic primes), twiddle
 * tables, the HIP stream and the scratch pool.  `device` is the HIP device ordinal.          */
int dftk_mi_basis_create(int nx, int ny, int nz, double unit_cell_volume, int device,
                         dftk_mi_basis** basis_out);
int dftk_mi_basis_destroy(dftk_mi_basis* basis);
int dftk_mi_basis_sync(dftk_mi_basis* basis);              /* synchronize_device (architecture.jl) */
/* Bands processed per FFT launch group (scratch = n * (T1 + T2) bytes); default 8. */
int dftk_mi_basis_set_fft_batch(dftk_mi_basis* basis, int n_bands_per_batch);
/* HIP stream used by this basis (hipStream_t as void*) -- lets the caller order its own work. */
void* dftk_mi_basis_stream(dftk_mi_basis* basis);

/* ---- k-block: replaces Kpoint(...) (src/Kpoint.jl:20-41) + DftHamiltonianBlock
 *      (src/terms/Hamiltonian.jl:22-57).  `mapping0_h` must be ascending (as Kpoint builds it). */
int dftk_mi_kblock_create(dftk_mi_basis* basis, int64_t n_G, const int64_t* mapping0_h,
                          const double* kinetic_h /* n_G: fourier_op.multiplier, kinetic.jl:31-35 */,
                          dftk_mi_kblock** kb_out);
int dftk_mi_kblock_destroy(dftk_mi_kblock* kb);
/* nonlocal_op (src/terms/operators.jl:119-129): P is n_G x n_p (borrowed device pointer, must
 * outlive the k-block or the next call), D is the dense real n_p x n_p coupling matrix
 * (nonlocal.jl:107-141) on the host; its band structure is detected and exploited. n_p = 0 clears. */
int dftk_mi_kblock_set_projectors(dftk_mi_kblock* kb, int n_p, const dftk_mi_cplx* P_d, int64_t ldP,
                                  const double* D_h);
/* local_op.potential (operators.jl:71-78, :213-222): the SUM of all local terms on the cube,
 * un-normalised (the 1/N of Hamiltonian.jl:152-153 is applied inside).  Copied. NULL clears. */
int dftk_mi_kblock_set_potential(dftk_mi_kblock* kb, const double* V_d);
/* The same potential for n k-blocks (all k-points of a basis apply ONE summed local potential, Hamiltonian.jl:36-57):
 * one call instead of one host round trip per k-block.  V_d must stay unchanged until the blocks' streams have run. */
int dftk_mi_kblocks_set_potential(int n_kblocks, dftk_mi_kblock* const* kbs, const double* V_d);

/* ---- mul!(Hpsi, H::DftHamiltonianBlock, psi)  (src/terms/Hamiltonian.jl:137-192) ------------- */
int dftk_mi_apply_H(dftk_mi_kblock* kb, int n_bands, const dftk_mi_cplx* psi_d, int64_t ld_psi,
                    dftk_mi_cplx* Hpsi_d, int64_t ld_Hpsi);
/* Pieces, for tests and profiling: which = 1 local (K1-K5: scatter, iFFT, V, FFT, gather),
 * 2 kinetic (K6), 4 nonlocal (K7-K9); OR-able.  dftk_mi_apply_H == which 7. */
int dftk_mi_apply_H_parts(dftk_mi_kblock* kb, int which, int n_bands, const dftk_mi_cplx* psi_d,
                          int64_t ld_psi, dftk_mi_cplx* Hpsi_d, int64_t ld_Hpsi);

/* ---- basis / term set-up ------------------------------------------------------------------
 * Kpoint(...) sphere enumeration (src/Kpoint.jl:20-41): all cube points with |B (G + k)|^2 / 2 <= Ecut in ascending
 * 0-based x-fastest linear index, their kinetic multipliers (src/terms/kinetic.jl:31-35) and integer G vectors
 * (3 per plane wave).  recip_lattice_h: 3x3 column-major (Julia's `model.recip_lattice`).  Host routine (no GPU
 * needed): call with cap = 0 / NULL buffers to get *n_G, then with buffers of that size. */
int dftk_mi_kpoint_sphere_host(int nx, int ny, int nz, const double* recip_lattice_h, const double* kcoord_h,
                               double Ecut, int64_t cap, int64_t* n_G, int64_t* mapping0_h, double* kinetic_h,
                               int32_t* G_h);
/* build_projection_vectors (src/terms/nonlocal.jl:166-244) for HGH pseudopotentials, written on the device:
 * P[g, c] = f_{l,i}(|q|) Y_lm(q) (-i)^l / sqrt(Omega) exp(-2 pi i (G + k).r_atom), q = B (G + k); columns atom by atom
 * in the order given, within an atom (l, m, i) as nonlocal.jl:228-229.  G_d: 3 ints per row (the rows of this rank's
 * slab for a sharded block); rp_h / n_proj_h: 4 entries (l = 0..3) per species: r_l and the number of radial
 * projectors; positions_h: 3 reduced coordinates per atom.  *n_p = number of columns (P_d may be NULL to query). */
int dftk_mi_build_projectors_hgh(dftk_mi_basis* basis, int64_t n_rows, const int32_t* G_d,
                                 const double* recip_lattice_h, const double* kcoord_h, double unit_cell_volume,
                                 int n_species, const double* rp_h, const int* n_proj_h, int n_atoms,
                                 const int* species_of_atom_h, const double* positions_h, dftk_mi_cplx* P_d,
                                 int64_t ldP, int* n_p);

/* Superposition of atomic form factors on the cube, real-space result:
 *   out(r) = irfft( enforce_real( sum_species ff_s(|G|) sum_{a in s} e^{-2 pi i G.r_a} / sqrt(Omega) ) )
 * kind 0: compute_local_potential (src/terms/local.jl:108-138) with the HGH local form factor
 *         (src/pseudo/PspHgh.jl:110-124); params_h[8 s + 0..5] = rloc, Zion, c1..c4
 * kind 1: Gaussian valence-density superposition of guess_density (src/density_methods.jl:111-125,158-181,236-244,
 *         un-normalised); params_h[8 s + 0..1] = decay length, valence charge.
 * Atoms grouped by species (species_of_atom_h non-decreasing); cube_kb spans the whole cube. */
int dftk_mi_atomic_superposition(dftk_mi_kblock* cube_kb, int kind, const double* recip_lattice_h, int n_species,
                                 const double* params_h, int n_atoms, const int* species_of_atom_h,
                                 const double* positions_h, double* out_d);

/* ---- local-potential pipeline of energy_hamiltonian (src/terms/Hamiltonian.jl:200-227) on the cube ----------
 * Hartree (src/terms/hartree.jl:50-59: V_H = irfft(green .* fft(rho)), E_H = 1/2 Re<V_H(G), rho(G)>), LDA exchange-
 * correlation (src/terms/xc.jl:84-160 with lda_x / lda_c_vwn / lda_c_pw, the functionals of `LDA()` and of the
 * reference's pinned tests), the local pseudopotential energy (src/terms/local.jl:15-16) and the summation into ONE
 * potential (src/terms/operators.jl:213-222):  V_out = V_loc + V_H + v_xc.
 * cube_kb: a k-block whose mapping is 0 .. N-1 (the library's cube FFT).  rho_d, V_loc_d, poisson_green_d
 * (4 pi / |G|^2 with the G = 0 and unpaired-G entries zeroed, hartree.jl:29-45), V_out_d: real cubes on the device;
 * V_loc_d / poisson_green_d / V_out_d may be NULL (term absent / energies only).  energies_h[3] = Hartree, Xc,
 * AtomicLocal (host); the call then synchronises the basis' stream.  energies_h may be NULL when V_out_d is not (the
 * Hamiltonian of an SCF step needs the potential only: no fetch, the call is asynchronous); the same holds for the
 * _collinear and _gga entries below. */
#define DFTK_MI_XC_LDA_X     1
#define DFTK_MI_XC_LDA_C_VWN 2
#define DFTK_MI_XC_LDA_C_PW  4
#define DFTK_MI_XC_LDA_XC_TETER93 32
int dftk_mi_local_potential(dftk_mi_kblock* cube_kb, const double* rho_d, const double* V_loc_d,
                            const double* poisson_green_d, int xc_functionals, double* V_out_d, double* energies_h);

/* Collinear spin (model.spin_polarization == :collinear, src/Model.jl:29-39): rho_d and V_out_d hold TWO cubes each,
 * (up, down) = rho[:, :, :, 1:2] of the reference.  V_out[s] = V_loc + V_H[rho_up + rho_down] + v_xc,s(rho_up, rho_down)
 * -- the potential of the k-blocks of spin s (ene_ops picks Vxc[:, :, :, kpt.spin], src/terms/xc.jl:163-175); the energies
 * are those of dftk_mi_local_potential with the total density in the Hartree and local terms.  Spin-polarised closed forms:
 * DFTK_MI_XC_LDA_X, DFTK_MI_XC_LDA_C_PW, DFTK_MI_XC_LDA_XC_TETER93 (anything else: DFTK_MI_EINVAL). */
int dftk_mi_local_potential_collinear(dftk_mi_kblock* cube_kb, const double* rho_d, const double* V_loc_d,
                                      const double* poisson_green_d, int xc_functionals, double* V_out_d,
                                      double* energies_h);

/* The same pipeline with GGA functionals (PBE() = gga_x_pbe + gga_c_pbe, DFTK_MI_XC_GGA_* bits, may be mixed with the
 * LDA bits): the density gradient and the divergence term of the potential (LibxcDensities src/terms/xc.jl:356-409,
 * xc_potential_real :140-150, divergence_real :576-584) are taken with i G multipliers between the library's cube FFTs,
 *   grad rho = irfft(i G_a fft(rho)),  v_xc = de/drho - 2 div(de/dsigma grad rho),  sigma = |grad rho|^2,
 * G cartesian from recip_lattice_h (3x3 column-major, Julia's model.recip_lattice; required with a GGA bit).  Points
 * with rho <= density_threshold contribute nothing to the GGA part.  Everything else as dftk_mi_local_potential. */
int dftk_mi_local_potential_gga(dftk_mi_kblock* cube_kb, const double* recip_lattice_h, const double* rho_d,
                                const double* V_loc_d, const double* poisson_green_d, int xc_functionals,
                                double density_threshold, double* V_out_d, double* energies_h);

/* ---- density-sized operations of the SCF glue ----------------------------------------------
 * symmetrize_rho(basis, rho; symmetries, do_lowpass) for one spin component (src/symmetry.jl:346-357): fft, then
 * accumulate_over_symmetries! (:282-319: out(G) = sum_s e^{-2 pi i G.tau_s} in(S_s^-1 G), terms whose S_s^-1 G leaves
 * the grid dropped) and lowpass_for_symmetry! (:323-343) as ONE kernel over G, / n_sym, irfft.  S_h: n_sym 3x3 integer
 * matrices, column-major (Julia's symop.S = W'); tau_h: 3 doubles each (symop.tau = -W^-1 w).  rho_out_d may alias
 * rho_in_d.  With identity-only symmetries the density is copied (all(isone, symmetries), :292-295). */
int dftk_mi_symmetrize_rho(dftk_mi_kblock* cube_kb, int n_sym, const int32_t* S_h, const double* tau_h, int do_lowpass,
                           const double* rho_in_d, double* rho_out_d);
/* mix_density(::KerkerMixing, basis, dF) (src/scf/mixing.jl:61-72, spin-unpolarised):
 *   d_rho = irfft(enforce_real!(G^2 / (kTF^2 + G^2) fft(dF))),  d_rho .+= mean(dF) - mean(d_rho)
 * (the last line = the G = 0 coefficient of dF kept as it is).  Multiplier evaluated on the fly from the integer G of
 * each cube entry and recip_lattice_h (column-major). */
int dftk_mi_mix_kerker(dftk_mi_kblock* cube_kb, const double* recip_lattice_h, double kTF, const double* dF_d,
                       double* drho_d);
/* mix_density(::DielectricMixing, ...) (mixing.jl:161-171): multiplier (kTF^2 - C0 G^2) / (eps_r kTF^2 - C0 G^2),
 * C0 = 1 - eps_r, DC component kept.  (eps_r = 1 and eps_r > 1/sqrt(eps) are the caller's special cases, :163-164.) */
int dftk_mi_mix_dielectric(dftk_mi_kblock* cube_kb, const double* recip_lattice_h, double kTF, double eps_r,
                           const double* dF_d, double* drho_d);
/* apply!(d_rho, ::DielectricModel, dV) kernel part (src/scf/chi0models.jl:66-77, identity localisation):
 *   out = irfft(C0 kTF^2 G^2 / (4 pi (kTF^2 - C0 G^2)) fft(dV)) */
int dftk_mi_chi0_dielectric_apply(dftk_mi_kblock* cube_kb, const double* recip_lattice_h, double kTF, double eps_r,
                                  const double* dV_d, double* out_d);
/* out = irfft(multiplier .* fft(f)) with a real multiplier cube on the device, e.g. apply_kernel(::TermHartree, ...)
 * with the Poisson Green's function (src/terms/hartree.jl:68-81) inside the chi0-mixing GMRES (mixing.jl:255-275). */
int dftk_mi_cube_fourier_filter(dftk_mi_kblock* cube_kb, const double* multiplier_d, const double* f_d, double* out_d);

/* GGA exchange-correlation point by point (the libxc call of src/terms/xc.jl:111 for PBE(): gga_x_pbe + gga_c_pbe):
 * e_d = energy density per volume, vrho_d = de/drho, vsigma_d = de/dsigma (sigma = |grad rho|^2) for n grid points;
 * derivatives by forward-mode differentiation of the closed forms on the device.  Points with rho <= density_threshold
 * give zeros.  The gradient / divergence of xc.jl:356-409,576-584 stay with the caller (cube FFTs). */
#define DFTK_MI_XC_GGA_X_PBE 8
#define DFTK_MI_XC_GGA_C_PBE 16
int dftk_mi_xc_gga(dftk_mi_basis* basis, int64_t n, const double* rho_d, const double* sigma_d, int xc_functionals,
                   double density_threshold, double* e_d, double* vrho_d, double* vsigma_d);

/* ---- sphere <-> cube transforms  (src/fft.jl:110-122 ifft!, :162-172 fft!; normalize=false) --
 * cube_d is nx*ny*nz complex, x fastest.  Test/diagnostic entry points (the hot path never
 * materialises the full cube in the caller's layout). */
int dftk_mi_ifft_sphere(dftk_mi_kblock* kb, const dftk_mi_cplx* c_d, dftk_mi_cplx* cube_d);
int dftk_mi_fft_sphere(dftk_mi_kblock* kb, const dftk_mi_cplx* cube_d, dftk_mi_cplx* c_d);

/* ---- compute_density inner loop (src/densities.jl:35-43) -------------------------------------
 * rho_d[nx*ny*nz] += sum_n weight_h[n] * |BFFT(pad(psi[:,n]))|^2, weight_h[n] =
 * occupation[n] * kweight * ifft_normalization^2 (bands with weight 0 are skipped). */
int dftk_mi_density_accumulate(dftk_mi_kblock* kb, int n_bands, const dftk_mi_cplx* psi_d,
                               int64_t ld_psi, const double* weight_h, double* rho_d);

/* The same with the spin index of the k-block spelled out: rho_d holds n_spin cubes (n_spin = 1 or 2,
 * model.n_spin_components), the bands of this block are added to cube `spin` (0-based: kpt.spin - 1) --
 * `rho[:, :, :, kpt.spin] .+= ...` of src/densities.jl:39; a collinear basis lists all spin-up k-blocks, then all
 * spin-down ones (src/PlaneWaveBasis.jl:50-53, build_kpoints src/Kpoint.jl:58-74). */
int dftk_mi_density_accumulate_spin(dftk_mi_kblock* kb, int n_bands, const dftk_mi_cplx* psi_d, int64_t ld_psi,
                                    const double* weight_h, double* rho_d, int spin, int n_spin);

/* ---- lobpcg_hyper(A, X0; prec=PreconditionerTPA, tol, miniter, maxiter, n_conv_check)
 *      (src/eigen/diag_lobpcg_hyper.jl:5-18 -> LOBPCG, src/eigen/lobpcg_hyper_impl.jl:354-582;
 *       PreconditionerTPA src/eigen/preconditioners.jl:27-78) ----------------------------------
 * X_d: in = guess (need not be orthonormal), out = Ritz vectors (orthonormal), n_G x M.
 * use_tpa: 1 = TPA preconditioner from the block's kinetic vector, 0 = none.
 * n_conv_check <= 0 means M.  seed drives the (rare) re-randomisation of null columns.
 * Outputs (host): lambda_h[M] ascending, resid_h[M] final residual norms, *n_iter, *converged,
 * *n_matvec (the "H psi applies" counter, lobpcg_hyper_impl.jl:377,417). */
int dftk_mi_lobpcg(dftk_mi_kblock* kb, int M, dftk_mi_cplx* X_d, int64_t ldX, double tol,
                   int miniter, int maxiter, int n_conv_check, int use_tpa, uint64_t seed,
                   double* lambda_h, double* resid_h, int* n_iter, int* converged,
                   int64_t* n_matvec);
/* Residual-norm history of the last dftk_mi_lobpcg call on this block (`resid_history` of
 * lobpcg_hyper_impl.jl:368,443-446, rows ordered like the returned eigenpairs): hist_h[i + M * it] for
 * it = 0 .. n_iter; cap = capacity of hist_h in doubles (hist_h may be NULL to query the sizes).
 * *n_svd = how many times ortho! took its SVD fallback (:226-231, :307-314) during the call. */
/* The loop over k-points of diagonalize_all_kblocks (src/eigen/diag.jl:24-48) in ONE call, for workloads of many small
 * k-blocks (k-point meshes of small cells: n_G ~ 1e3, a handful of bands -- launch-latency bound one at a time).  Every
 * k-block runs exactly the iteration of dftk_mi_lobpcg, but in lock-step with its siblings: device operations of the
 * same kind are merged into one launch over all k-blocks (batched small GEMMs / Cholesky / Jacobi kernels, one FFT
 * pipeline over the bands of all k-points), and the host waits once per round instead of once per k-block and
 * operation.  All k-blocks must belong to ONE basis handle.  Arrays of n_kblocks entries; lambda_h / resid_h hold M
 * values per k-block; status[i] = what dftk_mi_lobpcg would have returned for k-block i.  Same eigenpairs as n_kblocks
 * separate calls up to the round-off of the different summation orders. */
int dftk_mi_lobpcg_multi(int n_kblocks, dftk_mi_kblock* const* kbs, int M, dftk_mi_cplx* const* X_d, const int64_t* ldX,
                         double tol, int miniter, int maxiter, int n_conv_check, int use_tpa, const uint64_t* seeds,
                         double* lambda_h, double* resid_h, int* n_iter, int* converged, int64_t* n_matvec, int* status);
/* compute_density's loop over k-points (src/densities.jl:35-43) in ONE call: rho_d += sum_k sum_n weight[k][n]
 * |ifft(psi_k[:, n])|^2 with all bands of all k-blocks in one pipeline (bands of zero weight never enter it).
 * weights_h: the per-band weights (occupation * kweight * ifft_normalization^2) of k-block 0, then of k-block 1, ...
 * One basis handle, unsharded k-blocks.  Asynchronous like dftk_mi_density_accumulate up to the final round. */
int dftk_mi_density_accumulate_multi(int n_kblocks, dftk_mi_kblock* const* kbs, const int* n_bands,
                                     const dftk_mi_cplx* const* psi_d, const int64_t* ld_psi, const double* weights_h,
                                     double* rho_d);
/* The same with a SECOND weight set and a second cube accumulated in the same pass: rho2_d += sum_k sum_n weights2[k][n]
 * |ifft psi_kn|^2 -- compute_density and compute_ldos (src/postprocess/dos.jl:43-62: "compute_density with modified weights")
 * of an SCF step with LdosMixing transform every band once instead of twice.  weights2_h / rho2_d both NULL = the call above. */
int dftk_mi_density_accumulate_multi2(int n_kblocks, dftk_mi_kblock* const* kbs, const int* n_bands,
                                      const dftk_mi_cplx* const* psi_d, const int64_t* ld_psi, const double* weights_h,
                                      double* rho_d, const double* weights2_h, double* rho2_d);
/* Kinetic energy of every band of n k-blocks in one call: out_h = sum_G kin_G |psi_Gn|^2 for the bands of k-block 0,
 * then of k-block 1, ... (the band terms of the Kinetic energy, src/terms/kinetic.jl:49-54).  One basis handle. */
int dftk_mi_band_kinetic_multi(int n_kblocks, dftk_mi_kblock* const* kbs, const int* n_bands,
                               const dftk_mi_cplx* const* psi_d, const int64_t* ld_psi, double* out_h);
/* Counters of the calling thread's last batched call: scheduling rounds (= host synchronisations), recorded
 * operations, merged launches, operations that ran one by one (no batched form). */
int dftk_mi_batch_stats(int64_t* rounds, int64_t* ops, int64_t* merged_launches, int64_t* sequential_ops);
/* Small k-blocks (M <= 8 bands, n_G * M <= 65536, neither sharded nor Gamma-real) run a driver with ONE host
 * synchronisation per LOBPCG iteration: ortho!(X) / ortho!(X, Y) are one kernel each, Ritz values and statuses travel
 * with the residual norms (lobpcg.cpp: lobpcg_run_small; src/eigen/lobpcg_hyper_impl.jl:354-582 unchanged as an
 * algorithm).  Rare branches it only detects (drop_small!, SVD fallbacks) restart the call on the general driver.
 * Process-wide counters: calls that took the small-block driver, and how many of them restarted.
 * DFTK_MI_LOBPCG_SMALL=0 switches the driver off. */
int dftk_mi_lobpcg_small_stats(int64_t* calls, int64_t* restarts);
/* The fused orthogonalisation kernel of that driver on a stand-alone block: ortho!(X, Y) for ny > 0 (X is normalised by
 * norms_d -- its column norms on the device -- or by norms computed here when NULL, projected against the orthonormal Y
 * and orthonormalised, lobpcg_hyper_impl.jl:271-323), plain ortho!(X) for ny = 0 (:216-261); m <= 8, ny <= 16.
 * res4_h = {status (0 done, 1 a host-side branch is needed: X unusable, 2 non-finite), rounds of the ortho!(X, Y) loop,
 * Cholesky factorisations of the last ortho!(X), growth factor}. */
int dftk_mi_ortho_small(dftk_mi_basis* basis, int64_t n, int m, dftk_mi_cplx* X_d, int64_t ldx, int ny, const dftk_mi_cplx* Y_d,
                        int64_t ldy, const double* norms_d, double tol, double* res4_h);
int dftk_mi_lobpcg_history(dftk_mi_kblock* kb, int* M, int* n_iter, double* hist_h, size_t cap, int* n_svd);
/* One-shot promise for the NEXT dftk_mi_lobpcg call on this block: its X0 IS the X the last call returned (an SCF step
 * hands the orbitals of the previous step back; same number of bands).  The driver then starts from
 * A_new X = (A_old X) inv(R) + (V_new - V_old) X -- the kept A X of its last exit, the Cholesky factor of its own
 * `X = ortho!(copy(X))` and ONE local-only application of the potential difference -- instead of a full H X: the kinetic and
 * nonlocal parts of H do not change between SCF steps (src/scf/self_consistent_field.jl:80-129: only the density-dependent
 * potential does).  Silently ignored (full H X) when nothing is kept, shapes differ, the block is small / batched, or the
 * orthogonalisation needed more than one plain pass.  dftk_mi_kblock_set_projectors drops what is kept. */
int dftk_mi_kblock_reuse_AX(dftk_mi_kblock* kb, int on);
/* number of dftk_mi_lobpcg calls of this process that started from the kept A X (diagnostic / tests) */
int dftk_mi_ax_reuse_count(int64_t* calls);

/* Optional: device pointer to H*X of the last dftk_mi_lobpcg call on this block (n_G x M,
 * leading dimension n_G; valid until the next lobpcg call on the block). */
const dftk_mi_cplx* dftk_mi_lobpcg_last_AX(dftk_mi_kblock* kb);

/* ---- column helpers of LOBPCG as stand-alone calls (results on the host) ------------------------
 * columnwise_norms / columnwise_dots (src/common/linalg.jl:2-15, GPU forms src/gpu/linalg.jl:17-27):
 * X, A, B are n x m column-major device blocks. */
int dftk_mi_columnwise_norms(dftk_mi_basis* basis, int64_t n, int m, const dftk_mi_cplx* X_d, int64_t ldx,
                             double* norms_h /* [m] */);
int dftk_mi_columnwise_dots(dftk_mi_basis* basis, int64_t n, int m, const dftk_mi_cplx* A_d, int64_t lda,
                            const dftk_mi_cplx* B_d, int64_t ldb, dftk_mi_cplx* dots_h /* [m]: dot(A[:,i], B[:,i]) */);
/* ortho_qr (src/common/ortho.jl:1-9) / ortho!(X) (lobpcg_hyper_impl.jl:216-261): orthonormalise the columns of X
 * in place by Cholesky-QR with the reference's shift-and-retry and SVD fallback (same column space as Householder
 * QR; Q differs from LAPACK's by a unitary diagonal).  force_svd = 1 takes the SVD branch (X = U V').
 * *n_chol = Cholesky factorizations used (100 after an SVD fallback, as the reference reports). */
int dftk_mi_ortho_qr(dftk_mi_basis* basis, int64_t n, int m, dftk_mi_cplx* X_d, int64_t ldx, int force_svd,
                     int* n_chol, int* used_svd);
/* PreconditionerTPA (src/eigen/preconditioners.jl:27-78; GPU forms src/gpu/linalg.jl:25-43) on the block's kinetic
 * vector: precondprep! -> mean_kin_h[m]; ldiv!(Y, P, R) with mean_kin_h (NULL = not prepared yet: the
 * 1 / (kin + default_shift) form). */
int dftk_mi_tpa_precondprep(dftk_mi_kblock* kb, int m, const dftk_mi_cplx* X_d, int64_t ldx, double* mean_kin_h);
int dftk_mi_tpa_ldiv(dftk_mi_kblock* kb, int m, const dftk_mi_cplx* R_d, int64_t ldr, const double* mean_kin_h,
                     double default_shift, dftk_mi_cplx* Y_d, int64_t ldy);
/* The fused residual pass of one LOBPCG iteration (lobpcg_hyper_impl.jl:441-449): R = AX - X diag(lambda),
 * norms_h = column norms of R, and from the same read of X: mean_kin_h (precondprep!) and xx_h = <x, x>
 * (normalisation check :533); mean_kin_h / xx_h may be NULL. */
int dftk_mi_block_residual(dftk_mi_kblock* kb, int m, const dftk_mi_cplx* AX_d, int64_t lda,
                           const dftk_mi_cplx* X_d, int64_t ldx, const double* lambda_h, dftk_mi_cplx* R_d,
                           int64_t ldr, double* norms_h, double* mean_kin_h, double* xx_h);

/* ---- dense helpers exposed for tests (the LOBPCG building blocks) ----------------------------
 * zgemm: C = alpha*op(A)*B + beta*C, op(A) = A ('N') or A^H ('C'); f64 MFMA, deterministic split-K. */
int dftk_mi_zgemm(dftk_mi_basis* basis, char transA, int64_t m, int64_t n, int64_t k,
                  dftk_mi_cplx alpha, const dftk_mi_cplx* A_d, int64_t lda,
                  const dftk_mi_cplx* B_d, int64_t ldb, dftk_mi_cplx beta,
                  dftk_mi_cplx* C_d, int64_t ldc);
/* Structured variants used by the LOBPCG driver (lobpcg_hyper_impl.jl:141-145 Gram matrices that are
 * hermitised afterwards, :216-261 X*inv(R)); flags (may be combined):
 *   DFTK_MI_GEMM_UPPER     only the 128x64 tiles of C that intersect the upper triangle (i <= j) are
 *                          computed and written, the rest of C is left untouched;
 *   DFTK_MI_GEMM_B_UPPER   B is upper triangular (B[k][j] = 0 for k > j): the k loop stops at the diagonal;
 *   DFTK_MI_GEMM_REAL      the long operands are blocks of real-symmetric vectors in the half-sphere format of
 *                          dftk_mi_kblock_set_gamma_real, i.e. real matrices with two real rows per complex entry:
 *                          'C' computes Re(A^H B) (imaginary part stored as 0), 'N' computes A * Re(B); two real
 *                          matrix-core products per complex entry instead of three. */
#define DFTK_MI_GEMM_UPPER 1
#define DFTK_MI_GEMM_B_UPPER 2
#define DFTK_MI_GEMM_REAL 8
int dftk_mi_zgemm_ex(dftk_mi_basis* basis, char transA, int64_t m, int64_t n, int64_t k,
                     dftk_mi_cplx alpha, const dftk_mi_cplx* A_d, int64_t lda,
                     const dftk_mi_cplx* B_d, int64_t ldb, dftk_mi_cplx beta,
                     dftk_mi_cplx* C_d, int64_t ldc, int flags);
/* Hermitian eigen-decomposition (blocked parallel Jacobi): A (n x n, full storage, destroyed),
 * W_h[n] ascending eigenvalues (host), V_d n x n eigenvectors (columns, sorted like W). */
int dftk_mi_heev(dftk_mi_basis* basis, int n, dftk_mi_cplx* A_d, int64_t lda, double* W_h,
                 dftk_mi_cplx* V_d, int64_t ldv);
/* The LOWEST nev eigenpairs only -- what rayleigh_ritz consumes (lobpcg_hyper_impl.jl:141-153: `vectors[:, 1:N]`
 * of the 2N x 2N / 3N x 3N matrix Y'AY): one spectral split by a Newton-Schulz sign iteration on the f64 matrix cores
 * (sigma above the nev-th smallest diagonal entry), an orthonormal basis of the lower invariant subspace by
 * Cholesky-QR, the blocked Jacobi on the k x k projected matrix (nev <= k), one product back.  W_h[0 .. nev) and the
 * first nev columns of V_d are set; A is left intact.  Small problems (n < 384), nev > 0.6 n and inputs on which the
 * split fails its own checks go to dftk_mi_heev (A destroyed, all n pairs returned).  DFTK_MI_HEEV_PARTIAL=0 switches
 * the split off, DFTK_MI_HEEV_PARTIAL_MIN=<n> moves the size threshold. */
int dftk_mi_heev_lowest(dftk_mi_basis* basis, int n, int nev, dftk_mi_cplx* A_d, int64_t lda, double* W_h,
                        dftk_mi_cplx* V_d, int64_t ldv);
/* Host-only: the shift rule of dftk_mi_heev_lowest on a diagonal (sigma, estimated distance to the nearest
 * eigenvalue) and, for a norm bound of A - sigma I, the number of held iterations (CPU test-suite). */
int dftk_mi_heev_sigma_host(int n, const double* diag_h, int nev, double* sigma, double* gap_guess,
                            int* hold_iterations, double norm_bound);
/* Upper Cholesky A = R^H R in place (strict lower part left untouched) + inverse of R.
 * Returns DFTK_MI_NUM_CHOLESKY when a pivot is not positive / finite. */
int dftk_mi_potrf_trtri(dftk_mi_basis* basis, int n, dftk_mi_cplx* A_d, int64_t lda,
                        dftk_mi_cplx* invR_d, int64_t ldi);
/* The same for a REAL symmetric A stored as complex: the caller vouches that every imaginary part of the upper
 * triangle is exactly zero (the Gram matrices of the real-symmetric Gamma iteration); they are not read. */
int dftk_mi_potrf_trtri_real(dftk_mi_basis* basis, int n, dftk_mi_cplx* A_d, int64_t lda,
                             dftk_mi_cplx* invR_d, int64_t ldi);

/* ---- comm_kpts: replaces MPI.Init / mpi_sum!(rho, comm_kpts)
 *      (src/common/mpi.jl:19-32 at src/densities.jl:46) with RCCL over xGMI ---------------------
 * Rank 0 calls get_unique_id and ships the 128 bytes to the other ranks by any side channel. */
int dftk_mi_comm_get_unique_id(char id_out[128]);
int dftk_mi_comm_init_rank(const char id[128], int n_ranks, int rank, int device,
                           dftk_mi_comm** comm_out);
/* Host-staged communicator: the library copies device data to pinned host buffers and calls back -- the hook
 * for MPI (MPI.Allreduce! / MPI.Alltoallv! in a Julia shim) and for running several ranks on one GPU in tests.
 * allreduce: in-place sum of n doubles.  alltoallv: piece i of send_h (offset / count in DOUBLES) goes to rank i,
 * piece i of recv_h comes from rank i (alltoallv may be NULL if no k-block is sharded).  Return 0 on success. */
typedef int (*dftk_mi_allreduce_fn)(void* user, double* buf_h, size_t n);
typedef int (*dftk_mi_alltoallv_fn)(void* user, const double* send_h, const size_t* send_counts,
                                    const size_t* send_offsets, double* recv_h, const size_t* recv_counts,
                                    const size_t* recv_offsets);
int dftk_mi_comm_create_host(int n_ranks, int rank, int device, dftk_mi_allreduce_fn allreduce,
                             dftk_mi_alltoallv_fn alltoallv, void* user, dftk_mi_comm** comm_out);
int dftk_mi_comm_destroy(dftk_mi_comm* comm);
/* What the communicator is: *backend = 0 RCCL / 1 host callbacks; *n_ranks as RCCL itself reports it
 * (ncclCommCount) for an RCCL communicator; *version = ncclGetVersion code (0 for the host back end). */
int dftk_mi_comm_describe(const dftk_mi_comm* comm, int* backend, int* n_ranks, int* version);
int dftk_mi_comm_rank(const dftk_mi_comm* comm);
int dftk_mi_comm_size(const dftk_mi_comm* comm);
/* In-place sum all-reduce of n doubles on `stream` (hipStream_t as void*, NULL = default). */
int dftk_mi_allreduce_sum_f64(dftk_mi_comm* comm, double* buf_d, size_t n, void* stream);

/* ---- plane-wave sharding of ONE k-block over a communicator (Gamma-only cells) ----------
 * The reference can only duplicate a k-point on surplus ranks (src/PlaneWaveBasis.jl:190-203).  Here rank r owns
 * the rows [row_starts_h[r], row_starts_h[r+1]) of the sphere (row_starts_h has n_ranks + 1 entries, 0 .. n_G):
 * after this call every orbital block handed to dftk_mi_apply_H(_parts) / dftk_mi_lobpcg /
 * dftk_mi_density_accumulate is that row slab (packed: leading dimension == local rows for apply_H / density),
 * and dftk_mi_kblock_set_projectors takes the row slab of P.  Inside: products over n_G become local partial
 * sums + one small all-reduce, the small dense factorizations run replicated, and the FFT pipeline is fed by
 * a slab <-> band all-to-all (every rank transforms n_bands / n_ranks whole bands).  dftk_mi_density_accumulate
 * then adds only this rank's bands into rho_d: complete it with dftk_mi_allreduce_sum_f64 over the same
 * communicator.  Call before set_projectors; comm = NULL un-shards.  The communicator is borrowed. */
int dftk_mi_kblock_set_shard(dftk_mi_kblock* kb, dftk_mi_comm* comm, const int64_t* row_starts_h);

/* ---- real-symmetric orbitals of a Gamma-point block (EXTENSION: the reference has no Gamma special case) -------
 * At k = 0 the Hamiltonian of the reference's models (real local potential, projectors of real functions)
 * commutes with complex conjugation in real space: its eigenvectors can be chosen with psi(-G) = conj(psi(G)).
 * After dftk_mi_kblock_set_gamma_real(kb, 1), dftk_mi_lobpcg on this block projects the caller's X onto that
 * invariant subspace, iterates on its HALF-SPHERE image (row 0 = x(G = 0), real; row j > 0 = sqrt(2) x(G_j) for one
 * representative of every pair {G, -G}; n_half = (n_G + 1) / 2 rows), and hands back full-sphere vectors.
 * Eigenvalues, residual norms, density and energies are those of the general complex iteration (same operator,
 * same spectrum and multiplicities); every n_G-long product runs as a REAL matrix product over half the rows
 * (DFTK_MI_GEMM_REAL: a third of the matrix-core flops) and two bands share one FFT pipeline pass (a + i b).
 * Errors (DFTK_MI_EINVAL): sphere without inversion symmetry / with a Nyquist point, kinetic(G) != kinetic(-G)
 * (k != 0), projectors that are not real-symmetric (checked when they are first used).
 * Plane-wave sharded block: call AFTER dftk_mi_kblock_set_shard; the half-format rows are then split evenly over
 * the ranks (dftk_mi_gamma_half_size returns this rank's count; dftk_mi_gamma_compress / _expand / _apply_H become
 * collective and work on packed row slabs), the caller keeps handing full-sphere row slabs to dftk_mi_lobpcg.
 * dftk_mi_apply_H / dftk_mi_density_accumulate on the block keep their general complex semantics.
 * dftk_mi_lobpcg_last_AX returns NULL after a real-mode run. */
int dftk_mi_kblock_set_gamma_real(dftk_mi_kblock* kb, int on);
int dftk_mi_gamma_half_size(dftk_mi_kblock* kb, int64_t* n_half);
/* Host-only pair tables (CPU test-suite): row_h[j] / partner_row_h[j] = sphere rows of G_j / -G_j, j = 0 is G = 0;
 * pairs ascending in row_h.  NULL tables: only count. */
int dftk_mi_gamma_tables_host(int nx, int ny, int nz, int64_t n_G, const int64_t* mapping0_h, int64_t* n_half,
                              int32_t* row_h, int32_t* partner_row_h);
/* full sphere (n_G x m) -> half format (n_half x m): the real-symmetric part (x(G) + conj x(-G)) / 2, scaled; and
 * back (exact inverse on real-symmetric vectors). */
int dftk_mi_gamma_compress(dftk_mi_kblock* kb, int m, const dftk_mi_cplx* X_d, int64_t ldx, dftk_mi_cplx* Xh_d,
                           int64_t ldh);
/* What dftk_mi_lobpcg does with its start vectors on a Gamma-real block: every column is rotated by the global phase
 * exp(-i phi), exp(2 i phi) = s / |s|, s = sum_G x(G) x(-G), that maximises its real-symmetric part, then compressed.
 * A real field times any phase (e.g. an orbital of a complex iteration) keeps all of its norm; a column that is already
 * real-symmetric is compressed bit for bit as by dftk_mi_gamma_compress. */
int dftk_mi_gamma_compress_aligned(dftk_mi_kblock* kb, int m, const dftk_mi_cplx* X_d, int64_t ldx, dftk_mi_cplx* Xh_d,
                                   int64_t ldh);
int dftk_mi_gamma_expand(dftk_mi_kblock* kb, int m, const dftk_mi_cplx* Xh_d, int64_t ldh, dftk_mi_cplx* X_d,
                         int64_t ldx);
/* H psi on half-format blocks (`which` as dftk_mi_apply_H_parts). */
int dftk_mi_gamma_apply_H(dftk_mi_kblock* kb, int which, int n_bands, const dftk_mi_cplx* psih_d, int64_t ld_psi,
                          dftk_mi_cplx* Hpsih_d, int64_t ld_Hpsi);
/* dftk_mi_density_accumulate for REAL-SYMMETRIC columns in the full-sphere layout (what a real-mode dftk_mi_lobpcg
 * returns): bands 2p and 2p + 1 share one transform (rho += w_2p Re^2 + w_2p+1 Im^2).  The caller vouches for the
 * symmetry; general complex columns give a wrong density.  Needs no prior set_gamma_real. */
int dftk_mi_density_accumulate_real(dftk_mi_kblock* kb, int n_bands, const dftk_mi_cplx* psi_d, int64_t ld_psi,
                                    const double* weight_h, double* rho_d);

/* Host-only view of the slab <-> band transposition plan of a sharded block (CPU test-suite): for `rank` of
 * `n_ranks`, an n_bands block: band_starts[s] .. band_starts[s+1] = bands transformed by rank s; slab_off/cnt[s] =
 * piece of the packed n_loc x n_bands slab that goes to rank s; band_off/cnt[r] = piece of the packed band layout that
 * comes from rank r (offsets / counts in complex elements). */
int dftk_mi_shard_plan_host(int n_ranks, int rank, int n_bands, const int64_t* row_starts_h, int* band_starts,
                            int64_t* slab_off, int64_t* slab_cnt, int64_t* band_off, int64_t* band_cnt);

/* ---- per-family kernel timing with HIP events on the basis' stream (used by bench.py) ----------
 * family: 0 UNSTRUCTURED zgemm calls (work = 8mnk flops; 4mnk for a DFTK_MI_GEMM_REAL call = the flops of the
 * equivalent real product), 1..5 FFT stages A..E (work = algorithmic bytes of
 * the pruned pipeline), 6 density z-pass, 7 heev, 8 potrf+trtri, 9 whole apply_H
 * (work = bands), 10 zgemm operand bytes (no time), 11 STRUCTURED zgemm calls (UPPER / B_UPPER; work = flops of
 * the mathematically needed part only), 12 real flops the launched zgemm tiles execute (no time; 6 per complex
 * multiply-add in the 3M kernels, 4 in the REAL ones), 13 collectives of a sharded block (work = bytes), 14 useful
 * flops of all zgemm calls counted as complex products (no time; a REAL call counts twice its flops: what the
 * general complex iteration would need).  enable(1) resets. */
int dftk_mi_prof_enable(dftk_mi_basis* basis, int on);
int dftk_mi_prof_get(dftk_mi_basis* basis, int family, double* total_ms, double* work, int64_t* launches);
/* ---- density-sized solvers of the SCF glue (mix_kernels.hip) -----------------------------
 * Anderson acceleration of the density fixed-point iteration (src/scf/anderson.jl:36-130, ScfAndersonDensitySolver of
 * src/scf/scf_solvers.jl:68-102): history of up to m (iterate, preconditioned residual) pairs of n doubles on the
 * device; a step = one reduction kernel (the inner products with the new residual; the Gram matrix of the history is
 * kept), ONE host synchronisation, the m x m least-squares problem on the host (conditioning test on cond(R) =
 * sqrt(cond(M'M)), entries with error > errorfactor * min dropped, as the reference), one fused update kernel.
 * x_next_d must not alias the inputs.  m = 0: plain damping x + alpha Pf. */
typedef struct dftk_mi_anderson dftk_mi_anderson;
int dftk_mi_anderson_create(dftk_mi_basis* basis, int64_t n, int m, double maxcond, double errorfactor,
                            dftk_mi_anderson** out);
int dftk_mi_anderson_destroy(dftk_mi_anderson* acc);
int dftk_mi_anderson_reset(dftk_mi_anderson* acc);
int dftk_mi_anderson_history(const dftk_mi_anderson* acc);   /* entries held */
int dftk_mi_anderson_step(dftk_mi_anderson* acc, const double* x_d, double alpha, const double* Pf_d, double* x_next_d,
                          int* n_history /* may be NULL */);
/* chi0 mixing (src/scf/mixing.jl:228-290, LdosMixing / HybridMixing / Chi0Mixing with RPA = true):
 * d_rho = (1 - chi0 vc)^-1 dF by restarted GMRES in real space (KrylovKit linsolve: krylovdim, tol = max(1e-12,
 * reltol |dF - mean dF|), zero start vector), vc = the Hartree kernel through its multiplier cube poisson_d
 * (src/terms/hartree.jl:68-81; NULL: no Hartree term), chi0 = LdosModel (ldos_d: n_comp cubes = the local density of
 * states at the Fermi level, src/scf/chi0models.jl:21-45; NULL or |ldos| < sqrt(eps) everywhere: term absent) and /
 * or DielectricModel (dielectric != 0: kTF, eps_r, :54-80, needs recip_lattice_h, column-major 3 x 3).  n_comp = 2:
 * collinear spin, vectors are (up, down) stacked, the Hartree kernel sees the total.  With no term alive d_rho = dF
 * (simple mixing, chi0models.jl:32, mixing.jl:264-266).  Every Krylov step is 12 launches and ONE host
 * synchronisation (all inner products of the classical Gram-Schmidt step in one fetch; a second pass when cancellation
 * asks for it).  *n_applies = applications of the dielectric operator, *converged = 0 if maxiter cycles did not
 * reach the tolerance (d_rho is the last iterate, as the reference's mixing uses it). */
int dftk_mi_chi0_mix(dftk_mi_kblock* cube_kblock, int n_comp, const double* recip_lattice_h, const double* poisson_d,
                     const double* ldos_d, double dvol, int dielectric, double kTF, double eps_r, const double* dF_d,
                     double reltol, int krylovdim, int maxiter, double* drho_d, int* n_applies, int* converged);

/* Process-wide counters since load: kernel launches issued by the library and host synchronisations it waited on
 * (stream synchronisations of its drivers, result fetches, scheduling rounds of the batched k-point driver).  For the
 * many-small-k workloads these two ARE the cost model; either pointer may be NULL. */
int dftk_mi_launch_count(int64_t* launches, int64_t* host_syncs);

/* The two density-sized scalars an SCF step reads on the host besides the term energies, in one kernel and one
 * synchronisation: out_h[0] = sum_i a[i] b[i] (b_d may be NULL: 0), out_h[1] = sum_i (a[i] - c[i])^2 (c_d may be NULL: 0);
 * n doubles each, on the device.  The host mirror passes a = rho_out, b = V_in (the nonlocal energy from the Ritz values:
 * sum f eps - E_kin - int V_in rho_out) and c = rho_in (||rho_out - rho_in||, the convergence criterion of
 * src/scf/self_consistent_field.jl:229-236); the caller multiplies by the volume element. */
int dftk_mi_step_sums(dftk_mi_basis* basis, int64_t n, const double* a_d, const double* b_d, const double* c_d, double* out_h);

/* compute_occupation's Fermi-level search (src/occupation.jl:99-132, FermiBisection; host-only, no device call): bisection of
 *   excess(eF) = sum_k kweights[k] sum_n filled * smearing((eig[k][n] - eF) / temperature) - n_electrons
 * on the bracket [lo, hi] (excess(lo) < 0 <= excess(hi)) until the midpoint equals an end point (adjacent doubles; at most
 * 200 halvings), *eF_out = midpoint of the final bracket.  eig: the k-points' eigenvalues behind one another (n_bands[k]
 * each).  smearing: 1 = Fermi-Dirac, 2 = Gaussian (Smearing.jl:66-76, :86-92); temperature > 0.  The host mirror runs this
 * once per SCF step (56 trial levels of a 72-k-point mesh: 0.3 ms of interpreted code in a 4 ms step). */
int dftk_mi_fermi_bisection(int n_k, const int* n_bands, const double* eig, const double* kweights, int smearing,
                            double temperature, double filled, double n_electrons, double lo, double hi, double* eF_out);
/* dftk_mi_prof_enable(basis, 3) also books every zgemm call per SHAPE; this returns (and clears) that table: row i of
 * rows6 = { transA ('N' = 0, 'C' = 1), m, n, k, flags (UPPER | B_UPPER | DFTK_MI_GEMM_REAL), calls }, ms[i] = summed time; *count = shapes seen (<= cap
 * rows are written).  bench.py replays the table at 1 / N of the rows for its sharded-step measurement. */
int dftk_mi_prof_zgemm_shapes(dftk_mi_basis* basis, int cap, int64_t* rows6, double* ms, int* count);

/* Diagnostic: measured issue-rate ceiling of v_mfma_f64_16x16x4_f64 (TFLOP/s, no memory traffic). */
int dftk_mi_diag_mfma_peak(dftk_mi_basis* basis, int waves_per_simd, int iters, double* tflops);

/* ---- host-only introspection (no GPU needed; used by the CPU test-suite) ---------------------
 * Launch plan of dftk_mi_zgemm(_ex) for one shape: out[12] = { column-tile width, full tile rows, full tile
 * columns, right-strip tiles, bottom-strip tiles, interior {K chunks, chunk length, chunk->XCD placement},
 * border {K chunks, chunk length, placement}, shifted }.  shifted = 1: a ragged last tile column is covered by
 * one more FULL tile in the interior launch, shifted left to end at column n (it stores only the new columns),
 * instead of a right-strip launch. */
int dftk_mi_zgemm_plan_host(char transA, int64_t m, int64_t n, int64_t k, int flags, int* out12);
/* Schedule of the blocked Jacobi eigensolver for an n x n matrix: *n_blocks = number of 16-wide blocks (even,
 * padded); for round in [-1, *n_blocks - 2] (pass pairs = where = NULL to query n_blocks only):
 * pairs[2k], pairs[2k+1] = the k-th disjoint block pair of the round (round -1: (2k, 2k+1), then a round-robin
 * tournament); where[2b], where[2b+1] = (pair index, member 0/1) of block b -- the inverse map the look-ahead
 * pair solve uses to find last round's rotations of its two blocks. */
int dftk_mi_jacobi_schedule_host(int n, int round, int* n_blocks, int* pairs /* [n_blocks] */,
                                 int* where /* [2 n_blocks] */);
/*
 * 1-D plan for length n: radices (<= 32 entries) and the in-place permutation `pos[e]` such
 * that a decimation-in-time pass wants input element e at position pos[e] and a
 * decimation-in-frequency pass leaves output frequency k at position pos[k]. */
int dftk_mi_fft_plan_host(int n, int* n_radices, int* radices /* [32] */, int* pos /* [n] */);
/* Sphere pruning tables derived from a mapping: number of non-empty x-lines, number of distinct
 * z planes, and (optionally, may be NULL) the line ids (iy + ny*iz) / first-coefficient offsets. */
int dftk_mi_sphere_tables_host(int nx, int ny, int nz, int64_t n_G, const int64_t* mapping0_h,
                               int64_t* n_lines, int* n_zplanes, int64_t* line_id /* [n_lines] */,
                               int64_t* line_start /* [n_lines+1] */);

#ifdef __cplusplus
}
#endif
#endif /* DFTK_MI355X_H */
