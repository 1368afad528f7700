// common.h -- shared declarations of the MI355X plane-wave SCF hot-path library (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <string>
#include <map>
#include <vector>
#include "../../include/dftk_mi355x.h"

typedef double2 cd;   // complex fp64, (x, y) = (re, im); layout-compatible with dftk_mi_cplx

// Every kernel launch of the library is counted (dftk_mi_launch_count): for the many-small-k workloads launches and
// host synchronisations per SCF step ARE the cost model (bench.py --mode kpoints: roofline.bound = "latency").
#include <atomic>
extern std::atomic<int64_t> g_dftk_launches, g_dftk_host_syncs;
inline hipError_t dftk_counted_stream_sync(hipStream_t s) {
    g_dftk_host_syncs.fetch_add(1, std::memory_order_relaxed);
    return (hipStreamSynchronize)(s);
}
#define hipStreamSynchronize(s) dftk_counted_stream_sync(s)
#undef hipLaunchKernelGGL
#define hipLaunchKernelGGL(kernelName, ...)                               \
    do {                                                                  \
        g_dftk_launches.fetch_add(1, std::memory_order_relaxed);          \
        hipLaunchKernelGGLInternal((kernelName), __VA_ARGS__);            \
    } while (0)

void dftk_set_error(const char* fmt, ...);
// hipMalloc for library-owned scratch; DFTK_MI_POISON=1 fills it with 0xFF bytes (NaN doubles) so that any read
// of scratch that was not written in the current call shows up as a non-finite result (debugging aid)
hipError_t dftk_scratch_malloc(void** p, size_t bytes);

#define HIPCHK(expr)                                                                      \
    do {                                                                                  \
        hipError_t _e = (expr);                                                           \
        if (_e != hipSuccess) {                                                           \
            dftk_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return DFTK_MI_EHIP;                                                          \
        }                                                                                 \
    } while (0)

#define CHK(expr)                                 \
    do {                                          \
        int _s = (expr);                          \
        if (_s != 0) return _s;                   \
    } while (0)

// ------------------------------------------------------------------------------------ 1-D plans
#define DFTK_MAX_RADICES 32
struct FftAxis {            // passed to kernels by value
    int n;                  // transform length
    int nrad;               // number of stages
    int rad[DFTK_MAX_RADICES];
    const cd* tw;           // device: tw[t] = exp(+2 pi i t / n), t in [0, n)
    const int* pos;         // device: in-place permutation (see dftk_mi_fft_plan_host)
};

int  plan_radices(int n, int* nrad, int* rad);          // host: factorise into {5,4,3,2,primes}
void plan_positions(int n, int nrad, const int* rad, int* pos);

// ------------------------------------------------------------------------------------ handles
struct dftk_mi_basis {
    int nx, ny, nz, nxp;          // nxp = nx rounded up to a multiple of FFT_L (padded x pitch)
    double volume;
    int device;
    hipStream_t stream;
    FftAxis ax[3];                // x, y, z
    void* d_tables[6];            // device tw/pos buffers (owned)
    int fft_batch;                // bands per launch group
    // scratch pool (grown on demand)
    cd* T1; size_t T1_bytes;
    cd* T2; size_t T2_bytes;
    // general workspace for dense algebra (split-K slabs, small matrices)
    void* ws; size_t ws_bytes;
    double* d_scalars;            // small device buffer for reductions (256 doubles)
    double* h_scalars;            // pinned host mirror
    char* h_fetch;                // pinned, device-visible landing zone of host_fetch() (zero-copy device -> host results)
    int use_mfma;                 // 0 => naive GEMM kernels (env DFTK_MI_GEMM=naive)
    struct Prof* prof;            // per-family HIP-event timing (dftk_mi_prof_*)
    // workspace of the dense factorizations (heev ping-pong copies, rotation buffers); owned by the basis so
    // that several bases / devices in one process never share it (freed in dftk_mi_basis_destroy)
    void* dense_ws; size_t dense_ws_bytes;
    // workspace of the partial-spectrum eigensolver (eig_kernels.hip): its iterates live across calls into the routines
    // that use dense_ws (Cholesky, Jacobi on the projected matrix), so it cannot share that buffer
    void* eig_ws; size_t eig_ws_bytes;
    struct dftk_mi_comm* comm;    // plane-wave (row-slab) communicator of a sharded k-block, or null (borrowed)
    // device copy of the symmetry tables of the last cube_symmetrize call (owned; keyed by a hash of the operations: an SCF
    // symmetrises with the same group every step -- no upload and no host synchronisation after the first call)
    void* symm_tab; uint64_t symm_key; int symm_n;
};

// ------------------------------------------------------------------------------------ profiling
// Kernel families timed with HIP events on the basis' stream (bench.py roofline numbers).
enum ProfFamily {
    PROF_ZGEMM = 0,      // UNSTRUCTURED zgemm calls: work = 8 m n k flops
    PROF_FFT_A = 1,      // x-lines backward + scatter      (work = algorithmic bytes, dense 3-pass convention)
    PROF_FFT_B = 2,      // y backward
    PROF_FFT_C = 3,      // fused z backward * V * z forward
    PROF_FFT_D = 4,      // y forward
    PROF_FFT_E = 5,      // x forward + gather + kinetic
    PROF_DENS_Z = 6,     // z backward + |psi|^2 accumulate
    PROF_HEEV = 7,       // dense Hermitian eigensolver (whole call)
    PROF_CHOL = 8,       // potrf + trtri (whole call)
    PROF_APPLY_H = 9,    // whole dftk_mi_apply_H call (work = bands)
    PROF_ZGEMM_BYTES = 10,   // no timing: work = algorithmic operand bytes of the zgemm calls (A + B + C [+ C if beta != 0])
    PROF_ZGEMM_STRUCT = 11,  // structured calls (UPPER / B_UPPER), timed apart: work = flops of the part of the product
                             // that is mathematically needed (upper triangle of C; k <= j for triangular B)
    PROF_ZGEMM_EXEC = 12,    // no timing: real flops the launched tiles execute on the matrix pipe, all zgemm calls
                             // (3M kernel: 6 per complex multiply-add, 4M: 8; full tiles incl. shifted/border recompute)
    PROF_COMM = 13,          // collectives of a sharded k-block (work = bytes handed to the communicator)
    PROF_ZGEMM_CPLX = 14,    // no timing: useful flops of all zgemm calls as if none were REAL (a DFTK_MI_GEMM_REAL call
                             // stands for a complex product of twice its flops): what the general complex path needs
    PROF_EW = 15,            // n_G-sized element-wise / column-reduction kernels of the LOBPCG driver and the Gamma-real
                             // pack / unpack passes (work = algorithmic bytes): row-local, i.e. SHARDED work of a
                             // plane-wave-sharded block
    PROF_HOST_WAIT = 16,     // no events: launches = host synchronisations inside the library, ms = wall ms the host waited
    PROF_AR_MODEL = 17,      // no timing: the all-reduces a plane-wave-sharded run of the same calls performs (launches =
                             // calls, work = bytes) -- counted on ONE rank too: the input of bench.py's Amdahl model
    PROF_A2A_MODEL = 18,     // no timing: its slab <-> band transposes (launches, work = bytes of the blocks moved)
    PROF_NFAM = 24
};
struct Prof {
    bool on = false;
    struct Pair { hipEvent_t a, b; int fam; uint64_t tag; double work; };
    struct Shape { double ms = 0, work = 0; int64_t n = 0; };
    std::map<uint64_t, Shape> shapes;   // per-shape zgemm breakdown (env DFTK_MI_GEMM_SHAPES)
    std::vector<Pair> pending;
    std::vector<Pair> pool;
    int open = 0;                       // scopes begun and not yet ended: no flush while > 0 (slots index `pending`)
    int mute = 0;                       // > 0: prof_begin books nothing (inner calls of a routine that is booked as a whole)
    bool shape_tags = false;            // dftk_mi_prof_enable(b, 3): zgemm calls are also booked per shape (dftk_mi_prof_zgemm_shapes)
    double ms[PROF_NFAM] = {0};
    double work[PROF_NFAM] = {0};
    int64_t launches[PROF_NFAM] = {0};
};
int prof_begin(dftk_mi_basis* b, int fam, double work, uint64_t tag = 0);   // returns slot index or -1
void prof_end(dftk_mi_basis* b, int slot);
int prof_resolve(dftk_mi_basis* b);
struct ProfScope {   // RAII pair of prof_begin / prof_end
    dftk_mi_basis* b;
    int slot;
    ProfScope(dftk_mi_basis* basis, int fam, double work) : b(basis), slot(prof_begin(basis, fam, work)) {}
    ~ProfScope() { prof_end(b, slot); }
    ProfScope(const ProfScope&) = delete;
    ProfScope& operator=(const ProfScope&) = delete;
};
void prof_count(dftk_mi_basis* b, int fam, double work);   // count-only families (no events)
// Every host synchronisation of the library's drivers goes through these two (counted in PROF_HOST_WAIT):
// host_wait = hipStreamSynchronize(b->stream); host_fetch = small device -> host result WITHOUT a blit: a copy kernel
// writes into pinned host memory mapped into the device's address space (no __amd_rocclr_copyBuffer dispatch, no staging
// through pageable memory), then one stream synchronisation.
const size_t HOST_FETCH_BYTES = 256 * 1024;
int host_wait(dftk_mi_basis* b);
int host_fetch(dftk_mi_basis* b, void* dst_h, const void* src_d, size_t bytes);

// real-symmetric orbitals of a Gamma-point block (gamma_kernels.hip)
struct GammaReal {
    bool on = false;              // dftk_mi_lobpcg iterates in the half-sphere format
    int64_t n_half = 0;           // (n_G + 1) / 2
    int *d_g = nullptr, *d_mg = nullptr;   // [n_half] sphere rows of G_j and -G_j (j = 0: G = 0)
    double* d_kin_half = nullptr;
    cd* P_half = nullptr;         // n_half x n_p, scaled like the vectors (built on first use)
    const cd* P_src = nullptr;
    int P_n_p = 0;
    cd* buf = nullptr;            // pack / unpack scratch
    size_t buf_bytes = 0;
    // plane-wave sharded block: rank r owns the half-format rows [half_rows[r], half_rows[r + 1]) (split evenly);
    // P_half is then this rank's slab of the half-format projectors
    std::vector<int64_t> half_rows;
};

struct dftk_mi_kblock {
    dftk_mi_basis* basis;
    int device;                   // copy of basis->device (destroy must not touch the basis)
    int64_t n_G;
    int64_t n_lines;              // non-empty x-lines (iy, iz)
    int nzx;                      // distinct z planes touched by the sphere
    int z_lo;                     // the sphere planes are {0 .. z_lo-1} u {nz-(nzx-z_lo) .. nz-1} (always so for a sphere
                                  // of G vectors; -1 otherwise: the register-resident z kernels then stay off)
    // device tables (owned)
    int*   d_cpos;                // [n_G]   pos_x[ix] of each coefficient
    int*   d_line_start;          // [n_lines+1] first coefficient of each line
    int*   d_line_ypos;           // [n_lines] pos_y[iy] of each line
    int*   d_zls;                 // [nzx+1] first line of each z plane
    int*   d_zpos;                // [nzx]   pos_z[iz] of each plane
    int*   d_zval;                // [nzx]   iz of each plane (natural index)
    int*   d_line_yval;           // [n_lines] iy of each line (natural index)
    int*   d_cx;                  // [n_G]   ix of each coefficient (natural index)
    double* d_kin;                // [n_G]
    double* d_Vs;                 // [nz*ny*nxp] potential / N, padded pitch (owned, or shared: Vs_share) or null
    struct SharedVs* Vs_share;    // non-null: d_Vs is the buffer of this reference-counted object (dftk_mi_kblocks_set_potential)
    // nonlocal
    int n_p;
    const cd* P;                  // borrowed device pointer, n_G x n_p
    int64_t ldP;
    double* d_D;                  // [n_p*n_p] dense (owned)
    int D_bw;                     // half bandwidth of D
    // LOBPCG workspace (owned, grown on demand)
    cd* lob_buf; size_t lob_bytes;
    cd* last_AX;
    // A X of the last dftk_mi_lobpcg exit in the driver's own format (general driver; points into lob_buf), its shape, and the
    // padded potential that was bound then: the next call may start from A_new X = A_old X + (V_new - V_old) X instead of a
    // full H X (dftk_mi_kblock_reuse_AX: kinetic and nonlocal parts do not change between SCF steps)
    cd* ax_keep; int ax_M; int64_t ax_rows; int64_t ax_ld;
    double* d_Vs_ax;              // [nz*ny*nxp] snapshot of d_Vs at that exit (owned)
    double* d_dVs;                // [nz*ny*nxp] scratch: V_new - V_old (owned)
    bool ax_reuse_next;           // the caller's one-shot promise: X0 of the next call IS the X returned by the last one
    // plane-wave (row-slab) sharding of this block over a communicator (dftk_mi_kblock_set_shard): orbital blocks
    // handed to apply_H / lobpcg / density_accumulate and the projector matrix are the rows
    // [sh_rows[rank], sh_rows[rank + 1]) of the sphere; the sphere tables / potential above stay complete
    dftk_mi_comm* sh_comm;           // borrowed; null = not sharded
    std::vector<int64_t>* sh_rows;   // [n_ranks + 1] row offsets (owned)
    cd* sh_buf; size_t sh_bytes;     // transpose buffers (owned, grown on demand)
    // residual history of the last dftk_mi_lobpcg call: hist[i + M * it], it = 0 .. n_iter (host, owned)
    std::vector<double>* lob_hist; int lob_hist_M, lob_hist_iters, lob_n_svd;
    // host copies of the sphere (pair tables of the Gamma-real format are built from them on demand)
    std::vector<int64_t>* h_mapping;
    std::vector<double>* h_kin;
    GammaReal* gr;                   // owned; null until dftk_mi_kblock_set_gamma_real / density_accumulate_real
};

// ------------------------------------------------------------------------------------ internal API
// fft_kernels.hip
int fft_ensure_scratch(dftk_mi_basis* b, dftk_mi_kblock* kb, int nb);
int launch_local_apply(dftk_mi_kblock* kb, int nb, const cd* psi, int64_t ldpsi, cd* out,
                       int64_t ldout, bool add_kinetic, bool have_local);
int launch_ifft_to_cube(dftk_mi_kblock* kb, const cd* c, cd* cube, int nb = 1);
int launch_fft_from_cube(dftk_mi_kblock* kb, const cd* cube, cd* c, int nb = 1);
// w_im_h (optional): separate weights for the squared IMAGINARY parts (two real bands packed into one transform)
int launch_density(dftk_mi_kblock* kb, int nb, const cd* psi, int64_t ldpsi, const double* w_h, double* rho,
                   const double* w_im_h = nullptr, const double* w2_h = nullptr, double* rho2 = nullptr);
int launch_pad_potential(dftk_mi_kblock* kb, const double* V);
int launch_kinetic_only(dftk_mi_kblock* kb, int nb, const cd* psi, int64_t ldpsi, cd* out,
                        int64_t ldout, bool accumulate, bool use_kin);

// gemm_kernels.hip
// upper & 1: only the 128x64 tiles that intersect the upper triangle (i <= j) are computed and
//            written; the rest of C is left untouched (Gram matrices that are hermitised afterwards)
// upper & 2: B is upper triangular (B[k][j] = 0 for k > j): the k loop of a tile column stops early
int zgemm(dftk_mi_basis* b, char transA, int64_t m, int64_t n, int64_t k, cd alpha, const cd* A,
          int64_t lda, const cd* B, int64_t ldb, cd beta, cd* C, int64_t ldc, int upper = 0);
int ensure_ws(dftk_mi_basis* b, size_t bytes);

// dense_kernels.hip
int dense_potrf_trtri(dftk_mi_basis* b, int n, cd* A, int64_t lda, cd* invR, int64_t ldi,
                      double* normest_R, double* normest_invR, bool real_input = false);   // host outputs
int dense_heev(dftk_mi_basis* b, int n, cd* A, int64_t lda, double* W_h, cd* V, int64_t ldv);
// the blocked Jacobi itself (all n pairs), not booked in any profile family and never redirected
int dense_heev_full(dftk_mi_basis* b, int n, cd* A, int64_t lda, double* W_h, cd* V, int64_t ldv);
// sums over the n x n matrix: squared off-diagonal / diagonal magnitudes and squared imaginary parts (host outputs)
int dense_input_norms(dftk_mi_basis* b, int n, const cd* A, int64_t lda, double* off2, double* dg2, double* im2);
// eig_kernels.hip: the lowest nev eigenpairs only (spectral split on the matrix cores + Jacobi on the projected matrix);
// falls back to dense_heev where the split does not apply
int dense_heev_lowest(dftk_mi_basis* b, int n, int nev, cd* A, int64_t lda, double* W_h, cd* V, int64_t ldv);
int eig_choose_sigma_host(int n, const double* diag, int nev, double* sigma, double* gap_guess);
int eig_hold_iterations_host(double gap_over_norm);
struct ProfMute {    // RAII: nothing inside is booked (the enclosing scope books the whole routine)
    dftk_mi_basis* b;
    explicit ProfMute(dftk_mi_basis* basis) : b(basis) { if (b->prof) b->prof->mute += 1; }
    ~ProfMute() { if (b->prof) b->prof->mute -= 1; }
    ProfMute(const ProfMute&) = delete;
    ProfMute& operator=(const ProfMute&) = delete;
};
int jacobi_schedule_host(int n, int round, int* nb_out, int* pairs, int* where);
int apply_D(dftk_mi_kblock* kb, int n_bands, const cd* X /*n_p x nb*/, cd* Y);
// elementwise / reductions used by LOBPCG (all on b->stream)
int ew_colnorms(dftk_mi_basis* b, int64_t n, int m, const cd* X, int64_t ldx, double* out_d);
int ew_coldots(dftk_mi_basis* b, int64_t n, int m, const cd* X, int64_t ldx, const cd* Y, int64_t ldy,
               double* out_re_d);
// R = AX - X diag(lam), norms = column norms of R; optionally (kin / xx non-null) mean_kin = sum kin |X|^2 and
// xx = sum |X|^2 per column in the same pass
int ew_residual(dftk_mi_basis* b, int64_t n, int m, const cd* AX, int64_t lda, const cd* X, int64_t ldx,
                const double* lam_d, cd* R, int64_t ldr, double* norms_d, const double* kin, double* mean_kin_d,
                double* xx_d);
// dst = TPA-preconditioned src (kin == null: copy), norms = column norms of dst
// (mean_kin_d == null with kin != null: the default-shift form 1 / (kin + shift), preconditioners.jl:52-53)
int ew_tpa(dftk_mi_basis* b, int64_t n, int m, const cd* src, int64_t lds, cd* dst, int64_t ldd, const double* kin,
           const double* mean_kin_d, double* norms_d, double default_shift = 1.0);
int ew_scale_cols(dftk_mi_basis* b, int64_t n, int m, cd* X, int64_t ldx, const double* s_d, bool invert);
int ew_copy(dftk_mi_basis* b, int64_t n, int m, const cd* X, int64_t ldx, cd* Y, int64_t ldy);
int ew_add(dftk_mi_basis* b, int64_t n, int m, const cd* X, int64_t ldx, cd* Y, int64_t ldy);          // Y += X
int ew_sub_real(dftk_mi_basis* b, int64_t n, const double* a, const double* c, double* out);            // out = a - c
int ew_fill_zero(dftk_mi_basis* b, cd* X, size_t count);
int ew_sub_identity_shifted(dftk_mi_basis* b, int rows, int cols, cd* C, int64_t ldc, int row0);
int ew_gather_cols(dftk_mi_basis* b, int64_t n, int m, const cd* X, int64_t ldx, const int* perm_d,
                   cd* Y, int64_t ldy);
int ew_add_diag(dftk_mi_basis* b, int n, cd* A, int64_t lda, double shift);
int ew_frob2(dftk_mi_basis* b, int64_t n, int m, const cd* X, int64_t ldx, double* out_d);
int ew_hermitize_upper(dftk_mi_basis* b, int n, cd* A, int64_t lda);
int ew_conj_transpose(dftk_mi_basis* b, int n, const cd* A, int64_t lda, cd* B, int64_t ldb);   // B = A^H (n x n)
int ew_square(dftk_mi_basis* b, double* d, size_t n);
int ew_sqrt(dftk_mi_basis* b, double* d, size_t n);
// imaginary parts of the column-wise dots  Im <x_c, y_c>  (ew_coldots gives the real parts)
int ew_coldots_im(dftk_mi_basis* b, int64_t n, int m, const cd* X, int64_t ldx, const cd* Y, int64_t ldy,
                  double* out_im_d);
// mean_kin_c = sum_G kin_G |X_Gc|^2  (precondprep!, preconditioners.jl:75-77)
int ew_weighted_colsums(dftk_mi_basis* b, int64_t n, int m, const cd* X, int64_t ldx, const double* w_d,
                        double* out_d);

// comm.cpp (internal side; all on the basis' stream; null communicator / one rank = no-op)
int comm_size(const dftk_mi_comm* c);
int comm_rank(const dftk_mi_comm* c);
int comm_allreduce(dftk_mi_comm* c, dftk_mi_basis* b, double* d, size_t n);
int comm_allreduce_norms(dftk_mi_comm* c, dftk_mi_basis* b, double* d, size_t n);   // d holds sqrt(local sums)
int comm_alltoallv(dftk_mi_comm* c, dftk_mi_basis* b, const cd* send, const size_t* soff, const size_t* scnt,
                   cd* recv, const size_t* roff, const size_t* rcnt);

// xc_kernels.hip
int local_potential_collinear(dftk_mi_kblock* cube_kb, const double* rho, const double* vloc, const double* green,
                              int fun_mask, double* V_out, double* energies_h);
int local_potential_lda(dftk_mi_kblock* cube_kb, const double* recip_h, const double* rho, const double* vloc,
                        const double* green, int fun_mask, double threshold, double* V_out, double* energies_h);

// cube_kernels.hip: density-sized operations of the SCF glue (symmetrisation, mixing multipliers) on a full-cube k-block
int cube_symmetrize(dftk_mi_kblock* cube_kb, int n_sym, const int32_t* S_h, const double* tau_h, int do_lowpass,
                    const double* rho_in, double* rho_out);
int cube_fourier_filter(dftk_mi_kblock* cube_kb, int kind, const double* recip_h, double p0, double p1,
                        const double* mult_d, const double* f, double* out);
int xc_gga_pointwise(dftk_mi_basis* b, int64_t n, const double* rho, const double* sigma, int fun_mask,
                     double threshold, double* e, double* vrho, double* vsigma);

// setup_kernels.hip
int sphere_enumerate_host(int nx, int ny, int nz, const double* B, const double* k, double Ecut, int64_t cap,
                          int64_t* n_G_out, int64_t* mapping0, double* kinetic, int32_t* G_out);
int build_projectors_hgh(dftk_mi_basis* b, int64_t n_rows, const int32_t* G_d, const double* recip_h, const double* k_h,
                         double volume, int n_species, const double* rp_h, const int* nproj_h, int n_atoms,
                         const int* species_of_atom_h, const double* positions_h, cd* P_d, int64_t ldP, int* n_p_out);

int atomic_superposition(dftk_mi_kblock* cube_kb, int kind, const double* recip_h, int n_species, const double* par_h,
                         int n_atoms, const int* species_of_atom_h, const double* positions_h, double* out_d);

// gamma_kernels.hip
int gamma_tables_host(int nx, int ny, int nz, int64_t n_G, const int64_t* mapping, int64_t* n_half_out, int32_t* g,
                      int32_t* mg);
int gamma_enable(dftk_mi_kblock* kb, int on);
void gamma_destroy(GammaReal* gr);
int gamma_compress(dftk_mi_kblock* kb, int m, const cd* X, int64_t ldx, cd* H, int64_t ldh);
// ... after rotating every column by the global phase that maximises its real-symmetric part (LOBPCG entry)
int gamma_compress_aligned(dftk_mi_kblock* kb, int m, const cd* X, int64_t ldx, cd* H, int64_t ldh);
int gamma_expand(dftk_mi_kblock* kb, int m, const cd* H, int64_t ldh, cd* X, int64_t ldx);
int gamma_apply_H(dftk_mi_kblock* kb, int which, int nb, const cd* psi, int64_t ldpsi, cd* Hpsi, int64_t ldH);
int gamma_density(dftk_mi_kblock* kb, int nb, const cd* psi, int64_t ldpsi, const double* w_h, double* rho);
// local building blocks (whole bands on this rank)
int gamma_ensure_buf(dftk_mi_kblock* kb, size_t elems);
int gamma_pack_pairs(dftk_mi_kblock* kb, int nb, const cd* H, int64_t ldh, cd* Z, int64_t ldz);
int gamma_unpack_pairs(dftk_mi_kblock* kb, int nb, const cd* W, int64_t ldw, cd* H, int64_t ldh);
int gamma_pack_full(dftk_mi_kblock* kb, int nb, const cd* X, int64_t ldx, cd* Z, int64_t ldz);
int gamma_gather_P(dftk_mi_kblock* kb, int ncols, const cd* P, int64_t ldP, cd* Ph, int64_t ldh, double* asym_mag_h);
int gamma_density_bands(dftk_mi_kblock* kb, int nb, const cd* psi, int64_t ldpsi, const double* w_h, double* rho);
int64_t gamma_local_rows(const dftk_mi_kblock* kb);     // half-format rows held by this rank
int64_t gamma_row0(const dftk_mi_kblock* kb);
// api.cpp: the plane-wave sharded variants (slab <-> band all-to-alls around the local building blocks) and the
// entry / exit conversions of dftk_mi_lobpcg (caller's full-sphere block <-> half-format block, sharded or not)
int gamma_apply_H_sharded(dftk_mi_kblock* kb, int which, int nb, const cd* psi, int64_t ldpsi, cd* Hpsi, int64_t ldH);
int gamma_projectors_sharded(dftk_mi_kblock* kb);
int gamma_density_sharded(dftk_mi_kblock* kb, int nb, const cd* psi, int64_t ldpsi, const double* w_h, double* rho);
int gamma_lobpcg_load(dftk_mi_kblock* kb, int M, const cd* Xuser, int64_t ldX, cd* Xh, int64_t ldh, bool align = false);
int gamma_lobpcg_store(dftk_mi_kblock* kb, int M, const cd* Xh, int64_t ldh, cd* Xuser, int64_t ldX);
// api.cpp: Hpsi (+)= P D P' psi on `rows` rows of projector storage P (leading dimension ldP); gemm_flags is OR-ed
// into the two products (DFTK_MI_GEMM_REAL for half-format blocks); comm (nullable) all-reduces the projections
int apply_nonlocal_rows(dftk_mi_kblock* kb, int nb, const cd* P, int64_t ldP, int64_t rows, const cd* psi,
                        int64_t ldpsi, cd* Hpsi, int64_t ldH, bool accumulate, int gemm_flags, dftk_mi_comm* comm);

// lobpcg.cpp
// ortho!(X) (Cholesky-QR with the reference's shift-and-retry and SVD fallback) on a stand-alone block;
// force_svd = 1 takes the SVD branch directly (tests)
int lobpcg_ortho(dftk_mi_basis* b, int64_t n, int m, cd* X, int64_t ldx, int force_svd, int* n_chol, int* used_svd);
int lobpcg_run_multi(int n_kb, dftk_mi_kblock* const* kbs, int M, cd* const* X, const int64_t* ldX, double tol, int miniter,
                     int maxiter, int n_conv_check, int use_tpa, const uint64_t* seeds, double* lambda_h, double* resid_h,
                     int* n_iter, int* converged, int64_t* n_matvec, int* status);
int lobpcg_run(dftk_mi_kblock* kb, int M, cd* X, int64_t ldX, double tol, int miniter, int maxiter,
               int n_conv_check, int use_tpa, uint64_t seed, double* lambda_h, double* resid_h,
               int* n_iter, int* converged, int64_t* n_matvec);
