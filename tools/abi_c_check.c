/* abi_c_check.c -- a plain-C (C99) consumer of include/dftk_mi355x.h.
 *
 * Every other caller in this repository reaches the library through Python ctypes, which never compiles the header and
 * never passes `dftk_mi_cplx` BY VALUE the way a C compiler (or Julia's `ccall` with an isbits struct) does.  This
 * program is compiled by gcc against the header and drives, for N processes (one per rank, forked here; rank r uses
 * HIP device r mod #devices):
 *
 *   dftk_mi_comm_get_unique_id -> dftk_mi_comm_init_rank -> dftk_mi_comm_describe (ncclCommCount == N)
 *   -> dftk_mi_allreduce_sum_f64                                   (mpi_sum!, src/common/mpi.jl:20 at src/densities.jl:46)
 *   -> dftk_mi_zgemm with complex alpha / beta by value             (checked against a host triple loop)
 *   -> dftk_mi_kblock_set_shard + dftk_mi_apply_H on the row slab   (== the rows of the unsharded mul!, Hamiltonian.jl:137-192)
 *   -> dftk_mi_density_accumulate + all-reduce                      (== the unsharded compute_density, densities.jl:35-47)
 *   -> dftk_mi_lobpcg on the sharded block                          (eigenvalues == the unsharded block's)
 *
 * Usage: abi_c_check [N = 1].  Exit code 0 and "abi_c_check OK ranks=N" on success; without a GPU the first library call
 * fails with DFTK_MI_ENOGPU and the program exits 3 (no CPU fallback anywhere).  RCCL needs one device per rank: on a
 * one-GPU box only N = 1 can run (exit 4 = "more ranks than devices", nothing tested).
 */
#define _POSIX_C_SOURCE 200809L
#include "dftk_mi355x.h"

#include <math.h>
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/types.h>
#include <sys/wait.h>
#include <unistd.h>

/* the four HIP runtime entry points a caller needs to own device buffers (libamdhip64), declared here so that the file
 * stays plain C without the HIP headers */
extern int hipGetDeviceCount(int* count);
extern int hipSetDevice(int device);
extern int hipMalloc(void** ptr, size_t size);
extern int hipFree(void* ptr);
extern int hipMemcpy(void* dst, const void* src, size_t bytes, int kind); /* 1 = host->device, 2 = device->host */
extern int hipDeviceSynchronize(void);

#define CHECK(call)                                                                                         \
    do {                                                                                                    \
        int st_ = (call);                                                                                   \
        if (st_ != 0) {                                                                                     \
            fprintf(stderr, "[rank %d] %s -> %d: %s\n", g_rank, #call, st_, dftk_mi_last_error());          \
            return st_ == DFTK_MI_ENOGPU ? 3 : 1;                                                           \
        }                                                                                                   \
    } while (0)
#define REQUIRE(cond, ...)                                        \
    do {                                                          \
        if (!(cond)) {                                            \
            fprintf(stderr, "[rank %d] FAILED: ", g_rank);        \
            fprintf(stderr, __VA_ARGS__);                         \
            fprintf(stderr, "\n");                                \
            return 1;                                             \
        }                                                         \
    } while (0)

static int g_rank = 0;

/* deterministic pseudo-random numbers in (-1, 1): identical on every rank */
static uint64_t g_seed = 0x9E3779B97F4A7C15ull;
static double rnd(void) {
    g_seed ^= g_seed << 13;
    g_seed ^= g_seed >> 7;
    g_seed ^= g_seed << 17;
    return (double)(g_seed >> 11) / 9007199254740992.0 * 2.0 - 1.0;
}

static void* dev_upload(const void* h, size_t bytes) {
    void* d = NULL;
    if (hipMalloc(&d, bytes ? bytes : 16) != 0) return NULL;
    if (h && bytes && hipMemcpy(d, h, bytes, 1) != 0) return NULL;   /* h == NULL: uninitialised output buffer */
    return d;
}

static int worker(int rank, int n_ranks, const char id[128]) {
    enum { NX = 24, NB = 3, NP = 4, M = 4 };
    const double L = 10.0, Ecut = 5.0, two_pi = 6.283185307179586;
    const double volume = L * L * L;
    int n_dev = 0, backend = -1, n_comm = -1, version = 0;
    int64_t n_G = 0, i, j, k;
    g_rank = rank;
    if (hipGetDeviceCount(&n_dev) != 0 || n_dev < 1) return 3;
    if (n_ranks > n_dev) return 4;
    const int device = rank % n_dev;
    if (hipSetDevice(device) != 0) return 1;

    dftk_mi_basis* basis = NULL;
    CHECK(dftk_mi_basis_create(NX, NX, NX, volume, device, &basis));

    /* ---- communicator: RCCL, one rank per process --------------------------------------------------------- */
    dftk_mi_comm* comm = NULL;
    CHECK(dftk_mi_comm_init_rank(id, n_ranks, rank, device, &comm));
    CHECK(dftk_mi_comm_describe(comm, &backend, &n_comm, &version));
    REQUIRE(backend == 0 && n_comm == n_ranks, "communicator reports backend %d with %d ranks, expected RCCL with %d",
            backend, n_comm, n_ranks);
    REQUIRE(dftk_mi_comm_rank(comm) == rank && dftk_mi_comm_size(comm) == n_ranks, "rank / size mismatch");
    /* one line per rank: what a node operator checks first (every rank on its own device, one RCCL version, N ranks met) */
    printf("[rank %d/%d] device %d of %d, RCCL %d, ncclCommCount %d\n", rank, n_ranks, device, n_dev, version, n_comm);
    fflush(stdout);
    {
        enum { NV = 1000 };
        double h[NV];
        for (i = 0; i < NV; ++i) h[i] = (double)(rank + 1) * (double)(i + 1);
        double* d = (double*)dev_upload(h, sizeof h);
        REQUIRE(d != NULL, "hipMalloc");
        CHECK(dftk_mi_allreduce_sum_f64(comm, d, NV, dftk_mi_basis_stream(basis)));
        CHECK(dftk_mi_basis_sync(basis));
        hipMemcpy(h, d, sizeof h, 2);
        for (i = 0; i < NV; ++i)
            REQUIRE(h[i] == 0.5 * n_ranks * (n_ranks + 1) * (double)(i + 1), "all-reduce entry %lld = %g", (long long)i, h[i]);
        hipFree(d);
    }

    /* ---- complex scalars BY VALUE: C = alpha A^H B + beta C -------------------------------------------------- */
    {
        enum { KK = 96, MM = 5, NN = 7 };
        static dftk_mi_cplx A[KK * MM], B[KK * NN], Cm[MM * NN], want[MM * NN];
        const dftk_mi_cplx alpha = {0.5, -0.25}, beta = {2.0, 1.0};
        for (i = 0; i < KK * MM; ++i) { A[i].re = rnd(); A[i].im = rnd(); }
        for (i = 0; i < KK * NN; ++i) { B[i].re = rnd(); B[i].im = rnd(); }
        for (i = 0; i < MM * NN; ++i) { Cm[i].re = rnd(); Cm[i].im = rnd(); }
        for (j = 0; j < NN; ++j)
            for (i = 0; i < MM; ++i) {
                double sr = 0, si = 0;
                for (k = 0; k < KK; ++k) {   /* conj(A[k,i]) * B[k,j] */
                    const dftk_mi_cplx a = A[k + KK * i], b = B[k + KK * j];
                    sr += a.re * b.re + a.im * b.im;
                    si += a.re * b.im - a.im * b.re;
                }
                const dftk_mi_cplx c = Cm[i + MM * j];
                want[i + MM * j].re = alpha.re * sr - alpha.im * si + beta.re * c.re - beta.im * c.im;
                want[i + MM * j].im = alpha.re * si + alpha.im * sr + beta.re * c.im + beta.im * c.re;
            }
        dftk_mi_cplx *dA = (dftk_mi_cplx*)dev_upload(A, sizeof A), *dB = (dftk_mi_cplx*)dev_upload(B, sizeof B),
                     *dC = (dftk_mi_cplx*)dev_upload(Cm, sizeof Cm);
        REQUIRE(dA && dB && dC, "hipMalloc");
        CHECK(dftk_mi_zgemm(basis, 'C', MM, NN, KK, alpha, dA, KK, dB, KK, beta, dC, MM));
        CHECK(dftk_mi_basis_sync(basis));
        hipMemcpy(Cm, dC, sizeof Cm, 2);
        for (i = 0; i < MM * NN; ++i)
            REQUIRE(fabs(Cm[i].re - want[i].re) < 1e-12 && fabs(Cm[i].im - want[i].im) < 1e-12,
                    "zgemm with by-value alpha/beta: entry %lld = (%g, %g), expected (%g, %g)", (long long)i, Cm[i].re,
                    Cm[i].im, want[i].re, want[i].im);
        hipFree(dA); hipFree(dB); hipFree(dC);
    }

    /* ---- Kpoint sphere (Kpoint.jl:20-41), cubic cell of edge L, Gamma point ---------------------------------- */
    double recip[9] = {0}, kcoord[3] = {0, 0, 0};
    recip[0] = recip[4] = recip[8] = two_pi / L;
    CHECK(dftk_mi_kpoint_sphere_host(NX, NX, NX, recip, kcoord, Ecut, 0, &n_G, NULL, NULL, NULL));
    REQUIRE(n_G > 3 * M && n_G < NX * NX * NX, "sphere of %lld plane waves", (long long)n_G);
    int64_t* mapping = (int64_t*)malloc((size_t)n_G * sizeof(int64_t));
    double* kin = (double*)malloc((size_t)n_G * sizeof(double));
    int32_t* Gv = (int32_t*)malloc((size_t)n_G * 3 * sizeof(int32_t));
    CHECK(dftk_mi_kpoint_sphere_host(NX, NX, NX, recip, kcoord, Ecut, n_G, &n_G, mapping, kin, Gv));

    /* rows [row0, row1) of the sphere belong to this rank (split_evenly, PlaneWaveBasis.jl:208-209) */
    int64_t* row_starts = (int64_t*)malloc((size_t)(n_ranks + 1) * sizeof(int64_t));
    for (i = 0; i <= n_ranks; ++i) row_starts[i] = (n_G * i) / n_ranks;
    const int64_t row0 = row_starts[rank], n_loc = row_starts[rank + 1] - row0;

    dftk_mi_kblock *kb_full = NULL, *kb_shard = NULL;
    CHECK(dftk_mi_kblock_create(basis, n_G, mapping, kin, &kb_full));
    CHECK(dftk_mi_kblock_create(basis, n_G, mapping, kin, &kb_shard));
    CHECK(dftk_mi_kblock_set_shard(kb_shard, comm, row_starts));

    /* local potential on the cube, projectors P (n_G x NP) with a diagonal coupling matrix */
    const size_t N = (size_t)NX * NX * NX;
    double* V = (double*)malloc(N * sizeof(double));
    for (i = 0; i < (int64_t)N; ++i) V[i] = 0.3 * rnd();
    double* dV = (double*)dev_upload(V, N * sizeof(double));
    dftk_mi_cplx* P = (dftk_mi_cplx*)malloc((size_t)n_G * NP * sizeof(dftk_mi_cplx));
    for (i = 0; i < n_G * NP; ++i) { P[i].re = 0.2 * rnd(); P[i].im = 0.2 * rnd(); }
    double D[NP * NP] = {0};
    for (i = 0; i < NP; ++i) D[i + NP * i] = 0.1 * (double)(i + 1);
    dftk_mi_cplx* dP = (dftk_mi_cplx*)dev_upload(P, (size_t)n_G * NP * sizeof(dftk_mi_cplx));
    dftk_mi_cplx* Pslab = (dftk_mi_cplx*)malloc((size_t)n_loc * NP * sizeof(dftk_mi_cplx));
    for (j = 0; j < NP; ++j) memcpy(Pslab + n_loc * j, P + row0 + n_G * j, (size_t)n_loc * sizeof(dftk_mi_cplx));
    dftk_mi_cplx* dPslab = (dftk_mi_cplx*)dev_upload(Pslab, (size_t)n_loc * NP * sizeof(dftk_mi_cplx));
    REQUIRE(dV && dP && dPslab, "hipMalloc");
    CHECK(dftk_mi_kblock_set_potential(kb_full, dV));
    CHECK(dftk_mi_kblock_set_potential(kb_shard, dV));
    CHECK(dftk_mi_kblock_set_projectors(kb_full, NP, dP, n_G, D));
    CHECK(dftk_mi_kblock_set_projectors(kb_shard, NP, dPslab, n_loc, D));

    /* ---- mul!(Hpsi, H, psi): slab of the sharded apply == rows of the unsharded one --------------------------- */
    dftk_mi_cplx* psi = (dftk_mi_cplx*)malloc((size_t)n_G * NB * sizeof(dftk_mi_cplx));
    for (i = 0; i < n_G * NB; ++i) { psi[i].re = rnd(); psi[i].im = rnd(); }
    dftk_mi_cplx* psi_slab = (dftk_mi_cplx*)malloc((size_t)n_loc * NB * sizeof(dftk_mi_cplx));
    for (j = 0; j < NB; ++j) memcpy(psi_slab + n_loc * j, psi + row0 + n_G * j, (size_t)n_loc * sizeof(dftk_mi_cplx));
    dftk_mi_cplx *d_psi = (dftk_mi_cplx*)dev_upload(psi, (size_t)n_G * NB * sizeof(dftk_mi_cplx)),
                 *d_slab = (dftk_mi_cplx*)dev_upload(psi_slab, (size_t)n_loc * NB * sizeof(dftk_mi_cplx)),
                 *d_H = (dftk_mi_cplx*)dev_upload(NULL, (size_t)n_G * NB * sizeof(dftk_mi_cplx)),
                 *d_Hslab = (dftk_mi_cplx*)dev_upload(NULL, (size_t)n_loc * NB * sizeof(dftk_mi_cplx));
    REQUIRE(d_psi && d_slab && d_H && d_Hslab, "hipMalloc");
    CHECK(dftk_mi_apply_H(kb_full, NB, d_psi, n_G, d_H, n_G));
    CHECK(dftk_mi_apply_H(kb_shard, NB, d_slab, n_loc, d_Hslab, n_loc));
    CHECK(dftk_mi_basis_sync(basis));
    dftk_mi_cplx* H = (dftk_mi_cplx*)malloc((size_t)n_G * NB * sizeof(dftk_mi_cplx));
    dftk_mi_cplx* Hslab = (dftk_mi_cplx*)malloc((size_t)n_loc * NB * sizeof(dftk_mi_cplx));
    hipMemcpy(H, d_H, (size_t)n_G * NB * sizeof(dftk_mi_cplx), 2);
    hipMemcpy(Hslab, d_Hslab, (size_t)n_loc * NB * sizeof(dftk_mi_cplx), 2);
    {
        double err = 0, nrm = 0;
        for (j = 0; j < NB; ++j)
            for (i = 0; i < n_loc; ++i) {
                const dftk_mi_cplx a = Hslab[i + n_loc * j], b = H[row0 + i + n_G * j];
                err += (a.re - b.re) * (a.re - b.re) + (a.im - b.im) * (a.im - b.im);
                nrm += b.re * b.re + b.im * b.im;
            }
        REQUIRE(nrm > 0 && sqrt(err / nrm) < 1e-12, "sharded H psi differs from the unsharded rows: %g", sqrt(err / nrm));
    }

    /* ---- compute_density: this rank's bands + all-reduce == unsharded ------------------------------------------ */
    {
        const double w[NB] = {2.0 / volume, 1.5 / volume, 0.25 / volume};
        double *d_rho_full = (double*)dev_upload(NULL, N * sizeof(double)), *d_rho = (double*)dev_upload(NULL, N * sizeof(double));
        double* zeros = (double*)calloc(N, sizeof(double));
        REQUIRE(d_rho_full && d_rho && zeros, "alloc");
        hipMemcpy(d_rho_full, zeros, N * sizeof(double), 1);
        hipMemcpy(d_rho, zeros, N * sizeof(double), 1);
        CHECK(dftk_mi_density_accumulate(kb_full, NB, d_psi, n_G, w, d_rho_full));
        CHECK(dftk_mi_density_accumulate(kb_shard, NB, d_slab, n_loc, w, d_rho));
        CHECK(dftk_mi_allreduce_sum_f64(comm, d_rho, N, dftk_mi_basis_stream(basis)));
        CHECK(dftk_mi_basis_sync(basis));
        double *a = (double*)malloc(N * sizeof(double)), *b = (double*)malloc(N * sizeof(double));
        hipMemcpy(a, d_rho, N * sizeof(double), 2);
        hipMemcpy(b, d_rho_full, N * sizeof(double), 2);
        double err = 0, nrm = 0;
        for (i = 0; i < (int64_t)N; ++i) { err += (a[i] - b[i]) * (a[i] - b[i]); nrm += b[i] * b[i]; }
        REQUIRE(nrm > 0 && sqrt(err / nrm) < 1e-12, "sharded density differs from the unsharded one: %g", sqrt(err / nrm));
        free(a); free(b); free(zeros); hipFree(d_rho); hipFree(d_rho_full);
    }

    /* ---- lobpcg_hyper on the sharded block: same eigenvalues as the unsharded block ---------------------------- */
    {
        double lam_f[M], lam_s[M], res[M];
        int n_iter = 0, conv = 0;
        int64_t n_matvec = 0;
        dftk_mi_cplx* X = (dftk_mi_cplx*)malloc((size_t)n_G * M * sizeof(dftk_mi_cplx));
        for (i = 0; i < n_G * M; ++i) { X[i].re = rnd(); X[i].im = rnd(); }
        dftk_mi_cplx* Xs = (dftk_mi_cplx*)malloc((size_t)n_loc * M * sizeof(dftk_mi_cplx));
        for (j = 0; j < M; ++j) memcpy(Xs + n_loc * j, X + row0 + n_G * j, (size_t)n_loc * sizeof(dftk_mi_cplx));
        dftk_mi_cplx *dX = (dftk_mi_cplx*)dev_upload(X, (size_t)n_G * M * sizeof(dftk_mi_cplx)),
                     *dXs = (dftk_mi_cplx*)dev_upload(Xs, (size_t)n_loc * M * sizeof(dftk_mi_cplx));
        REQUIRE(dX && dXs, "hipMalloc");
        CHECK(dftk_mi_lobpcg(kb_full, M, dX, n_G, 1e-9, 1, 200, M, 1, 7, lam_f, res, &n_iter, &conv, &n_matvec));
        REQUIRE(conv == 1, "unsharded LOBPCG did not converge");
        CHECK(dftk_mi_lobpcg(kb_shard, M, dXs, n_loc, 1e-9, 1, 200, M, 1, 7, lam_s, res, &n_iter, &conv, &n_matvec));
        REQUIRE(conv == 1, "sharded LOBPCG did not converge");
        for (i = 0; i < M; ++i)
            REQUIRE(fabs(lam_f[i] - lam_s[i]) < 1e-8, "eigenvalue %lld: %.12f (one block) vs %.12f (sharded)", (long long)i,
                    lam_f[i], lam_s[i]);
        free(X); free(Xs); hipFree(dX); hipFree(dXs);
    }

    CHECK(dftk_mi_kblock_destroy(kb_shard));
    CHECK(dftk_mi_kblock_destroy(kb_full));
    CHECK(dftk_mi_comm_destroy(comm));
    CHECK(dftk_mi_basis_destroy(basis));
    hipFree(dV); hipFree(dP); hipFree(dPslab); hipFree(d_psi); hipFree(d_slab); hipFree(d_H); hipFree(d_Hslab);
    free(mapping); free(kin); free(Gv); free(row_starts); free(V); free(P); free(Pslab); free(psi); free(psi_slab);
    free(H); free(Hslab);
    if (rank == 0)
        printf("abi_c_check OK ranks=%d rccl=%d n_G=%lld (%s)\n", n_ranks, version, (long long)n_G, dftk_mi_version());
    fflush(stdout);   /* the ranks leave through _exit */
    return 0;
}

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 1;
    if (n < 1 || n > 64) {
        fprintf(stderr, "usage: %s [n_ranks = 1]\n", argv[0]);
        return 2;
    }
    /* the parent never touches HIP (fork after HIP initialisation is undefined): rank 0 creates the unique id and
     * the parent relays its 128 bytes to the other ranks -- "any side channel" of the header's contract */
    int (*up)[2] = malloc((size_t)n * sizeof *up), (*down)[2] = malloc((size_t)n * sizeof *down);
    pid_t* pids = malloc((size_t)n * sizeof(pid_t));
    int r, status = 0;
    signal(SIGPIPE, SIG_IGN);   /* a rank that has already given up (no GPU) must not kill the parent relaying the id */
    for (r = 0; r < n; ++r) {
        if (pipe(up[r]) != 0 || pipe(down[r]) != 0) return 2;
        pids[r] = fork();
        if (pids[r] < 0) return 2;
        if (pids[r] == 0) {
            char id[128];
            close(up[r][0]);
            close(down[r][1]);
            if (r == 0) {
                int n_dev = 0;
                g_rank = 0;
                if (hipGetDeviceCount(&n_dev) != 0 || n_dev < 1) {
                    fprintf(stderr, "abi_c_check: no HIP device -- the library has no CPU fallback (DFTK_MI_ENOGPU)\n");
                    memset(id, 0, sizeof id);
                    if (write(up[0][1], id, sizeof id) != (ssize_t)sizeof id) _exit(2);
                    _exit(3);
                }
                const int st = dftk_mi_comm_get_unique_id(id);
                if (st != 0) {
                    fprintf(stderr, "[rank 0] dftk_mi_comm_get_unique_id -> %d: %s\n", st, dftk_mi_last_error());
                    memset(id, 0, sizeof id);
                    if (write(up[0][1], id, sizeof id) != (ssize_t)sizeof id) _exit(2);
                    _exit(st == DFTK_MI_ENOGPU ? 3 : 1);
                }
                if (write(up[0][1], id, sizeof id) != (ssize_t)sizeof id) _exit(2);
            }
            if (read(down[r][0], id, sizeof id) != (ssize_t)sizeof id) _exit(2);
            _exit(worker(r, n, id));
        }
        close(up[r][1]);
        close(down[r][0]);
    }
    {
        char id[128];
        if (read(up[0][0], id, sizeof id) != (ssize_t)sizeof id) memset(id, 0, sizeof id);
        for (r = 0; r < n; ++r)
            if (write(down[r][1], id, sizeof id) != (ssize_t)sizeof id) status = status ? status : 0;   /* rank gone: its exit code tells */
    }
    for (r = 0; r < n; ++r) {
        int ws = 0;
        waitpid(pids[r], &ws, 0);
        const int code = WIFEXITED(ws) ? WEXITSTATUS(ws) : 1;
        if (code != 0 && status == 0) status = code;
    }
    if (status == 4) fprintf(stderr, "abi_c_check: %d ranks need %d HIP devices (RCCL: one device per rank)\n", n, n);
    free(up); free(down); free(pids);
    return status;
}
