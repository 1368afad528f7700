// batch.h -- lock-step batching of many small k-blocks ("must batch across k-points and bands
// inside one launch to beat launch latency"; the reference's loop over k-points is sequential, src/eigen/diag.jl:24-48).
//
// The LOBPCG driver (lobpcg.cpp) stays the single statement of the algorithm.  dftk_mi_lobpcg_multi runs one instance
// of it per k-block as a FIBER on the calling thread.  While a fiber runs, every device operation it issues (the
// internal entry points zgemm, ew_*, dense_potrf_trtri, dense_heev, apply_H, host<->device copies) is RECORDED instead
// of launched; when the fiber needs a result on the host it yields.  Once every fiber has yielded (or finished) the
// recorded queues are merged position by position: operations of the same kind become ONE launch over all k-blocks
// (batched kernels of batch_kernels.hip / the multi-job FFT kernels), anything without a batched form runs as
// recorded, one after the other.  One stream synchronisation per round instead of one per k-block and operation.
#pragma once
#include "common.h"
#include <vector>

enum BOpType {
    BOP_ZGEMM = 0,
    BOP_COLRED,      // ew_colnorms / coldots / coldots_im / weighted_colsums / frob2 (k_col_reduce modes)
    BOP_RESIDUAL,
    BOP_TPA,
    BOP_SCALE,
    BOP_COPY,
    BOP_FILL0,
    BOP_SUBID,
    BOP_GATHER,
    BOP_ADDDIAG,
    BOP_HERMIT,
    BOP_CTRANS,
    BOP_H2D,         // small host -> device copy (payload copied at record time)
    BOP_D2H,         // device -> host copy, visible when the fiber resumes
    BOP_POTRF,       // dense_potrf_trtri (outputs on the host)
    BOP_HEEV,        // dense_heev (eigenvalues on the host)
    BOP_APPLYH,      // dftk_mi_apply_H_parts
    BOP_DENSITY,     // launch_density
    BOP_APPLYD,      // apply_D: Y = D X with the banded real D of a k-block (internal to the batched apply_H)
    BOP_ORTHO,       // fused ortho!(X) / ortho!(X, Y) of a small block, whole adaptive loop inside ONE kernel (batched form only)
    BOP_NTYPES
};

// Field use per type (all pointers are device pointers unless named host*):
//   ZGEMM    trans, gm, gn, gk, alpha, A, lda, B, ldb, beta, C, ldc, flags (upper | REAL bits of zgemm())
//   COLRED   mode (k_col_reduce), n rows, m columns, A = X, lda, B = Y, ldb, W = weights, C = out (doubles)
//   RESIDUAL n, m, A = AX, lda, B = X, ldb, W = lam, C = R, ldc, D = norms, W2 = kin, E = mean_kin out, F = <x,x> out
//   TPA      n, m, A = src, lda, C = dst, ldc, W = kin, W2 = mean_kin, D = norms, s0 = default shift
//   SCALE    n, m, C = X, ldc, W = s, flags = invert        COPY / GATHER  n, m, A = X, lda, C = Y, ldc (, W = perm)
//   FILL0    C, bytes     SUBID  n = rows, m = cols, C, ldc, i0 = row0     ADDDIAG m = n, C, ldc, s0 = shift
//   HERMIT   m = n, C, ldc        CTRANS  m = n, A, lda, C = B, ldc
//   H2D      C = destination, payload       D2H  A = source, host, bytes
//   POTRF    m = n, C = A, ldc, D = invR, ldb = ldi, host = double[2] {normest R, normest inv R}
//   HEEV     m = n, C = A, ldc, D = V, ldb = ldv, host = eigenvalues (n doubles)
//   APPLYH   kb, flags = which, m = bands, A = psi, lda, C = H psi, ldc
//   APPLYD   kb, m = bands, A = X (n_p x m), C = Y
//   DENSITY  kb, m = bands, A = psi, lda, C = rho, payload = m weights (+ m weights of the imaginary parts, flags = 1)
//   ORTHO    n rows, m columns of X (C, ldc; m <= 8), k columns of Y (A, lda; k <= 16, 0 = ortho!(X) alone), W = column norms of X
//            (optional), s0 = tol, host = double[4] {status, ortho!(X, Y) rounds, Cholesky count, growth factor} (after the round);
//            status 0 = done, 1 = a rare branch is needed (drop_small!, SVD fallback: X is NOT usable), 2 = non-finite input
// Extensions used by the small-block LOBPCG driver (lobpcg_run_small): RESIDUAL with W3 != null takes lam[c] = W[c] / W3[c];
// HEEV with E != null also leaves the eigenvalues in that DEVICE array (the residual pass of the same round reads them).
struct BOp {
    int type = 0;
    dftk_mi_basis* b = nullptr;
    dftk_mi_kblock* kb = nullptr;
    char trans = 'N';
    int64_t n = 0, lda = 0, ldb = 0, ldc = 0;
    int64_t gm = 0, gn = 0, gk = 0;   // ZGEMM: C (gm x gn) = alpha op(A) B + beta C, inner dimension gk
    int m = 0, k = 0, i0 = 0, mode = 0, flags = 0;
    cd alpha = {0.0, 0.0}, beta = {0.0, 0.0};
    double s0 = 0.0;
    const void *A = nullptr, *B = nullptr, *W = nullptr, *W2 = nullptr, *W3 = nullptr;
    void *C = nullptr, *D = nullptr, *E = nullptr, *F = nullptr;
    void* host = nullptr;        // D2H destination / POTRF, HEEV host outputs
    void* host2 = nullptr;
    size_t bytes = 0;
    std::vector<char> payload;   // H2D source (copied when recorded: the caller may reuse its buffer)
    bool join_next = false;      // the fiber's NEXT operation (same type, independent of this one) joins this one's launch
    int status = 0;              // numerical status of POTRF / HEEV (set by the executor)
    int* status_out = nullptr;   // where the issuing fiber reads it after resuming (on its own stack)
};

// true while the calling thread runs inside a fiber of a batched call
bool batching();
// enqueue on the current fiber; returns 0
int batch_record(BOp&& op);
// the operation recorded last and the one recorded next (same type, independent: e.g. X = Y c and AX = AY c) share a launch
void batch_join_next();
// enqueue, yield until the round has executed; returns the op's status
int batch_record_sync(BOp&& op);
// plain "wait for everything I have issued": yields (all queued work of the fiber is complete on return)
int batch_sync();
// host <-> device traffic of host-driven drivers: recorded inside a fiber of a batched call, plain HIP calls otherwise
int dev_h2d(dftk_mi_basis* b, void* dst_d, const void* src_h, size_t bytes);
int dev_d2h_sync(dftk_mi_basis* b, void* dst_h, const void* src_d, size_t bytes);
// the same without giving up the fiber: inside a batched call dst_h is valid after the fiber's NEXT synchronising call
// (the copy rides on that round); outside it is a plain synchronous fetch
int dev_d2h_async(dftk_mi_basis* b, void* dst_h, const void* src_d, size_t bytes);
int dev_stream_sync(dftk_mi_basis* b);

// batched executors (batch_kernels.hip / fft_kernels.hip); return 0 if the whole group was launched, 1 if the group
// has no batched form (the caller then runs the ops one by one), < 0 on errors
struct BatchCtx;   // staging rings + per-call scratch (batch.cpp)
int batch_exec_group(BatchCtx* ctx, hipStream_t stream, int type, std::vector<BOp*>& ops);
int batch_exec_apply_H(BatchCtx* ctx, hipStream_t stream, std::vector<BOp*>& ops);     // fft_kernels.hip
int batch_exec_density(BatchCtx* ctx, hipStream_t stream, std::vector<BOp*>& ops);     // fft_kernels.hip
// helpers for the executors
const void* batch_stage(BatchCtx* c, const void* src, size_t bytes);       // host table -> device (pinned ring + async copy)
void* batch_result_slot(BatchCtx* c, size_t bytes, void** host_twin);      // device slot + pinned host twin for results
size_t batch_result_room(BatchCtx* c);                                      // bytes of result slots left in this round
int batch_results_fetch(BatchCtx* c);                                       // enqueue ONE copy of all result slots of the round
// device scratch of the executors: a bump allocator.  A pointer stays valid until the enclosing BatchScratchScope ends
// (later operations of the round then reuse the space in stream order) or, without a scope, until the round ends; it is
// never freed or moved under a holder: a request that does not fit retires the buffer until the round's synchronisation
void* batch_scratch(BatchCtx* c, size_t bytes);
void batch_scratch_mark(BatchCtx* c, size_t* off, int* gen);
void batch_scratch_release(BatchCtx* c, size_t off, int gen);
struct BatchScratchScope {
    BatchCtx* c;
    size_t off;
    int gen;
    explicit BatchScratchScope(BatchCtx* ctx) : c(ctx) { batch_scratch_mark(c, &off, &gen); }
    ~BatchScratchScope() { batch_scratch_release(c, off, gen); }
    BatchScratchScope(const BatchScratchScope&) = delete;
    BatchScratchScope& operator=(const BatchScratchScope&) = delete;
};
#include <functional>
void batch_add_fixup(BatchCtx* c, std::function<void()> f);                // runs on the host after the round's sync
// run the bodies as fibers of one batched call on the stream of `b`
int batch_run(dftk_mi_basis* b, std::vector<std::function<int()>>& bodies, std::vector<int>& rets);
