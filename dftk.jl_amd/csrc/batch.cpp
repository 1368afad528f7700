// batch.cpp -- fibers, recording and merged execution for many small k-blocks (see batch.h).
#include "batch.h"
#include <ucontext.h>
#include <memory>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>

namespace {
struct Fiber {
    ucontext_t ctx;
    char* stack = nullptr;      // borrowed from the thread's stack pool (fiber_stack)
    std::vector<BOp> fifo;
    size_t head = 0;
    bool done = false;
    int ret = 0;
    std::function<int()> body;
};
// Fiber stacks are pooled per host thread and never returned: a value-initialised 1 MiB vector per fiber and call was
// 72 MiB of page faults (and an munmap) per batched call of a 72-k-point workload -- more than the call's device work.
const size_t FIBER_STACK_BYTES = 1u << 20;
char* fiber_stack(size_t i) {
    static thread_local std::vector<std::unique_ptr<char[]>> pool;
    while (pool.size() <= i) pool.emplace_back(new char[FIBER_STACK_BYTES]);   // (uninitialised: pages touched on use)
    return pool[i].get();
}
}  // namespace

// staging for batched launches: argument tables travel host -> device through one pinned ring, small results come
// back through another; both are reset after the stream synchronisation that ends a round
struct BatchCtx {
    hipStream_t stream = nullptr;
    char *h_ring = nullptr, *d_ring = nullptr;
    size_t cap = 0, off = 0;
    char *h_res = nullptr, *d_res = nullptr;
    bool res_mapped = false;     // d_res is the device view of h_res (zero-copy result slots)
    size_t res_cap = 0, res_off = 0;
    // device scratch of the executors: a bump allocator per round (every batch_scratch() pointer stays valid until the
    // round's stream synchronisation; a request that does not fit opens a larger buffer and RETIRES the current one,
    // which is freed only after that synchronisation)
    void* d_scratch = nullptr;
    size_t scratch_bytes = 0, scratch_off = 0;
    int scratch_gen = 0;             // bumped when a larger buffer replaces the current one (see BatchScratchScope)
    std::vector<void*> retired;
    std::vector<std::function<void()>> fixups;   // run after the round's stream synchronisation
    bool failed = false;
    int64_t n_rounds = 0, n_ops = 0, n_launch_groups = 0, n_sequential = 0;   // dftk_mi_batch_stats
    double t_fibers = 0, t_exec = 0, t_sync = 0, t_fix = 0;                     // DFTK_MI_KBATCH_TRACE: seconds per phase
};

const void* batch_stage(BatchCtx* c, const void* src, size_t bytes) {
    const size_t al = (bytes + 255) & ~(size_t)255;
    if (c->off + al > c->cap) {
        // ring exhausted inside a round: drain the stream (the tables already queued are consumed) and start over
        if (hipStreamSynchronize(c->stream) != hipSuccess) c->failed = true;
        c->off = 0;
        if (al > c->cap) {
            c->failed = true;
            return nullptr;
        }
    }
    memcpy(c->h_ring + c->off, src, bytes);
    // small tables are read by the kernels straight from the pinned ring (host memory mapped into the device's address
    // space): no copy to enqueue in front of every launch; large ones (job tables of thousands of bands) are copied
    const bool zero_copy = true;
    const void* d;
    if (zero_copy && bytes <= 16384) {
        d = c->h_ring + c->off;
    } else {
        if (hipMemcpyAsync(c->d_ring + c->off, c->h_ring + c->off, bytes, hipMemcpyHostToDevice, c->stream) != hipSuccess)
            c->failed = true;
        d = c->d_ring + c->off;
    }
    c->off += al;
    return d;
}
void* batch_result_slot(BatchCtx* c, size_t bytes, void** host_twin) {
    const size_t al = (bytes + 255) & ~(size_t)255;
    if (c->res_off + al > c->res_cap) {
        c->failed = true;
        return nullptr;
    }
    void* d = c->d_res + c->res_off;
    *host_twin = c->h_res + c->res_off;
    c->res_off += al;
    return d;
}
size_t batch_result_room(BatchCtx* c) { return c->res_cap - c->res_off; }
int batch_results_fetch(BatchCtx* c) {
    if (c->res_off == 0 || c->res_mapped) return 0;
    HIPCHK(hipMemcpyAsync(c->h_res, c->d_res, c->res_off, hipMemcpyDeviceToHost, c->stream));
    return 0;
}
void batch_add_fixup(BatchCtx* c, std::function<void()> f) { c->fixups.push_back(std::move(f)); }
void* batch_scratch(BatchCtx* c, size_t bytes) {
    const size_t al = (bytes + 255) & ~(size_t)255;
    if (c->scratch_off + al > c->scratch_bytes) {
        // nothing queued on the stream may lose its buffer: the old one is retired, not freed (batch_scratch_reset)
        if (c->d_scratch) c->retired.push_back(c->d_scratch);
        c->d_scratch = nullptr;
        const size_t want = std::max(al + al / 4, 2 * c->scratch_bytes);
        c->scratch_bytes = c->scratch_off = 0;
        c->scratch_gen += 1;
        if (dftk_scratch_malloc(&c->d_scratch, want) != hipSuccess) {
            c->d_scratch = nullptr;
            c->failed = true;
            return nullptr;
        }
        c->scratch_bytes = want;
    }
    void* p = static_cast<char*>(c->d_scratch) + c->scratch_off;
    c->scratch_off += al;
    return p;
}
// Executors bracket their requests with a scope: what an executor took is handed out again to the NEXT operation of the
// round (stream order makes the reuse safe), while requests made INSIDE the scope (the split-K partials of the products an
// apply_H executor runs) stack on top of it.  A mark taken in a buffer that has been replaced since releases to offset 0
// of the new one (the outer allocation lives in the retired buffer).
void batch_scratch_mark(BatchCtx* c, size_t* off, int* gen) {
    *off = c->scratch_off;
    *gen = c->scratch_gen;
}
void batch_scratch_release(BatchCtx* c, size_t off, int gen) { c->scratch_off = gen == c->scratch_gen ? off : 0; }
// end of a round, AFTER its stream synchronisation: everything handed out is dead
static void batch_scratch_reset(BatchCtx* c) {
    c->scratch_off = 0;
    for (void* p : c->retired) hipFree(p);
    c->retired.clear();
}

namespace {
struct Recorder {
    std::vector<Fiber> fibers;
    int cur = -1;
    ucontext_t main_ctx;
    BatchCtx ctx;
};
thread_local Recorder* g_rec = nullptr;
thread_local bool g_suspended = false;   // executors call the original entry points: recording is off meanwhile

void fiber_entry(unsigned lo, unsigned hi) {
    Fiber* f = reinterpret_cast<Fiber*>(((uintptr_t)hi << 32) | (uintptr_t)lo);
    f->ret = f->body();
    f->done = true;   // falls back to the scheduler through uc_link
}

// the original entry point of a recorded operation (recording suspended by the caller)
int exec_one(BOp& o) {
    cd* Cc = reinterpret_cast<cd*>(o.C);
    switch (o.type) {
        case BOP_ZGEMM:
            return zgemm(o.b, o.trans, o.gm, o.gn, o.gk, o.alpha, (const cd*)o.A, o.lda, (const cd*)o.B, o.ldb, o.beta, Cc,
                         o.ldc, o.flags);
        case BOP_COLRED:
            switch (o.mode) {
                case 0: return ew_colnorms(o.b, o.n, o.m, (const cd*)o.A, o.lda, (double*)o.C);
                case 1: return ew_coldots(o.b, o.n, o.m, (const cd*)o.A, o.lda, (const cd*)o.B, o.ldb, (double*)o.C);
                case 2: return ew_weighted_colsums(o.b, o.n, o.m, (const cd*)o.A, o.lda, (const double*)o.W, (double*)o.C);
                case 3: return ew_frob2(o.b, o.n, o.m, (const cd*)o.A, o.lda, (double*)o.C);
                default: return ew_coldots_im(o.b, o.n, o.m, (const cd*)o.A, o.lda, (const cd*)o.B, o.ldb, (double*)o.C);
            }
        case BOP_RESIDUAL:
            return ew_residual(o.b, o.n, o.m, (const cd*)o.A, o.lda, (const cd*)o.B, o.ldb, (const double*)o.W, Cc, o.ldc,
                               (double*)o.D, (const double*)o.W2, (double*)o.E, (double*)o.F);
        case BOP_TPA:
            return ew_tpa(o.b, o.n, o.m, (const cd*)o.A, o.lda, Cc, o.ldc, (const double*)o.W, (const double*)o.W2,
                          (double*)o.D, o.s0);
        case BOP_SCALE: return ew_scale_cols(o.b, o.n, o.m, Cc, o.ldc, (const double*)o.W, o.flags != 0);
        case BOP_COPY: return ew_copy(o.b, o.n, o.m, (const cd*)o.A, o.lda, Cc, o.ldc);
        case BOP_FILL0: return ew_fill_zero(o.b, Cc, o.bytes / sizeof(cd));
        case BOP_SUBID: return ew_sub_identity_shifted(o.b, (int)o.n, o.m, Cc, o.ldc, o.i0);
        case BOP_GATHER: return ew_gather_cols(o.b, o.n, o.m, (const cd*)o.A, o.lda, (const int*)o.W, Cc, o.ldc);
        case BOP_ADDDIAG: return ew_add_diag(o.b, o.m, Cc, o.ldc, o.s0);
        case BOP_HERMIT: return ew_hermitize_upper(o.b, o.m, Cc, o.ldc);
        case BOP_CTRANS: return ew_conj_transpose(o.b, o.m, (const cd*)o.A, o.lda, Cc, o.ldc);
        case BOP_H2D:
            // (pageable source: staged by the runtime before the call returns; the payload outlives the round anyway)
            HIPCHK(hipMemcpyAsync(o.C, o.payload.data(), o.payload.size(), hipMemcpyHostToDevice, o.b->stream));
            return 0;
        case BOP_D2H:
            HIPCHK(hipMemcpyAsync(o.host, o.A, o.bytes, hipMemcpyDeviceToHost, o.b->stream));
            return 0;
        case BOP_POTRF: {
            double* out = reinterpret_cast<double*>(o.host);
            o.status = dense_potrf_trtri(o.b, o.m, Cc, o.ldc, (cd*)o.D, o.ldb, out, out + 1);
            return (o.status < 0) ? o.status : 0;   // numerical (> 0) results travel to the fiber in o.status
        }
        case BOP_HEEV:
            o.status = dense_heev(o.b, o.m, Cc, o.ldc, (double*)o.host, (cd*)o.D, o.ldb);
            return (o.status < 0) ? o.status : 0;
        case BOP_APPLYH:
            return dftk_mi_apply_H_parts(o.kb, o.flags, o.m, (const dftk_mi_cplx*)o.A, o.lda, (dftk_mi_cplx*)o.C, o.ldc);
        case BOP_APPLYD: return apply_D(o.kb, o.m, (const cd*)o.A, Cc);
        case BOP_DENSITY: {
            const double* w = reinterpret_cast<const double*>(o.payload.data());
            const double* wim = (o.flags & 1) ? w + o.m : nullptr;
            const double* w2 = (o.flags & 2) ? w + (size_t)((o.flags & 1) ? 2 : 1) * o.m : nullptr;
            return launch_density(o.kb, o.m, (const cd*)o.A, o.lda, w, (double*)o.C, wim, w2, (double*)o.D);
        }
        default: return DFTK_MI_EINVAL;
    }
}

// Merge the queues position by position and execute; ONE stream synchronisation at the end of the round.
int flush(Recorder* r) {
    BatchCtx* c = &r->ctx;
    const bool no_merge = getenv("DFTK_MI_KBATCH_SEQUENTIAL") != nullptr;   // debugging: every op as recorded
    int status = 0;
    g_suspended = true;
    const auto tp0 = std::chrono::steady_clock::now();
    std::vector<BOp*> groups[BOP_NTYPES];
    std::vector<BOp*> all;
    for (;;) {
        bool any = false;
        for (auto& g : groups) g.clear();
        for (auto& f : r->fibers)
            while (f.head < f.fifo.size()) {
                BOp* op = &f.fifo[f.head++];
                groups[op->type].push_back(op);
                all.push_back(op);
                any = true;
                // (batch_join_next: the fiber's next operation is independent of this one and of the same type)
                if (!(op->join_next && f.head < f.fifo.size() && f.fifo[f.head].type == op->type)) break;
            }
        if (!any) break;
        for (int t = 0; t < BOP_NTYPES && status == 0; ++t) {
            auto& g = groups[t];
            if (g.empty()) continue;
            c->n_ops += (int64_t)g.size();
            int st = 1;
            if (!no_merge) st = batch_exec_group(c, c->stream, t, g);
            if (st < 0) status = st;
            if (st == 0) c->n_launch_groups += 1;
            if (st == 1) {
                c->n_sequential += (int64_t)g.size();
                for (BOp* op : g) {
                    const int s1 = exec_one(*op);
                    if (s1 != 0) {
                        status = s1;
                        break;
                    }
                }
            }
        }
        if (status != 0) break;
    }
    if (status == 0 && batch_results_fetch(c) != 0) status = DFTK_MI_EHIP;
    g_suspended = false;
    const auto tp1 = std::chrono::steady_clock::now();
    c->t_exec += std::chrono::duration<double>(tp1 - tp0).count();
    if (hipStreamSynchronize(c->stream) != hipSuccess) {
        dftk_set_error("batched round: stream synchronisation failed: %s", hipGetErrorString(hipGetLastError()));
        status = status ? status : DFTK_MI_EHIP;
    }
    if (c->failed && status == 0) {
        dftk_set_error("batched round: staging buffers exhausted or a copy failed");
        status = DFTK_MI_EHIP;
    }
    const auto tp2 = std::chrono::steady_clock::now();
    c->t_sync += std::chrono::duration<double>(tp2 - tp1).count();
    if (status == 0) {
        for (auto& fx : c->fixups) fx();
        for (BOp* op : all)
            if (op->status_out) *op->status_out = op->status;
    }
    c->t_fix += std::chrono::duration<double>(std::chrono::steady_clock::now() - tp2).count();
    c->fixups.clear();
    c->off = 0;
    c->res_off = 0;
    batch_scratch_reset(c);
    c->n_rounds += 1;
    for (auto& f : r->fibers) {
        f.fifo.clear();
        f.head = 0;
    }
    return status;
}

void yield_to_main() {
    Recorder* r = g_rec;
    Fiber& f = r->fibers[r->cur];
    swapcontext(&f.ctx, &r->main_ctx);
}
}  // namespace

bool batching() { return g_rec != nullptr && !g_suspended && g_rec->cur >= 0; }

int batch_record(BOp&& op) {
    Recorder* r = g_rec;
    r->fibers[r->cur].fifo.push_back(std::move(op));
    return 0;
}

void batch_join_next() {
    Recorder* r = g_rec;
    if (!r || g_suspended || r->cur < 0) return;
    auto& q = r->fibers[r->cur].fifo;
    if (!q.empty()) q.back().join_next = true;
}

int batch_record_sync(BOp&& op) {
    int st = 0;   // lives on the fiber's stack; the scheduler writes it after the round
    op.status_out = &st;
    Recorder* r = g_rec;
    r->fibers[r->cur].fifo.push_back(std::move(op));
    yield_to_main();
    return st;
}

int batch_sync() {
    yield_to_main();
    return 0;
}

int dev_h2d(dftk_mi_basis* b, void* dst_d, const void* src_h, size_t bytes) {
    if (batching()) {
        BOp o;
        o.b = b;
        o.type = BOP_H2D;
        o.C = dst_d;
        o.payload.assign(reinterpret_cast<const char*>(src_h), reinterpret_cast<const char*>(src_h) + bytes);
        return batch_record(std::move(o));
    }
    HIPCHK(hipMemcpyAsync(dst_d, src_h, bytes, hipMemcpyHostToDevice, b->stream));
    return 0;
}
int dev_d2h_async(dftk_mi_basis* b, void* dst_h, const void* src_d, size_t bytes) {
    if (batching()) {
        BOp o;
        o.b = b;
        o.type = BOP_D2H;
        o.A = src_d;
        o.host = dst_h;
        o.bytes = bytes;
        return batch_record(std::move(o));
    }
    return host_fetch(b, dst_h, src_d, bytes);
}
int dev_stream_sync(dftk_mi_basis* b) {
    if (batching()) return batch_sync();
    return host_wait(b);
}
int dev_d2h_sync(dftk_mi_basis* b, void* dst_h, const void* src_d, size_t bytes) {
    if (batching()) {
        BOp o;
        o.b = b;
        o.type = BOP_D2H;
        o.A = src_d;
        o.host = dst_h;
        o.bytes = bytes;
        return batch_record_sync(std::move(o));
    }
    return host_fetch(b, dst_h, src_d, bytes);
}

static thread_local int64_t g_last_stats[4] = {0, 0, 0, 0};
extern "C" int dftk_mi_batch_stats(int64_t* rounds, int64_t* ops, int64_t* merged_launches, int64_t* sequential_ops) {
    if (rounds) *rounds = g_last_stats[0];
    if (ops) *ops = g_last_stats[1];
    if (merged_launches) *merged_launches = g_last_stats[2];
    if (sequential_ops) *sequential_ops = g_last_stats[3];
    return 0;
}

int batch_run(dftk_mi_basis* b, std::vector<std::function<int()>>& bodies, std::vector<int>& rets) {
    if (g_rec) {
        dftk_set_error("batched calls do not nest");
        return DFTK_MI_EINVAL;
    }
    Recorder rec;
    BatchCtx* c = &rec.ctx;
    c->stream = b->stream;
    // staging rings and scratch are kept per (thread, device) across calls: pinned allocations cost milliseconds, a
    // batched call of a k-point workload a few
    struct Pool {
        int device = -1;
        char *h_ring = nullptr, *d_ring = nullptr, *h_res = nullptr, *d_res = nullptr;
        void* d_scratch = nullptr;
        size_t scratch_bytes = 0;
        bool res_mapped = false;
    };
    static thread_local Pool pool;
    const size_t cap = 16u << 20, res_cap = 8u << 20;
    if (pool.device != b->device) {
        if (pool.h_ring) {
            hipHostFree(pool.h_ring);
            hipFree(pool.d_ring);
            hipHostFree(pool.h_res);
            if (!pool.res_mapped) hipFree(pool.d_res);
            if (pool.d_scratch) hipFree(pool.d_scratch);
            pool = Pool();
        }
        HIPCHK(hipHostMalloc((void**)&pool.h_ring, cap));
        HIPCHK(hipMalloc((void**)&pool.d_ring, cap));
        // result slots: pinned host memory mapped into the device's address space -- the executors' kernels write their few
        // bytes (statuses, norms, eigenvalues) straight into it, the round's stream synchronisation makes them visible: no
        // device -> host blit per round (as host_fetch does for the single-block drivers; measured on the Al workload:
        // 35.7 - 38.1 -> 38.7 - 38.8 SCF it/s)
        HIPCHK(hipHostMalloc((void**)&pool.h_res, res_cap, hipHostMallocMapped));
        void* dp = nullptr;
        HIPCHK(hipHostGetDevicePointer(&dp, pool.h_res, 0));
        pool.d_res = reinterpret_cast<char*>(dp);
        pool.res_mapped = true;
        pool.device = b->device;
    }
    c->cap = cap;
    c->res_cap = res_cap;
    c->h_ring = pool.h_ring;
    c->d_ring = pool.d_ring;
    c->h_res = pool.h_res;
    c->d_res = pool.d_res;
    c->res_mapped = pool.res_mapped;
    c->d_scratch = pool.d_scratch;
    c->scratch_bytes = pool.scratch_bytes;
    const size_t n = bodies.size();
    rec.fibers.resize(n);
    const size_t stack_bytes = FIBER_STACK_BYTES;
    for (size_t i = 0; i < n; ++i) {
        Fiber& f = rec.fibers[i];
        f.body = bodies[i];
        f.stack = fiber_stack(i);
        getcontext(&f.ctx);
        f.ctx.uc_stack.ss_sp = f.stack;
        f.ctx.uc_stack.ss_size = stack_bytes;
        f.ctx.uc_link = &rec.main_ctx;
        const uintptr_t p = reinterpret_cast<uintptr_t>(&f);
        makecontext(&f.ctx, (void (*)())fiber_entry, 2, (unsigned)(p & 0xFFFFFFFFu), (unsigned)(p >> 32));
    }
    g_rec = &rec;
    int status = 0;
    for (;;) {
        bool alive = false;
        const auto tf0 = std::chrono::steady_clock::now();
        for (size_t i = 0; i < n; ++i) {
            Fiber& f = rec.fibers[i];
            if (f.done) continue;
            alive = true;
            rec.cur = (int)i;
            swapcontext(&rec.main_ctx, &f.ctx);   // runs until the fiber yields or finishes
            // a fiber that ended with an error may have left copies queued whose host destinations died with its frames
            // (dev_d2h_async): nothing it recorded since its last synchronisation is executed
            if (f.done && f.ret != 0) {
                f.fifo.clear();
                f.head = 0;
            }
        }
        rec.cur = -1;
        c->t_fibers += std::chrono::duration<double>(std::chrono::steady_clock::now() - tf0).count();
        bool queued = false;
        for (auto& f : rec.fibers) queued = queued || !f.fifo.empty();
        if (!alive && !queued) break;
        status = flush(&rec);
        if (status != 0) break;   // (fibers still suspended are abandoned with their stacks)
    }
    g_rec = nullptr;
    g_last_stats[0] = c->n_rounds;
    g_last_stats[1] = c->n_ops;
    g_last_stats[2] = c->n_launch_groups;
    g_last_stats[3] = c->n_sequential;
    hipStreamSynchronize(c->stream);
    batch_scratch_reset(c);
    if (getenv("DFTK_MI_KBATCH_TRACE"))
        fprintf(stderr, "[kbatch] %zu fibers: %lld rounds, %lld ops, %lld merged launches, %lld one-by-one ops; ms: fibers %.2f, "
                "launching %.2f, waiting %.2f, fix-ups %.2f\n", n, (long long)c->n_rounds, (long long)c->n_ops,
                (long long)c->n_launch_groups, (long long)c->n_sequential, 1e3 * c->t_fibers, 1e3 * c->t_exec, 1e3 * c->t_sync,
                1e3 * c->t_fix);
    pool.d_scratch = c->d_scratch;          // (may have grown during the call)
    pool.scratch_bytes = c->scratch_bytes;
    rets.resize(n);
    for (size_t i = 0; i < n; ++i) rets[i] = rec.fibers[i].done ? rec.fibers[i].ret : DFTK_MI_EHIP;
    return status;
}
