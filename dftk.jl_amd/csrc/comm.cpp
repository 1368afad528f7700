// comm.cpp -- communicators of the hot path.
//
//   * comm_kpts: mpi_sum!(rho, comm_kpts)  (src/common/mpi.jl:19-32, called at src/densities.jl:46) -> one RCCL
//     all-reduce over xGMI per SCF step;
//   * the plane-wave (row-slab) communicator of a SHARDED k-block (Gamma-only cells; the
//     reference can only duplicate the k-point, src/PlaneWaveBasis.jl:190-203): small all-reduces after every
//     product whose inner dimension is n_G, and the slab <-> band all-to-all around the FFT pipeline.
//
// Two back ends behind one handle:
//   RCCL  -- one process per GPU; the unique id travels by any side channel (the Python host mirror uses the
//            torch.distributed store, a Julia shim MPI.bcast).  Bound at run time (dlopen) so that the library
//            loads on hosts without RCCL and shares the copy another runtime (PyTorch) may already have loaded.
//   host  -- caller-supplied callbacks on pinned HOST buffers (the library stages device <-> host around them):
//            what a Julia shim plugs MPI.Allreduce! / MPI.Alltoallv! into, and what the test-suite uses to run two
//            ranks on ONE GPU over gloo (RCCL refuses two ranks on one device).
#include "common.h"
#include <dlfcn.h>
#include <algorithm>
#include <cstring>

namespace {
typedef struct { char internal[128]; } ncclUniqueId_t;
typedef void* ncclComm_t_;
typedef int (*fn_GetUniqueId)(ncclUniqueId_t*);
typedef int (*fn_CommInitRank)(ncclComm_t_*, int, ncclUniqueId_t, int);
typedef int (*fn_CommDestroy)(ncclComm_t_);
typedef int (*fn_AllReduce)(const void*, void*, size_t, int, int, ncclComm_t_, hipStream_t);
typedef int (*fn_Send)(const void*, size_t, int, int, ncclComm_t_, hipStream_t);
typedef int (*fn_Recv)(void*, size_t, int, int, ncclComm_t_, hipStream_t);
typedef int (*fn_Group)(void);
typedef const char* (*fn_GetErrorString)(int);
typedef int (*fn_GetVersion)(int*);
typedef int (*fn_CommCount)(ncclComm_t_, int*);

struct Rccl {
    void* h = nullptr;
    fn_GetUniqueId GetUniqueId = nullptr;
    fn_CommInitRank CommInitRank = nullptr;
    fn_CommDestroy CommDestroy = nullptr;
    fn_AllReduce AllReduce = nullptr;
    fn_Send Send = nullptr;
    fn_Recv Recv = nullptr;
    fn_Group GroupStart = nullptr, GroupEnd = nullptr;
    fn_GetErrorString GetErrorString = nullptr;
    fn_GetVersion GetVersion = nullptr;
    fn_CommCount CommCount = nullptr;
};
Rccl g_rccl;

int load_rccl() {
    if (g_rccl.h) return 0;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
        g_rccl.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (g_rccl.h) break;
    }
    if (!g_rccl.h) {
        dftk_set_error("cannot load RCCL: %s", dlerror());
        return DFTK_MI_ERCCL;
    }
    g_rccl.GetUniqueId = (fn_GetUniqueId)dlsym(g_rccl.h, "ncclGetUniqueId");
    g_rccl.CommInitRank = (fn_CommInitRank)dlsym(g_rccl.h, "ncclCommInitRank");
    g_rccl.CommDestroy = (fn_CommDestroy)dlsym(g_rccl.h, "ncclCommDestroy");
    g_rccl.AllReduce = (fn_AllReduce)dlsym(g_rccl.h, "ncclAllReduce");
    g_rccl.Send = (fn_Send)dlsym(g_rccl.h, "ncclSend");
    g_rccl.Recv = (fn_Recv)dlsym(g_rccl.h, "ncclRecv");
    g_rccl.GroupStart = (fn_Group)dlsym(g_rccl.h, "ncclGroupStart");
    g_rccl.GroupEnd = (fn_Group)dlsym(g_rccl.h, "ncclGroupEnd");
    g_rccl.GetErrorString = (fn_GetErrorString)dlsym(g_rccl.h, "ncclGetErrorString");
    g_rccl.GetVersion = (fn_GetVersion)dlsym(g_rccl.h, "ncclGetVersion");
    g_rccl.CommCount = (fn_CommCount)dlsym(g_rccl.h, "ncclCommCount");
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.CommDestroy || !g_rccl.AllReduce || !g_rccl.Send ||
        !g_rccl.Recv || !g_rccl.GroupStart || !g_rccl.GroupEnd) {
        dftk_set_error("RCCL symbols missing");
        return DFTK_MI_ERCCL;
    }
    return 0;
}

int rccl_fail(const char* what, int code) {
    dftk_set_error("%s failed: %s", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(code) : "?");
    return DFTK_MI_ERCCL;
}
const int NCCL_F64 = 8, NCCL_SUM = 0;   // ncclFloat64 / ncclSum (nccl.h enums, stable across NCCL / RCCL 2.x)
}  // namespace

struct dftk_mi_comm {
    int backend;   // 0 = RCCL, 1 = host callbacks
    ncclComm_t_ comm;
    int n_ranks, rank, device;
    dftk_mi_allreduce_fn allreduce;
    dftk_mi_alltoallv_fn alltoallv;
    void* user;
    double* stage[2];   // pinned host staging buffers (host back end), grown on demand
    size_t stage_bytes[2];
};

static int stage_ensure(dftk_mi_comm* c, int which, size_t bytes) {
    if (bytes <= c->stage_bytes[which]) return 0;
    if (c->stage[which]) HIPCHK(hipHostFree(c->stage[which]));
    c->stage[which] = nullptr;
    c->stage_bytes[which] = 0;
    const size_t want = bytes + bytes / 4 + 4096;
    HIPCHK(hipHostMalloc((void**)&c->stage[which], want));
    c->stage_bytes[which] = want;
    return 0;
}

extern "C" int dftk_mi_comm_get_unique_id(char id_out[128]) {
    if (!id_out) return DFTK_MI_EINVAL;
    CHK(load_rccl());
    ncclUniqueId_t id;
    int rc = g_rccl.GetUniqueId(&id);
    if (rc != 0) return rccl_fail("ncclGetUniqueId", rc);
    memcpy(id_out, id.internal, 128);
    return 0;
}

extern "C" int dftk_mi_comm_init_rank(const char id[128], int n_ranks, int rank, int device, dftk_mi_comm** out) {
    if (!id || !out || n_ranks < 1 || rank < 0 || rank >= n_ranks) return DFTK_MI_EINVAL;
    CHK(load_rccl());
    HIPCHK(hipSetDevice(device));
    ncclUniqueId_t uid;
    memcpy(uid.internal, id, 128);
    dftk_mi_comm* c = new dftk_mi_comm();
    memset(c, 0, sizeof(*c));
    c->backend = 0;
    c->n_ranks = n_ranks;
    c->rank = rank;
    c->device = device;
    int rc = g_rccl.CommInitRank(&c->comm, n_ranks, uid, rank);
    if (rc != 0) {
        delete c;
        return rccl_fail("ncclCommInitRank", rc);
    }
    *out = c;
    return 0;
}

extern "C" int dftk_mi_comm_create_host(int n_ranks, int rank, int device, dftk_mi_allreduce_fn allreduce,
                                        dftk_mi_alltoallv_fn alltoallv, void* user, dftk_mi_comm** out) {
    if (!out || n_ranks < 1 || rank < 0 || rank >= n_ranks || !allreduce) return DFTK_MI_EINVAL;
    dftk_mi_comm* c = new dftk_mi_comm();
    memset(c, 0, sizeof(*c));
    c->backend = 1;
    c->n_ranks = n_ranks;
    c->rank = rank;
    c->device = device;
    c->allreduce = allreduce;
    c->alltoallv = alltoallv;
    c->user = user;
    *out = c;
    return 0;
}

extern "C" int dftk_mi_comm_destroy(dftk_mi_comm* c) {
    if (!c) return 0;
    if (c->backend == 0 && g_rccl.CommDestroy) g_rccl.CommDestroy(c->comm);
    for (int i = 0; i < 2; ++i)
        if (c->stage[i]) hipHostFree(c->stage[i]);
    delete c;
    return 0;
}

extern "C" int dftk_mi_comm_describe(const dftk_mi_comm* c, int* backend, int* n_ranks, int* version) {
    if (!c) return DFTK_MI_EINVAL;
    if (backend) *backend = c->backend;
    if (n_ranks) {
        *n_ranks = c->n_ranks;
        // ask RCCL itself how many ranks met in this communicator (not what the caller told us)
        if (c->backend == 0 && g_rccl.CommCount) {
            int cnt = -1;
            if (g_rccl.CommCount(c->comm, &cnt) == 0) *n_ranks = cnt;
        }
    }
    if (version) {
        *version = 0;
        if (c->backend == 0 && g_rccl.GetVersion) g_rccl.GetVersion(version);
    }
    return 0;
}

extern "C" int dftk_mi_comm_rank(const dftk_mi_comm* c) { return c ? c->rank : -1; }
extern "C" int dftk_mi_comm_size(const dftk_mi_comm* c) { return c ? c->n_ranks : -1; }

extern "C" int dftk_mi_allreduce_sum_f64(dftk_mi_comm* c, double* buf_d, size_t n, void* stream) {
    if (!c || !buf_d) return DFTK_MI_EINVAL;
    if (n == 0) return 0;
    HIPCHK(hipSetDevice(c->device));
    hipStream_t st = (hipStream_t)stream;
    if (c->backend == 0) {
        int rc = g_rccl.AllReduce(buf_d, buf_d, n, NCCL_F64, NCCL_SUM, c->comm, st);
        if (rc != 0) return rccl_fail("ncclAllReduce", rc);
        return 0;
    }
    CHK(stage_ensure(c, 0, n * sizeof(double)));
    HIPCHK(hipMemcpyAsync(c->stage[0], buf_d, n * sizeof(double), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (c->allreduce(c->user, c->stage[0], n) != 0) {
        dftk_set_error("host all-reduce callback failed");
        return DFTK_MI_ERCCL;
    }
    HIPCHK(hipMemcpyAsync(buf_d, c->stage[0], n * sizeof(double), hipMemcpyHostToDevice, st));
    HIPCHK(hipStreamSynchronize(st));   // the staging buffer is reused by the next call
    return 0;
}

// ---- internal (library-side) entry points: everything on the basis' stream ---------------------------------
int comm_size(const dftk_mi_comm* c) { return c ? c->n_ranks : 1; }
int comm_rank(const dftk_mi_comm* c) { return c ? c->rank : 0; }

int comm_allreduce(dftk_mi_comm* c, dftk_mi_basis* b, double* d, size_t n) {
    if (!c || c->n_ranks == 1 || n == 0) return 0;
    const int ps = prof_begin(b, PROF_COMM, 8.0 * (double)n);
    const int st = dftk_mi_allreduce_sum_f64(c, d, n, (void*)b->stream);
    prof_end(b, ps);
    return st;
}

int comm_allreduce_norms(dftk_mi_comm* c, dftk_mi_basis* b, double* d, size_t n) {
    if (!c || c->n_ranks == 1 || n == 0) return 0;
    CHK(ew_square(b, d, n));
    CHK(comm_allreduce(c, b, d, n));
    return ew_sqrt(b, d, n);
}

// Variable all-to-all of complex elements: piece s of `send` (offset / count in ELEMENTS) goes to rank s, piece r
// of `recv` comes from rank r.  send and recv must not overlap.
int comm_alltoallv(dftk_mi_comm* c, dftk_mi_basis* b, const cd* send, const size_t* soff, const size_t* scnt,
                   cd* recv, const size_t* roff, const size_t* rcnt) {
    if (!c) return DFTK_MI_EINVAL;
    const int p = c->n_ranks, me = c->rank;
    size_t stot = 0, rtot = 0;
    for (int i = 0; i < p; ++i) {
        stot = std::max(stot, soff[i] + scnt[i]);
        rtot = std::max(rtot, roff[i] + rcnt[i]);
    }
    const int ps = prof_begin(b, PROF_COMM, 16.0 * (double)(stot + rtot));
    struct G {
        dftk_mi_basis* b;
        int s;
        ~G() { prof_end(b, s); }
    } guard{b, ps};
    if (scnt[me] != rcnt[me]) {
        dftk_set_error("alltoallv: self piece mismatch");
        return DFTK_MI_EINVAL;
    }
    if (c->backend == 0) {
        if (scnt[me])
            HIPCHK(hipMemcpyAsync(recv + roff[me], send + soff[me], scnt[me] * sizeof(cd), hipMemcpyDeviceToDevice,
                                  b->stream));
        if (p == 1) return 0;
        int rc = g_rccl.GroupStart();
        if (rc != 0) return rccl_fail("ncclGroupStart", rc);
        // an error inside the group must still close it (an open group poisons every later RCCL call of the process)
        int bad = 0;
        const char* what = nullptr;
        for (int i = 0; i < p && !bad; ++i) {
            if (i == me) continue;
            if (scnt[i]) {
                bad = g_rccl.Send(send + soff[i], 2 * scnt[i], NCCL_F64, i, c->comm, b->stream);
                if (bad) what = "ncclSend";
            }
            if (!bad && rcnt[i]) {
                bad = g_rccl.Recv(recv + roff[i], 2 * rcnt[i], NCCL_F64, i, c->comm, b->stream);
                if (bad) what = "ncclRecv";
            }
        }
        rc = g_rccl.GroupEnd();
        if (bad) return rccl_fail(what, bad);
        if (rc != 0) return rccl_fail("ncclGroupEnd", rc);
        return 0;
    }
    if (!c->alltoallv) {
        dftk_set_error("host communicator without an alltoallv callback");
        return DFTK_MI_EINVAL;
    }
    CHK(stage_ensure(c, 0, stot * sizeof(cd)));
    CHK(stage_ensure(c, 1, rtot * sizeof(cd)));
    if (stot) HIPCHK(hipMemcpyAsync(c->stage[0], send, stot * sizeof(cd), hipMemcpyDeviceToHost, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));
    std::vector<size_t> so(p), sc(p), ro(p), rc2(p);   // the callback counts DOUBLES
    for (int i = 0; i < p; ++i) {
        so[i] = 2 * soff[i];
        sc[i] = 2 * scnt[i];
        ro[i] = 2 * roff[i];
        rc2[i] = 2 * rcnt[i];
    }
    if (c->alltoallv(c->user, c->stage[0], sc.data(), so.data(), c->stage[1], rc2.data(), ro.data()) != 0) {
        dftk_set_error("host all-to-all callback failed");
        return DFTK_MI_ERCCL;
    }
    if (rtot) HIPCHK(hipMemcpyAsync(recv, c->stage[1], rtot * sizeof(cd), hipMemcpyHostToDevice, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));
    return 0;
}
