// comm.cpp -- RCCL (xGMI) replacement of DFTK's MPI wrappers on the hot path:
//   mpi_sum!(rho, comm_kpts)   src/common/mpi.jl:19-32, called at src/densities.jl:46
// One process per GPU; the unique id travels by any side channel (the Python host mirror uses
// the torch.distributed store, a Julia shim would use MPI.bcast or a file).
// RCCL is bound at run time (dlopen) so that the library loads on hosts without RCCL and shares
// the copy another runtime in the process (e.g. PyTorch) may already have loaded.
#include "common.h"
#include <dlfcn.h>
#include <cstring>

namespace {
typedef struct { char internal[128]; } ncclUniqueId_t;
typedef void* ncclComm_t_;
typedef int (*fn_GetUniqueId)(ncclUniqueId_t*);
typedef int (*fn_CommInitRank)(ncclComm_t_*, int, ncclUniqueId_t, int);
typedef int (*fn_CommDestroy)(ncclComm_t_);
typedef int (*fn_AllReduce)(const void*, void*, size_t, int, int, ncclComm_t_, hipStream_t);
typedef const char* (*fn_GetErrorString)(int);

struct Rccl {
    void* h = nullptr;
    fn_GetUniqueId GetUniqueId = nullptr;
    fn_CommInitRank CommInitRank = nullptr;
    fn_CommDestroy CommDestroy = nullptr;
    fn_AllReduce AllReduce = nullptr;
    fn_GetErrorString GetErrorString = nullptr;
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
    g_rccl.GetErrorString = (fn_GetErrorString)dlsym(g_rccl.h, "ncclGetErrorString");
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.CommDestroy || !g_rccl.AllReduce) {
        dftk_set_error("RCCL symbols missing");
        return DFTK_MI_ERCCL;
    }
    return 0;
}

int rccl_fail(const char* what, int code) {
    dftk_set_error("%s failed: %s", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(code) : "?");
    return DFTK_MI_ERCCL;
}
}  // namespace

struct dftk_mi_comm {
    ncclComm_t_ comm;
    int n_ranks, rank, device;
};

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

extern "C" int dftk_mi_comm_destroy(dftk_mi_comm* c) {
    if (!c) return 0;
    if (g_rccl.CommDestroy) g_rccl.CommDestroy(c->comm);
    delete c;
    return 0;
}

extern "C" int dftk_mi_allreduce_sum_f64(dftk_mi_comm* c, double* buf_d, size_t n, void* stream) {
    if (!c || !buf_d) return DFTK_MI_EINVAL;
    if (n == 0) return 0;
    HIPCHK(hipSetDevice(c->device));
    // ncclFloat64 = 8, ncclSum = 0 (nccl.h enums, stable across NCCL/RCCL 2.x)
    int rc = g_rccl.AllReduce(buf_d, buf_d, n, 8, 0, c->comm, (hipStream_t)stream);
    if (rc != 0) return rccl_fail("ncclAllReduce", rc);
    return 0;
}
