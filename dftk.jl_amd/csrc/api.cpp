// api.cpp -- the extern "C" boundary (include/dftk_mi355x.h), handles and host-side planning.
#include "common.h"
#include <chrono>
#include "batch.h"
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

static thread_local char g_err[1024] = "";

void dftk_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* dftk_mi_last_error(void) { return g_err; }

hipError_t dftk_scratch_malloc(void** p, size_t bytes) {
    static const bool poison = getenv("DFTK_MI_POISON") != nullptr;
    hipError_t e = hipMalloc(p, bytes);
    if (e == hipSuccess && poison) e = hipMemset(*p, 0xFF, bytes);
    return e;
}
#ifndef DFTK_MI_SRC_HASH
#define DFTK_MI_SRC_HASH "unknown"
#endif
extern "C" const char* dftk_mi_version(void) {
    return "dftk_mi355x 0.2.0 (gfx950; fp64; pruned batched FFT pipeline, f64-MFMA zgemm, blocked-Jacobi heev, "
           "plane-wave sharded LOBPCG) src=" DFTK_MI_SRC_HASH;
}

// ------------------------------------------------------------------------------------ profiling
int prof_begin(dftk_mi_basis* b, int fam, double work, uint64_t tag) {
    Prof* p = b->prof;
    if (!p || !p->on || p->mute > 0) return -1;
    // slots index `pending`, and scopes nest (apply_H / heev hold one across inner zgemm / FFT scopes): never
    // flush while a scope is open
    if (p->pending.size() >= 60000 && p->open == 0) prof_resolve(b);
    Prof::Pair pr;
    if (!p->pool.empty()) {
        pr = p->pool.back();
        p->pool.pop_back();
    } else {
        if (hipEventCreate(&pr.a) != hipSuccess || hipEventCreate(&pr.b) != hipSuccess) return -1;
    }
    pr.fam = fam;
    pr.tag = tag;
    pr.work = work;
    p->work[fam] += work;
    p->launches[fam] += 1;
    hipEventRecord(pr.a, b->stream);
    p->pending.push_back(pr);
    p->open += 1;
    return (int)p->pending.size() - 1;
}
void prof_end(dftk_mi_basis* b, int slot) {
    if (slot < 0) return;
    Prof* p = b->prof;
    if (p->open > 0) p->open -= 1;
    if ((size_t)slot < p->pending.size()) hipEventRecord(p->pending[slot].b, b->stream);
}
void prof_count(dftk_mi_basis* b, int fam, double work) {
    Prof* p = b->prof;
    if (!p || !p->on) return;
    p->work[fam] += work;
    p->launches[fam] += 1;
}
int host_wait(dftk_mi_basis* b) {
    Prof* p = b->prof;
    if (!p || !p->on) {
        HIPCHK(hipStreamSynchronize(b->stream));
        return 0;
    }
    const auto t0 = std::chrono::steady_clock::now();
    HIPCHK(hipStreamSynchronize(b->stream));
    p->ms[PROF_HOST_WAIT] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    p->launches[PROF_HOST_WAIT] += 1;
    return 0;
}
int prof_resolve(dftk_mi_basis* b) {
    Prof* p = b->prof;
    if (!p) return 0;
    HIPCHK(hipStreamSynchronize(b->stream));
    for (auto& pr : p->pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, pr.a, pr.b) == hipSuccess) {
            p->ms[pr.fam] += ms;
            if (pr.tag) {
                auto& sh = p->shapes[pr.tag];
                sh.ms += ms;
                sh.work += pr.work;
                sh.n += 1;
            }
        }
        p->pool.push_back(pr);
    }
    p->pending.clear();
    p->open = 0;
    return 0;
}
extern "C" int dftk_mi_prof_enable(dftk_mi_basis* b, int on) {
    if (!b) return DFTK_MI_EINVAL;
    CHK(prof_resolve(b));
    if (on) {
        for (int i = 0; i < PROF_NFAM; ++i) {
            b->prof->ms[i] = 0;
            b->prof->work[i] = 0;
            b->prof->launches[i] = 0;
        }
    }
    b->prof->shape_tags = (on & 2) != 0;
    static const bool shapes_env = getenv("DFTK_MI_GEMM_SHAPES") != nullptr;
    if (!on && shapes_env && !b->prof->shapes.empty()) {
        // DFTK_MI_GEMM_SHAPES: per-shape zgemm table (tag = trans | m | n | k)
        for (auto& kv : b->prof->shapes) {
            const uint64_t t = kv.first;
            fprintf(stderr, "[zgemm-shape] %c m=%llu n=%llu k=%llu flags=%d calls=%lld ms=%.3f TF/s=%.2f\n",
                    (t >> 63) ? 'C' : 'N', (unsigned long long)((t >> 42) & 0xFFFFF),
                    (unsigned long long)((t >> 22) & 0x3FFFF), (unsigned long long)(t & 0x3FFFFF), (int)((t >> 40) & 3),
                    (long long)kv.second.n, kv.second.ms, kv.second.work / (kv.second.ms * 1e9));
        }
        b->prof->shapes.clear();
    }
    b->prof->on = on != 0;
    return 0;
}
// Per-shape zgemm table accumulated since dftk_mi_prof_enable(b, 3): row i = { transA ('N' = 0, 'C' = 1), m, n, k, flags & 3,
// calls }, ms[i] = summed HIP-event time.  Returns the number of shapes in *count (at most cap rows are written) and
// clears the table.
extern "C" int dftk_mi_prof_zgemm_shapes(dftk_mi_basis* b, int cap, int64_t* rows6, double* ms, int* count) {
    if (!b || cap < 0 || !count || (cap > 0 && (!rows6 || !ms))) return DFTK_MI_EINVAL;
    CHK(prof_resolve(b));
    int i = 0;
    for (auto& kv : b->prof->shapes) {
        if (i < cap) {
            const uint64_t t = kv.first;
            rows6[6 * i + 0] = (int64_t)(t >> 63);
            rows6[6 * i + 1] = (int64_t)((t >> 42) & 0x7FFFF);
            rows6[6 * i + 2] = (int64_t)((t >> 22) & 0x3FFFF);
            rows6[6 * i + 3] = (int64_t)(t & 0x3FFFFF);
            rows6[6 * i + 4] = (int64_t)((t >> 40) & 3) | (((t >> 61) & 1) ? DFTK_MI_GEMM_REAL : 0);
            rows6[6 * i + 5] = (int64_t)kv.second.n;
            ms[i] = kv.second.ms;
        }
        ++i;
    }
    *count = i;
    b->prof->shapes.clear();
    return 0;
}
std::atomic<int64_t> g_dftk_launches{0}, g_dftk_host_syncs{0};
extern "C" int dftk_mi_launch_count(int64_t* launches, int64_t* host_syncs) {
    if (launches) *launches = g_dftk_launches.load(std::memory_order_relaxed);
    if (host_syncs) *host_syncs = g_dftk_host_syncs.load(std::memory_order_relaxed);
    return 0;
}
extern "C" int dftk_mi_prof_get(dftk_mi_basis* b, int family, double* total_ms, double* work, int64_t* launches) {
    if (!b || family < 0 || family >= PROF_NFAM) return DFTK_MI_EINVAL;
    CHK(prof_resolve(b));
    if (total_ms) *total_ms = b->prof->ms[family];
    if (work) *work = b->prof->work[family];
    if (launches) *launches = b->prof->launches[family];
    return 0;
}

// ------------------------------------------------------------------------------------ 1-D plans
// Factorise n into the butterflies the kernels have in registers -- 8, 6, 5, 4, 3, 2 (then primes <= 64
// for the generic kernel) -- with as FEW stages as possible: every stage is one LDS round trip of the
// whole tile, which is what bounds the FFT kernels.  Largest radix first (the first DIT stage has no
// twiddles).
int plan_radices(int n, int* nrad, int* rad) {
    int m = n, a2 = 0, a3 = 0, a5 = 0;
    while (m % 2 == 0 && m > 1) { m /= 2; ++a2; }
    while (m % 3 == 0 && m > 1) { m /= 3; ++a3; }
    while (m % 5 == 0 && m > 1) { m /= 5; ++a5; }
    int best8 = 0, best6 = 0, best = 1 << 30;
    for (int n8 = 0; n8 <= a2 / 3; ++n8)
        for (int n6 = 0; n6 <= (a2 - 3 * n8 < a3 ? a2 - 3 * n8 : a3); ++n6) {
            const int rem2 = a2 - 3 * n8 - n6;
            const int stages = n8 + n6 + rem2 / 2 + rem2 % 2 + (a3 - n6) + a5;
            if (stages < best) {
                best = stages;
                best8 = n8;
                best6 = n6;
            }
        }
    const int rem2 = a2 - 3 * best8 - best6;
    int counts[6][2] = {{8, best8}, {6, best6}, {5, a5}, {4, rem2 / 2}, {3, a3 - best6}, {2, rem2 % 2}};
    int k = 0;
    // an ODD radix first when there is one: with the unpadded 8-line tiles of the y/z kernels the first
    // stage (stride R between neighbouring butterflies) is LDS-bank-conflict-free only for odd R
    for (int oi : {2, 4})
        if (counts[oi][1] > 0) {
            rad[k++] = counts[oi][0];
            counts[oi][1] -= 1;
            break;
        }
    for (auto& c : counts)
        for (int i = 0; i < c[1]; ++i) {
            if (k >= DFTK_MAX_RADICES) return -1;
            rad[k++] = c[0];
        }
    for (int p = 7; m > 1; p += 2)
        while (m % p == 0) {
            if (k >= DFTK_MAX_RADICES || p > 64) return -1;
            rad[k++] = p;
            m /= p;
        }
    *nrad = k;
    return 0;
}

// pos[e] = position at which a decimation-in-time pass expects input element e, equivalently the
// position at which a decimation-in-frequency pass leaves output frequency e.
// position p has digits q_s = (p / m_s) % r_s with m_s = r_0...r_{s-1}; it holds element
// e = q_{k-1} + r_{k-1} (q_{k-2} + r_{k-2} ( ... q_0)).
void plan_positions(int n, int nrad, const int* rad, int* pos) {
    if (nrad == 0) {
        for (int i = 0; i < n; ++i) pos[i] = i;
        return;
    }
    std::vector<int> m(nrad + 1, 1);
    for (int s = 0; s < nrad; ++s) m[s + 1] = m[s] * rad[s];
    for (int p = 0; p < n; ++p) {
        int e = (p / m[0]) % rad[0];
        for (int s = 1; s < nrad; ++s) e = (p / m[s]) % rad[s] + rad[s] * e;
        pos[e] = p;
    }
}

extern "C" int dftk_mi_jacobi_schedule_host(int n, int round, int* n_blocks, int* pairs, int* where) {
    return jacobi_schedule_host(n, round, n_blocks, pairs, where);
}

extern "C" int dftk_mi_fft_plan_host(int n, int* n_radices, int* radices, int* pos) {
    if (n < 1 || !n_radices || !radices || !pos) return DFTK_MI_EINVAL;
    if (plan_radices(n, n_radices, radices) != 0) {
        dftk_set_error("cannot plan FFT of length %d (prime factor > 64)", n);
        return DFTK_MI_EINVAL;
    }
    plan_positions(n, *n_radices, radices, pos);
    return 0;
}

// ------------------------------------------------------------------------------------ sphere tables
struct SphereTables {
    std::vector<int64_t> line_id;      // iy + ny*iz
    std::vector<int64_t> line_start;   // [n_lines+1]
    std::vector<int> zval;             // distinct iz, ascending
    std::vector<int> zls;              // [nzx+1] first line of each plane
};

static int build_sphere_tables(int nx, int ny, int nz, int64_t n_G, const int64_t* mapping, SphereTables& t) {
    const int64_t N = (int64_t)nx * ny * nz;
    int64_t prev = -1, prev_line = -1;
    int prev_z = -1;
    for (int64_t c = 0; c < n_G; ++c) {
        const int64_t g = mapping[c];
        if (g < 0 || g >= N || g <= prev) {
            dftk_set_error("mapping must be strictly ascending and within the cube (entry %lld = %lld)", (long long)c,
                           (long long)g);
            return DFTK_MI_EINVAL;
        }
        prev = g;
        const int64_t line = g / nx;
        if (line != prev_line) {
            t.line_id.push_back(line);
            t.line_start.push_back(c);
            prev_line = line;
            const int iz = (int)(line / ny);
            if (iz != prev_z) {
                t.zval.push_back(iz);
                t.zls.push_back((int)t.line_id.size() - 1);
                prev_z = iz;
            }
        }
    }
    t.line_start.push_back(n_G);
    t.zls.push_back((int)t.line_id.size());
    return 0;
}

extern "C" int dftk_mi_sphere_tables_host(int nx, int ny, int nz, int64_t n_G, const int64_t* mapping0_h,
                                          int64_t* n_lines, int* n_zplanes, int64_t* line_id, int64_t* line_start) {
    if (nx < 1 || ny < 1 || nz < 1 || n_G < 0 || (n_G > 0 && !mapping0_h)) return DFTK_MI_EINVAL;
    SphereTables t;
    CHK(build_sphere_tables(nx, ny, nz, n_G, mapping0_h, t));
    if (n_lines) *n_lines = (int64_t)t.line_id.size();
    if (n_zplanes) *n_zplanes = (int)t.zval.size();
    if (line_id) std::copy(t.line_id.begin(), t.line_id.end(), line_id);
    if (line_start) std::copy(t.line_start.begin(), t.line_start.end(), line_start);
    return 0;
}

// ------------------------------------------------------------------------------------ basis
template <typename T>
static int upload(const std::vector<T>& v, T** d) {
    *d = nullptr;
    if (v.empty()) return 0;
    HIPCHK(hipMalloc((void**)d, v.size() * sizeof(T)));
    HIPCHK(hipMemcpy(*d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return 0;
}

static int check_device(int device) {
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
        dftk_set_error("no HIP device visible (%s): this library has no CPU fallback",
                       e == hipSuccess ? "count = 0" : hipGetErrorString(e));
        return DFTK_MI_ENOGPU;
    }
    if (device < 0 || device >= count) {
        dftk_set_error("device %d out of range (%d visible)", device, count);
        return DFTK_MI_EINVAL;
    }
    return 0;
}

extern "C" int dftk_mi_basis_create(int nx, int ny, int nz, double unit_cell_volume, int device,
                                    dftk_mi_basis** out) {
    if (!out || nx < 1 || ny < 1 || nz < 1 || !(unit_cell_volume > 0)) return DFTK_MI_EINVAL;
    CHK(check_device(device));
    HIPCHK(hipSetDevice(device));
    dftk_mi_basis* b = new dftk_mi_basis();
    memset(b, 0, sizeof(*b));
    b->nx = nx;
    b->ny = ny;
    b->nz = nz;
    b->nxp = ((nx + 7) / 8) * 8;
    b->volume = unit_cell_volume;
    b->device = device;
    // bands per launch group of the FFT pipeline.  32 since round 5: at 8 the launches of the x stages last 48 - 61 us and
    // every stage loses bandwidth to its fill / drain (tools/fft_bench.py 5 256 <group>: pipeline + density of 256 bands
    // 149.1 ms at 8, 143.3 at 16, 134.5 at 32, 135.6 at 64); scratch = group x (T1 + T2) = 2.8 GB at the 1000-electron cell
    b->fft_batch = 32;
    b->prof = new Prof();
    const char* g = getenv("DFTK_MI_GEMM");
    b->use_mfma = (g && strcmp(g, "naive") == 0) ? 0 : 1;
    HIPCHK(hipStreamCreate(&b->stream));
    const int dims[3] = {nx, ny, nz};
    for (int a = 0; a < 3; ++a) {
        const int n = dims[a];
        FftAxis& ax = b->ax[a];
        ax.n = n;
        if (plan_radices(n, &ax.nrad, ax.rad) != 0) {
            dftk_set_error("cannot plan FFT axis of length %d", n);
            return DFTK_MI_EINVAL;
        }
        std::vector<int> pos(n);
        plan_positions(n, ax.nrad, ax.rad, pos.data());
        std::vector<double> tw(2 * (size_t)n);
        for (int t = 0; t < n; ++t) {
            // exact octant reduction keeps the table accurate to ~1 ulp
            const long double ang = 2.0L * 3.14159265358979323846264338327950288L * (long double)t / (long double)n;
            tw[2 * t] = (double)cosl(ang);
            tw[2 * t + 1] = (double)sinl(ang);
        }
        double* dtw;
        int* dpos;
        CHK(upload(tw, &dtw));
        CHK(upload(pos, &dpos));
        ax.tw = reinterpret_cast<const cd*>(dtw);
        ax.pos = dpos;
        b->d_tables[2 * a] = dtw;
        b->d_tables[2 * a + 1] = dpos;
    }
    HIPCHK(hipMalloc((void**)&b->d_scalars, 256 * sizeof(double)));
    HIPCHK(hipHostMalloc((void**)&b->h_scalars, 256 * sizeof(double)));
    HIPCHK(hipHostMalloc((void**)&b->h_fetch, HOST_FETCH_BYTES, hipHostMallocMapped));
    *out = b;
    return 0;
}

extern "C" int dftk_mi_basis_destroy(dftk_mi_basis* b) {
    if (!b) return 0;
    hipSetDevice(b->device);
    hipStreamSynchronize(b->stream);
    for (void* p : b->d_tables)
        if (p) hipFree(p);
    if (b->T1) hipFree(b->T1);
    if (b->T2) hipFree(b->T2);
    if (b->ws) hipFree(b->ws);
    if (b->dense_ws) hipFree(b->dense_ws);
    if (b->eig_ws) hipFree(b->eig_ws);
    if (b->symm_tab) hipFree(b->symm_tab);
    if (b->d_scalars) hipFree(b->d_scalars);
    if (b->h_scalars) hipHostFree(b->h_scalars);
    if (b->h_fetch) hipHostFree(b->h_fetch);
    if (b->prof) {
        prof_resolve(b);
        for (auto& pr : b->prof->pool) {
            hipEventDestroy(pr.a);
            hipEventDestroy(pr.b);
        }
        delete b->prof;
    }
    hipStreamDestroy(b->stream);
    delete b;
    return 0;
}

extern "C" int dftk_mi_basis_sync(dftk_mi_basis* b) {
    if (!b) return DFTK_MI_EINVAL;
    HIPCHK(hipStreamSynchronize(b->stream));
    return 0;
}

extern "C" int dftk_mi_basis_set_fft_batch(dftk_mi_basis* b, int n) {
    if (!b || n < 1 || n > 256) return DFTK_MI_EINVAL;
    b->fft_batch = n;
    return 0;
}

extern "C" void* dftk_mi_basis_stream(dftk_mi_basis* b) { return b ? (void*)b->stream : nullptr; }

// ------------------------------------------------------------------------------------ k-block
extern "C" int dftk_mi_kblock_create(dftk_mi_basis* b, int64_t n_G, const int64_t* mapping0_h, const double* kinetic_h,
                                     dftk_mi_kblock** out) {
    if (!b || !out || n_G < 1 || !mapping0_h) return DFTK_MI_EINVAL;
    if (n_G > INT32_MAX) return DFTK_MI_EINVAL;
    HIPCHK(hipSetDevice(b->device));
    SphereTables t;
    CHK(build_sphere_tables(b->nx, b->ny, b->nz, n_G, mapping0_h, t));
    // per-axis permutation tables (host copies)
    std::vector<int> pos[3];
    const int dims[3] = {b->nx, b->ny, b->nz};
    for (int a = 0; a < 3; ++a) {
        pos[a].resize(dims[a]);
        plan_positions(dims[a], b->ax[a].nrad, b->ax[a].rad, pos[a].data());
    }
    const int64_t n_lines = (int64_t)t.line_id.size();
    std::vector<int> cpos(n_G), cx(n_G), line_start(n_lines + 1), line_ypos(n_lines), line_yval(n_lines);
    for (int64_t l = 0; l < n_lines; ++l) {
        const int iy = (int)(t.line_id[l] % b->ny);
        line_yval[l] = iy;
        line_ypos[l] = pos[1][iy];
        line_start[l] = (int)t.line_start[l];
        for (int64_t c = t.line_start[l]; c < t.line_start[l + 1]; ++c) {
            const int ix = (int)(mapping0_h[c] - t.line_id[l] * b->nx);
            cx[c] = ix;
            cpos[c] = pos[0][ix];
        }
    }
    line_start[n_lines] = (int)n_G;
    std::vector<int> zpos(t.zval.size());
    for (size_t i = 0; i < t.zval.size(); ++i) zpos[i] = pos[2][t.zval[i]];

    dftk_mi_kblock* kb = new dftk_mi_kblock();
    memset(kb, 0, sizeof(*kb));
    kb->basis = b;
    kb->device = b->device;
    kb->n_G = n_G;
    kb->n_lines = n_lines;
    kb->nzx = (int)t.zval.size();
    {   // wrap-around contiguity of the sphere's z planes (index arithmetic instead of a table in k_zpass_reg)
        int lo = 0;
        while (lo < kb->nzx && t.zval[lo] == lo) ++lo;
        bool ok = true;
        for (int i = lo; i < kb->nzx; ++i) ok = ok && t.zval[i] == b->nz - (kb->nzx - i);
        kb->z_lo = ok ? lo : -1;
    }
    CHK(upload(cpos, &kb->d_cpos));
    CHK(upload(cx, &kb->d_cx));
    CHK(upload(line_start, &kb->d_line_start));
    CHK(upload(line_ypos, &kb->d_line_ypos));
    CHK(upload(line_yval, &kb->d_line_yval));
    CHK(upload(t.zls, &kb->d_zls));
    CHK(upload(zpos, &kb->d_zpos));
    CHK(upload(t.zval, &kb->d_zval));
    std::vector<double> kin(n_G, 0.0);
    if (kinetic_h) std::copy(kinetic_h, kinetic_h + n_G, kin.begin());
    CHK(upload(kin, &kb->d_kin));
    kb->h_mapping = new std::vector<int64_t>(mapping0_h, mapping0_h + n_G);
    kb->h_kin = new std::vector<double>(kin);
    *out = kb;
    return 0;
}

// one padded potential shared by the k-blocks of a dftk_mi_kblocks_set_potential call (they all apply the SAME V)
struct SharedVs {
    double* p = nullptr;
    int refs = 0;
};
// the block lets go of its padded potential (the caller has made sure no kernel reading it is in flight)
static void release_Vs(dftk_mi_kblock* kb) {
    if (kb->Vs_share) {
        if (--kb->Vs_share->refs == 0) {
            if (kb->Vs_share->p) hipFree(kb->Vs_share->p);
            delete kb->Vs_share;
        }
        kb->Vs_share = nullptr;
    } else if (kb->d_Vs) {
        hipFree(kb->d_Vs);
    }
    kb->d_Vs = nullptr;
}

extern "C" int dftk_mi_kblock_destroy(dftk_mi_kblock* kb) {
    if (!kb) return 0;
    hipSetDevice(kb->device);
    hipDeviceSynchronize();   // the basis may already be gone: never dereference it here
    release_Vs(kb);
    if (kb->d_Vs_ax) hipFree(kb->d_Vs_ax);
    if (kb->d_dVs) hipFree(kb->d_dVs);
    void* ptrs[] = {kb->d_cpos, kb->d_cx, kb->d_line_start, kb->d_line_ypos, kb->d_line_yval, kb->d_zls,
                    kb->d_zpos, kb->d_zval, kb->d_kin, kb->d_D, kb->lob_buf};
    for (void* p : ptrs)
        if (p) hipFree(p);
    if (kb->sh_buf) hipFree(kb->sh_buf);
    delete kb->sh_rows;
    delete kb->lob_hist;
    delete kb->h_mapping;
    delete kb->h_kin;
    gamma_destroy(kb->gr);
    delete kb;
    return 0;
}

// ------------------------------------------------------------------------------------ plane-wave sharding
static int64_t local_rows(const dftk_mi_kblock* kb) {
    if (!kb->sh_comm) return kb->n_G;
    const int r = comm_rank(kb->sh_comm);
    return (*kb->sh_rows)[r + 1] - (*kb->sh_rows)[r];
}

extern "C" int dftk_mi_kblock_set_shard(dftk_mi_kblock* kb, dftk_mi_comm* comm, const int64_t* row_starts_h) {
    if (!kb) return DFTK_MI_EINVAL;
    HIPCHK(hipSetDevice(kb->basis->device));
    HIPCHK(hipStreamSynchronize(kb->basis->stream));
    if (!comm || comm_size(comm) == 1) {      // a one-rank communicator owns the whole sphere: nothing to shard
        kb->sh_comm = nullptr;
        return 0;
    }
    if (kb->gr && kb->gr->on) {
        dftk_set_error("set_shard: call before dftk_mi_kblock_set_gamma_real (the half-format rows are split when it is switched on)");
        return DFTK_MI_EINVAL;
    }
    const int p = comm_size(comm);
    if (!row_starts_h || row_starts_h[0] != 0 || row_starts_h[p] != kb->n_G) {
        dftk_set_error("set_shard: row_starts must run from 0 to n_G over n_ranks + 1 entries");
        return DFTK_MI_EINVAL;
    }
    for (int r = 0; r < p; ++r)
        if (row_starts_h[r + 1] <= row_starts_h[r]) {
            dftk_set_error("set_shard: every rank needs a non-empty row slab");
            return DFTK_MI_EINVAL;
        }
    if (kb->n_p != 0) {
        dftk_set_error("set_shard: call before dftk_mi_kblock_set_projectors (P becomes the row slab of this rank)");
        return DFTK_MI_EINVAL;
    }
    if (!kb->sh_rows) kb->sh_rows = new std::vector<int64_t>();
    kb->sh_rows->assign(row_starts_h, row_starts_h + p + 1);
    kb->sh_comm = comm;
    return 0;
}

// columns [c0[s], c0[s+1]) of an nb-column block are transformed by rank s (split_evenly)
static void band_split(int nb, int p, std::vector<int>& c0) {
    c0.assign(p + 1, 0);
    const int base = nb / p, rem = nb % p;
    for (int s = 0; s < p; ++s) c0[s + 1] = c0[s] + base + (s < rem ? 1 : 0);
}

static int shard_buffers(dftk_mi_kblock* kb, int nb, cd** R1, cd** F, cd** G) {
    const int p = comm_size(kb->sh_comm);
    const size_t maxc = (size_t)(nb + p - 1) / p;
    const size_t each = (size_t)kb->n_G * (maxc ? maxc : 1);
    const size_t need = 3 * each * sizeof(cd);
    if (need > kb->sh_bytes) {
        HIPCHK(hipStreamSynchronize(kb->basis->stream));
        if (kb->sh_buf) HIPCHK(hipFree(kb->sh_buf));
        kb->sh_buf = nullptr;
        kb->sh_bytes = 0;
        HIPCHK(dftk_scratch_malloc((void**)&kb->sh_buf, need));
        kb->sh_bytes = need;
    }
    *R1 = kb->sh_buf;
    *F = kb->sh_buf + each;
    *G = kb->sh_buf + 2 * each;
    return 0;
}

// Offsets / counts (complex elements) of the two layouts of an nb-band block of a sharded k-block on rank `me`:
// slab side  -- this rank's rows of all bands, packed n_loc x nb: the columns of destination s are one contiguous piece;
// band side  -- all rows of this rank's bands: the piece from source r is its rows x mine columns, packed.
static void shard_plan(const std::vector<int64_t>& rows, int p, int me, int nb, std::vector<int>& c0,
                       std::vector<size_t>& slab_off, std::vector<size_t>& slab_cnt, std::vector<size_t>& band_off,
                       std::vector<size_t>& band_cnt) {
    band_split(nb, p, c0);
    const int64_t nloc = rows[me + 1] - rows[me];
    const int mine = c0[me + 1] - c0[me];
    slab_off.resize(p), slab_cnt.resize(p), band_off.resize(p), band_cnt.resize(p);
    for (int s = 0; s < p; ++s) {
        slab_off[s] = (size_t)c0[s] * nloc;
        slab_cnt[s] = (size_t)(c0[s + 1] - c0[s]) * nloc;
        band_off[s] = (size_t)rows[s] * mine;
        band_cnt[s] = (size_t)(rows[s + 1] - rows[s]) * mine;
    }
}

extern "C" int dftk_mi_shard_plan_host(int n_ranks, int rank, int n_bands, const int64_t* row_starts_h,
                                       int* band_starts /* [n_ranks + 1] */, int64_t* slab_off, int64_t* slab_cnt,
                                       int64_t* band_off, int64_t* band_cnt /* [n_ranks] each */) {
    if (n_ranks < 1 || rank < 0 || rank >= n_ranks || n_bands < 0 || !row_starts_h || !band_starts || !slab_off ||
        !slab_cnt || !band_off || !band_cnt)
        return DFTK_MI_EINVAL;
    std::vector<int64_t> rows(row_starts_h, row_starts_h + n_ranks + 1);
    std::vector<int> c0;
    std::vector<size_t> so, sc, bo, bc;
    shard_plan(rows, n_ranks, rank, n_bands, c0, so, sc, bo, bc);
    for (int s = 0; s <= n_ranks; ++s) band_starts[s] = c0[s];
    for (int s = 0; s < n_ranks; ++s) {
        slab_off[s] = (int64_t)so[s];
        slab_cnt[s] = (int64_t)sc[s];
        band_off[s] = (int64_t)bo[s];
        band_cnt[s] = (int64_t)bc[s];
    }
    return 0;
}

// slab layout (n_loc x nb, this rank's rows of every band) -> band layout (n_G x mine, all rows of this rank's
// bands) and back: one all-to-all each way, pieces are contiguous column groups on the slab side
struct Transposer {
    dftk_mi_kblock* kb;
    int p, me, nb, mine;
    int64_t nloc, ntot;
    const std::vector<int64_t>* rows;   // [p + 1] row offsets of the sharded dimension (full sphere, or half format)
    std::vector<int> c0;
    std::vector<size_t> slab_off, slab_cnt, band_off, band_cnt;
    Transposer(dftk_mi_kblock* k, int nbands, const std::vector<int64_t>* row_offsets = nullptr)
        : kb(k), nb(nbands), rows(row_offsets ? row_offsets : k->sh_rows) {
        p = comm_size(kb->sh_comm);
        me = comm_rank(kb->sh_comm);
        nloc = (*rows)[me + 1] - (*rows)[me];
        ntot = (*rows)[p];
        shard_plan(*rows, p, me, nb, c0, slab_off, slab_cnt, band_off, band_cnt);
        mine = c0[me + 1] - c0[me];
    }
    // psi_loc (ld == nloc required by the caller) -> F (ntot x mine, ld ntot); R1 is scratch
    int to_bands(const cd* slab, cd* R1, cd* F) {
        dftk_mi_basis* b = kb->basis;
        CHK(comm_alltoallv(kb->sh_comm, b, slab, slab_off.data(), slab_cnt.data(), R1, band_off.data(),
                           band_cnt.data()));
        for (int r = 0; r < p; ++r) {
            const int64_t nr = (*rows)[r + 1] - (*rows)[r];
            CHK(ew_copy(b, nr, mine, R1 + band_off[r], nr, F + (*rows)[r], ntot));
        }
        return 0;
    }
    int to_slabs(const cd* Gfull, cd* R1, cd* slab) {
        dftk_mi_basis* b = kb->basis;
        for (int r = 0; r < p; ++r) {
            const int64_t nr = (*rows)[r + 1] - (*rows)[r];
            CHK(ew_copy(b, nr, mine, Gfull + (*rows)[r], ntot, R1 + band_off[r], nr));
        }
        return comm_alltoallv(kb->sh_comm, b, R1, band_off.data(), band_cnt.data(), slab, slab_off.data(),
                              slab_cnt.data());
    }
};

// ---- real-symmetric Gamma orbitals on a plane-wave sharded block (gamma_kernels.hip has the local pieces) ----------
// Half-format blocks are sharded by half-format rows; whole bands exist only between two all-to-alls, where the
// pair packing / FFT pipeline / compress / expand kernels run on this rank's share of the bands.
int gamma_apply_H_sharded(dftk_mi_kblock* kb, int which, int nb, const cd* psi, int64_t ldpsi, cd* Hpsi, int64_t ldH) {
    GammaReal* gr = kb->gr;
    dftk_mi_basis* b = kb->basis;
    const int64_t hl = gamma_local_rows(kb);
    if (ldpsi != hl || ldH != hl) {
        dftk_set_error("sharded gamma apply_H: blocks must be packed (leading dimension == local half rows %lld)", (long long)hl);
        return DFTK_MI_EINVAL;
    }
    const int slot = prof_begin(b, PROF_APPLY_H, (double)nb);
    struct G {
        dftk_mi_basis* b;
        int s;
        ~G() { prof_end(b, s); }
    } guard{b, slot};
    const bool local = (which & 1) && kb->d_Vs != nullptr;
    const bool kinetic = which & 2;
    if ((which & 4) && kb->n_p > 0) CHK(gamma_projectors_sharded(kb));   // (first: it may regrow the shared buffers)
    cd *R1, *F, *Gf;
    CHK(shard_buffers(kb, nb, &R1, &F, &Gf));
    Transposer t(kb, nb, &gr->half_rows);
    const int mine2 = (t.mine + 1) / 2;
    CHK(gamma_ensure_buf(kb, 2 * (size_t)kb->n_G * (mine2 ? mine2 : 1)));
    cd* Z = gr->buf;
    cd* W = gr->buf + (size_t)kb->n_G * (mine2 ? mine2 : 1);
    CHK(t.to_bands(psi, R1, F));                                   // F: n_half x mine
    if (t.mine > 0) {
        CHK(gamma_pack_pairs(kb, t.mine, F, gr->n_half, Z, kb->n_G));
        CHK(launch_local_apply(kb, mine2, Z, kb->n_G, W, kb->n_G, kinetic, local));
        CHK(gamma_unpack_pairs(kb, t.mine, W, kb->n_G, Gf, gr->n_half));
    }
    CHK(t.to_slabs(Gf, R1, Hpsi));
    if ((which & 4) && kb->n_p > 0)
        CHK(apply_nonlocal_rows(kb, nb, gr->P_half, hl, hl, psi, ldpsi, Hpsi, ldH, true, DFTK_MI_GEMM_REAL, kb->sh_comm));
    return 0;
}

// this rank's slab of the half-format projectors from its slab of the full-sphere ones (two all-to-alls, once)
int gamma_projectors_sharded(dftk_mi_kblock* kb) {
    GammaReal* gr = kb->gr;
    if (gr->P_src == kb->P && gr->P_n_p == kb->n_p && gr->P_half) return 0;
    dftk_mi_basis* b = kb->basis;
    const int64_t hl = gamma_local_rows(kb), fl = local_rows(kb);
    const int n_p = kb->n_p;
    HIPCHK(hipStreamSynchronize(b->stream));
    if (gr->P_half) HIPCHK(hipFree(gr->P_half));
    gr->P_half = nullptr;
    HIPCHK(hipMalloc((void**)&gr->P_half, (size_t)hl * n_p * sizeof(cd)));
    cd *R1, *F, *Gf;
    CHK(shard_buffers(kb, n_p, &R1, &F, &Gf));
    Transposer tf(kb, n_p), th(kb, n_p, &gr->half_rows);
    // packed copy of the slab (the caller's leading dimension may exceed the local rows)
    CHK(gamma_ensure_buf(kb, (size_t)fl * n_p));
    CHK(ew_copy(b, fl, n_p, kb->P, kb->ldP, gr->buf, fl));
    CHK(tf.to_bands(gr->buf, R1, F));                              // F: n_G x mine columns of P
    double h[2] = {0.0, 0.0};
    CHK(gamma_gather_P(kb, tf.mine, F, kb->n_G, Gf, gr->n_half, h));
    double bad = (std::isfinite(h[1]) && h[0] <= 1e-10 * (h[1] > 0 ? h[1] : 1.0)) ? 0.0 : 1.0;
    HIPCHK(hipMemcpyAsync(b->d_scalars, &bad, sizeof(double), hipMemcpyHostToDevice, b->stream));
    CHK(comm_allreduce(kb->sh_comm, b, b->d_scalars, 1));
    HIPCHK(hipMemcpyAsync(&bad, b->d_scalars, sizeof(double), hipMemcpyDeviceToHost, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));
    if (bad != 0.0) {
        dftk_set_error("gamma_real: the projectors are not real-symmetric (max |P(-G) - conj P(G)| = %.3e on this rank)", h[0]);
        HIPCHK(hipFree(gr->P_half));
        gr->P_half = nullptr;
        return DFTK_MI_EINVAL;
    }
    CHK(th.to_slabs(Gf, R1, gr->P_half));
    gr->P_src = kb->P;
    gr->P_n_p = n_p;
    return 0;
}

int gamma_density_sharded(dftk_mi_kblock* kb, int nb, const cd* psi, int64_t ldpsi, const double* w_h, double* rho) {
    if (ldpsi != local_rows(kb)) {
        dftk_set_error("sharded density: blocks must be packed (leading dimension == local rows)");
        return DFTK_MI_EINVAL;
    }
    cd *R1, *F, *Gf;
    CHK(shard_buffers(kb, nb, &R1, &F, &Gf));
    Transposer t(kb, nb);
    CHK(t.to_bands(psi, R1, F));
    if (t.mine == 0) return 0;
    return gamma_density_bands(kb, t.mine, F, kb->n_G, w_h + t.c0[t.me], rho);
}

// caller's full-sphere block (row slab of it on a sharded block) -> half-format block of dftk_mi_lobpcg, and back
int gamma_lobpcg_load(dftk_mi_kblock* kb, int M, const cd* Xuser, int64_t ldX, cd* Xh, int64_t ldh, bool align) {
    if (!kb->sh_comm)
        return align ? gamma_compress_aligned(kb, M, Xuser, ldX, Xh, ldh) : gamma_compress(kb, M, Xuser, ldX, Xh, ldh);
    GammaReal* gr = kb->gr;
    dftk_mi_basis* b = kb->basis;
    const int64_t hl = gamma_local_rows(kb), fl = local_rows(kb);
    if (ldh != hl) return DFTK_MI_EINVAL;
    cd *R1, *F, *Gf;
    CHK(shard_buffers(kb, M, &R1, &F, &Gf));
    Transposer tf(kb, M), th(kb, M, &gr->half_rows);
    CHK(gamma_ensure_buf(kb, (size_t)fl * M));
    CHK(ew_copy(b, fl, M, Xuser, ldX, gr->buf, fl));
    CHK(tf.to_bands(gr->buf, R1, F));
    // (every rank holds WHOLE bands here: the phase of a band is decided by the rank that owns it)
    CHK(align ? gamma_compress_aligned(kb, tf.mine, F, kb->n_G, Gf, gr->n_half)
              : gamma_compress(kb, tf.mine, F, kb->n_G, Gf, gr->n_half));
    return th.to_slabs(Gf, R1, Xh);
}

int gamma_lobpcg_store(dftk_mi_kblock* kb, int M, const cd* Xh, int64_t ldh, cd* Xuser, int64_t ldX) {
    if (!kb->sh_comm) return gamma_expand(kb, M, Xh, ldh, Xuser, ldX);
    GammaReal* gr = kb->gr;
    dftk_mi_basis* b = kb->basis;
    const int64_t hl = gamma_local_rows(kb), fl = local_rows(kb);
    if (ldh != hl) return DFTK_MI_EINVAL;
    cd *R1, *F, *Gf;
    CHK(shard_buffers(kb, M, &R1, &F, &Gf));
    Transposer tf(kb, M), th(kb, M, &gr->half_rows);
    CHK(th.to_bands(Xh, R1, F));                                   // F: n_half x mine
    CHK(gamma_expand(kb, th.mine, F, gr->n_half, Gf, kb->n_G));    // Gf: n_G x mine
    CHK(gamma_ensure_buf(kb, (size_t)fl * M));
    CHK(tf.to_slabs(Gf, R1, gr->buf));
    return ew_copy(b, fl, M, gr->buf, fl, Xuser, ldX);
}

extern "C" int dftk_mi_kblock_set_projectors(dftk_mi_kblock* kb, int n_p, const dftk_mi_cplx* P_d, int64_t ldP,
                                             const double* D_h) {
    if (!kb || n_p < 0) return DFTK_MI_EINVAL;
    HIPCHK(hipSetDevice(kb->basis->device));
    HIPCHK(hipStreamSynchronize(kb->basis->stream));
    if (kb->d_D) {
        HIPCHK(hipFree(kb->d_D));
        kb->d_D = nullptr;
    }
    kb->n_p = 0;
    kb->P = nullptr;
    kb->ax_keep = nullptr;                  // (an A X kept by the LOBPCG driver belongs to the old nonlocal term)
    kb->ax_reuse_next = false;
    if (kb->gr) kb->gr->P_src = nullptr;   // the half-format copy is rebuilt on its next use
    if (n_p == 0) return 0;
    if (!P_d || !D_h || ldP < local_rows(kb)) return DFTK_MI_EINVAL;
    int bw = 0;
    for (int j = 0; j < n_p; ++j)
        for (int i = 0; i < n_p; ++i)
            if (D_h[i + (size_t)j * n_p] != 0.0) bw = std::max(bw, std::abs(i - j));
    std::vector<double> D(D_h, D_h + (size_t)n_p * n_p);
    CHK(upload(D, &kb->d_D));
    kb->D_bw = bw;
    kb->n_p = n_p;
    kb->P = reinterpret_cast<const cd*>(P_d);
    kb->ldP = ldP;
    return 0;
}

extern "C" int dftk_mi_kblock_set_potential(dftk_mi_kblock* kb, const double* V_d) {
    if (!kb) return DFTK_MI_EINVAL;
    dftk_mi_basis* b = kb->basis;
    HIPCHK(hipSetDevice(b->device));
    if (!V_d || kb->Vs_share) {   // (a block that shares its buffer gets one of its own again: the others keep theirs)
        HIPCHK(hipStreamSynchronize(b->stream));
        release_Vs(kb);
        if (!V_d) return 0;
    }
    if (!kb->d_Vs) HIPCHK(hipMalloc((void**)&kb->d_Vs, (size_t)b->nz * b->ny * b->nxp * sizeof(double)));
    return launch_pad_potential(kb, V_d);
}

// The SAME summed local potential for many k-blocks (every k-point of a basis applies one V): one call instead of a
// host round trip per k-block.  The unsharded blocks of ONE basis (one stream) share ONE padded copy -- one pad kernel per
// SCF step instead of one per k-point (72 launches of ~3 us for BASELINE configs[2]); a later dftk_mi_kblock_set_potential
// on one of them detaches it.  Blocks of different bases (lanes) keep their own copies.
extern "C" int dftk_mi_kblocks_set_potential(int n_kblocks, dftk_mi_kblock* const* kbs, const double* V_d) {
    if (n_kblocks < 0 || (n_kblocks > 0 && !kbs) || !V_d) return DFTK_MI_EINVAL;
    for (int i = 0; i < n_kblocks; ++i)
        if (!kbs[i]) return DFTK_MI_EINVAL;
    bool one_basis = n_kblocks >= 2;
    for (int i = 1; i < n_kblocks && one_basis; ++i) one_basis = kbs[i]->basis == kbs[0]->basis;
    if (!one_basis) {
        for (int i = 0; i < n_kblocks; ++i) CHK(dftk_mi_kblock_set_potential(kbs[i], V_d));
        return 0;
    }
    dftk_mi_basis* b = kbs[0]->basis;
    HIPCHK(hipSetDevice(b->device));
    // already one shared buffer held by exactly these blocks (the call of the previous SCF step): refill it
    SharedVs* sh = kbs[0]->Vs_share;
    bool reuse = sh != nullptr && sh->refs == n_kblocks;
    for (int i = 0; i < n_kblocks && reuse; ++i) reuse = kbs[i]->Vs_share == sh;
    if (!reuse) {
        HIPCHK(hipStreamSynchronize(b->stream));
        for (int i = 0; i < n_kblocks; ++i) release_Vs(kbs[i]);
        sh = new SharedVs();
        if (hipMalloc((void**)&sh->p, (size_t)b->nz * b->ny * b->nxp * sizeof(double)) != hipSuccess) {
            delete sh;
            dftk_set_error("dftk_mi_kblocks_set_potential: out of device memory");
            return DFTK_MI_EHIP;
        }
        for (int i = 0; i < n_kblocks; ++i) {
            kbs[i]->Vs_share = sh;
            kbs[i]->d_Vs = sh->p;
            sh->refs += 1;
        }
    }
    return launch_pad_potential(kbs[0], V_d);     // asynchronous on the basis' stream
}

// ------------------------------------------------------------------------------------ H psi
int apply_nonlocal_rows(dftk_mi_kblock* kb, int nb, const cd* P, int64_t ldP, int64_t rows, const cd* psi,
                        int64_t ldpsi, cd* Hpsi, int64_t ldH, bool accumulate, int gemm_flags, dftk_mi_comm* comm) {
    // Hpsi (+)= P (D (P' psi))   (operators.jl:126-128)
    dftk_mi_basis* b = kb->basis;
    const cd one = {1.0, 0.0}, zero = {0.0, 0.0};
    if (kb->n_p == 0) {
        if (!accumulate)
            for (int c = 0; c < nb; ++c)
                HIPCHK(hipMemsetAsync(Hpsi + (int64_t)c * ldH, 0, rows * sizeof(cd), b->stream));
        return 0;
    }
    // scratch for the two n_p x nb panels lives in T1 (free outside the FFT pipeline)
    const size_t need = 2 * (size_t)kb->n_p * nb * sizeof(cd);
    if (need > b->T1_bytes) {
        HIPCHK(hipStreamSynchronize(b->stream));
        if (b->T1) HIPCHK(hipFree(b->T1));
        b->T1 = nullptr;
        b->T1_bytes = 0;
        HIPCHK(dftk_scratch_malloc((void**)&b->T1, need));
        b->T1_bytes = need;
    }
    cd* Ppsi = b->T1;
    cd* DPpsi = b->T1 + (size_t)kb->n_p * nb;
    // (sharded block: P, psi, Hpsi are row slabs; the projections are partial sums -> one small all-reduce)
    CHK(zgemm(b, 'C', kb->n_p, nb, rows, one, P, ldP, psi, ldpsi, zero, Ppsi, kb->n_p, gemm_flags));
    if (comm) CHK(comm_allreduce(comm, b, reinterpret_cast<double*>(Ppsi), 2 * (size_t)kb->n_p * nb));
    CHK(apply_D(kb, nb, Ppsi, DPpsi));
    CHK(zgemm(b, 'N', rows, nb, kb->n_p, one, P, ldP, DPpsi, kb->n_p, accumulate ? one : zero, Hpsi, ldH, gemm_flags));
    return 0;
}

static int apply_nonlocal(dftk_mi_kblock* kb, int nb, const cd* psi, int64_t ldpsi, cd* Hpsi, int64_t ldH,
                          bool accumulate) {
    return apply_nonlocal_rows(kb, nb, kb->P, kb->ldP, local_rows(kb), psi, ldpsi, Hpsi, ldH, accumulate, 0, kb->sh_comm);
}

extern "C" int dftk_mi_apply_H_parts(dftk_mi_kblock* kb, int which, int n_bands, const dftk_mi_cplx* psi_d,
                                     int64_t ld_psi, dftk_mi_cplx* Hpsi_d, int64_t ld_Hpsi) {
    if (!kb || !psi_d || !Hpsi_d || n_bands < 0 || (which & ~7)) return DFTK_MI_EINVAL;
    const int64_t rows = local_rows(kb);
    if (ld_psi < rows || ld_Hpsi < rows) return DFTK_MI_EINVAL;
    if (n_bands == 0) return 0;   // "Nothing to do if psi empty" (Hamiltonian.jl:141)
    if (batching() && !kb->sh_comm) {   // fiber of a batched multi-k call: all k-blocks' bands go through ONE pipeline later
        BOp o;
        o.b = kb->basis;
        o.kb = kb;
        o.type = BOP_APPLYH; o.flags = which; o.m = n_bands; o.A = psi_d; o.lda = ld_psi; o.C = Hpsi_d; o.ldc = ld_Hpsi;
        return batch_record(std::move(o));
    }
    HIPCHK(hipSetDevice(kb->basis->device));
    const cd* psi = reinterpret_cast<const cd*>(psi_d);
    cd* H = reinterpret_cast<cd*>(Hpsi_d);
    const bool local = (which & 1) && kb->d_Vs != nullptr;
    const bool kinetic = which & 2;
    const bool nonlocal = which & 4;
    const int slot = prof_begin(kb->basis, PROF_APPLY_H, (double)n_bands);
    struct G {
        dftk_mi_basis* b;
        int s;
        ~G() { prof_end(b, s); }
    } guard{kb->basis, slot};
    if (!kb->sh_comm) {
        // local (+ kinetic fused into the gather epilogue), or kinetic only / zero
        if (local) {   // what the sharded branch below would move and reduce (input of bench.py's Amdahl model)
            prof_count(kb->basis, PROF_A2A_MODEL, 16.0 * (double)kb->n_G * n_bands);
            prof_count(kb->basis, PROF_A2A_MODEL, 16.0 * (double)kb->n_G * n_bands);
        }
        if (nonlocal && kb->n_p > 0) prof_count(kb->basis, PROF_AR_MODEL, 16.0 * (double)kb->n_p * n_bands);
        CHK(launch_local_apply(kb, n_bands, psi, ld_psi, H, ld_Hpsi, kinetic, local));
    } else {
        // row-slab sharded block: the FFT pipeline needs whole bands -> slab -> band all-to-all, every rank
        // transforms its share of the bands, band -> slab all-to-all back
        if (ld_psi != rows || ld_Hpsi != rows) {
            dftk_set_error("sharded apply_H: blocks must be packed (leading dimension == local rows %lld)", (long long)rows);
            return DFTK_MI_EINVAL;
        }
        cd *R1, *F, *Gf;
        CHK(shard_buffers(kb, n_bands, &R1, &F, &Gf));
        Transposer t(kb, n_bands);
        CHK(t.to_bands(psi, R1, F));
        if (t.mine > 0) CHK(launch_local_apply(kb, t.mine, F, kb->n_G, Gf, kb->n_G, kinetic, local));
        CHK(t.to_slabs(Gf, R1, H));
    }
    if (nonlocal) CHK(apply_nonlocal(kb, n_bands, psi, ld_psi, H, ld_Hpsi, true));
    return 0;
}

extern "C" int dftk_mi_apply_H(dftk_mi_kblock* kb, int n_bands, const dftk_mi_cplx* psi_d, int64_t ld_psi,
                               dftk_mi_cplx* Hpsi_d, int64_t ld_Hpsi) {
    return dftk_mi_apply_H_parts(kb, 7, n_bands, psi_d, ld_psi, Hpsi_d, ld_Hpsi);
}

extern "C" int dftk_mi_local_potential(dftk_mi_kblock* cube_kb, const double* rho_d, const double* V_loc_d,
                                       const double* poisson_green_d, int xc_functionals, double* V_out_d,
                                       double* energies_h) {
    if (!cube_kb || !rho_d || (!energies_h && !V_out_d) || (xc_functionals & ~(7 | 32)) || cube_kb->sh_comm) return DFTK_MI_EINVAL;
    HIPCHK(hipSetDevice(cube_kb->basis->device));
    return local_potential_lda(cube_kb, nullptr, rho_d, V_loc_d, poisson_green_d, xc_functionals, 0.0, V_out_d,
                               energies_h);
}

extern "C" int dftk_mi_local_potential_collinear(dftk_mi_kblock* cube_kb, const double* rho_d, const double* V_loc_d,
                                                 const double* poisson_green_d, int xc_functionals, double* V_out_d,
                                                 double* energies_h) {
    if (!cube_kb || !rho_d || (!energies_h && !V_out_d) || cube_kb->sh_comm) return DFTK_MI_EINVAL;
    HIPCHK(hipSetDevice(cube_kb->basis->device));
    return local_potential_collinear(cube_kb, rho_d, V_loc_d, poisson_green_d, xc_functionals, V_out_d, energies_h);
}

// Julia's column-major 3x3 (entry (i, j) at i + 3 j) -> the kernels' row-major copy
static void rowmajor3(const double* colmajor, double* out) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) out[3 * i + j] = colmajor[i + 3 * j];
}

extern "C" int dftk_mi_local_potential_gga(dftk_mi_kblock* cube_kb, const double* recip_lattice_h, const double* rho_d,
                                           const double* V_loc_d, const double* poisson_green_d, int xc_functionals,
                                           double density_threshold, double* V_out_d, double* energies_h) {
    if (!cube_kb || !rho_d || (!energies_h && !V_out_d) || (xc_functionals & ~31) || cube_kb->sh_comm) return DFTK_MI_EINVAL;
    if ((xc_functionals & 24) && !recip_lattice_h) return DFTK_MI_EINVAL;
    HIPCHK(hipSetDevice(cube_kb->basis->device));
    double B[9] = {0};
    if (recip_lattice_h) rowmajor3(recip_lattice_h, B);
    return local_potential_lda(cube_kb, B, rho_d, V_loc_d, poisson_green_d, xc_functionals, density_threshold, V_out_d,
                               energies_h);
}

extern "C" int dftk_mi_symmetrize_rho(dftk_mi_kblock* cube_kb, int n_sym, const int32_t* S_h, const double* tau_h,
                                      int do_lowpass, const double* rho_in_d, double* rho_out_d) {
    if (!cube_kb || n_sym < 1 || !S_h || !tau_h || !rho_in_d || !rho_out_d || cube_kb->sh_comm) return DFTK_MI_EINVAL;
    HIPCHK(hipSetDevice(cube_kb->basis->device));
    std::vector<int32_t> S((size_t)9 * n_sym);
    for (int s = 0; s < n_sym; ++s)
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) S[9 * (size_t)s + 3 * i + j] = S_h[9 * (size_t)s + i + 3 * j];
    return cube_symmetrize(cube_kb, n_sym, S.data(), tau_h, do_lowpass, rho_in_d, rho_out_d);
}

static int filter_entry(dftk_mi_kblock* cube_kb, int kind, const double* recip_lattice_h, double p0, double p1,
                        const double* mult_d, const double* f_d, double* out_d) {
    if (!cube_kb || !f_d || !out_d || cube_kb->sh_comm) return DFTK_MI_EINVAL;
    if (kind == 3 ? !mult_d : !recip_lattice_h) return DFTK_MI_EINVAL;
    HIPCHK(hipSetDevice(cube_kb->basis->device));
    double B[9] = {0};
    if (recip_lattice_h) rowmajor3(recip_lattice_h, B);
    return cube_fourier_filter(cube_kb, kind, B, p0, p1, mult_d, f_d, out_d);
}

extern "C" int dftk_mi_mix_kerker(dftk_mi_kblock* cube_kb, const double* recip_lattice_h, double kTF, const double* dF_d,
                                  double* drho_d) {
    if (!(kTF >= 0)) return DFTK_MI_EINVAL;
    return filter_entry(cube_kb, 0, recip_lattice_h, kTF, 0.0, nullptr, dF_d, drho_d);
}

extern "C" int dftk_mi_mix_dielectric(dftk_mi_kblock* cube_kb, const double* recip_lattice_h, double kTF, double eps_r,
                                      const double* dF_d, double* drho_d) {
    if (!(kTF >= 0) || !(eps_r > 0)) return DFTK_MI_EINVAL;
    return filter_entry(cube_kb, 1, recip_lattice_h, kTF, eps_r, nullptr, dF_d, drho_d);
}

extern "C" int dftk_mi_chi0_dielectric_apply(dftk_mi_kblock* cube_kb, const double* recip_lattice_h, double kTF,
                                             double eps_r, const double* dV_d, double* out_d) {
    if (!(kTF >= 0) || !(eps_r > 0)) return DFTK_MI_EINVAL;
    return filter_entry(cube_kb, 2, recip_lattice_h, kTF, eps_r, nullptr, dV_d, out_d);
}

extern "C" int dftk_mi_cube_fourier_filter(dftk_mi_kblock* cube_kb, const double* multiplier_d, const double* f_d,
                                           double* out_d) {
    return filter_entry(cube_kb, 3, nullptr, 0.0, 0.0, multiplier_d, f_d, out_d);
}

extern "C" int dftk_mi_kpoint_sphere_host(int nx, int ny, int nz, const double* recip_lattice_h, const double* kcoord_h,
                                          double Ecut, int64_t cap, int64_t* n_G, int64_t* mapping0_h,
                                          double* kinetic_h, int32_t* G_h) {
    if (nx < 1 || ny < 1 || nz < 1 || !recip_lattice_h || !kcoord_h || !n_G || cap < 0) return DFTK_MI_EINVAL;
    return sphere_enumerate_host(nx, ny, nz, recip_lattice_h, kcoord_h, Ecut, cap, n_G, mapping0_h, kinetic_h, G_h);
}

extern "C" int dftk_mi_build_projectors_hgh(dftk_mi_basis* b, int64_t n_rows, const int32_t* G_d,
                                            const double* recip_lattice_h, const double* kcoord_h,
                                            double unit_cell_volume, int n_species, const double* rp_h,
                                            const int* n_proj_h, int n_atoms, const int* species_of_atom_h,
                                            const double* positions_h, dftk_mi_cplx* P_d, int64_t ldP, int* n_p) {
    if (!b || n_rows < 0 || !recip_lattice_h || !kcoord_h || !(unit_cell_volume > 0) || n_species < 1 || !rp_h ||
        !n_proj_h || n_atoms < 0 || (n_atoms > 0 && (!species_of_atom_h || !positions_h)) || !n_p ||
        (P_d && !G_d))
        return DFTK_MI_EINVAL;
    HIPCHK(hipSetDevice(b->device));
    return build_projectors_hgh(b, n_rows, G_d, recip_lattice_h, kcoord_h, unit_cell_volume, n_species, rp_h, n_proj_h,
                                n_atoms, species_of_atom_h, positions_h, reinterpret_cast<cd*>(P_d), ldP, n_p);
}

extern "C" int dftk_mi_atomic_superposition(dftk_mi_kblock* cube_kb, int kind, const double* recip_lattice_h,
                                            int n_species, const double* params_h, int n_atoms,
                                            const int* species_of_atom_h, const double* positions_h, double* out_d) {
    if (!cube_kb || (kind != 0 && kind != 1) || !recip_lattice_h || n_species < 1 || !params_h || n_atoms < 1 ||
        !species_of_atom_h || !positions_h || !out_d || cube_kb->sh_comm)
        return DFTK_MI_EINVAL;
    HIPCHK(hipSetDevice(cube_kb->basis->device));
    return atomic_superposition(cube_kb, kind, recip_lattice_h, n_species, params_h, n_atoms, species_of_atom_h,
                                positions_h, out_d);
}

extern "C" int dftk_mi_xc_gga(dftk_mi_basis* b, int64_t n, const double* rho_d, const double* sigma_d, int xc_functionals,
                              double density_threshold, double* e_d, double* vrho_d, double* vsigma_d) {
    if (!b || n < 0 || !rho_d || !sigma_d || !e_d || !vrho_d || !vsigma_d || (xc_functionals & ~24) || !xc_functionals)
        return DFTK_MI_EINVAL;
    if (n == 0) return 0;
    HIPCHK(hipSetDevice(b->device));
    return xc_gga_pointwise(b, n, rho_d, sigma_d, xc_functionals, density_threshold, e_d, vrho_d, vsigma_d);
}

extern "C" int dftk_mi_ifft_sphere(dftk_mi_kblock* kb, const dftk_mi_cplx* c_d, dftk_mi_cplx* cube_d) {
    if (!kb || !c_d || !cube_d) return DFTK_MI_EINVAL;
    HIPCHK(hipSetDevice(kb->basis->device));
    return launch_ifft_to_cube(kb, reinterpret_cast<const cd*>(c_d), reinterpret_cast<cd*>(cube_d));
}

extern "C" int dftk_mi_fft_sphere(dftk_mi_kblock* kb, const dftk_mi_cplx* cube_d, dftk_mi_cplx* c_d) {
    if (!kb || !c_d || !cube_d) return DFTK_MI_EINVAL;
    HIPCHK(hipSetDevice(kb->basis->device));
    return launch_fft_from_cube(kb, reinterpret_cast<const cd*>(cube_d), reinterpret_cast<cd*>(c_d));
}

extern "C" int dftk_mi_density_accumulate(dftk_mi_kblock* kb, int n_bands, const dftk_mi_cplx* psi_d, int64_t ld_psi,
                                          const double* weight_h, double* rho_d) {
    if (!kb || !psi_d || !weight_h || !rho_d || n_bands < 0 || ld_psi < local_rows(kb)) return DFTK_MI_EINVAL;
    if (n_bands == 0) return 0;
    HIPCHK(hipSetDevice(kb->basis->device));
    if (!kb->sh_comm) {
        prof_count(kb->basis, PROF_A2A_MODEL, 16.0 * (double)kb->n_G * n_bands);   // (the transpose of the sharded branch)
        return launch_density(kb, n_bands, reinterpret_cast<const cd*>(psi_d), ld_psi, weight_h, rho_d);
    }
    // sharded block: every rank accumulates |psi|^2 of its share of the bands into ITS rho (partial sum);
    // the caller's density all-reduce (mpi_sum!(rho, comm), densities.jl:46) completes it
    if (ld_psi != local_rows(kb)) {
        dftk_set_error("sharded density: blocks must be packed (leading dimension == local rows)");
        return DFTK_MI_EINVAL;
    }
    cd *R1, *F, *Gf;
    CHK(shard_buffers(kb, n_bands, &R1, &F, &Gf));
    Transposer t(kb, n_bands);
    CHK(t.to_bands(reinterpret_cast<const cd*>(psi_d), R1, F));
    if (t.mine == 0) return 0;
    return launch_density(kb, t.mine, F, kb->n_G, weight_h + t.c0[t.me], rho_d);
}

// compute_density's accumulation with the spin index of the k-block spelled out (densities.jl:39:
// rho[:, :, :, kpt.spin] += ...): rho_d holds n_spin cubes, the bands of this block go to cube `spin` (0-based)
extern "C" int dftk_mi_density_accumulate_spin(dftk_mi_kblock* kb, int n_bands, const dftk_mi_cplx* psi_d, int64_t ld_psi,
                                               const double* weight_h, double* rho_d, int spin, int n_spin) {
    if (!kb || !rho_d || n_spin < 1 || n_spin > 2 || spin < 0 || spin >= n_spin) return DFTK_MI_EINVAL;
    const int64_t N = (int64_t)kb->basis->nx * kb->basis->ny * kb->basis->nz;
    return dftk_mi_density_accumulate(kb, n_bands, psi_d, ld_psi, weight_h, rho_d + (int64_t)spin * N);
}

// ------------------------------------------------------------------------------------ Gamma-real extension
extern "C" int dftk_mi_kblock_set_gamma_real(dftk_mi_kblock* kb, int on) {
    if (!kb) return DFTK_MI_EINVAL;
    HIPCHK(hipSetDevice(kb->basis->device));
    return gamma_enable(kb, on);
}

extern "C" int dftk_mi_gamma_tables_host(int nx, int ny, int nz, int64_t n_G, const int64_t* mapping0_h,
                                         int64_t* n_half, int32_t* row_h, int32_t* partner_row_h) {
    if (nx < 1 || ny < 1 || nz < 1 || n_G < 1 || !mapping0_h || !n_half) return DFTK_MI_EINVAL;
    if ((row_h == nullptr) != (partner_row_h == nullptr)) return DFTK_MI_EINVAL;
    return gamma_tables_host(nx, ny, nz, n_G, mapping0_h, n_half, row_h, partner_row_h);
}

static int gamma_ready(dftk_mi_kblock* kb) {
    if (!kb || !kb->gr || !kb->gr->d_g) {
        dftk_set_error("gamma_real: call dftk_mi_kblock_set_gamma_real(kb, 1) first");
        return DFTK_MI_EINVAL;
    }
    return 0;
}

extern "C" int dftk_mi_gamma_half_size(dftk_mi_kblock* kb, int64_t* n_half) {
    if (!n_half) return DFTK_MI_EINVAL;
    CHK(gamma_ready(kb));
    *n_half = gamma_local_rows(kb);     // (the rows of this rank on a plane-wave sharded block)
    return 0;
}

extern "C" int dftk_mi_gamma_compress(dftk_mi_kblock* kb, int m, const dftk_mi_cplx* X_d, int64_t ldx,
                                      dftk_mi_cplx* Xh_d, int64_t ldh) {
    CHK(gamma_ready(kb));
    if (m < 0 || !X_d || !Xh_d || ldx < local_rows(kb) || ldh < gamma_local_rows(kb)) return DFTK_MI_EINVAL;
    HIPCHK(hipSetDevice(kb->basis->device));
    if (kb->sh_comm)   // row slabs in, row slabs out (collective over the block's communicator)
        return gamma_lobpcg_load(kb, m, reinterpret_cast<const cd*>(X_d), ldx, reinterpret_cast<cd*>(Xh_d), ldh);
    return gamma_compress(kb, m, reinterpret_cast<const cd*>(X_d), ldx, reinterpret_cast<cd*>(Xh_d), ldh);
}

extern "C" int dftk_mi_gamma_compress_aligned(dftk_mi_kblock* kb, int m, const dftk_mi_cplx* X_d, int64_t ldx,
                                              dftk_mi_cplx* Xh_d, int64_t ldh) {
    CHK(gamma_ready(kb));
    if (m < 0 || !X_d || !Xh_d || ldx < local_rows(kb) || ldh < gamma_local_rows(kb)) return DFTK_MI_EINVAL;
    HIPCHK(hipSetDevice(kb->basis->device));
    return gamma_lobpcg_load(kb, m, reinterpret_cast<const cd*>(X_d), ldx, reinterpret_cast<cd*>(Xh_d), ldh, true);
}

extern "C" int dftk_mi_gamma_expand(dftk_mi_kblock* kb, int m, const dftk_mi_cplx* Xh_d, int64_t ldh,
                                    dftk_mi_cplx* X_d, int64_t ldx) {
    CHK(gamma_ready(kb));
    if (m < 0 || !X_d || !Xh_d || ldx < local_rows(kb) || ldh < gamma_local_rows(kb)) return DFTK_MI_EINVAL;
    HIPCHK(hipSetDevice(kb->basis->device));
    if (kb->sh_comm)
        return gamma_lobpcg_store(kb, m, reinterpret_cast<const cd*>(Xh_d), ldh, reinterpret_cast<cd*>(X_d), ldx);
    return gamma_expand(kb, m, reinterpret_cast<const cd*>(Xh_d), ldh, reinterpret_cast<cd*>(X_d), ldx);
}

extern "C" int dftk_mi_gamma_apply_H(dftk_mi_kblock* kb, int which, int n_bands, const dftk_mi_cplx* psih_d,
                                     int64_t ld_psi, dftk_mi_cplx* Hpsih_d, int64_t ld_Hpsi) {
    CHK(gamma_ready(kb));
    if (!psih_d || !Hpsih_d || n_bands < 0 || (which & ~7) || ld_psi < gamma_local_rows(kb) ||
        ld_Hpsi < gamma_local_rows(kb))
        return DFTK_MI_EINVAL;
    HIPCHK(hipSetDevice(kb->basis->device));
    return gamma_apply_H(kb, which, n_bands, reinterpret_cast<const cd*>(psih_d), ld_psi,
                         reinterpret_cast<cd*>(Hpsih_d), ld_Hpsi);
}

extern "C" int dftk_mi_density_accumulate_real(dftk_mi_kblock* kb, int n_bands, const dftk_mi_cplx* psi_d,
                                               int64_t ld_psi, const double* weight_h, double* rho_d) {
    if (!kb || !psi_d || !weight_h || !rho_d || n_bands < 0 || ld_psi < local_rows(kb)) return DFTK_MI_EINVAL;
    if (n_bands == 0) return 0;
    HIPCHK(hipSetDevice(kb->basis->device));
    if (!kb->sh_comm) prof_count(kb->basis, PROF_A2A_MODEL, 8.0 * (double)kb->n_G * n_bands);   // half-format slabs -> bands
    return gamma_density(kb, n_bands, reinterpret_cast<const cd*>(psi_d), ld_psi, weight_h, rho_d);
}

extern "C" int dftk_mi_lobpcg(dftk_mi_kblock* kb, int M, dftk_mi_cplx* X_d, int64_t ldX, double tol, int miniter,
                              int maxiter, int n_conv_check, int use_tpa, uint64_t seed, double* lambda_h,
                              double* resid_h, int* n_iter, int* converged, int64_t* n_matvec) {
    if (!kb || !X_d || M < 1 || ldX < local_rows(kb) || !lambda_h || !resid_h || !n_iter || !converged || !n_matvec ||
        maxiter < 0)
        return DFTK_MI_EINVAL;
    HIPCHK(hipSetDevice(kb->basis->device));
    return lobpcg_run(kb, M, reinterpret_cast<cd*>(X_d), ldX, tol, miniter, maxiter, n_conv_check, use_tpa, seed,
                      lambda_h, resid_h, n_iter, converged, n_matvec);
}

extern "C" int dftk_mi_lobpcg_multi(int n_kblocks, dftk_mi_kblock* const* kbs, int M, dftk_mi_cplx* const* X_d,
                                    const int64_t* ldX, double tol, int miniter, int maxiter, int n_conv_check, int use_tpa,
                                    const uint64_t* seeds, double* lambda_h, double* resid_h, int* n_iter, int* converged,
                                    int64_t* n_matvec, int* status) {
    if (n_kblocks < 0 || (n_kblocks > 0 && (!kbs || !X_d || !ldX || !lambda_h || !resid_h || !n_iter || !converged ||
                                            !n_matvec || !status)) || M < 1 || maxiter < 0)
        return DFTK_MI_EINVAL;
    if (n_kblocks == 0) return 0;
    for (int i = 0; i < n_kblocks; ++i)
        if (!kbs[i] || !X_d[i] || ldX[i] < local_rows(kbs[i])) return DFTK_MI_EINVAL;
    HIPCHK(hipSetDevice(kbs[0]->basis->device));
    return lobpcg_run_multi(n_kblocks, kbs, M, reinterpret_cast<cd* const*>(X_d), ldX, tol, miniter, maxiter, n_conv_check,
                            use_tpa, seeds, lambda_h, resid_h, n_iter, converged, n_matvec, status);
}

// kinetic energy of every band of n k-blocks: out_h = [sum_G kin_G |psi_Gn|^2 for the bands of k-block 0, then 1, ...]
// (the per-band terms of ene_ops(::KineticOperator), src/terms/kinetic.jl:49-54; the reduction of precondprep!)
extern "C" int dftk_mi_band_kinetic_multi(int n_kblocks, dftk_mi_kblock* const* kbs, const int* n_bands,
                                          const dftk_mi_cplx* const* psi_d, const int64_t* ld_psi, double* out_h) {
    if (n_kblocks < 0 || (n_kblocks > 0 && (!kbs || !n_bands || !psi_d || !ld_psi || !out_h))) return DFTK_MI_EINVAL;
    if (n_kblocks == 0) return 0;
    dftk_mi_basis* b = kbs[0] ? kbs[0]->basis : nullptr;
    if (!b) return DFTK_MI_EINVAL;
    size_t total = 0;
    for (int i = 0; i < n_kblocks; ++i) {
        if (!kbs[i] || kbs[i]->basis != b || kbs[i]->sh_comm || !psi_d[i] || n_bands[i] < 0 || ld_psi[i] < kbs[i]->n_G)
            return DFTK_MI_EINVAL;
        total += (size_t)n_bands[i];
    }
    HIPCHK(hipSetDevice(b->device));
    CHK(ensure_ws(b, total * sizeof(double)));
    double* d = reinterpret_cast<double*>(b->ws);
    std::vector<std::function<int()>> bodies;
    size_t off = 0;
    for (int i = 0; i < n_kblocks; ++i) {
        const size_t o = off;
        off += (size_t)n_bands[i];
        bodies.push_back([=]() {
            if (n_bands[i] == 0) return 0;
            CHK(ew_weighted_colsums(b, kbs[i]->n_G, n_bands[i], reinterpret_cast<const cd*>(psi_d[i]), ld_psi[i], kbs[i]->d_kin,
                                    d + o));
            return dev_d2h_sync(b, out_h + o, d + o, (size_t)n_bands[i] * sizeof(double));
        });
    }
    std::vector<int> rets;
    CHK(batch_run(b, bodies, rets));
    for (int r : rets)
        if (r != 0) return r;
    return 0;
}

extern "C" int dftk_mi_density_accumulate_multi(int n_kblocks, dftk_mi_kblock* const* kbs, const int* n_bands,
                                                const dftk_mi_cplx* const* psi_d, const int64_t* ld_psi,
                                                const double* weights_h, double* rho_d) {
    return dftk_mi_density_accumulate_multi2(n_kblocks, kbs, n_bands, psi_d, ld_psi, weights_h, rho_d, nullptr, nullptr);
}

extern "C" int dftk_mi_density_accumulate_multi2(int n_kblocks, dftk_mi_kblock* const* kbs, const int* n_bands,
                                                 const dftk_mi_cplx* const* psi_d, const int64_t* ld_psi,
                                                 const double* weights_h, double* rho_d, const double* weights2_h,
                                                 double* rho2_d) {
    if (n_kblocks < 0 || (n_kblocks > 0 && (!kbs || !n_bands || !psi_d || !ld_psi || !weights_h)) || !rho_d ||
        ((weights2_h == nullptr) != (rho2_d == nullptr)) || (rho2_d && rho2_d == rho_d))
        return DFTK_MI_EINVAL;
    if (n_kblocks == 0) return 0;
    dftk_mi_basis* b = kbs[0] ? kbs[0]->basis : nullptr;
    if (!b) return DFTK_MI_EINVAL;
    for (int i = 0; i < n_kblocks; ++i)
        if (!kbs[i] || kbs[i]->basis != b || kbs[i]->sh_comm || !psi_d[i] || n_bands[i] < 0 || ld_psi[i] < kbs[i]->n_G)
            return DFTK_MI_EINVAL;
    HIPCHK(hipSetDevice(b->device));
    // every k-block records its accumulation; the queue is merged into ONE pipeline over all bands (batch.h)
    std::vector<std::function<int()>> bodies;
    size_t off = 0;
    for (int i = 0; i < n_kblocks; ++i) {
        const double* w = weights_h + off;
        const double* w2 = weights2_h ? weights2_h + off : nullptr;
        off += (size_t)n_bands[i];
        bodies.push_back([=]() {
            return launch_density(kbs[i], n_bands[i], reinterpret_cast<const cd*>(psi_d[i]), ld_psi[i], w, rho_d, nullptr, w2,
                                  rho2_d);
        });
    }
    std::vector<int> rets;
    CHK(batch_run(b, bodies, rets));
    for (int r : rets)
        if (r != 0) return r;
    return 0;
}

extern "C" int dftk_mi_fermi_bisection(int n_k, const int* n_bands, const double* eig, const double* kweights, int smearing,
                                       double temperature, double filled, double n_electrons, double lo, double hi,
                                       double* eF_out) {
    if (n_k < 1 || !n_bands || !eig || !kweights || !eF_out || !(temperature > 0.0) || (smearing != 1 && smearing != 2) ||
        !(lo <= hi))
        return DFTK_MI_EINVAL;
    auto excess = [&](double eF) {
        double total = 0.0;
        const double* e = eig;
        for (int k = 0; k < n_k; ++k) {
            double sk = 0.0;
            for (int i = 0; i < n_bands[k]; ++i) {
                const double x = (e[i] - eF) / temperature;
                double f;
                if (smearing == 2) {
                    f = 0.5 * std::erfc(x);
                } else if (x > 0) {                     // overflow-safe Fermi-Dirac (Smearing.jl:66-76)
                    const double y = std::exp(-x);
                    f = y / (1.0 + y);
                } else {
                    f = 1.0 / (1.0 + std::exp(x));
                }
                sk += filled * f;
            }
            total += kweights[k] * sk;
            e += n_bands[k];
        }
        return total - n_electrons;
    };
    for (int it = 0; it < 200; ++it) {
        const double mid = 0.5 * (lo + hi);
        if (mid == lo || mid == hi) break;
        if (excess(mid) < 0)
            lo = mid;
        else
            hi = mid;
    }
    *eF_out = 0.5 * (lo + hi);
    return 0;
}

extern "C" int dftk_mi_kblock_reuse_AX(dftk_mi_kblock* kb, int on) {
    if (!kb) return DFTK_MI_EINVAL;
    kb->ax_reuse_next = on != 0 && kb->ax_keep != nullptr;
    return 0;
}

extern "C" const dftk_mi_cplx* dftk_mi_lobpcg_last_AX(dftk_mi_kblock* kb) {
    return kb ? reinterpret_cast<const dftk_mi_cplx*>(kb->last_AX) : nullptr;
}

// ------------------------------------------------------------------------------------ dense helpers
extern "C" int dftk_mi_zgemm(dftk_mi_basis* b, char transA, int64_t m, int64_t n, int64_t k, dftk_mi_cplx alpha,
                             const dftk_mi_cplx* A_d, int64_t lda, const dftk_mi_cplx* B_d, int64_t ldb,
                             dftk_mi_cplx beta, dftk_mi_cplx* C_d, int64_t ldc) {
    if (!b || m < 0 || n < 0 || k < 0 || !C_d) return DFTK_MI_EINVAL;
    HIPCHK(hipSetDevice(b->device));
    cd al = {alpha.re, alpha.im}, be = {beta.re, beta.im};
    return zgemm(b, transA, m, n, k, al, reinterpret_cast<const cd*>(A_d), lda, reinterpret_cast<const cd*>(B_d), ldb,
                 be, reinterpret_cast<cd*>(C_d), ldc);
}

int zgemm_plan_host(char transA, int64_t m, int64_t n, int64_t k, int flags, int* out);   // gemm_kernels.hip
extern "C" int dftk_mi_zgemm_plan_host(char transA, int64_t m, int64_t n, int64_t k, int flags, int* out12) {
    return zgemm_plan_host(transA, m, n, k, flags, out12);
}

extern "C" int dftk_mi_zgemm_ex(dftk_mi_basis* b, char transA, int64_t m, int64_t n, int64_t k, dftk_mi_cplx alpha,
                                const dftk_mi_cplx* A_d, int64_t lda, const dftk_mi_cplx* B_d, int64_t ldb,
                                dftk_mi_cplx beta, dftk_mi_cplx* C_d, int64_t ldc, int flags) {
    if (!b || m < 0 || n < 0 || k < 0 || !C_d || (flags & ~(3 | DFTK_MI_GEMM_REAL))) return DFTK_MI_EINVAL;
    HIPCHK(hipSetDevice(b->device));
    cd al = {alpha.re, alpha.im}, be = {beta.re, beta.im};
    return zgemm(b, transA, m, n, k, al, reinterpret_cast<const cd*>(A_d), lda, reinterpret_cast<const cd*>(B_d), ldb,
                 be, reinterpret_cast<cd*>(C_d), ldc, flags);
}

extern "C" int dftk_mi_heev(dftk_mi_basis* b, int n, dftk_mi_cplx* A_d, int64_t lda, double* W_h, dftk_mi_cplx* V_d,
                            int64_t ldv) {
    if (!b || n < 1 || !A_d || !W_h || !V_d || lda < n || ldv < n) return DFTK_MI_EINVAL;
    HIPCHK(hipSetDevice(b->device));
    return dense_heev(b, n, reinterpret_cast<cd*>(A_d), lda, W_h, reinterpret_cast<cd*>(V_d), ldv);
}

extern "C" int dftk_mi_heev_lowest(dftk_mi_basis* b, int n, int nev, dftk_mi_cplx* A_d, int64_t lda, double* W_h,
                                   dftk_mi_cplx* V_d, int64_t ldv) {
    if (!b || n < 1 || nev < 1 || nev > n || !A_d || !W_h || !V_d || lda < n || ldv < n) return DFTK_MI_EINVAL;
    HIPCHK(hipSetDevice(b->device));
    return dense_heev_lowest(b, n, nev, reinterpret_cast<cd*>(A_d), lda, W_h, reinterpret_cast<cd*>(V_d), ldv);
}
extern "C" int dftk_mi_heev_sigma_host(int n, const double* diag_h, int nev, double* sigma, double* gap_guess,
                                       int* hold_iterations, double norm_bound) {
    const int st = eig_choose_sigma_host(n, diag_h, nev, sigma, gap_guess);
    if (st != 0) return st;
    if (hold_iterations) *hold_iterations = norm_bound > 0.0 ? eig_hold_iterations_host(*gap_guess / norm_bound) : 0;
    return 0;
}

extern "C" int dftk_mi_potrf_trtri(dftk_mi_basis* b, int n, dftk_mi_cplx* A_d, int64_t lda, dftk_mi_cplx* invR_d,
                                   int64_t ldi) {
    if (!b || n < 1 || !A_d || !invR_d || lda < n || ldi < n) return DFTK_MI_EINVAL;
    HIPCHK(hipSetDevice(b->device));
    double a, c;
    return dense_potrf_trtri(b, n, reinterpret_cast<cd*>(A_d), lda, reinterpret_cast<cd*>(invR_d), ldi, &a, &c);
}

extern "C" int dftk_mi_potrf_trtri_real(dftk_mi_basis* b, int n, dftk_mi_cplx* A_d, int64_t lda, dftk_mi_cplx* invR_d,
                                        int64_t ldi) {
    if (!b || n < 1 || !A_d || !invR_d || lda < n || ldi < n) return DFTK_MI_EINVAL;
    HIPCHK(hipSetDevice(b->device));
    double a, c;
    return dense_potrf_trtri(b, n, reinterpret_cast<cd*>(A_d), lda, reinterpret_cast<cd*>(invR_d), ldi, &a, &c, true);
}

// ------------------------------------------------------------------------------------ LOBPCG building blocks
extern "C" int dftk_mi_lobpcg_history(dftk_mi_kblock* kb, int* M_out, int* n_iter_out, double* hist_h, size_t cap,
                                      int* n_svd_out) {
    if (!kb || !kb->lob_hist) return DFTK_MI_EINVAL;
    if (M_out) *M_out = kb->lob_hist_M;
    if (n_iter_out) *n_iter_out = kb->lob_hist_iters;
    if (n_svd_out) *n_svd_out = kb->lob_n_svd;
    if (hist_h) {
        if (cap < kb->lob_hist->size()) return DFTK_MI_EINVAL;
        std::copy(kb->lob_hist->begin(), kb->lob_hist->end(), hist_h);
    }
    return 0;
}

static int fetch(dftk_mi_basis* b, const double* d, double* h, size_t n) {
    HIPCHK(hipMemcpyAsync(h, d, n * sizeof(double), hipMemcpyDeviceToHost, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));
    return 0;
}

extern "C" int dftk_mi_columnwise_norms(dftk_mi_basis* b, int64_t n, int m, const dftk_mi_cplx* X_d, int64_t ldx,
                                        double* norms_h) {
    if (!b || !X_d || !norms_h || n < 0 || m < 0 || ldx < n || m > 100000) return DFTK_MI_EINVAL;
    if (m == 0) return 0;
    HIPCHK(hipSetDevice(b->device));
    CHK(ensure_ws(b, (size_t)m * sizeof(double)));
    double* d = reinterpret_cast<double*>(b->ws);
    CHK(ew_colnorms(b, n, m, reinterpret_cast<const cd*>(X_d), ldx, d));
    return fetch(b, d, norms_h, m);
}

extern "C" int dftk_mi_columnwise_dots(dftk_mi_basis* b, int64_t n, int m, const dftk_mi_cplx* A_d, int64_t lda,
                                       const dftk_mi_cplx* B_d, int64_t ldb, dftk_mi_cplx* dots_h) {
    if (!b || !A_d || !B_d || !dots_h || n < 0 || m < 0 || lda < n || ldb < n) return DFTK_MI_EINVAL;
    if (m == 0) return 0;
    HIPCHK(hipSetDevice(b->device));
    CHK(ensure_ws(b, 2 * (size_t)m * sizeof(double)));
    double* d = reinterpret_cast<double*>(b->ws);
    CHK(ew_coldots(b, n, m, reinterpret_cast<const cd*>(A_d), lda, reinterpret_cast<const cd*>(B_d), ldb, d));
    CHK(ew_coldots_im(b, n, m, reinterpret_cast<const cd*>(A_d), lda, reinterpret_cast<const cd*>(B_d), ldb, d + m));
    std::vector<double> h(2 * (size_t)m);
    CHK(fetch(b, d, h.data(), h.size()));
    for (int i = 0; i < m; ++i) {
        dots_h[i].re = h[i];
        dots_h[i].im = h[m + i];
    }
    return 0;
}

extern "C" int dftk_mi_ortho_qr(dftk_mi_basis* b, int64_t n, int m, dftk_mi_cplx* X_d, int64_t ldx, int force_svd,
                                int* n_chol, int* used_svd) {
    if (!b || !X_d || n < 1 || m < 0 || ldx < n || m > n) return DFTK_MI_EINVAL;
    HIPCHK(hipSetDevice(b->device));
    return lobpcg_ortho(b, n, m, reinterpret_cast<cd*>(X_d), ldx, force_svd, n_chol, used_svd);
}

extern "C" int dftk_mi_tpa_precondprep(dftk_mi_kblock* kb, int m, const dftk_mi_cplx* X_d, int64_t ldx,
                                       double* mean_kin_h) {
    if (!kb || !X_d || !mean_kin_h || m < 0 || ldx < kb->n_G || kb->sh_comm) return DFTK_MI_EINVAL;
    if (m == 0) return 0;
    dftk_mi_basis* b = kb->basis;
    HIPCHK(hipSetDevice(b->device));
    CHK(ensure_ws(b, (size_t)m * sizeof(double)));
    double* d = reinterpret_cast<double*>(b->ws);
    CHK(ew_weighted_colsums(b, kb->n_G, m, reinterpret_cast<const cd*>(X_d), ldx, kb->d_kin, d));
    return fetch(b, d, mean_kin_h, m);
}

extern "C" int dftk_mi_tpa_ldiv(dftk_mi_kblock* kb, int m, const dftk_mi_cplx* R_d, int64_t ldr,
                                const double* mean_kin_h, double default_shift, dftk_mi_cplx* Y_d, int64_t ldy) {
    if (!kb || !R_d || !Y_d || m < 0 || ldr < kb->n_G || ldy < kb->n_G || kb->sh_comm) return DFTK_MI_EINVAL;
    if (m == 0) return 0;
    dftk_mi_basis* b = kb->basis;
    HIPCHK(hipSetDevice(b->device));
    CHK(ensure_ws(b, 2 * (size_t)m * sizeof(double)));
    double* d = reinterpret_cast<double*>(b->ws);
    if (mean_kin_h) {
        HIPCHK(hipMemcpyAsync(d, mean_kin_h, m * sizeof(double), hipMemcpyHostToDevice, b->stream));
        HIPCHK(hipStreamSynchronize(b->stream));
    }
    return ew_tpa(b, kb->n_G, m, reinterpret_cast<const cd*>(R_d), ldr, reinterpret_cast<cd*>(Y_d), ldy, kb->d_kin,
                  mean_kin_h ? d : nullptr, d + m, default_shift);
}

extern "C" int dftk_mi_block_residual(dftk_mi_kblock* kb, int m, const dftk_mi_cplx* AX_d, int64_t lda,
                                      const dftk_mi_cplx* X_d, int64_t ldx, const double* lambda_h, dftk_mi_cplx* R_d,
                                      int64_t ldr, double* norms_h, double* mean_kin_h, double* xx_h) {
    if (!kb || !AX_d || !X_d || !lambda_h || !R_d || !norms_h || m < 0 || lda < kb->n_G || ldx < kb->n_G ||
        ldr < kb->n_G || kb->sh_comm)
        return DFTK_MI_EINVAL;
    if (m == 0) return 0;
    dftk_mi_basis* b = kb->basis;
    HIPCHK(hipSetDevice(b->device));
    CHK(ensure_ws(b, 4 * (size_t)m * sizeof(double)));
    double* d = reinterpret_cast<double*>(b->ws);
    HIPCHK(hipMemcpyAsync(d, lambda_h, m * sizeof(double), hipMemcpyHostToDevice, b->stream));
    CHK(ew_residual(b, kb->n_G, m, reinterpret_cast<const cd*>(AX_d), lda, reinterpret_cast<const cd*>(X_d), ldx, d,
                    reinterpret_cast<cd*>(R_d), ldr, d + m, kb->d_kin, d + 2 * m, d + 3 * m));
    std::vector<double> h(3 * (size_t)m);
    CHK(fetch(b, d + m, h.data(), h.size()));
    for (int i = 0; i < m; ++i) {
        norms_h[i] = h[i];
        if (mean_kin_h) mean_kin_h[i] = h[m + i];
        if (xx_h) xx_h[i] = h[2 * m + i];
    }
    return 0;
}
