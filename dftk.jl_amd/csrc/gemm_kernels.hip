// gemm_kernels.hip -- complex fp64 GEMM on the f64 matrix cores (v_mfma_f64_16x16x4_f64, gfx950).
//
// Covers every BLAS3 call of the reference's hot path:
//   P' * psi, P * (D P' psi)                    src/terms/operators.jl:126-128          (K7, K9)
//   mul_hermi(Y', AY), BY' * X                  src/eigen/lobpcg_hyper_impl.jl:90-121,278 (K10, K15)
//   Y * cX, X * invR                            :124-132, 234                           (K12, K14)
// as  C = alpha * op(A) * B + beta * C  with op(A) in {A, A^H}, B never transposed, column-major.
//
// Kernels:
//   k_zgemm_3m   default.  Three real MFMA streams per complex product (Karatsuba), 128 x 32 workgroup
//                tiles, operands staged through LDS, software pipeline with one mid-tile barrier.
//   k_zgemm_lds  the classic four-product kernel on 128 x 64 tiles (same staging / pipeline), selected
//                with DFTK_MI_GEMM_4M=1; kept as the measured baseline of the 3M kernel.
//   k_zgemm_naive  one thread per C entry, no matrix cores (DFTK_MI_GEMM=naive): debugging reference.
//   k_zgemm_reduce fixed-order sum of the split-K slabs => bitwise reproducible results.
// Host side: tiling into full tiles + ragged border, a cost-model K split that fills the resident
// workgroup slots once, XCD-aware 1-D grids (zgemm(), gemm_tiling(), gemm_plan_split()).
#include "common.h"
#include "batch.h"
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <map>
#include <vector>
#include <cmath>
#include <algorithm>
#include <mutex>

typedef double v4d __attribute__((ext_vector_type(4)));

#define GEMM_RM 2
#define GEMM_RN 4
#define GEMM_WAVES 4
#define GEMM_BM (GEMM_WAVES * GEMM_RM * 16)   // 128
#define GEMM_BN (GEMM_RN * 16)                // 64

// lane mapping of v_mfma_f64_16x16x4_f64:
//   A operand: lane l holds A[i = l & 15][k = l >> 4];  B operand: lane l holds B[k = l >> 4][j = l & 15]
//   C/D: 4 values per lane, value r is C[row = (l >> 4) + 4 r][col = l & 15]

// ------------------------------------------------------------------------------------------------
// LDS-staged four-product kernel.  A wave owns a 32 x 64 output tile (two accumulator sets of 8 MFMA
// tiles = 128 VGPRs), a workgroup is 4 waves stacked along m.  The K-tile (8 deep) of both operands is
// staged through LDS:
//   * global reads are fully coalesced (8 lanes x 16 B = one 128-B line per column of a K-major
//     operand; 128 consecutive rows of an M-major one) and each element is fetched ONCE per
//     workgroup instead of once per wave that needs it;
//   * LDS image is [k][column] with a 16-element-aligned pitch, so a wave's MFMA fragment read
//     (16 columns x 2 k per lane group) touches 16 distinct 16-B slots: conflict-free
//     ds_read_b128; K-major operands are stored with the column index XOR (k & 7) so that the
//     transposing ds_write_b128 of 8 consecutive-k lanes also hits 8 distinct slots;
//   * register-staged software pipeline: the global loads of tile t+1 are in flight while the 64
//     MFMAs of tile t issue; one barrier per tile, two LDS buffers.
// Diagnostic (DFTK_MI_GEMM_CLOCK=1 in tools/lab): shader clock actually sustained inside the LDS kernel.
// Workgroup 0 of every launch stores its shader-cycle and 100 MHz wall-tick counts.
__device__ long long g_gemm_clk[4];
int zgemm_debug_clock(double* mhz, double* us) {
    long long h[4];
    HIPCHK(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_gemm_clk), sizeof(h)));
    const double ticks = (double)(h[3] - h[2]);
    *us = ticks / 100.0;
    *mhz = ticks > 0 ? (double)(h[1] - h[0]) / (ticks / 100.0) : 0.0;
    return 0;
}
#define LT_KT 8
// FULL: launched only over tiles that lie entirely inside C -> no per-sub-tile predicates (runtime
// predicates become a branch per MFMA and break the MFMA issue stream).  gm x gn is the tile
// sub-grid of this launch, (rt0, ct0) its origin in tiles (see lsplit below for the border launch).
template <bool CONJA, int MODE>
__global__ __launch_bounds__(GEMM_WAVES * 64) void k_zgemm_lds(int m, int n, int K, int kchunk, int gm, int gn,
                                                               int rt0, int ct0, int lsplit, int upper,
                                                               int nsplit, const cd* __restrict__ A, int64_t lda,
                                                               const cd* __restrict__ B, int64_t ldb,
                                                               cd* __restrict__ C, int64_t ldc, cd alpha, cd beta,
                                                               cd* __restrict__ slab) {
    __shared__ cd sA[2][LT_KT][GEMM_BM];
    __shared__ cd sB[2][LT_KT][GEMM_BN];
    const int id = blockIdx.x;
    const int xcd = id & 7, slot = id >> 3;
    int z, row_t, col_t;
    if (upper & 4) {
        // K-split launches: all tiles of one k-chunk on the SAME XCD, so that both the A chunk and the
        // B chunk are fetched from HBM once and shared through that XCD's L2
        const int per = gm * gn;
        z = (slot / per) * 8 + xcd;
        const int rem = slot % per;
        row_t = rem / gn;
        col_t = rem - row_t * gn;
        if (z >= nsplit) return;   // whole workgroup
    } else {
        // column tiles of one (k-chunk, row panel) on the same XCD (shares the A panel)
        col_t = slot % gn;
        const int R = (slot / gn) * 8 + xcd;
        if (R >= gm * nsplit) return;   // whole workgroup
        z = R / gm;
        row_t = R - z * gm;
    }

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
#ifdef GEMM_EXP_CLOCK
    if (blockIdx.x == 0 && tid == 0) {
        g_gemm_clk[0] = clock64();
        g_gemm_clk[2] = wall_clock64();
    }
#endif
    // rectangle of tiles at (rt0, ct0), or (lsplit >= 0) the L-shaped ragged border as a list:
    // entries < lsplit are the right strip (tile column ct0), the rest the bottom strip (tile row rt0)
    const int tr = lsplit < 0 ? row_t + rt0 : (row_t < lsplit ? row_t : rt0);
    const int tcn = lsplit < 0 ? col_t + ct0 : (row_t < lsplit ? ct0 : row_t - lsplit);
    const int I0 = tr * GEMM_BM, J0 = tcn * GEMM_BN;
    if ((upper & 1) && I0 >= J0 + GEMM_BN) return;   // tile strictly below the diagonal (whole workgroup)
    const int i0 = I0 + wave * (GEMM_RM * 16);
    const int kbeg = z * kchunk;
    // bit 1 of `upper`: B is upper triangular (B[k][j] = 0 for k > j) -> this tile column stops at k = J0 + BN
    const int kend = min(min(K, kbeg + kchunk), (upper & 2) ? J0 + GEMM_BN : K);
    // MODE 1: every tile of the launch is full (no predicates anywhere).  MODE 0: predicated only.
    constexpr bool FULL = MODE == 1;
    const int rmv = FULL ? GEMM_RM : min(GEMM_RM, max(0, (m - i0 + 15) >> 4));
    const int rnv = FULL ? GEMM_RN : min(GEMM_RN, max(0, (n - J0 + 15) >> 4));
    const bool active = FULL || (rmv > 0 && rnv > 0);

    v4d accR[GEMM_RM][GEMM_RN], accI[GEMM_RM][GEMM_RN];
#pragma unroll
    for (int a = 0; a < GEMM_RM; ++a)
#pragma unroll
        for (int b = 0; b < GEMM_RN; ++b) {
            accR[a][b] = (v4d){0.0, 0.0, 0.0, 0.0};
            accI[a][b] = (v4d){0.0, 0.0, 0.0, 0.0};
        }

    // ---- global -> register staging assignment
    // K-major operand tile (LT_KT x W columns): thread handles k = tid & 7, columns (tid >> 3) + 32 r
    // M-major A tile (LT_KT x 128 rows):        thread handles row = tid & 127, k = (tid >> 7) + 2 r
    const int tk = tid & 7, tc = tid >> 3;
    // running per-thread source pointers (named scalars: arrays captured by lambdas end up in scratch);
    // every load advances them by one k-tile
    const cd *pA0, *pA1, *pA2, *pA3, *pB0, *pB1;
    {
        auto a_ptr = [&](int r) -> const cd* {
            if (CONJA) {
                int c = I0 + tc + 32 * r;
                if (c > m - 1) c = m - 1;
                return A + (int64_t)c * lda + kbeg + tk;
            } else {
                int i = I0 + (tid & 127);
                if (i > m - 1) i = m - 1;
                return A + i + (int64_t)(kbeg + (tid >> 7) + 2 * r) * lda;
            }
        };
        auto b_ptr = [&](int r) -> const cd* {
            int c = J0 + tc + 32 * r;
            if (c > n - 1) c = n - 1;
            return B + (int64_t)c * ldb + kbeg + tk;
        };
        pA0 = a_ptr(0);
        pA1 = a_ptr(1);
        pA2 = a_ptr(2);
        pA3 = a_ptr(3);
        pB0 = b_ptr(0);
        pB1 = b_ptr(1);
    }
    const int64_t stepA = CONJA ? (int64_t)LT_KT : (int64_t)LT_KT * lda;
    struct Stage {
        cd a0, a1, a2, a3, b0, b1;
    };
    auto advance = [&]() {
        pA0 += stepA;
        pA1 += stepA;
        pA2 += stepA;
        pA3 += stepA;
        pB0 += LT_KT;
        pB1 += LT_KT;
    };
    // tile entirely inside [kbeg, kend): plain loads
    auto load_fast = [&]() -> Stage {
        Stage st;
        st.a0 = *pA0;
        st.a1 = *pA1;
        st.a2 = *pA2;
        st.a3 = *pA3;
        st.b0 = *pB0;
        st.b1 = *pB1;
        advance();
        return st;
    };
    // any tile: k indices beyond kend-1 are clamped to kend-1 (and zeroed later by mask_tile)
    auto load_tile = [&](int k0) -> Stage {
        Stage st;
        const int oB = max(0, k0 + tk - (kend - 1));
        if (CONJA) {
            st.a0 = *(pA0 - oB);
            st.a1 = *(pA1 - oB);
            st.a2 = *(pA2 - oB);
            st.a3 = *(pA3 - oB);
        } else {
            const int kk = k0 + (tid >> 7) - (kend - 1);
            st.a0 = *(pA0 - (int64_t)max(0, kk) * lda);
            st.a1 = *(pA1 - (int64_t)max(0, kk + 2) * lda);
            st.a2 = *(pA2 - (int64_t)max(0, kk + 4) * lda);
            st.a3 = *(pA3 - (int64_t)max(0, kk + 6) * lda);
        }
        st.b0 = *(pB0 - oB);
        st.b1 = *(pB1 - oB);
        advance();
        return st;
    };
    // zero the entries whose k lies beyond the K range (only the last tile of a chunk needs it)
    auto mask_tile = [&](Stage st, int k0) -> Stage {
        const cd czero = make_double2(0.0, 0.0);
        const bool vB = (k0 + tk) < kend;
        if (CONJA) {
            if (!vB) st.a0 = st.a1 = st.a2 = st.a3 = czero;
        } else {
            const int kk = k0 + (tid >> 7);
            if (kk >= kend) st.a0 = czero;
            if (kk + 2 >= kend) st.a1 = czero;
            if (kk + 4 >= kend) st.a2 = czero;
            if (kk + 6 >= kend) st.a3 = czero;
        }
        if (!vB) st.b0 = st.b1 = czero;
        return st;
    };
    auto store_tile = [&](int buf, const Stage& st) {
        if (CONJA) {
            sA[buf][tk][(tc) ^ tk] = st.a0;
            sA[buf][tk][(tc + 32) ^ tk] = st.a1;
            sA[buf][tk][(tc + 64) ^ tk] = st.a2;
            sA[buf][tk][(tc + 96) ^ tk] = st.a3;
        } else {
            const int kk = tid >> 7, ii = tid & 127;
            sA[buf][kk][ii] = st.a0;
            sA[buf][kk + 2][ii] = st.a1;
            sA[buf][kk + 4][ii] = st.a2;
            sA[buf][kk + 6][ii] = st.a3;
        }
        sB[buf][tk][(tc) ^ tk] = st.b0;
        sB[buf][tk][(tc + 32) ^ tk] = st.b1;
    };
    // MFMA fragments of one k-half of a tile: half h holds k = 2*lk + h (lk = lane >> 4)
    struct Frag {
        cd a[GEMM_RM], b[GEMM_RN];
    };
    auto read_frag = [&](int buf, int h) -> Frag {
        Frag f;
        const int kk = 2 * lk + h;
#pragma unroll
        for (int a = 0; a < GEMM_RM; ++a) {
            const int c = wave * (GEMM_RM * 16) + a * 16 + li;
            f.a[a] = sA[buf][kk][CONJA ? (c ^ kk) : c];
        }
#pragma unroll
        for (int b = 0; b < GEMM_RN; ++b) {
            const int c = b * 16 + li;
            f.b[b] = sB[buf][kk][c ^ kk];
        }
        return f;
    };
    auto mfma_half = [&](const Frag& f, auto nopred_tag) {
        constexpr bool NOPRED = decltype(nopred_tag)::value;
#pragma unroll
        for (int a = 0; a < GEMM_RM; ++a) {
            if (NOPRED || a < rmv) {
                const double ar = f.a[a].x;
                const double ai = f.a[a].y;
                const double nai = -ai;
#pragma unroll
                for (int b = 0; b < GEMM_RN; ++b)
                    if (NOPRED || b < rnv) accR[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(ar, f.b[b].x, accR[a][b], 0, 0, 0);
#pragma unroll
                for (int b = 0; b < GEMM_RN; ++b)
                    if (NOPRED || b < rnv) accI[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(ar, f.b[b].y, accI[a][b], 0, 0, 0);
#pragma unroll
                for (int b = 0; b < GEMM_RN; ++b)
                    if (NOPRED || b < rnv)
                        accR[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(CONJA ? ai : nai, f.b[b].y, accR[a][b], 0, 0, 0);
#pragma unroll
                for (int b = 0; b < GEMM_RN; ++b)
                    if (NOPRED || b < rnv)
                        accI[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(CONJA ? nai : ai, f.b[b].x, accI[a][b], 0, 0, 0);
            }
        }
    };

    // Software pipeline (two LDS buffers, ONE barrier per tile, placed in the middle of the tile's
    // MFMA stream so that nothing waits on a fresh LDS/global access):
    //   iteration t:  read half-1 fragments of tile t | write tile t+1 (registers) to the other buffer,
    //                 issue the global loads of tile t+2 | 32 MFMAs on half 0 | barrier |
    //                 read half-0 fragments of tile t+1 | 32 MFMAs on half 1
    const int nt = (kend - kbeg + LT_KT - 1) / LT_KT;
    if (nt > 0) {
        Stage st = load_tile(kbeg);
        if (nt == 1) st = mask_tile(st, kbeg);
        store_tile(0, st);
        if (nt > 1) st = load_tile(kbeg + LT_KT);
        __syncthreads();
        Frag f0 = read_frag(0, 0);
        int t = 0;
        if (FULL) {
            // steady state (tiles t+1, t+2, t+3 exist): one branch-free block, and the
            // scheduler is told to drop one memory instruction into the shadow of each MFMA so that this
            // wave alone keeps the matrix pipe fed (the two workgroups of a CU run in lockstep, so
            // "the other wave covers my memory phase" does not happen by itself).
            for (; t + 3 < nt; ++t) {   // tile t+2 is not the last one: it lies entirely inside the chunk
                Frag f1 = read_frag(t & 1, 1);
                store_tile((t + 1) & 1, st);
                st = load_fast();
                mfma_half(f0, std::true_type{});
#pragma unroll
                for (int i = 0; i < GEMM_RM + GEMM_RN; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // 1 DS read
                }
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
                    __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);   // 1 DS write
                }
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // 1 VMEM read
                    __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);   // pointer advance
                }
                __syncthreads();
                f0 = read_frag((t + 1) & 1, 0);
                mfma_half(f1, std::true_type{});
#pragma unroll
                for (int i = 0; i < GEMM_RM + GEMM_RN; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 1);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 1);
                }
            }
        }
        for (; t < nt; ++t) {
            const bool more = (t + 1) < nt;
            Frag f1 = read_frag(t & 1, 1);
            if (more) {
                if (t + 2 == nt) st = mask_tile(st, kbeg + (t + 1) * LT_KT);
                store_tile((t + 1) & 1, st);
                if (t + 2 < nt) st = load_tile(kbeg + (t + 2) * LT_KT);
            }
            if (active) mfma_half(f0, std::integral_constant<bool, FULL>{});
            __syncthreads();
            if (more) f0 = read_frag((t + 1) & 1, 0);
            if (active) mfma_half(f1, std::integral_constant<bool, FULL>{});
        }
    }
    if (!active) return;

    // epilogue
#ifdef GEMM_EXP_CLOCK
    if (blockIdx.x == 0 && tid == 0) {
        g_gemm_clk[1] = clock64();
        g_gemm_clk[3] = wall_clock64();
    }
#endif
#ifdef GEMM_EXP_NOSTORE
    if (accR[0][0][0] != 1.2345e300) return;   // timing experiment: skip the C write (8 % of a k = 256 product)
#endif
    const int j0 = J0;
    const bool direct = (slab == nullptr);
    cd* sl = direct ? nullptr : slab + (int64_t)z * m * n;
#pragma unroll
    for (int a = 0; a < GEMM_RM; ++a)
#pragma unroll
        for (int b = 0; b < GEMM_RN; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int gi = i0 + a * 16 + lk + 4 * r;
                const int gj = j0 + b * 16 + li;
                if (gi < m && gj < n) {
                    const double vr = accR[a][b][r], vi = accI[a][b][r];
                    if (direct) {
                        cd* c = C + gi + (int64_t)gj * ldc;
                        cd o = make_double2(alpha.x * vr - alpha.y * vi, alpha.x * vi + alpha.y * vr);
                        if (beta.x != 0.0 || beta.y != 0.0) {
                            const cd old = *c;
                            o.x += beta.x * old.x - beta.y * old.y;
                            o.y += beta.x * old.y + beta.y * old.x;
                        }
                        *c = o;
                    } else {
                        sl[gi + (int64_t)gj * m] = make_double2(vr, vi);
                    }
                }
            }
}

// Default interior kernel: the complex product with THREE real MFMAs per tile step instead of four
// (Karatsuba / "3M"), on 128 x 32 workgroup tiles (wave tile 32 x 32: 12 accumulator tiles = 96 VGPRs).
// 15-23 % faster than the 4-product kernel on every LOBPCG / projector shape; the rounding error bound
// grows by a factor ~2 in the imaginary part (still eps * sum |a||b|).  DFTK_MI_GEMM_4M=1 selects the
// 4-product kernel (k_zgemm_lds MODE 1).
#define M3_RN 2
#ifdef M3_NO_HINTS
#define M3_HINT(a, b, c) ((void)0)
#else
#define M3_HINT(a, b, c) __builtin_amdgcn_sched_group_barrier(a, b, c)
#endif
#define M3_BN (16 * M3_RN)
#define M3_RN_REAL 4
#define M3_BN_REAL (16 * M3_RN_REAL)
// workgroups per CU the kernel is compiled for.  3 for the M-major variant (168 VGPRs) is as fast as 2,
// but with 12 instead of 8 row panels in flight per XCD the column tiles of a panel drift apart in k and
// re-fetch their A tiles: FETCH_SIZE 4.0x the operand bytes against 1.55x -> 2.  The K-major variant
// spills at 168 registers (-12 %).
#ifndef M3_N_BLOCKS
#define M3_N_BLOCKS 2
#endif
#define M3_MIN_BLOCKS(CONJA) ((CONJA) ? 2 : M3_N_BLOCKS)
// REAL (flag DFTK_MI_GEMM_REAL of zgemm): the operands are blocks of REAL-SYMMETRIC plane-wave vectors in the
// half-sphere format (gamma_kernels.hip), i.e. really REAL matrices with two real rows per complex entry:
//   conj(A)' B -> Re(A^H B) = Ar' Br + Ai' Bi   (the imaginary part vanishes mathematically; stored as 0)
//   A B        -> A * Re(B): Re = Ar Br, Im = Ai Br   (B is a real coefficient matrix stored as complex)
// TWO real MFMA streams per complex entry instead of three -- exactly the flops of the equivalent dgemm.
template <bool CONJA, int MODE, bool REAL, int RN>
__global__ __launch_bounds__(GEMM_WAVES * 64, M3_MIN_BLOCKS(CONJA)) void k_zgemm_3m(int m, int n, int K, int kchunk, int gm, int gn,
                                                               int rt0, int ct0, int lsplit, int upper,
                                                               int shift_ct, int shift_rt, int nsplit, const cd* __restrict__ A, int64_t lda,
                                                               const cd* __restrict__ B, int64_t ldb,
                                                               cd* __restrict__ C, int64_t ldc, cd alpha, cd beta,
                                                               cd* __restrict__ slab) {
    constexpr int BN = 16 * RN;      // column-tile width: 32 (3M complex), 64 (REAL: the third product's registers
                                     // hold a second pair of accumulator columns -> 1.7x the flops per operand byte)
    static_assert(RN == 2 || RN == 4, "B staging handles one or two columns per thread");
    __shared__ cd sA[2][LT_KT][GEMM_BM];
    __shared__ cd sB[2][LT_KT][BN];
    const int id = blockIdx.x;
    const int xcd = id & 7, slot = id >> 3;
    int z, row_t, col_t;
    if (upper & 16) {
        // compact UPPER launch: only the LIVE tiles exist in the grid (row panel r keeps its column tiles >= r BM / BN), the
        // work items (k-chunk, row panel, live column tile) in this order are dealt to the XCDs in 8 contiguous blocks: an
        // XCD sees at most two k-chunks, the column tiles of a row panel sit next to each other, every XCD gets the same
        // number of items and no dead workgroup passes through the dispatcher (with the full grid below 38-44 % of the
        // workgroups of an UPPER launch leave at once; measured: the live ones then spread unevenly over the CUs as soon
        // as an XCD holds more than ~50 of them)
        constexpr int q = GEMM_BM / BN;
        int L = 0;
        for (int r = 0; r < gm; ++r) L += max(0, gn - r * q);
        const int N = L * nsplit, per = (N + 7) >> 3;
        const int item = xcd * per + slot;
        if (slot >= per || item >= N) return;   // whole workgroup
        z = item / L;
        int t = item - z * L, r = 0;
        while (t >= gn - r * q) {
            t -= gn - r * q;
            ++r;
        }
        row_t = r;
        col_t = r * q + t;
    } else if (upper & 4) {
        // K-split launches: all tiles of one k-chunk on the SAME XCD, so that both the A chunk and the
        // B chunk are fetched from HBM once and shared through that XCD's L2
        const int per = gm * gn;
        z = (slot / per) * 8 + xcd;
        const int rem = slot % per;
        row_t = rem / gn;
        col_t = rem - row_t * gn;
        if (z >= nsplit) return;   // whole workgroup
    } else {
        // column tiles of one (k-chunk, row panel) on the same XCD (shares the A panel)
        col_t = slot % gn;
        const int R = (slot / gn) * 8 + xcd;
        if (R >= gm * nsplit) return;   // whole workgroup
        z = R / gm;
        row_t = R - z * gm;
        // upper & 8: rotate the row panels by floor(z gcd(8, gm) / 8).  An XCD sees the groups R = xcd + 8 t, i.e. only
        // gm / gcd(8, gm) different row panels; with UPPER the panels hold different numbers of live tiles (8, 6, 4, 2
        // for a 503^2 Gram matrix), so without the rotation XCDs 0 and 4 do four times the work of XCDs 3 and 7
        if (upper & 8) {
            const int g = (gm % 8 == 0) ? 8 : (gm % 4 == 0) ? 4 : (gm % 2 == 0) ? 2 : 1;
            row_t = (row_t + (z * g) / 8) % gm;
        }
    }

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
#ifdef GEMM_EXP_CLOCK
    if (blockIdx.x == 0 && tid == 0) {
        g_gemm_clk[0] = clock64();
        g_gemm_clk[2] = wall_clock64();
    }
#endif
    // rectangle of tiles at (rt0, ct0), or (lsplit >= 0) the L-shaped ragged border as a list:
    // entries < lsplit are the right strip (tile column ct0), the rest the bottom strip (tile row rt0)
    const int tr = lsplit < 0 ? row_t + rt0 : (row_t < lsplit ? row_t : rt0);
    const int tcn = lsplit < 0 ? col_t + ct0 : (row_t < lsplit ? ct0 : row_t - lsplit);
    // tile column shift_ct (the one past the last full column) is SHIFTED LEFT to end at column n: it overlaps
    // its neighbour, recomputes up to 31 columns and stores only columns >= jmin.  A ragged n then needs no
    // right-strip launch (which would stream all of A a second time for a handful of columns).
    const bool shifted = tcn == shift_ct;
    // likewise tile row shift_rt (the one past the last full row) is SHIFTED UP to end at row m and stores only
    // rows >= imin: a ragged m needs no bottom-strip launch (a second, serialized launch of slow predicated
    // workgroups whose k chains are as long as the interior's)
    const bool rshifted = tr == shift_rt;
    const int I0 = rshifted ? m - GEMM_BM : tr * GEMM_BM, J0 = shifted ? n - BN : tcn * BN;
    const int jmin = shifted ? tcn * BN : 0, imin = tr * GEMM_BM;
    if ((upper & 1) && imin >= J0 + BN) return;   // no needed entry (i <= j) in this tile (whole workgroup)
    const int i0 = I0 + wave * (GEMM_RM * 16);
    const int kbeg = z * kchunk;
    // bit 1 of `upper`: B is upper triangular (B[k][j] = 0 for k > j) -> this tile column stops at k = J0 + BN
    const int kend = min(min(K, kbeg + kchunk), (upper & 2) ? J0 + BN : K);
    // MODE 1: every tile of the launch is full (no predicates anywhere).  MODE 0: predicated only.
    constexpr bool FULL = MODE == 1;
    const int rmv = FULL ? GEMM_RM : min(GEMM_RM, max(0, (m - i0 + 15) >> 4));
    const int rnv = FULL ? RN : min(RN, max(0, (n - J0 + 15) >> 4));
    const bool active = FULL || (rmv > 0 && rnv > 0);

    // three real products per complex one (Karatsuba / "3M"):
    //   P1 = sum Ar Br, P2 = sum Ai Bi, P3 = sum (Ar +/- Ai)(Br + Bi)   (- for conj(A))
    //   A B      : Re = P1 - P2, Im = P3 - P1 - P2 ;   conj(A) B: Re = P1 + P2, Im = P3 - P1 + P2
    v4d acc1[GEMM_RM][RN], acc2[GEMM_RM][RN], acc3[GEMM_RM][RN];
#pragma unroll
    for (int a = 0; a < GEMM_RM; ++a)
#pragma unroll
        for (int b = 0; b < RN; ++b) {
            acc1[a][b] = (v4d){0.0, 0.0, 0.0, 0.0};
            acc2[a][b] = (v4d){0.0, 0.0, 0.0, 0.0};
            acc3[a][b] = (v4d){0.0, 0.0, 0.0, 0.0};
        }

    // ---- global -> register staging assignment
    // K-major operand tile (LT_KT x W columns): thread handles k = tid & 7, columns (tid >> 3) + 32 r
    // M-major A tile (LT_KT x 128 rows):        thread handles row = tid & 127, k = (tid >> 7) + 2 r
    const int tk = tid & 7, tc = tid >> 3;
    // running per-thread source pointers (named scalars: arrays captured by lambdas end up in scratch);
    // every load advances them by one k-tile
    const cd *pA0, *pA1, *pA2, *pA3, *pB0, *pB1;
    {
        auto a_ptr = [&](int r) -> const cd* {
            if (CONJA) {
                int c = I0 + tc + 32 * r;
                if (c > m - 1) c = m - 1;
                return A + (int64_t)c * lda + kbeg + tk;
            } else {
                int i = I0 + (tid & 127);
                if (i > m - 1) i = m - 1;
                return A + i + (int64_t)(kbeg + (tid >> 7) + 2 * r) * lda;
            }
        };
        auto b_ptr = [&](int r) -> const cd* {
            int c = J0 + tc + 32 * r;
            if (c > n - 1) c = n - 1;
            return B + (int64_t)c * ldb + kbeg + tk;
        };
        pA0 = a_ptr(0);
        pA1 = a_ptr(1);
        pA2 = a_ptr(2);
        pA3 = a_ptr(3);
        pB0 = b_ptr(0);
        pB1 = b_ptr(RN == 4 ? 1 : 0);
    }
    const int64_t stepA = CONJA ? (int64_t)LT_KT : (int64_t)LT_KT * lda;
    struct Stage {
        cd a0, a1, a2, a3, b0, b1;
    };
    auto advance = [&]() {
        pA0 += stepA;
        pA1 += stepA;
        pA2 += stepA;
        pA3 += stepA;
        pB0 += LT_KT;
        if (RN == 4) pB1 += LT_KT;
    };
    // tile entirely inside [kbeg, kend): plain loads
    auto load_fast = [&]() -> Stage {
        Stage st;
        st.a0 = *pA0;
        st.a1 = *pA1;
        st.a2 = *pA2;
        st.a3 = *pA3;
        st.b0 = *pB0;
        if (RN == 4) st.b1 = *pB1;
        advance();
        return st;
    };
    // any tile: k indices beyond kend-1 are clamped to kend-1 (and zeroed later by mask_tile)
    auto load_tile = [&](int k0) -> Stage {
        Stage st;
        const int oB = max(0, k0 + tk - (kend - 1));
        if (CONJA) {
            st.a0 = *(pA0 - oB);
            st.a1 = *(pA1 - oB);
            st.a2 = *(pA2 - oB);
            st.a3 = *(pA3 - oB);
        } else {
            const int kk = k0 + (tid >> 7) - (kend - 1);
            st.a0 = *(pA0 - (int64_t)max(0, kk) * lda);
            st.a1 = *(pA1 - (int64_t)max(0, kk + 2) * lda);
            st.a2 = *(pA2 - (int64_t)max(0, kk + 4) * lda);
            st.a3 = *(pA3 - (int64_t)max(0, kk + 6) * lda);
        }
        st.b0 = *(pB0 - oB);
        if (RN == 4) st.b1 = *(pB1 - oB);
        advance();
        return st;
    };
    // zero the entries whose k lies beyond the K range (only the last tile of a chunk needs it)
    auto mask_tile = [&](Stage st, int k0) -> Stage {
        const cd czero = make_double2(0.0, 0.0);
        const bool vB = (k0 + tk) < kend;
        if (CONJA) {
            if (!vB) st.a0 = st.a1 = st.a2 = st.a3 = czero;
        } else {
            const int kk = k0 + (tid >> 7);
            if (kk >= kend) st.a0 = czero;
            if (kk + 2 >= kend) st.a1 = czero;
            if (kk + 4 >= kend) st.a2 = czero;
            if (kk + 6 >= kend) st.a3 = czero;
        }
        if (!vB) st.b0 = st.b1 = czero;
        return st;
    };
    auto store_tile = [&](int buf, const Stage& st) {
        if (CONJA) {
            sA[buf][tk][(tc) ^ tk] = st.a0;
            sA[buf][tk][(tc + 32) ^ tk] = st.a1;
            sA[buf][tk][(tc + 64) ^ tk] = st.a2;
            sA[buf][tk][(tc + 96) ^ tk] = st.a3;
        } else {
            const int kk = tid >> 7, ii = tid & 127;
            sA[buf][kk][ii] = st.a0;
            sA[buf][kk + 2][ii] = st.a1;
            sA[buf][kk + 4][ii] = st.a2;
            sA[buf][kk + 6][ii] = st.a3;
        }
        sB[buf][tk][(tc) ^ tk] = st.b0;
        if (RN == 4) sB[buf][tk][(tc + 32) ^ tk] = st.b1;
    };
    // MFMA fragments of one k-half of a tile: half h holds k = 2*lk + h (lk = lane >> 4)
    struct Frag {
        cd a[GEMM_RM], b[RN];
    };
    auto read_frag = [&](int buf, int h) -> Frag {
        Frag f;
        const int kk = 2 * lk + h;
#pragma unroll
        for (int a = 0; a < GEMM_RM; ++a) {
            const int c = wave * (GEMM_RM * 16) + a * 16 + li;
            f.a[a] = sA[buf][kk][CONJA ? (c ^ kk) : c];
        }
#pragma unroll
        for (int b = 0; b < RN; ++b) {
            const int c = b * 16 + li;
            f.b[b] = sB[buf][kk][c ^ kk];
        }
        return f;
    };
    auto mfma_half = [&](const Frag& f, auto nopred_tag) {
        constexpr bool NOPRED = decltype(nopred_tag)::value;
        double bs[RN];
#pragma unroll
        for (int b = 0; b < RN; ++b) bs[b] = f.b[b].x + f.b[b].y;
#pragma unroll
        for (int a = 0; a < GEMM_RM; ++a) {
            if (NOPRED || a < rmv) {
                const double ar = f.a[a].x, ai = f.a[a].y;
                const double as = CONJA ? ar - ai : ar + ai;
#pragma unroll
                for (int b = 0; b < RN; ++b)
                    if (NOPRED || b < rnv) acc1[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(ar, f.b[b].x, acc1[a][b], 0, 0, 0);
#pragma unroll
                for (int b = 0; b < RN; ++b)
                    if (NOPRED || b < rnv) {
                        if (REAL && CONJA)   // Re(conj(a) b) = ar br + ai bi: ONE accumulator
                            acc1[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(ai, f.b[b].y, acc1[a][b], 0, 0, 0);
                        else
                            acc2[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(ai, REAL ? f.b[b].x : f.b[b].y, acc2[a][b],
                                                                              0, 0, 0);
                    }
                if (!REAL) {
#pragma unroll
                    for (int b = 0; b < RN; ++b)
                        if (NOPRED || b < rnv) acc3[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(as, bs[b], acc3[a][b], 0, 0, 0);
                }
            }
        }
    };

    // Software pipeline (two LDS buffers, ONE barrier per tile, placed in the middle of the tile's
    // MFMA stream so that nothing waits on a fresh LDS/global access):
    //   iteration t:  read half-1 fragments of tile t | write tile t+1 (registers) to the other buffer,
    //                 issue the global loads of tile t+2 | 32 MFMAs on half 0 | barrier |
    //                 read half-0 fragments of tile t+1 | 32 MFMAs on half 1
    const int nt = (kend - kbeg + LT_KT - 1) / LT_KT;
    if (nt > 0) {
        Stage st = load_tile(kbeg);
        if (nt == 1) st = mask_tile(st, kbeg);
        store_tile(0, st);
        if (nt > 1) st = load_tile(kbeg + LT_KT);
        __syncthreads();
        Frag f0 = read_frag(0, 0);
        int t = 0;
        if (FULL) {
            // steady state (tiles t+1, t+2, t+3 exist): one branch-free block, and the
            // scheduler is told to drop one memory instruction into the shadow of each MFMA so that this
            // wave alone keeps the matrix pipe fed (the two workgroups of a CU run in lockstep, so
            // "the other wave covers my memory phase" does not happen by itself).
            for (; t + 3 < nt; ++t) {   // tile t+2 is not the last one: it lies entirely inside the chunk
                Frag f1 = read_frag(t & 1, 1);
                store_tile((t + 1) & 1, st);
                st = load_fast();
                mfma_half(f0, std::true_type{});
#pragma unroll
                for (int i = 0; i < GEMM_RM + RN; ++i) {
                    M3_HINT(0x008, 1, 0);   // 1 MFMA
                    M3_HINT(0x100, 1, 0);   // 1 DS read
                }
#pragma unroll
                for (int i = 0; i < 4 + RN / 2; ++i) {
                    M3_HINT(0x008, 1, 0);   // 1 MFMA
                    M3_HINT(0x200, 1, 0);   // 1 DS write
                    M3_HINT(0x020, 1, 0);   // 1 VMEM read
                    M3_HINT(0x002, 2, 0);   // pointer advance
                }
                __syncthreads();
                f0 = read_frag((t + 1) & 1, 0);
                mfma_half(f1, std::true_type{});
#pragma unroll
                for (int i = 0; i < GEMM_RM + RN; ++i) {
                    M3_HINT(0x008, 2, 1);
                    M3_HINT(0x100, 1, 1);
                }
            }
        }
        for (; t < nt; ++t) {
            const bool more = (t + 1) < nt;
            Frag f1 = read_frag(t & 1, 1);
            if (more) {
                if (t + 2 == nt) st = mask_tile(st, kbeg + (t + 1) * LT_KT);
                store_tile((t + 1) & 1, st);
                if (t + 2 < nt) st = load_tile(kbeg + (t + 2) * LT_KT);
            }
            if (active) mfma_half(f0, std::integral_constant<bool, FULL>{});
            __syncthreads();
            if (more) f0 = read_frag((t + 1) & 1, 0);
            if (active) mfma_half(f1, std::integral_constant<bool, FULL>{});
        }
    }
    if (!active) return;

    // epilogue
#ifdef GEMM_EXP_CLOCK
    if (blockIdx.x == 0 && tid == 0) {
        g_gemm_clk[1] = clock64();
        g_gemm_clk[3] = wall_clock64();
    }
#endif
#ifdef GEMM_EXP_NOSTORE
    if (acc1[0][0][0] != 1.2345e300) return;   // timing experiment: skip the C write (8 % of a k = 256 product)
#endif
    const int j0 = J0;
    const bool direct = (slab == nullptr);
    cd* sl = direct ? nullptr : slab + (int64_t)z * m * n;
#pragma unroll
    for (int a = 0; a < GEMM_RM; ++a)
#pragma unroll
        for (int b = 0; b < RN; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int gi = i0 + a * 16 + lk + 4 * r;
                const int gj = j0 + b * 16 + li;
                if (gi < m && gj < n && gj >= jmin && gi >= imin) {
                    const double p1 = acc1[a][b][r], p2 = acc2[a][b][r], p3 = acc3[a][b][r];
                    const double vr = REAL ? p1 : (CONJA ? p1 + p2 : p1 - p2);
                    const double vi = REAL ? (CONJA ? 0.0 : p2) : (CONJA ? p3 - p1 + p2 : p3 - p1 - p2);
                    if (direct) {
                        cd* c = C + gi + (int64_t)gj * ldc;
                        cd o = make_double2(alpha.x * vr - alpha.y * vi, alpha.x * vi + alpha.y * vr);
                        if (beta.x != 0.0 || beta.y != 0.0) {
                            const cd old = *c;
                            o.x += beta.x * old.x - beta.y * old.y;
                            o.y += beta.x * old.y + beta.y * old.x;
                        }
                        *c = o;
                    } else {
                        sl[gi + (int64_t)gj * m] = make_double2(vr, vi);
                    }
                }
            }
}

// C = alpha * sum_z slab[z] + beta * C   (fixed summation order).  The interior (i < mi, j < nj) and
// the ragged border have their own split counts / slabs; a count < 0 means that region was written
// directly by the GEMM kernel and is skipped here.
__global__ void k_zgemm_reduce(int m, int n, int mi, int nj, int nsI, const cd* __restrict__ slabI, int nsB,
                               const cd* __restrict__ slabB, cd* __restrict__ C, int64_t ldc, cd alpha, cd beta,
                               int upper) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)m * n) return;
    const int j = (int)(idx / m);
    const int i = (int)(idx - (int64_t)j * m);
    if ((upper & 1) && (i / GEMM_BM) * GEMM_BM >= (j / GEMM_BN) * GEMM_BN + GEMM_BN) return;   // tile not computed
    const bool interior = i < mi && j < nj;
    const int nsplit = interior ? nsI : nsB;
    const cd* slab = interior ? slabI : slabB;
    if (nsplit < 0) return;
    double sr = 0.0, si = 0.0;
    for (int z = 0; z < nsplit; ++z) {
        const cd v = slab[(int64_t)z * m * n + idx];
        sr += v.x;
        si += v.y;
    }
    cd* c = C + i + (int64_t)j * ldc;
    cd o = make_double2(alpha.x * sr - alpha.y * si, alpha.x * si + alpha.y * sr);
    if (beta.x != 0.0 || beta.y != 0.0) {
        const cd old = *c;
        o.x += beta.x * old.x - beta.y * old.y;
        o.y += beta.x * old.y + beta.y * old.x;
    }
    *c = o;
}

// Reference kernel without matrix cores (debug path, env DFTK_MI_GEMM=naive): one thread per C entry.
template <bool CONJA>
__global__ void k_zgemm_naive(int m, int n, int K, const cd* __restrict__ A, int64_t lda,
                              const cd* __restrict__ B, int64_t ldb, cd* __restrict__ C, int64_t ldc, cd alpha,
                              cd beta, int real) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)m * n) return;
    const int j = (int)(idx / m);
    const int i = (int)(idx - (int64_t)j * m);
    double sr = 0.0, si = 0.0;
    for (int k = 0; k < K; ++k) {
        cd a = CONJA ? A[k + (int64_t)i * lda] : A[i + (int64_t)k * lda];
        if (CONJA) a.y = -a.y;
        cd b = B[k + (int64_t)j * ldb];
        if (real && !CONJA) b.y = 0.0;
        sr += a.x * b.x - a.y * b.y;
        si += a.x * b.y + a.y * b.x;
    }
    if (real && CONJA) si = 0.0;
    cd* c = C + i + (int64_t)j * ldc;
    cd o = make_double2(alpha.x * sr - alpha.y * si, alpha.x * si + alpha.y * sr);
    if (beta.x != 0.0 || beta.y != 0.0) {
        const cd old = *c;
        o.x += beta.x * old.x - beta.y * old.y;
        o.y += beta.x * old.y + beta.y * old.x;
    }
    *c = o;
}

// ---- diagnostic: pure v_mfma_f64_16x16x4_f64 issue rate (no memory traffic) -> measured ceiling
__global__ __launch_bounds__(256) void k_mfma_peak(int iters, double* out) {
    v4d acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = (v4d){0.0, 0.0, 0.0, 0.0};
    double a = 1.0 + threadIdx.x * 1e-9, bb = 1.0 - threadIdx.x * 1e-9;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bb, acc[i], 0, 0, 0);
    }
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678) out[0] = s;   // keep the chain alive
}

// cycles per MFMA on one wave: out[1] = shader clocks, out[2] = 100 MHz wall ticks for `iters` x 8 MFMAs
template <int NACC>
__global__ __launch_bounds__(256) void k_mfma_cycles(int iters, double* out) {
    v4d acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = (v4d){0.0, 0.0, 0.0, 0.0};
    double a = 1.0 + threadIdx.x * 1e-9, bb = 1.0 - threadIdx.x * 1e-9;
    const long long c0 = clock64();
    const long long w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
            acc[i % NACC] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bb, acc[i % NACC], 0, 0, 0);
    }
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    const long long c1 = clock64();
    const long long w1 = wall_clock64();
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        out[1] = (double)(c1 - c0);
        out[2] = (double)(w1 - w0);
        out[3] = s;
    }
}

extern "C" int dftk_mi_diag_mfma_peak(dftk_mi_basis* b, int waves_per_simd, int iters, double* tflops) {
    if (!b || !tflops || waves_per_simd < 1 || iters < 1) return DFTK_MI_EINVAL;
    HIPCHK(hipSetDevice(b->device));
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0));
    HIPCHK(hipEventCreate(&e1));
    const int blocks = 256 * waves_per_simd;       // 256 CUs x (4 waves per block = 1 per SIMD)
    hipLaunchKernelGGL(k_mfma_peak, dim3(blocks), dim3(256), 0, b->stream, 10, b->d_scalars);
    HIPCHK(hipEventRecord(e0, b->stream));
    hipLaunchKernelGGL(k_mfma_peak, dim3(blocks), dim3(256), 0, b->stream, iters, b->d_scalars);
    HIPCHK(hipEventRecord(e1, b->stream));
    HIPCHK(hipEventSynchronize(e1));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    const double flops = (double)blocks * 4.0 * iters * 8.0 * 2048.0;
    *tflops = flops / (ms * 1e-3) / 1e12;
    for (int nacc : {8, 2, 1}) {
        if (nacc == 8) hipLaunchKernelGGL(k_mfma_cycles<8>, dim3(blocks), dim3(256), 0, b->stream, iters, b->d_scalars);
        if (nacc == 2) hipLaunchKernelGGL(k_mfma_cycles<2>, dim3(blocks), dim3(256), 0, b->stream, iters, b->d_scalars);
        if (nacc == 1) hipLaunchKernelGGL(k_mfma_cycles<1>, dim3(blocks), dim3(256), 0, b->stream, iters, b->d_scalars);
        HIPCHK(hipMemcpyAsync(b->h_scalars, b->d_scalars, 4 * sizeof(double), hipMemcpyDeviceToHost, b->stream));
        HIPCHK(hipStreamSynchronize(b->stream));
        fprintf(stderr, "[diag] %d wave(s)/SIMD, %d independent accumulators: %.1f shader clocks per MFMA, clock %.0f MHz\n",
                waves_per_simd, nacc, b->h_scalars[1] / (8.0 * iters), b->h_scalars[1] / (b->h_scalars[2] / 100.0));
    }
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    return 0;
}

int ensure_ws(dftk_mi_basis* b, size_t bytes) {
    if (bytes <= b->ws_bytes) return 0;
    HIPCHK(hipStreamSynchronize(b->stream));
    if (b->ws) HIPCHK(hipFree(b->ws));
    b->ws = nullptr;
    b->ws_bytes = 0;
    size_t want = bytes + bytes / 4;
    HIPCHK(dftk_scratch_malloc(&b->ws, want));
    b->ws_bytes = want;
    return 0;
}

// ---------------------------------------------------------------------------------------- planning
// K split: chosen per launch by a small cost model of the actual workgroup -> XCD placement (8 XCDs x
// 64 resident workgroups, 2 per CU).  Candidates: ns chunks with either mapping of the kernel
//   row-major: (chunk, tile row) pairs round-robin over the XCDs, their column tiles together;
//   z-major  : whole chunks round-robin over the XCDs (A and B chunk both shared through one L2).
// cost = (rounds of the fullest XCD) x (chunk length + prologue/epilogue) + slab traffic.
// Short K is latency-bound: chunks of >= 8 tiles.  Plans are cached per shape.  Host-only code
// (dftk_mi_zgemm_plan_host exposes it to the CPU test-suite).
struct Split {
    int nsplit, kchunk;
    bool zmajor;
    cd* slab;
    bool compact = false;   // UPPER interior launch over the live tiles only (k_zgemm_3m, `upper & 16`)
    int live = 0;           // live tiles of the launch (compact: grid = 8 ceil(live nsplit / 8))
};
// UPPER launches of the 3M / REAL kernels that cannot take the compact live-tile grid rotate the row panels over the
// XCDs (k_zgemm_3m, `upper & 8`)
static bool gemm_rotate_rows() { return true; }
static int64_t gemm_slots2() { return 512; }   // resident workgroups of the 2-per-CU kernels (256 CUs)
static Split gemm_plan_split(int64_t m, int64_t n, int64_t k, int upper, const std::vector<int>& live_rows, int kind,
                             int64_t slots, bool compact_ok = false) {
    const int64_t plane = (int64_t)m * n * (int64_t)sizeof(cd);
    static std::map<std::vector<int64_t>, std::pair<int, int>> plan_cache;   // key -> (nsplit, zmajor)
    static std::mutex plan_mutex;   // host-only cache shared by every basis / thread of the process
    std::lock_guard<std::mutex> plan_lock(plan_mutex);
    // (measured, REAL Gram products of the 1000-electron cell: 503^2 UPPER 2.29 -> 1.90 ms with the rotation; at 1006^2 /
    //  1509^2 -- 8 / 12 row panels -- the k-major mapping with whole chunks per XCD stays as fast or faster, so the rotated
    //  mapping only competes where an XCD would otherwise see at most 4 different row panels)
    const bool rot_rows = (upper & 1) && kind >= 3 && gemm_rotate_rows() && (int)live_rows.size() <= 4;
    int64_t total = 0;
    for (int v : live_rows) total += v;
    const int gm_s = (int)live_rows.size();
    // mapping of the launch: 0 row-major, 1 z-major, 2 compact (live tiles only, UPPER interior launches of the 3M family)
    int best_ns = 1, best_zm = compact_ok ? 2 : 0;
    if (k >= 128 && total > 0 && total < slots) {
        const std::vector<int64_t> key = {m, n, k, (int64_t)(upper & 1), (int64_t)kind, slots, (int64_t)rot_rows,
                                          (int64_t)compact_ok};
        auto it = plan_cache.find(key);
        if (it != plan_cache.end()) {
            best_ns = it->second.first;
            best_zm = it->second.second;
        } else {
            int64_t max_ns = k >= 2048 ? k / 256 : k / 64;
            if (max_ns > 1024) max_ns = 1024;
            const int64_t max_by_ws = (int64_t)(1024ull << 20) / plane;   // all slabs of a launch <= 1 GiB
            if (max_ns > max_by_ws) max_ns = max_by_ws;
            if (max_ns > 2 * slots / total + 8) max_ns = 2 * slots / total + 8;
            if (max_ns < 1) max_ns = 1;
            const double per_xcd = (double)slots / 8.0;
            const double slab_cost = 2.0 * (double)plane / 5e12 / 3.7e-6;   // k-tiles of time per extra chunk
            double best = 1e300;
            for (int ns = 1; ns <= max_ns; ++ns) {
                int kc = (int)((k + ns - 1) / ns);
                kc = (kc + 7) & ~7;
                if ((int)((k + kc - 1) / kc) != ns) continue;
                for (int zm = 0; zm < 3; ++zm) {
                    if (zm == 1 && (ns < 8 || k < 2048)) continue;
                    if (zm == 2 && !compact_ok) continue;
                    int64_t load[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                    if (zm == 2) {
                        load[0] = (total * ns + 7) / 8;   // equal contiguous blocks of work items
                    } else if (zm) {
                        for (int z = 0; z < ns; ++z) load[z & 7] += total;
                    } else {
                        const int64_t R = (int64_t)gm_s * ns;
                        const int g8 = (gm_s % 8 == 0) ? 8 : (gm_s % 4 == 0) ? 4 : (gm_s % 2 == 0) ? 2 : 1;
                        for (int64_t r = 0; r < R; ++r) {
                            const int64_t z = r / gm_s;
                            const int64_t row = rot_rows ? (r % gm_s + (z * g8) / 8) % gm_s : r % gm_s;
                            load[r & 7] += live_rows[row];
                        }
                    }
                    int64_t mx = 0;
                    for (int x = 0; x < 8; ++x) mx = load[x] > mx ? load[x] : mx;
                    const double rounds = std::ceil((double)mx / per_xcd);
                    const double cost = rounds * (kc / 8.0 + 12.0) + (ns > 1 ? slab_cost * ns : 0.0) - 1e-3 * zm;
                    if (cost < best) {
                        best = cost;
                        best_ns = ns;
                        best_zm = zm;
                    }
                }
            }
            if (plan_cache.size() > 4096) plan_cache.clear();
            plan_cache[key] = {best_ns, best_zm};
        }
    }
    int kc = (int)((k + best_ns - 1) / best_ns);
    kc = (kc + 7) & ~7;
    const int ns = (int)((k + kc - 1) / kc);
    Split sp{ns, kc, best_zm == 1 && ns >= 8, nullptr};
    sp.compact = best_zm == 2;
    sp.live = (int)total;
    return sp;
}

// Tiling of one product for the default (3M) kernel family: interior rectangle of full 128 x BN tiles, ragged
// border as a list, and the K split of each.
struct GemmTiling {
    int BNt, gm, gmf, gnf, gnt, nright, nbottom;
    int shift;   // 1: the ragged last tile column is a full tile shifted left (3M kernel), no right strip
    int gnI;     // tile columns of the interior launch = gnf + shift
    int shift_r; // 1: the ragged last tile row is a full tile shifted up, no bottom strip
    int gmI;     // tile rows of the interior launch = gmf + shift_r
    Split I, B;
};
// column tiles of the REAL kernels: 16 * RN wide (RN = 4: 128 x 64 workgroup tiles for both operand layouts)
static int real_rn(bool) { return M3_RN_REAL == 2 ? 2 : 4; }
static GemmTiling gemm_tiling(bool conja, int64_t m, int64_t n, int64_t k, int upper, bool use3m, bool real = false) {
    GemmTiling t;
    const int64_t slots2 = gemm_slots2();
    t.BNt = real ? 16 * real_rn(conja) : use3m ? M3_BN : GEMM_BN;   // column-tile width of this kernel family
    t.gm = (int)((m + GEMM_BM - 1) / GEMM_BM);
    t.gnt = (int)((n + t.BNt - 1) / t.BNt);
    t.gmf = (int)(m / GEMM_BM);
    t.gnf = (int)(n / t.BNt);
    t.shift = (use3m && t.gnt > t.gnf && t.gnf >= 1) ? 1 : 0;
    t.gnI = t.gnf + t.shift;
    t.shift_r = (use3m && t.gm > t.gmf && t.gmf >= 1) ? 1 : 0;
    t.gmI = t.gmf + t.shift_r;
    t.nright = (t.gnt > t.gnf && !t.shift) ? t.gm : 0;
    t.nbottom = (t.gm > t.gmf && !t.shift_r) ? t.gnI : 0;
    // live column tiles per tile row of each launch (upper: only tiles that intersect the upper triangle)
    auto live = [&](int tr, int tc) {
        const int64_t jend = (t.shift && tc == t.gnf) ? n : (int64_t)tc * t.BNt + t.BNt;
        return !(upper & 1) || (int64_t)tr * GEMM_BM < jend;
    };
    std::vector<int> rowsI(t.gmI, 0), rowsB(t.nright + t.nbottom, 0);
    for (int tr = 0; tr < t.gmI; ++tr)
        for (int tc = 0; tc < t.gnI; ++tc) rowsI[tr] += live(tr, tc) ? 1 : 0;
    for (int e = 0; e < t.nright; ++e) rowsB[e] = live(e, t.gnf) ? 1 : 0;
    for (int e = 0; e < t.nbottom; ++e) rowsB[t.nright + e] = live(t.gmf, e) ? 1 : 0;
    // the 3M kernel is compiled for M3_MIN_BLOCKS workgroups per CU
    const int64_t slots3 = use3m ? (slots2 / 2) * M3_MIN_BLOCKS(conja) : slots2;
    // compact UPPER launches (k_zgemm_3m, `upper & 16`): the kernel rebuilds the live tile list as "row panel r keeps its
    // column tiles >= r BM / BN" -- only used when that is exactly the list above
    bool compact_ok = use3m && (upper & 1) && GEMM_BM % t.BNt == 0;
    for (int tr = 0; tr < t.gmI && compact_ok; ++tr)
        compact_ok = rowsI[tr] == std::max(0, t.gnI - tr * (GEMM_BM / t.BNt));
    t.I = gemm_plan_split(m, n, k, upper, rowsI, use3m ? 3 : 1, slots3, compact_ok);
    t.B = gemm_plan_split(m, n, k, upper, rowsB, use3m ? 4 : 2, slots3);
    return t;
}
int zgemm_plan_host(char transA, int64_t m, int64_t n, int64_t k, int flags, int* out) {
    if (m <= 0 || n <= 0 || k <= 0 || (flags & ~(3 | DFTK_MI_GEMM_REAL)) || !out) return DFTK_MI_EINVAL;
    const bool conja = (transA == 'C' || transA == 'c');
    const GemmTiling t = gemm_tiling(conja, m, n, k, flags & 3,
                                     getenv("DFTK_MI_GEMM_4M") == nullptr || (flags & DFTK_MI_GEMM_REAL),
                                     (flags & DFTK_MI_GEMM_REAL) != 0);
    const int v[12] = {t.BNt, t.gmf, t.gnf, t.nright, t.nbottom, t.I.nsplit, t.I.kchunk, t.I.compact ? 2 : t.I.zmajor ? 1 : 0,
                       t.B.nsplit, t.B.kchunk, t.B.zmajor ? 1 : 0, t.shift};
    for (int i = 0; i < 12; ++i) out[i] = v[i];
    return 0;
}

int zgemm(dftk_mi_basis* b, char transA, int64_t m, int64_t n, int64_t k, cd alpha, const cd* A, int64_t lda,
          const cd* B, int64_t ldb, cd beta, cd* C, int64_t ldc, int upper_in) {
    const int upper = upper_in & 3;
    const bool real = (upper_in & DFTK_MI_GEMM_REAL) != 0;   // operands are real-symmetric half-sphere blocks
    if (m <= 0 || n <= 0) return 0;
    const bool conja = (transA == 'C' || transA == 'c');
    if (!conja && !(transA == 'N' || transA == 'n')) {
        dftk_set_error("zgemm: transA must be 'N' or 'C'");
        return DFTK_MI_EINVAL;
    }
    if (m > INT32_MAX || n > INT32_MAX || k > INT32_MAX) return DFTK_MI_EINVAL;
    if (batching()) {   // a fiber of a batched multi-k call: recorded, merged with its siblings' products later
        BOp o;
        o.b = b;
        o.type = BOP_ZGEMM; o.trans = conja ? 'C' : 'N'; o.gm = m; o.gn = n; o.gk = k; o.alpha = alpha; o.A = A; o.lda = lda;
        o.B = B; o.ldb = ldb; o.beta = beta; o.C = C; o.ldc = ldc; o.flags = upper_in;
        return batch_record(std::move(o));
    }
    if (k <= 0) {   // C = beta * C : run the reduce kernel over zero slabs
        hipLaunchKernelGGL(k_zgemm_reduce, dim3((unsigned)((m * n + 255) / 256)), dim3(256), 0, b->stream, (int)m,
                           (int)n, (int)m, (int)n, 0, (const cd*)nullptr, 0, (const cd*)nullptr, C, ldc, alpha, beta, 0);
        HIPCHK(hipGetLastError());
        return 0;
    }
    static const bool trace = getenv("DFTK_MI_TRACE_GEMM") != nullptr;
    if (trace) fprintf(stderr, "[zgemm] %c %lld %lld %lld\n", transA, (long long)m, (long long)n, (long long)k);
    static const bool shapes_env = getenv("DFTK_MI_GEMM_SHAPES") != nullptr;
    const bool shapes = shapes_env || (b->prof && b->prof->shape_tags);
    const uint64_t tag = !shapes ? 0
                                 // (bit 63 transA, 62 "tagged", 61 REAL, 42..60 m, 40..41 UPPER / B_UPPER, 22..39 n, 0..21 k)
                                 : ((uint64_t)conja << 63) | ((uint64_t)(m & 0x7FFFF) << 42) | ((uint64_t)(real ? 1 : 0) << 61) |
                                       ((uint64_t)(n & 0x3FFFF) << 22) | ((uint64_t)(upper_in & 3) << 40) |
                                       (uint64_t)(k & 0x3FFFFF) | (1ull << 62);
    static const bool use3m_env = getenv("DFTK_MI_GEMM_4M") == nullptr;   // DFTK_MI_GEMM_4M=1: classic 4-product kernels
    const bool use3m = use3m_env || real;                                  // (the REAL product only exists in the 3M family)
    const GemmTiling til = gemm_tiling(conja, m, n, k, upper, use3m, real);
    // Flop booking (bench.py roofline).  `useful` = the part of the product that is mathematically needed:
    // 8mnk for an unstructured call, only the (i <= j) entries of C for UPPER, only k <= j for a triangular B.
    // `executed` = what the launched tiles really run on the matrix pipe (whole tiles, shifted-tile and border
    // recompute included; 6 real flops per complex multiply-add in the 3M kernels, 8 in the 4M ones).
    double useful = 0.0, executed = 0.0;
    if (b->prof && b->prof->on) {
        // a REAL call is a real GEMM with twice the rows (or inner dimension): 2 * (2 m n k) flops, no 3M saving
        const double mac_useful = real ? 4.0 : 8.0;
        if (!(upper & 3)) {
            useful = mac_useful * (double)m * (double)n * (double)k;
        } else {
            for (int64_t j = 0; j < n; ++j) {
                const double rows = (upper & 1) ? (double)std::min<int64_t>(m, j + 1) : (double)m;
                const double kk = (upper & 2) ? (double)std::min<int64_t>(k, j + 1) : (double)k;
                useful += mac_useful * rows * kk;
            }
        }
        const double per_mac = real ? 4.0 : (b->use_mfma && use3m) ? 6.0 : 8.0;
        const int BNt = til.BNt;
        auto kext = [&](int tc) {   // k range a tile column runs (triangular B stops at the diagonal)
            const int64_t j0 = (til.shift && tc == til.gnf) ? n - BNt : (int64_t)tc * BNt;
            return (double)((upper & 2) ? std::min<int64_t>(k, j0 + BNt) : k);
        };
        auto live = [&](int64_t i0, int tc) {
            const int64_t jend = (til.shift && tc == til.gnf) ? n : std::min<int64_t>(n, (int64_t)tc * BNt + BNt);
            return !(upper & 1) || i0 < jend;
        };
        for (int tr = 0; tr < til.gmI; ++tr)
            for (int tc = 0; tc < til.gnI; ++tc)
                if (live((int64_t)tr * GEMM_BM, tc)) executed += per_mac * GEMM_BM * BNt * kext(tc);
        const int64_t mrem = m - (int64_t)til.gmf * GEMM_BM;
        const double rows_b = (double)(((mrem + 31) / 32) * 32);   // border: 32-row wave tiles that hold rows run
        for (int tc = 0; tc < til.nbottom; ++tc)
            if (live((int64_t)til.gmf * GEMM_BM, tc)) executed += per_mac * rows_b * BNt * kext(tc);
        for (int tr = 0; tr < til.nright; ++tr) {
            const double rr = tr < til.gmf ? (double)GEMM_BM : rows_b;
            if (live((int64_t)tr * GEMM_BM, til.gnf)) executed += per_mac * rr * BNt * kext(til.gnf);
        }
        if (!b->use_mfma) executed = useful;
    }
    const int slot = prof_begin(b, (upper & 3) ? PROF_ZGEMM_STRUCT : PROF_ZGEMM, useful, tag);
    if (slot >= 0) {
        b->prof->work[PROF_ZGEMM_BYTES] += 16.0 * ((double)m * k + (double)k * n +
                                                   (double)m * n * ((beta.x != 0.0 || beta.y != 0.0) ? 2.0 : 1.0));
        b->prof->work[PROF_ZGEMM_EXEC] += executed;
        b->prof->work[PROF_ZGEMM_CPLX] += real ? 2.0 * useful : useful;
    }
    struct ProfGuard {
        dftk_mi_basis* b;
        int s;
        ~ProfGuard() { prof_end(b, s); }
    } guard{b, slot};
    if (!b->use_mfma) {
        const unsigned blocks = (unsigned)((m * n + 255) / 256);
        if (conja)
            hipLaunchKernelGGL(k_zgemm_naive<true>, dim3(blocks), dim3(256), 0, b->stream, (int)m, (int)n, (int)k, A,
                               lda, B, ldb, C, ldc, alpha, beta, real ? 1 : 0);
        else
            hipLaunchKernelGGL(k_zgemm_naive<false>, dim3(blocks), dim3(256), 0, b->stream, (int)m, (int)n, (int)k,
                               A, lda, B, ldb, C, ldc, alpha, beta, real ? 1 : 0);
        HIPCHK(hipGetLastError());
        return 0;
    }
    const int64_t plane = (int64_t)m * n * (int64_t)sizeof(cd);
    // XCD-aware 1-D grid over a gm_s x gn_s sub-grid of tiles (x nsplit K chunks)
    auto grid_for = [&](int gm_s, int gn_s, int ns) -> int64_t {
        const int64_t rows_total = (int64_t)gm_s * ns;
        return ((rows_total + 7) / 8) * 8 * gn_s;
    };
    // Full tiles run the predicate-free instantiation (a ragged last tile column as one more full tile,
    // shifted left); the remaining ragged strips the predicated one, as ONE list-shaped launch (right strip:
    // all tile rows of the last tile column; bottom strip: the tile columns of the last tile row) with its own
    // K split.  (One mixed launch over all tiles was measured 2-7 % faster on the block updates, but the light
    // border workgroups run ahead of their siblings through k and FETCH_SIZE rose from 2.3x to 3.5x the
    // operand bytes -- dropped.)
    const int BNt = til.BNt, gmf = til.gmf, gnf = til.gnf, nright = til.nright, nbottom = til.nbottom;
    Split spI = til.I, spB = til.B;
    const size_t bytesI = spI.nsplit > 1 ? (size_t)spI.nsplit * plane : 0;
    const size_t bytesB = spB.nsplit > 1 ? (size_t)spB.nsplit * plane : 0;
    if (bytesI + bytesB) CHK(ensure_ws(b, bytesI + bytesB));
    if (bytesI) spI.slab = (cd*)b->ws;
    if (bytesB) spB.slab = (cd*)((char*)b->ws + bytesI);
    auto launch = [&](int mode, int gm_s, int gn_s, int rt0, int ct0, int lsplit, const Split& sp) -> int {
        if (gm_s <= 0 || gn_s <= 0) return 0;
        const bool zmajor = sp.zmajor;
        const bool compact = sp.compact && mode == 1 && lsplit < 0 && use3m;
        const int64_t nblk = compact  ? (((int64_t)sp.live * sp.nsplit + 7) / 8) * 8
                             : zmajor ? (int64_t)((sp.nsplit + 7) / 8) * 8 * gm_s * gn_s
                                      : grid_for(gm_s, gn_s, sp.nsplit);
        if (nblk > INT32_MAX) return DFTK_MI_EINVAL;
        dim3 grid((unsigned)nblk);
        const int upper = (upper_in & 3) | (compact ? 16 : zmajor ? 4 : 0) |
                          ((!compact && !zmajor && (upper_in & 1) && use3m && gemm_rotate_rows() && gm_s <= 4) ? 8 : 0);
#define DFTK_LAUNCH_LDS(CJ, FL)                                                                                        \
    hipLaunchKernelGGL((k_zgemm_lds<CJ, FL>), grid, dim3(GEMM_WAVES * 64), 0, b->stream, (int)m, (int)n, (int)k, \
                       sp.kchunk, gm_s, gn_s, rt0, ct0, lsplit, upper, sp.nsplit, A, lda, B, ldb, C, ldc, alpha, beta, sp.slab)
#define DFTK_LAUNCH_3M(CJ, MD)                                                                                         \
    if (real && BNt == 64) DFTK_LAUNCH_3M_(CJ, MD, true, 4);                                                           \
    else if (real) DFTK_LAUNCH_3M_(CJ, MD, true, 2);                                                                   \
    else DFTK_LAUNCH_3M_(CJ, MD, false, M3_RN)
#define DFTK_LAUNCH_3M_(CJ, MD, RL, RNN)                                                                               \
    hipLaunchKernelGGL((k_zgemm_3m<CJ, MD, RL, RNN>), grid, dim3(GEMM_WAVES * 64), 0, b->stream, (int)m, (int)n, (int)k,  \
                       sp.kchunk, gm_s, gn_s, rt0, ct0, lsplit, upper, til.shift ? gnf : -1,                           \
                       (til.shift_r && lsplit < 0) ? gmf : -1, sp.nsplit, A, lda, B, ldb, C,                           \
                       ldc, alpha, beta, sp.slab)
        // mode 1 = full tiles, 0 = predicated border
        if (use3m) {
            if (conja) {
                if (mode == 1) { DFTK_LAUNCH_3M(true, 1); }
                else { DFTK_LAUNCH_3M(true, 0); }
            } else {
                if (mode == 1) { DFTK_LAUNCH_3M(false, 1); }
                else { DFTK_LAUNCH_3M(false, 0); }
            }
        } else if (conja) {
            if (mode == 1) DFTK_LAUNCH_LDS(true, 1);
            else DFTK_LAUNCH_LDS(true, 0);
        } else {
            if (mode == 1) DFTK_LAUNCH_LDS(false, 1);
            else DFTK_LAUNCH_LDS(false, 0);
        }
#undef DFTK_LAUNCH_3M
#undef DFTK_LAUNCH_3M_
#undef DFTK_LAUNCH_LDS
        return 0;
    };
    CHK(launch(1, til.gmI, til.gnI, 0, 0, -1, spI));
    CHK(launch(0, nright + nbottom, 1, gmf, gnf, nright, spB));
    if (spI.slab || spB.slab)
        hipLaunchKernelGGL(k_zgemm_reduce, dim3((unsigned)((m * n + 255) / 256)), dim3(256), 0, b->stream, (int)m,
                           (int)n, til.shift_r ? (int)m : gmf * GEMM_BM, til.shift ? (int)n : gnf * BNt, spI.slab ? spI.nsplit : -1, spI.slab,
                           spB.slab ? spB.nsplit : -1, spB.slab, C, ldc, alpha, beta, upper);
    HIPCHK(hipGetLastError());
    return 0;
}
