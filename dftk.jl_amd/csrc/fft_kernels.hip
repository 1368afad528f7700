// fft_kernels.hip -- pruned, batched sphere<->cube 3-D FFT pipeline with the local-potential
// multiply fused in (gfx950).  Replaces the per-band loop of DFTK's
//   mul!(Hpsi, ::DftHamiltonianBlock, psi)   src/terms/Hamiltonian.jl:155-163
//   ifft!/fft! on the sphere                 src/fft.jl:110-122, 162-172
//   compute_density inner loop               src/densities.jl:35-43
//
// Pipeline per batch of bands ("FFT pipeline"):
//   A  x-lines that intersect the sphere:  scatter coefficients -> LDS, backward FFT_x -> T1
//   B  (x-tile, z-plane in sphere extent):  backward FFT_y of the non-empty lines       -> T2
//   C  (x-tile, y): backward FFT_z, multiply by V/N, forward FFT_z, keep sphere planes  -> T2
//   D  forward FFT_y, keep the sphere's lines                                            -> T1
//   E  forward FFT_x, gather coefficients, add the kinetic diagonal                      -> Hpsi
// Only the part of the zero-padded cube that can be non-zero is ever read or written.
//
// One workgroup = 256 threads = a tile of FFT_L = 8 lines held in LDS as buf[elem*FFT_LS + line];
// each line is transformed in place by a mixed-radix Stockham-free scheme: decimation in time
// (input placed at the permuted position `pos`, natural output) for backward transforms and
// decimation in frequency (natural input, output read back through `pos`) for forward ones.
#include "common.h"
#include "batch.h"
#include <algorithm>
#include <cstring>
#include <type_traits>

#define FFT_L 8
#ifndef ZPASS_MIN_BLOCKS
#define ZPASS_MIN_BLOCKS 6   // workgroups per CU the z pass is compiled for (80 VGPRs; 5 -> 6 resident workgroups: -9 %)
#endif
#ifndef DENS_MIN_BLOCKS
#define DENS_MIN_BLOCKS 1   // (6 spills: slower; prefetching the next band's planes into registers: 130 VGPRs, 165 vs 150 us)
#endif
#ifndef YFWD_MIN_BLOCKS
#define YFWD_MIN_BLOCKS 1   // (7 spills: slower)
#endif
#define FFT_LS_X 9     // x kernels: odd pitch keeps the transposed tile I/O (lanes along x) conflict-free
#define FFT_LS_YZ 8    // y/z kernels: no padding; with an odd first radix the butterfly accesses of a
                       // ds_read_b128 lane group (4 quads of lines x 4 consecutive butterflies) hit 16 distinct slots
#define FFT_THREADS 256
#define FFT_TPL (FFT_THREADS / FFT_L)
#define DENS_MAXACC 12   // supports nz <= 12*32 = 384

__device__ __forceinline__ cd cmul(cd a, cd w) {
    return make_double2(fma(a.x, w.x, -a.y * w.y), fma(a.x, w.y, a.y * w.x));
}
__device__ __forceinline__ cd cadd(cd a, cd b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ cd csub(cd a, cd b) { return make_double2(a.x - b.x, a.y - b.y); }
// multiply by (+i*s) : (x + i y) * i s = (-y s, x s)
__device__ __forceinline__ cd cmuli(cd a, double s) { return make_double2(-a.y * s, a.x * s); }

template <int R>
__device__ __forceinline__ void dft_small(cd (&a)[R], double sgn, const cd* __restrict__ tw, int nOverR) {
    if constexpr (R == 2) {
        cd t = a[0];
        a[0] = cadd(t, a[1]);
        a[1] = csub(t, a[1]);
    } else if constexpr (R == 3) {
        const double s3 = 0.86602540378443864676;   // sin(2 pi / 3)
        cd t1 = cadd(a[1], a[2]);
        cd t2 = make_double2(a[0].x - 0.5 * t1.x, a[0].y - 0.5 * t1.y);
        cd t3 = cmuli(csub(a[1], a[2]), sgn * s3);
        a[0] = cadd(a[0], t1);
        a[1] = cadd(t2, t3);
        a[2] = csub(t2, t3);
    } else if constexpr (R == 4) {
        cd s02 = cadd(a[0], a[2]), d02 = csub(a[0], a[2]);
        cd s13 = cadd(a[1], a[3]), d13 = cmuli(csub(a[1], a[3]), sgn);
        a[0] = cadd(s02, s13);
        a[2] = csub(s02, s13);
        a[1] = cadd(d02, d13);
        a[3] = csub(d02, d13);
    } else if constexpr (R == 5) {
        const double c1 = 0.30901699437494742410, c2 = -0.80901699437494742410;
        const double s1 = 0.95105651629515357212, s2 = 0.58778525229247312917;
        cd t1 = cadd(a[1], a[4]), t2 = cadd(a[2], a[3]);
        cd t3 = csub(a[1], a[4]), t4 = csub(a[2], a[3]);
        cd m1 = make_double2(a[0].x + c1 * t1.x + c2 * t2.x, a[0].y + c1 * t1.y + c2 * t2.y);
        cd m2 = make_double2(a[0].x + c2 * t1.x + c1 * t2.x, a[0].y + c2 * t1.y + c1 * t2.y);
        cd n1 = make_double2(s1 * t3.x + s2 * t4.x, s1 * t3.y + s2 * t4.y);
        cd n2 = make_double2(s2 * t3.x - s1 * t4.x, s2 * t3.y - s1 * t4.y);
        cd in1 = cmuli(n1, sgn), in2 = cmuli(n2, sgn);
        a[0] = make_double2(a[0].x + t1.x + t2.x, a[0].y + t1.y + t2.y);
        a[1] = cadd(m1, in1);
        a[4] = csub(m1, in1);
        a[2] = cadd(m2, in2);
        a[3] = csub(m2, in2);
    } else if constexpr (R == 6) {
        // 6 = 2 x 3: X[k] = E[k] + w^k O[k], X[k+3] = E[k] - w^k O[k], E/O = DFT_3 of the even/odd inputs
        const double s3 = 0.86602540378443864676;
        cd e[3] = {a[0], a[2], a[4]}, o[3] = {a[1], a[3], a[5]};
        dft_small<3>(e, sgn, tw, 0);
        dft_small<3>(o, sgn, tw, 0);
        const cd w1o = cmul(o[1], make_double2(0.5, sgn * s3));
        const cd w2o = cmul(o[2], make_double2(-0.5, sgn * s3));
        a[0] = cadd(e[0], o[0]);
        a[3] = csub(e[0], o[0]);
        a[1] = cadd(e[1], w1o);
        a[4] = csub(e[1], w1o);
        a[2] = cadd(e[2], w2o);
        a[5] = csub(e[2], w2o);
    } else if constexpr (R == 8) {
        // 8 = 2 x 4: X[k] = E[k] + w^k O[k], X[k+4] = E[k] - w^k O[k], w = exp(sgn i pi/4)
        const double h = 0.70710678118654752440;
        cd e[4] = {a[0], a[2], a[4], a[6]}, o[4] = {a[1], a[3], a[5], a[7]};
        dft_small<4>(e, sgn, tw, 0);
        dft_small<4>(o, sgn, tw, 0);
        const cd w1o = cmul(o[1], make_double2(h, sgn * h));
        const cd w2o = cmuli(o[2], sgn);
        const cd w3o = cmul(o[3], make_double2(-h, sgn * h));
        a[0] = cadd(e[0], o[0]);
        a[4] = csub(e[0], o[0]);
        a[1] = cadd(e[1], w1o);
        a[5] = csub(e[1], w1o);
        a[2] = cadd(e[2], w2o);
        a[6] = csub(e[2], w2o);
        a[3] = cadd(e[3], w3o);
        a[7] = csub(e[3], w3o);
    } else {
        cd b[R];
#pragma unroll
        for (int p = 0; p < R; ++p) {
            cd acc = a[0];
#pragma unroll
            for (int q = 1; q < R; ++q) {
                cd w = tw[((p * q) % R) * nOverR];
                w.y *= sgn;
                cd t = cmul(a[q], w);
                acc.x += t.x;
                acc.y += t.y;
            }
            b[p] = acc;
        }
#pragma unroll
        for (int p = 0; p < R; ++p) a[p] = b[p];
    }
}

// One radix-R pass over the tile.  m = product of the radices already combined (DIT) or still
// to be split (DIF).  Thread (l, j) owns line l and butterflies j, j+TPL, ...
// VMUL (last stage of a backward transform only): the natural-order outputs are multiplied by the real
// potential column vcol[z * vstride] and pushed through the first butterfly of the forward transform
// before they go back to LDS (fused V*psi: saves two LDS round trips; see fft_tile's `skip`).
template <int FFT_LS, int R, bool DIF, bool VMUL = false>
__device__ __forceinline__ void fft_stage(cd* buf, const cd* __restrict__ tw, int n, int m, double sgn,
                                          int l, int j, const double* __restrict__ vcol = nullptr,
                                          int64_t vstride = 0) {
    const int nb = n / R;
    const int twstep = n / (R * m);
    const int nOverR = n / R;
    for (int b = j; b < nb; b += FFT_TPL) {
        const int g = b / m;
        const int jj = b - g * m;
        const int base = g * R * m + jj;
        double vv[R];
        if (VMUL) {   // issued first: the global loads fly while the butterfly is computed
#pragma unroll
            for (int p = 0; p < R; ++p) vv[p] = vcol[(int64_t)(base + p * m) * vstride];
        }
        cd a[R];
#pragma unroll
        for (int q = 0; q < R; ++q) a[q] = buf[(base + q * m) * FFT_LS + l];
        if (!DIF && m > 1) {
#pragma unroll
            for (int q = 1; q < R; ++q) {
                cd w = tw[q * jj * twstep];
                w.y *= sgn;
                a[q] = cmul(a[q], w);
            }
        }
        dft_small<R>(a, sgn, tw, nOverR);
        if (DIF && m > 1) {
#pragma unroll
            for (int p = 1; p < R; ++p) {
                cd w = tw[p * jj * twstep];
                w.y *= sgn;
                a[p] = cmul(a[p], w);
            }
        }
        if (VMUL) {
            // middle pass of the fused local apply: these R natural-order outputs of the backward transform
            // are exactly the inputs of the first decimation-in-frequency butterfly of the forward transform
            // (same radix, same m): multiply by V and run that butterfly here, one LDS round trip for both
#pragma unroll
            for (int p = 0; p < R; ++p) {
                a[p].x *= vv[p];
                a[p].y *= vv[p];
            }
            dft_small<R>(a, -sgn, tw, nOverR);
            if (m > 1) {
#pragma unroll
                for (int p = 1; p < R; ++p) {
                    cd w = tw[p * jj * twstep];
                    w.y *= -sgn;
                    a[p] = cmul(a[p], w);
                }
            }
        }
#pragma unroll
        for (int p = 0; p < R; ++p) buf[(base + p * m) * FFT_LS + l] = a[p];
    }
}

// Arbitrary (prime) radix up to 64: slow path used only for non 2-3-5-7-11-13 sizes.
template <int FFT_LS, bool DIF>
__device__ void fft_stage_generic(cd* buf, const cd* __restrict__ tw, int n, int m, int R, double sgn,
                                  int l, int j) {
    const int nb = n / R;
    const int twstep = n / (R * m);
    const int nOverR = n / R;
    for (int b = j; b < nb; b += FFT_TPL) {
        const int g = b / m;
        const int jj = b - g * m;
        const int base = g * R * m + jj;
        cd a[64], o[64];
        for (int q = 0; q < R; ++q) {
            cd v = buf[(base + q * m) * FFT_LS + l];
            if (!DIF && m > 1 && q > 0) {
                cd w = tw[q * jj * twstep];
                w.y *= sgn;
                v = cmul(v, w);
            }
            a[q] = v;
        }
        for (int p = 0; p < R; ++p) {
            cd acc = a[0];
            for (int q = 1; q < R; ++q) {
                cd w = tw[((p * q) % R) * nOverR];
                w.y *= sgn;
                cd t = cmul(a[q], w);
                acc.x += t.x;
                acc.y += t.y;
            }
            if (DIF && m > 1 && p > 0) {
                cd w = tw[p * jj * twstep];
                w.y *= sgn;
                acc = cmul(acc, w);
            }
            o[p] = acc;
        }
        for (int p = 0; p < R; ++p) buf[(base + p * m) * FFT_LS + l] = o[p];
    }
}

// GEN = false: 2-3-5-smooth lengths only (the hot path; keeps the register footprint small).
// GEN = true : additionally radix 7 and the generic (scratch-backed) prime radix.
// last backward stage with the fused potential multiply (2-3-5 radices only; see k_zpass)
template <int FFT_LS>
__device__ __forceinline__ bool fft_stage_vmul(int R, cd* buf, const cd* tw, int n, int m, double sgn, int l, int j,
                                               const double* vcol, int64_t vstride) {
    switch (R) {
        case 2: fft_stage<FFT_LS, 2, false, true>(buf, tw, n, m, sgn, l, j, vcol, vstride); return true;
        case 3: fft_stage<FFT_LS, 3, false, true>(buf, tw, n, m, sgn, l, j, vcol, vstride); return true;
        case 4: fft_stage<FFT_LS, 4, false, true>(buf, tw, n, m, sgn, l, j, vcol, vstride); return true;
        case 5: fft_stage<FFT_LS, 5, false, true>(buf, tw, n, m, sgn, l, j, vcol, vstride); return true;
        case 6: fft_stage<FFT_LS, 6, false, true>(buf, tw, n, m, sgn, l, j, vcol, vstride); return true;
        case 8: fft_stage<FFT_LS, 8, false, true>(buf, tw, n, m, sgn, l, j, vcol, vstride); return true;
        default: return false;
    }
}

template <int FFT_LS, bool DIF, bool GEN>
__device__ __forceinline__ void fft_stage_dispatch(int R, cd* buf, const cd* tw, int n, int m, double sgn,
                                                   int l, int j) {
    switch (R) {
        case 2: fft_stage<FFT_LS, 2, DIF>(buf, tw, n, m, sgn, l, j); break;
        case 3: fft_stage<FFT_LS, 3, DIF>(buf, tw, n, m, sgn, l, j); break;
        case 4: fft_stage<FFT_LS, 4, DIF>(buf, tw, n, m, sgn, l, j); break;
        case 5: fft_stage<FFT_LS, 5, DIF>(buf, tw, n, m, sgn, l, j); break;
        case 6: fft_stage<FFT_LS, 6, DIF>(buf, tw, n, m, sgn, l, j); break;
        case 8: fft_stage<FFT_LS, 8, DIF>(buf, tw, n, m, sgn, l, j); break;
        default:
            if constexpr (GEN) {
                if (R == 7)
                    fft_stage<FFT_LS, 7, DIF>(buf, tw, n, m, sgn, l, j);
                else
                    fft_stage_generic<FFT_LS, DIF>(buf, tw, n, m, R, sgn, l, j);
            }
            break;
    }
}

// In-place transform of the whole tile; caller has synchronised after filling buf.
// Ends with a __syncthreads().
// vcol != nullptr (DIT only): the last stage also multiplies by vcol[z * vstride] and runs the first
// forward butterfly (the following forward fft_tile must then be called with skip_first); returns false if
// that stage's radix has no fused variant (the caller then multiplies in a separate pass).
template <int FFT_LS, bool DIF, bool GEN>
__device__ __forceinline__ bool fft_tile(cd* buf, const cd* tw, const FftAxis& ax, double sgn, int l, int j,
                                         const double* vcol = nullptr, int64_t vstride = 0,
                                         bool skip_first = false) {
    const int n = ax.n;
    bool fused = false;
    if (!DIF) {
        int m = 1;
        for (int s = 0; s < ax.nrad; ++s) {
            const int R = ax.rad[s];
            if (vcol != nullptr && s == ax.nrad - 1)
                fused = fft_stage_vmul<FFT_LS>(R, buf, tw, n, m, sgn, l, j, vcol, vstride);
            if (!fused) fft_stage_dispatch<FFT_LS, false, GEN>(R, buf, tw, n, m, sgn, l, j);
            __syncthreads();
            m *= R;
        }
    } else {
        int m = n;
        for (int s = ax.nrad - 1; s >= 0; --s) {
            const int R = ax.rad[s];
            m /= R;
            if (skip_first && s == ax.nrad - 1) continue;   // already done by the fused middle pass
            fft_stage_dispatch<FFT_LS, true, GEN>(R, buf, tw, n, m, sgn, l, j);
            __syncthreads();
        }
    }
    return fused;
}

template <int FFT_LS>
__device__ __forceinline__ void tile_prologue(cd* buf, cd* tw, const FftAxis& ax, bool zero, bool copy_tw = true) {
    const int n = ax.n;
    if (copy_tw)
        for (int t = threadIdx.x; t < n; t += FFT_THREADS) tw[t] = ax.tw[t];
    if (zero) {
        const cd z = make_double2(0.0, 0.0);
        for (int t = threadIdx.x; t < n * FFT_LS; t += FFT_THREADS) buf[t] = z;
    }
}

extern __shared__ __attribute__((aligned(16))) char dftk_smem[];

// Multi-k launches (batch.h): the bands of MANY k-blocks go through one launch of each stage.  `jobs` (nullable) holds
// one entry per band: the sphere tables of its k-block and its own pointers replace the scalar kernel arguments; the
// grid is sized for the largest k-block and surplus workgroups of smaller ones leave at once.
struct FftJob {
    const int *line_start, *cpos, *line_ypos, *zls, *zpos;
    int z_lo;                 // register-resident z kernels: sphere planes with z < nz/2
    const double* kin;    // stage E: kinetic multiplier (or null)
    const double* Vs;     // stage C: this k-block's padded potential / N
    const cd* psi;        // the band's sphere coefficients (stage A input, stage E kinetic term)
    cd* out;              // stage E output
    int n_lines, nzx;
    double w, wim;        // density weights of the band
    double w2;            // weight of the band in the SECOND accumulated cube (two-weight pass: density + LDOS), or 0
};

// ---------------------------------------------------------------------------------------- stage A
template <bool GEN>
__global__ __launch_bounds__(FFT_THREADS) void k_xbwd_scatter(FftAxis ax, int nxp, int n_lines,
                                                              const int* __restrict__ line_start,
                                                              const int* __restrict__ cpos,
                                                              const cd* __restrict__ psi, int64_t ldpsi,
                                                              cd* __restrict__ T1, int64_t T1_stride,
                                                              const FftJob* __restrict__ jobs) {
    constexpr int FFT_LS = FFT_LS_X;
    cd* buf = reinterpret_cast<cd*>(dftk_smem);
    cd* tw = buf + ax.n * FFT_LS;
    const int tid = threadIdx.x, l = tid & (FFT_L - 1), j = tid >> 3;
    const int band = blockIdx.y;
    const int l0 = blockIdx.x * FFT_L;
    const cd* p = psi + (int64_t)band * ldpsi;
    if (jobs) {
        const FftJob jb = jobs[band];
        n_lines = jb.n_lines;
        line_start = jb.line_start;
        cpos = jb.cpos;
        p = jb.psi;
        if (l0 >= n_lines) return;   // whole workgroup, before any barrier
    }
    tile_prologue<FFT_LS>(buf, tw, ax, true);
    __syncthreads();
    const int line = l0 + l;
    if (line < n_lines) {
        const int c0 = line_start[line], c1 = line_start[line + 1];
        for (int c = c0 + j; c < c1; c += FFT_TPL) buf[cpos[c] * FFT_LS + l] = p[c];
    }
    __syncthreads();
    fft_tile<FFT_LS, false, GEN>(buf, tw, ax, +1.0, l, j);
    cd* out = T1 + (int64_t)band * T1_stride + (int64_t)l0 * nxp;
    const int nlv = min(FFT_L, n_lines - l0);
    const int n = ax.n;
    for (int idx = tid; idx < nlv * nxp; idx += FFT_THREADS) {
        const int ll = idx / nxp;
        const int x = idx - ll * nxp;
        out[idx] = (x < n) ? buf[x * FFT_LS + ll] : make_double2(0.0, 0.0);
    }
}

// ---------------------------------------------------------------------------------------- stage B
template <bool GEN>
__global__ __launch_bounds__(FFT_THREADS) void k_ybwd(FftAxis ay, int nxp, int ny,
                                                      const int* __restrict__ zls,
                                                      const int* __restrict__ line_ypos,
                                                      const cd* __restrict__ T1, int64_t T1_stride,
                                                      cd* __restrict__ T2, int64_t T2_stride,
                                                      const FftJob* __restrict__ jobs) {
    constexpr int FFT_LS = FFT_LS_YZ;
    cd* buf = reinterpret_cast<cd*>(dftk_smem);
    cd* tw = buf + ay.n * FFT_LS;
    const int tid = threadIdx.x, l = tid & (FFT_L - 1), j = tid >> 3;
    const int x = blockIdx.x * FFT_L + l;
    const int zi = blockIdx.y, band = blockIdx.z;
    if (jobs) {
        const FftJob jb = jobs[band];
        if (zi >= jb.nzx) return;   // whole workgroup
        zls = jb.zls;
        line_ypos = jb.line_ypos;
    }
    const cd* t1 = T1 + (int64_t)band * T1_stride;
    const int ln0 = zls[zi] + j, ln1 = zls[zi + 1];
    // lines of this plane to registers before the prologue (HBM latency overlaps twiddles + zero fill);
    // up to 3 per thread (96 lines per plane), otherwise the plain loop
    const bool early = ln1 - zls[zi] <= 3 * FFT_TPL;
    cd e0, e1, e2;
    if (early) {
        if (ln0 < ln1) e0 = t1[(int64_t)ln0 * nxp + x];
        if (ln0 + FFT_TPL < ln1) e1 = t1[(int64_t)(ln0 + FFT_TPL) * nxp + x];
        if (ln0 + 2 * FFT_TPL < ln1) e2 = t1[(int64_t)(ln0 + 2 * FFT_TPL) * nxp + x];
    }
    tile_prologue<FFT_LS>(buf, tw, ay, true);
    __syncthreads();
    if (early) {
        if (ln0 < ln1) buf[line_ypos[ln0] * FFT_LS + l] = e0;
        if (ln0 + FFT_TPL < ln1) buf[line_ypos[ln0 + FFT_TPL] * FFT_LS + l] = e1;
        if (ln0 + 2 * FFT_TPL < ln1) buf[line_ypos[ln0 + 2 * FFT_TPL] * FFT_LS + l] = e2;
    } else {
        for (int ln = ln0; ln < ln1; ln += FFT_TPL) buf[line_ypos[ln] * FFT_LS + l] = t1[(int64_t)ln * nxp + x];
    }
    __syncthreads();
    fft_tile<FFT_LS, false, GEN>(buf, tw, ay, +1.0, l, j);
    cd* t2 = T2 + (int64_t)band * T2_stride + (int64_t)zi * ny * nxp + x;
    for (int y = j; y < ny; y += FFT_TPL) t2[(int64_t)y * nxp] = buf[y * FFT_LS + l];
}

// ---------------------------------------------------------------------------------------- stage C
// MODE 0: backward z, multiply by Vs, forward z (fused local apply), T2 -> T2 in place
// MODE 1: backward z only, natural-order output to an (nx,ny,nz) cube
// MODE 2: forward z only from an (nx,ny,nz) cube, sphere planes -> T2
template <int MODE, bool GEN>
__global__ __launch_bounds__(FFT_THREADS, ZPASS_MIN_BLOCKS) void k_zpass(FftAxis az, int nx, int nxp, int ny, int nzx,
                                                       int nbands, const int* __restrict__ zpos,
                                                       const double* __restrict__ Vs,
                                                       cd* __restrict__ T2, int64_t T2_stride,
                                                       cd* __restrict__ cube, const FftJob* __restrict__ jobs = nullptr) {
    constexpr int FFT_LS = FFT_LS_YZ;
    cd* buf = reinterpret_cast<cd*>(dftk_smem);
    cd* tw = buf + az.n * FFT_LS;
    const int tid = threadIdx.x, l = tid & (FFT_L - 1), j = tid >> 3;
    // 1-D XCD-aware grid: workgroup id -> (xcd = id % 8, slot = id / 8); the bands of one (x tile, y) column
    // are consecutive slots of the SAME XCD, so the potential tile they all multiply with is fetched into
    // that XCD's L2 once (with a (bands, x tiles, y) grid the 8 bands of a batch land on 8 different XCDs
    // and each L2 fetches the tile itself: PMC FETCH_SIZE 419 MB per launch against 246 MB of operands)
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int band = slot % nbands;
    const int grp = (slot / nbands) * 8 + xcd;
    const int nxt = nxp / FFT_L;
    if (grp >= nxt * ny) return;   // whole workgroup, before any barrier
    if (jobs) {
        const FftJob jb = jobs[band];
        zpos = jb.zpos;
        nzx = jb.nzx;
        Vs = jb.Vs;
    }
    const int y = grp / nxt;
    const int x = (grp - y * nxt) * FFT_L + l;
    const int nz = az.n;
    const int64_t plane = (int64_t)ny * nxp;
    cd* t2 = T2 + (int64_t)band * T2_stride + (int64_t)y * nxp + x;
    // the sphere planes of this column go to registers BEFORE the prologue (twiddles + zero fill), so that the
    // HBM latency of the tile overlaps it (up to 3 planes per thread: nzx <= 96; more -> plain loop below)
    const bool early = MODE != 2 && nzx <= 3 * FFT_TPL;
    cd e0, e1, e2;
    if (early) {
        if (j < nzx) e0 = t2[(int64_t)j * plane];
        if (j + FFT_TPL < nzx) e1 = t2[(int64_t)(j + FFT_TPL) * plane];
        if (j + 2 * FFT_TPL < nzx) e2 = t2[(int64_t)(j + 2 * FFT_TPL) * plane];
    }
    tile_prologue<FFT_LS>(buf, tw, az, MODE != 2, true);
    __syncthreads();
    bool fused_v = false;
    if (MODE != 2) {
        if (early) {
            if (j < nzx) buf[zpos[j] * FFT_LS + l] = e0;
            if (j + FFT_TPL < nzx) buf[zpos[j + FFT_TPL] * FFT_LS + l] = e1;
            if (j + 2 * FFT_TPL < nzx) buf[zpos[j + 2 * FFT_TPL] * FFT_LS + l] = e2;
        } else {
            for (int zi = j; zi < nzx; zi += FFT_TPL) buf[zpos[zi] * FFT_LS + l] = t2[(int64_t)zi * plane];
        }
        __syncthreads();
        fused_v = fft_tile<FFT_LS, false, GEN>(buf, tw, az, +1.0, l, j,
                                       (MODE == 0 && !GEN) ? Vs + (int64_t)y * nxp + x : nullptr, plane);
    }
    if (MODE == 0 && !fused_v) {
        const double* v = Vs + (int64_t)y * nxp + x;
        for (int z = j; z < nz; z += FFT_TPL) {
            const double s = v[(int64_t)z * plane];
            cd a = buf[z * FFT_LS + l];
            a.x *= s;
            a.y *= s;
            buf[z * FFT_LS + l] = a;
        }
        __syncthreads();
    }
    if (MODE == 1 || MODE == 2) cube += (int64_t)band * nz * ny * nx;   // several cubes per launch: one behind the other
    if (MODE == 1) {
        if (x < nx)
            for (int z = j; z < nz; z += FFT_TPL) cube[((int64_t)z * ny + y) * nx + x] = buf[z * FFT_LS + l];
        return;
    }
    if (MODE == 2) {
        for (int z = j; z < nz; z += FFT_TPL)
            buf[z * FFT_LS + l] = (x < nx) ? cube[((int64_t)z * ny + y) * nx + x] : make_double2(0.0, 0.0);
        __syncthreads();
    }
    fft_tile<FFT_LS, true, GEN>(buf, tw, az, -1.0, l, j, nullptr, 0, MODE == 0 && fused_v);
    for (int zi = j; zi < nzx; zi += FFT_TPL) t2[(int64_t)zi * plane] = buf[zpos[zi] * FFT_LS + l];
}

// ---------------------------------------------------------------------------------------- stage C, register-resident
// "Four-step" z pass for n = R1 * R2 (R1 = R1A * R1B, R2 = R2A * R2B, all factors in {2, 3, 4, 5}): every thread holds a
// whole R1- or R2-point sub-transform in registers (Cooley-Tukey R?A x R?B with COMPILE-TIME twiddles), the tile goes
// through LDS only for the transposition between the two sub-transforms:
//   backward:  thread n2 < R2 loads x[R2 n1 + n2] straight from T2 (sphere planes by index arithmetic, zeros elsewhere),
//              DFT_R1 over n1, twiddle w^(n2 k1), LDS [k1][n2];  thread k1 < R1 reads its row, DFT_R2 over n2 ->
//              the natural-order values psi(z = k1 + R1 k2) -- multiplied by V(z) in registers;
//   forward:   the SAME thread already holds the inputs of its R2-point sub-transform (over k2): DFT_R2, twiddle
//              w^-(k1 k1'), LDS [k1'][k1];  thread k1' < R2 reads its row, DFT_R1 over k1 -> frequencies
//              z' = k1' + R2 k2', the sphere planes of which go straight back to T2.
// Two LDS round trips and three barriers per tile instead of five passes with a barrier each, no zero fill, no position
// tables, and 45 % fewer instructions (the 16- and 12-point butterflies need no twiddle loads).
namespace ctw {
constexpr double PI = 3.14159265358979323846264338327950288;
constexpr double sin_small(double x) {   // |x| <= pi/4: Taylor series to x^25
    const double x2 = x * x;
    double term = x, sum = x;
    for (int k = 1; k <= 12; ++k) {
        term *= -x2 / (double)((2 * k) * (2 * k + 1));
        sum += term;
    }
    return sum;
}
constexpr double cos_small(double x) {
    const double x2 = x * x;
    double term = 1.0, sum = 1.0;
    for (int k = 1; k <= 12; ++k) {
        term *= -x2 / (double)((2 * k - 1) * (2 * k));
        sum += term;
    }
    return sum;
}
struct CS {
    double c, s;
};
constexpr CS cs(int m, int N) {   // exp(2 pi i m / N), evaluated by the compiler
    m %= N;
    if (m < 0) m += N;
    const int k = (8 * m + N) / (2 * N);   // nearest quarter turn
    const double delta = 2.0 * PI * (double)(4 * m - k * N) / (4.0 * (double)N);
    const double c = cos_small(delta), s = sin_small(delta);
    switch (k & 3) {
        case 0: return CS{c, s};
        case 1: return CS{-s, c};
        case 2: return CS{-c, -s};
        default: return CS{s, -c};
    }
}
}   // namespace ctw

template <int B, int E, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        static_for<B + 1, E>(f);
    }
}
// a * exp(SGN 2 pi i M / N) with the twiddle folded into the instruction stream
template <int N, int SGN, int M>
__device__ __forceinline__ cd tw_mul_c(cd a) {
    constexpr int m = ((M % N) + N) % N;
    if constexpr (m == 0) {
        return a;
    } else if constexpr (2 * m == N) {
        return make_double2(-a.x, -a.y);
    } else if constexpr (4 * m == N) {
        return SGN > 0 ? make_double2(-a.y, a.x) : make_double2(a.y, -a.x);
    } else if constexpr (4 * m == 3 * N) {
        return SGN > 0 ? make_double2(a.y, -a.x) : make_double2(-a.y, a.x);
    } else {
        constexpr ctw::CS w = ctw::cs(m, N);
        constexpr double c = w.c, s = (SGN > 0 ? w.s : -w.s);
        return make_double2(fma(a.x, c, -a.y * s), fma(a.x, s, a.y * c));
    }
}
// In-register DFT of RA * RB points: input index n = RB a + b, result x[RB c + d] = X[c + RA d]
template <int RA, int RB, int SGN>
__device__ __forceinline__ void dft_ct(cd (&x)[RA * RB]) {
    static_for<0, RB>([&](auto bi) {
        constexpr int b = decltype(bi)::value;
        cd t[RA];
#pragma unroll
        for (int a = 0; a < RA; ++a) t[a] = x[RB * a + b];
        dft_small<RA>(t, (double)SGN, nullptr, 0);
        static_for<0, RA>([&](auto ci) {
            constexpr int c = decltype(ci)::value;
            x[RB * c + b] = tw_mul_c<RA * RB, SGN, b * c>(t[c]);
        });
    });
    static_for<0, RA>([&](auto ci) {
        constexpr int c = decltype(ci)::value;
        cd u[RB];
#pragma unroll
        for (int b = 0; b < RB; ++b) u[b] = x[RB * c + b];
        dft_small<RB>(u, (double)SGN, nullptr, 0);
#pragma unroll
        for (int d = 0; d < RB; ++d) x[RB * c + d] = u[d];
    });
}

#ifndef ZREG_MIN_BLOCKS
#define ZREG_MIN_BLOCKS 3   // waves per SIMD the register-resident kernels are compiled for (second argument of HIP's
                            // __launch_bounds__ = waves per execution unit; LDS allows 6 workgroups of 2 waves at 192)
#endif
// One tile (FFT_L lines) of four-step transforms of length N = R1 R2.  Threads (l, j): l = line, j < max(R1, R2).
template <int R1A, int R1B, int R2A, int R2B>
struct FourStep {
    static constexpr int R1 = R1A * R1B, R2 = R2A * R2B, N = R1 * R2, L = FFT_L;
    static constexpr int TPL = R1 > R2 ? R1 : R2, THREADS = L * TPL;
    // row pitches of the two transposition images: = 8 (mod 16) elements, so that the two rows a 16-lane group of a
    // ds_read_b128 touches (8 lines x 2 consecutive j) lie 32 banks apart (a pitch of "row + 1" puts lane (l + 1, j) and
    // lane (l, j + 1) on the same bank: PMC showed 35 % of the LDS cycles as bank conflicts)
    static constexpr int P1 = R2 * L + ((R2 % 2 == 0) ? 8 : 0), P2 = R1 * L + ((R1 % 2 == 0) ? 8 : 0);
    static constexpr int LDS_ELEMS = R1 * P1 > R2 * P2 ? R1 * P1 : R2 * P2;
    // natural index of register p of a finished sub-transform (dft_ct's output order)
    static constexpr int k2_of(int p) { return p / R2B + R2A * (p % R2B); }    // DFT_R2 results: index k2
    static constexpr int k2p_of(int p) { return p / R1B + R1A * (p % R1B); }   // DFT_R1 results: index k2'

    // Backward transform (exp(+i)).  in(n1): input element e = R2 n1 + j of this thread (n1 a std::integral_constant; zero
    // where the pruned pipeline has nothing).
    // Thread j < R1 receives out[p] = X[j + R1 k2_of(p)].  Ends after the LDS reads: the caller places a barrier before the
    // tile image is written again.
    template <class IN>
    static __device__ __forceinline__ void backward(cd* buf, const cd* __restrict__ twg, int l, int j, IN in, cd (&out)[R2]) {
        if (j < R2) {
            cd a[R1];
            static_for<0, R1>([&](auto ni) { a[decltype(ni)::value] = in(ni); });   // element e = R2 n1 + j
            const cd w1 = twg[j], wA = twg[j * R1A];   // bases of the twiddles w^(j k1), k1 = c + R1A d
            dft_ct<R1A, R1B, +1>(a);
            cd wd = make_double2(1.0, 0.0);
            static_for<0, R1B>([&](auto di) {
                constexpr int d = decltype(di)::value;
                if constexpr (d > 0) wd = (d == 1) ? wA : cmul(wd, wA);
                cd w = wd;
                static_for<0, R1A>([&](auto ci) {
                    constexpr int c = decltype(ci)::value;
                    if constexpr (c > 0) w = cmul(w, w1);
                    constexpr int k1 = c + R1A * d;
                    buf[k1 * P1 + j * L + l] = (k1 == 0) ? a[R1B * c + d] : cmul(a[R1B * c + d], w);
                });
            });
        }
        __syncthreads();
        if (j < R1) {
#pragma unroll
            for (int n2 = 0; n2 < R2; ++n2) out[n2] = buf[j * P1 + n2 * L + l];
            dft_ct<R2A, R2B, +1>(out);
        }
    }
    // Forward transform (exp(-i)) of f[k2] = x[j + R1 k2] held by thread j < R1 (natural k2 order).  Thread j < R2 receives
    // out[p] = X[j + R2 k2p_of(p)].  Starts by writing the tile image: barrier before if it may still be read.
    static __device__ __forceinline__ void forward(cd* buf, const cd* __restrict__ twg, int l, int j, cd (&f)[R2],
                                                   cd (&out)[R1]) {
        if (j < R1) {
            const cd w1 = twg[j], wA = twg[j * R2A];   // twiddles conj(w)^(j k1'), k1' = c + R2A d
            dft_ct<R2A, R2B, -1>(f);
            cd wd = make_double2(1.0, 0.0);
            static_for<0, R2B>([&](auto di) {
                constexpr int d = decltype(di)::value;
                if constexpr (d > 0) wd = (d == 1) ? wA : cmul(wd, wA);
                cd w = wd;
                static_for<0, R2A>([&](auto ci) {
                    constexpr int c = decltype(ci)::value;
                    if constexpr (c > 0) w = cmul(w, w1);
                    constexpr int k1p = c + R2A * d;
                    buf[k1p * P2 + j * L + l] = (k1p == 0) ? f[R2B * c + d] : cmul(f[R2B * c + d], make_double2(w.x, -w.y));
                });
            });
        }
        __syncthreads();
        if (j < R2) {
#pragma unroll
            for (int k1 = 0; k1 < R1; ++k1) out[k1] = buf[j * P2 + k1 * L + l];
            dft_ct<R1A, R1B, -1>(out);
        }
    }
};
// element at a 32-bit BYTE offset from a workgroup-uniform base: scalar base + one vector offset in the memory instruction
// instead of a 64-bit multiply-add per element (every T2 / V / rho column of a launch spans < 2^32 bytes from its tile origin)
template <class T>
__device__ __forceinline__ T ld_off(const T* base, unsigned byte_off) {
    return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + byte_off);
}
template <class T>
__device__ __forceinline__ void st_off(T* base, unsigned byte_off, T v) {
    *reinterpret_cast<T*>(reinterpret_cast<char*>(base) + byte_off) = v;
}

// stage C: backward z, multiply by V / N, forward z, in place on the sphere planes of T2
template <int R1A, int R1B, int R2A, int R2B>
__global__ __launch_bounds__(FFT_L * ((R1A * R1B > R2A * R2B) ? R1A * R1B : R2A * R2B), ZREG_MIN_BLOCKS)
void k_zpass_reg(FftAxis az, int nx, int nxp, int ny, int nzx, int z_lo, int nbands, const double* __restrict__ Vs,
                 cd* __restrict__ T2, int64_t T2_stride, const FftJob* __restrict__ jobs) {
    typedef FourStep<R1A, R1B, R2A, R2B> FS;
    cd* buf = reinterpret_cast<cd*>(dftk_smem);
    const int tid = threadIdx.x, l = tid & (FFT_L - 1), j = tid >> 3;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int band = slot % nbands;
    const int grp = (slot / nbands) * 8 + xcd;
    const int nxt = nxp / FFT_L;
    if (grp >= nxt * ny) return;   // whole workgroup, before any barrier
    if (jobs) {
        const FftJob jb = jobs[band];
        nzx = jb.nzx;
        z_lo = jb.z_lo;
        Vs = jb.Vs;
    }
    const int y = grp / nxt;
    const int x = (grp - y * nxt) * FFT_L + l;
    const unsigned plane = (unsigned)ny * (unsigned)nxp;
    const int64_t col0 = (int64_t)y * nxp + (x - l);               // tile origin: uniform over the workgroup
    cd* __restrict__ t2 = T2 + (int64_t)band * T2_stride + col0;
    // byte offsets of this thread's planes from the tile origin: plane zi = e for e < z_lo, e - (N - nzx) for e >= zneg0; the
    // elements of a thread are R2 planes apart, i.e. a workgroup-uniform step
    const unsigned planeB = plane * (unsigned)sizeof(cd), stepB = (unsigned)FS::R2 * planeB;
    const unsigned off_lo = (unsigned)j * planeB + (unsigned)l * (unsigned)sizeof(cd);
    const unsigned off_hi = off_lo - (unsigned)(FS::N - nzx) * planeB;   // (mod 2^32; only used where it is a valid offset)
    const int zneg0 = FS::N - (nzx - z_lo);
    auto in = [&](auto ni) -> cd {
        constexpr int n1 = decltype(ni)::value;
        const int e = FS::R2 * n1 + j;
        const bool lo = e < z_lo, hi = e >= zneg0;
        cd v = make_double2(0.0, 0.0);
        if (lo || hi) v = ld_off(t2, (lo ? off_lo : off_hi) + (unsigned)n1 * stepB);
        return v;
    };
    cd psi[FS::R2];
    double vv[FS::R2];
    if (j < FS::R1) {   // potential column of this thread's z = j + R1 k2 (in flight during the backward transform)
        const double* __restrict__ vcol = Vs + col0;
        const unsigned vb = ((unsigned)j * plane + (unsigned)l) * 8u;
#pragma unroll
        for (int k2 = 0; k2 < FS::R2; ++k2) vv[k2] = ld_off(vcol, vb + (unsigned)(FS::R1 * k2) * plane * 8u);
    }
    FS::backward(buf, az.tw, l, j, in, psi);
    cd f[FS::R2];
    if (j < FS::R1) {
        static_for<0, FS::R2>([&](auto pi) {
            constexpr int p = decltype(pi)::value;
            constexpr int k2 = FS::k2_of(p);
            f[k2] = make_double2(psi[p].x * vv[k2], psi[p].y * vv[k2]);
        });
    }
    __syncthreads();   // every row of the first image has been read
    cd out[FS::R1];
    FS::forward(buf, az.tw, l, j, f, out);
    if (j < FS::R2) {
        static_for<0, FS::R1>([&](auto pi) {
            constexpr int p = decltype(pi)::value;
            constexpr int k2p = FS::k2p_of(p);
            const int e = j + FS::R2 * k2p;
            const bool lo = e < z_lo, hi = e >= zneg0;
            if (lo || hi) st_off(t2, (lo ? off_lo : off_hi) + (unsigned)k2p * stepB, out[p]);
        });
    }
}

// density: rho[z, y, x] += sum_band w Re^2 + wim Im^2 of the backward z transform; thread j < R1 accumulates its R2 planes
template <int R1A, int R1B, int R2A, int R2B>
__global__ __launch_bounds__(FFT_L * ((R1A * R1B > R2A * R2B) ? R1A * R1B : R2A * R2B), ZREG_MIN_BLOCKS)
void k_zdensity_reg(FftAxis az, int nx, int nxp, int ny, int nzx, int z_lo, int nb, const double* __restrict__ w,
                    const double* __restrict__ wim, const cd* __restrict__ T2, int64_t T2_stride,
                    double* __restrict__ rho, const FftJob* __restrict__ jobs, double* __restrict__ part,
                    double* __restrict__ rho2 = nullptr) {
    typedef FourStep<R1A, R1B, R2A, R2B> FS;
    cd* buf = reinterpret_cast<cd*>(dftk_smem);
    const int tid = threadIdx.x, l = tid & (FFT_L - 1), j = tid >> 3;
    const int x = blockIdx.x * FFT_L + l;
    const int y = blockIdx.y;
    const unsigned plane = (unsigned)ny * (unsigned)nxp;
    double acc[FS::R2], acc2[FS::R2];
#pragma unroll
    for (int p = 0; p < FS::R2; ++p) acc[p] = acc2[p] = 0.0;
    // part != null: gridDim.z band groups (bands blockIdx.z, + gridDim.z, ...), each WRITES its partial cube part[group]
    // (k_dens_reduce adds them to rho in group order) -- small cubes have too few (x tile, y) columns to fill the chip
    for (int ib = blockIdx.z; ib < nb; ib += gridDim.z) {
        double wb, wi, wb2 = 0.0;
        if (jobs) {
            const FftJob jb = jobs[ib];
            wb = jb.w;
            wi = jb.wim;
            wb2 = jb.w2;
            nzx = jb.nzx;
            z_lo = jb.z_lo;
        } else {
            wb = w[ib];
            wi = wim ? wim[ib] : wb;
        }
        if (wb == 0.0 && wi == 0.0 && wb2 == 0.0) continue;   // uniform across the block
        const cd* __restrict__ t2 = T2 + (int64_t)ib * T2_stride + (int64_t)y * nxp + (x - l);   // tile origin: uniform
        const unsigned planeB = plane * (unsigned)sizeof(cd), stepB = (unsigned)FS::R2 * planeB;
        const unsigned off_lo = (unsigned)j * planeB + (unsigned)l * (unsigned)sizeof(cd);
        const unsigned off_hi = off_lo - (unsigned)(FS::N - nzx) * planeB;
        const int zneg0 = FS::N - (nzx - z_lo);
        auto in = [&](auto ni) -> cd {
            constexpr int n1 = decltype(ni)::value;
            const int e = FS::R2 * n1 + j;
            const bool lo = e < z_lo, hi = e >= zneg0;
            cd v = make_double2(0.0, 0.0);
            if (lo || hi) v = ld_off(t2, (lo ? off_lo : off_hi) + (unsigned)n1 * stepB);
            return v;
        };
        cd psi[FS::R2];
        FS::backward(buf, az.tw, l, j, in, psi);
        if (j < FS::R1) {
#pragma unroll
            for (int p = 0; p < FS::R2; ++p) {
                acc[p] = fma(wb, psi[p].x * psi[p].x, fma(wi, psi[p].y * psi[p].y, acc[p]));
                if (rho2) acc2[p] = fma(wb2, psi[p].x * psi[p].x + psi[p].y * psi[p].y, acc2[p]);
            }
        }
        __syncthreads();   // the tile image is rewritten by the next band
    }
    if (j < FS::R1 && x < nx) {
        double* __restrict__ r0 = (part ? part + (int64_t)blockIdx.z * az.n * ny * nx : rho) + (int64_t)y * nx + (x - l);   // tile origin of this y row: uniform
        const unsigned zB = (unsigned)ny * (unsigned)nx * 8u;         // bytes between z planes of rho (the cube is < 2^32 bytes:
        const unsigned ob = (unsigned)j * zB + (unsigned)l * 8u;     //  192^3 doubles = 57 MB; checked by the launcher)
        static_for<0, FS::R2>([&](auto pi) {
            constexpr int p = decltype(pi)::value;
            const unsigned off = ob + (unsigned)(FS::R1 * FS::k2_of(p)) * zB;
            st_off(r0, off, part ? acc[p] : ld_off((const double*)r0, off) + acc[p]);
        });
        if (rho2) {   // second cube: the partial cubes of the groups follow those of the first (gridDim.z of them)
            double* __restrict__ q0 = (part ? part + ((int64_t)gridDim.z + blockIdx.z) * az.n * ny * nx : rho2) + (int64_t)y * nx + (x - l);
            static_for<0, FS::R2>([&](auto pi) {
                constexpr int p = decltype(pi)::value;
                const unsigned off = ob + (unsigned)(FS::R1 * FS::k2_of(p)) * zB;
                st_off(q0, off, part ? acc2[p] : ld_off((const double*)q0, off) + acc2[p]);
            });
        }
    }
}

// ---------------------------------------------------------------------------------------- stage D
template <bool GEN>
__global__ __launch_bounds__(FFT_THREADS, YFWD_MIN_BLOCKS) void k_yfwd(FftAxis ay, int nxp, int ny,
                                                      const int* __restrict__ zls,
                                                      const int* __restrict__ line_ypos,
                                                      const cd* __restrict__ T2, int64_t T2_stride,
                                                      cd* __restrict__ T1, int64_t T1_stride,
                                                      const FftJob* __restrict__ jobs) {
    constexpr int FFT_LS = FFT_LS_YZ;
    cd* buf = reinterpret_cast<cd*>(dftk_smem);
    cd* tw = buf + ay.n * FFT_LS;
    const int tid = threadIdx.x, l = tid & (FFT_L - 1), j = tid >> 3;
    const int x = blockIdx.x * FFT_L + l;
    const int zi = blockIdx.y, band = blockIdx.z;
    if (jobs) {
        const FftJob jb = jobs[band];
        if (zi >= jb.nzx) return;   // whole workgroup
        zls = jb.zls;
        line_ypos = jb.line_ypos;
    }
    tile_prologue<FFT_LS>(buf, tw, ay, false);
    const cd* t2 = T2 + (int64_t)band * T2_stride + (int64_t)zi * ny * nxp + x;
    for (int y = j; y < ny; y += FFT_TPL) buf[y * FFT_LS + l] = t2[(int64_t)y * nxp];
    __syncthreads();
    fft_tile<FFT_LS, true, GEN>(buf, tw, ay, -1.0, l, j);
    cd* t1 = T1 + (int64_t)band * T1_stride;
    const int ln1 = zls[zi + 1];
    for (int ln = zls[zi] + j; ln < ln1; ln += FFT_TPL)
        t1[(int64_t)ln * nxp + x] = buf[line_ypos[ln] * FFT_LS + l];
}

// ---------------------------------------------------------------------------------------- stage E
// out[c] = FFT_x(line)[ix_c] (+ kin[c] * psi[c]) (+ out[c] if accumulate)
template <bool GEN>
__global__ __launch_bounds__(FFT_THREADS, YFWD_MIN_BLOCKS) void k_xfwd_gather(FftAxis ax, int nxp, int n_lines,
                                                             const int* __restrict__ line_start,
                                                             const int* __restrict__ cpos,
                                                             const cd* __restrict__ T1, int64_t T1_stride,
                                                             const double* __restrict__ kin,
                                                             const cd* __restrict__ psi, int64_t ldpsi,
                                                             cd* __restrict__ out, int64_t ldout,
                                                             const FftJob* __restrict__ jobs) {
    constexpr int FFT_LS = FFT_LS_X;
    cd* buf = reinterpret_cast<cd*>(dftk_smem);
    cd* tw = buf + ax.n * FFT_LS;
    const int tid = threadIdx.x, l = tid & (FFT_L - 1), j = tid >> 3;
    const int band = blockIdx.y;
    const int l0 = blockIdx.x * FFT_L;
    const int n = ax.n;
    cd* o = out + (int64_t)band * ldout;
    const cd* p = psi + (int64_t)band * ldpsi;
    if (jobs) {
        const FftJob jb = jobs[band];
        n_lines = jb.n_lines;
        line_start = jb.line_start;
        cpos = jb.cpos;
        kin = jb.kin;
        p = jb.psi;
        o = jb.out;
        if (l0 >= n_lines) return;   // whole workgroup, before any barrier
    }
    tile_prologue<FFT_LS>(buf, tw, ax, false);
    const cd* in = T1 + (int64_t)band * T1_stride + (int64_t)l0 * nxp;
    const int nlv = min(FFT_L, n_lines - l0);
    for (int idx = tid; idx < FFT_L * nxp; idx += FFT_THREADS) {
        const int ll = idx / nxp;
        const int x = idx - ll * nxp;
        if (x < n) buf[x * FFT_LS + ll] = (ll < nlv) ? in[idx] : make_double2(0.0, 0.0);
    }
    __syncthreads();
    fft_tile<FFT_LS, true, GEN>(buf, tw, ax, -1.0, l, j);
    const int line = l0 + l;
    if (line < n_lines) {
        const int c0 = line_start[line], c1 = line_start[line + 1];
        for (int c = c0 + j; c < c1; c += FFT_TPL) {
            cd v = buf[cpos[c] * FFT_LS + l];
            if (kin != nullptr) {
                const double k = kin[c];
                const cd q = p[c];
                v.x = fma(k, q.x, v.x);
                v.y = fma(k, q.y, v.y);
            }
            o[c] = v;
        }
    }
}

// ---------------------------------------------------------------------------------------- density
// rho[z,y,x] += sum_band w[band] |BFFT_z(T2[band])|^2 ; one workgroup owns an (x-tile, y) column set.
// wim (nullable): the squared imaginary part is weighted with wim[band] instead (two real-symmetric orbitals packed
// as a + i b into one transform, gamma_kernels.hip: rho += w Re^2 + wim Im^2).
template <bool GEN>
__global__ __launch_bounds__(FFT_THREADS, DENS_MIN_BLOCKS) void k_zdensity(FftAxis az, int nx, int nxp, int ny, int nzx,
                                                          const int* __restrict__ zpos, int nb,
                                                          const double* __restrict__ w,
                                                          const double* __restrict__ wim,
                                                          const cd* __restrict__ T2, int64_t T2_stride,
                                                          double* __restrict__ rho, const FftJob* __restrict__ jobs,
                                                          double* __restrict__ part, double* __restrict__ rho2 = nullptr) {
    constexpr int FFT_LS = FFT_LS_YZ;
    cd* buf = reinterpret_cast<cd*>(dftk_smem);
    cd* tw = buf + az.n * FFT_LS;
    const int tid = threadIdx.x, l = tid & (FFT_L - 1), j = tid >> 3;
    const int x = blockIdx.x * FFT_L + l;
    const int y = blockIdx.y;
    const int nz = az.n;
    const int64_t plane = (int64_t)ny * nxp;
    double acc[DENS_MAXACC], acc2[DENS_MAXACC];
#pragma unroll
    for (int k = 0; k < DENS_MAXACC; ++k) acc[k] = acc2[k] = 0.0;
    for (int t = tid; t < nz; t += FFT_THREADS) tw[t] = az.tw[t];
    // (band groups over gridDim.z with partial cubes: see k_zdensity_reg; rho2: the second cube of a two-weight pass)
    for (int ib = blockIdx.z; ib < nb; ib += gridDim.z) {
        double wb, wi, wb2 = 0.0;
        if (jobs) {               // multi-k launch: the band's k-block decides the z planes, its entry the weights
            const FftJob jb = jobs[ib];
            wb = jb.w;
            wi = jb.wim;
            wb2 = jb.w2;
            zpos = jb.zpos;
            nzx = jb.nzx;
        } else {
            wb = w[ib];
            wi = wim ? wim[ib] : wb;
        }
        if (wb == 0.0 && wi == 0.0 && wb2 == 0.0) continue;   // uniform across the block
        const cd z0 = make_double2(0.0, 0.0);
        for (int t = tid; t < nz * FFT_LS; t += FFT_THREADS) buf[t] = z0;
        __syncthreads();
        const cd* t2 = T2 + (int64_t)ib * T2_stride + (int64_t)y * nxp + x;
        for (int zi = j; zi < nzx; zi += FFT_TPL) buf[zpos[zi] * FFT_LS + l] = t2[(int64_t)zi * plane];
        __syncthreads();
        fft_tile<FFT_LS, false, GEN>(buf, tw, az, +1.0, l, j);
#pragma unroll
        for (int k = 0; k < DENS_MAXACC; ++k) {
            const int z = j + k * FFT_TPL;
            if (z < nz) {
                const cd v = buf[z * FFT_LS + l];
                acc[k] = fma(wb, v.x * v.x, fma(wi, v.y * v.y, acc[k]));
                if (rho2) acc2[k] = fma(wb2, v.x * v.x + v.y * v.y, acc2[k]);
            }
        }
        __syncthreads();
    }
    if (x < nx) {
        double* __restrict__ dst = part ? part + (int64_t)blockIdx.z * nz * ny * nx : rho;
#pragma unroll
        for (int k = 0; k < DENS_MAXACC; ++k) {
            const int z = j + k * FFT_TPL;
            if (z < nz) {
                const int64_t at = ((int64_t)z * ny + y) * nx + x;
                dst[at] = part ? acc[k] : dst[at] + acc[k];
            }
        }
        if (rho2) {
            double* __restrict__ dst2 = part ? part + ((int64_t)gridDim.z + blockIdx.z) * nz * ny * nx : rho2;
#pragma unroll
            for (int k = 0; k < DENS_MAXACC; ++k) {
                const int z = j + k * FFT_TPL;
                if (z < nz) {
                    const int64_t at = ((int64_t)z * ny + y) * nx + x;
                    dst2[at] = part ? acc2[k] : dst2[at] + acc2[k];
                }
            }
        }
    }
}

// rho += part[0] + part[1] + ... (group order: deterministic)
__global__ void k_dens_reduce(int64_t n, int ngroups, const double* __restrict__ part, double* __restrict__ rho) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double s = 0.0;
    for (int g = 0; g < ngroups; ++g) s += part[(int64_t)g * n + i];
    rho[i] += s;
}

__global__ void k_pad_potential(int nx, int nxp, int64_t rows, double scale, const double* __restrict__ V,
                                double* __restrict__ Vs) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * nxp) return;
    const int64_t r = i / nxp;
    const int x = (int)(i - r * nxp);
    Vs[i] = (x < nx) ? V[r * nx + x] * scale : 0.0;
}

__global__ void k_kinetic(int64_t n, int nb, const double* __restrict__ kin, const cd* __restrict__ psi,
                          int64_t ldpsi, cd* __restrict__ out, int64_t ldout, int accumulate) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double k = kin ? kin[i] : 0.0;
    for (int b = 0; b < nb; ++b) {
        cd q = psi[(int64_t)b * ldpsi + i];
        cd v = make_double2(k * q.x, k * q.y);
        if (accumulate) {
            cd o = out[(int64_t)b * ldout + i];
            v.x += o.x;
            v.y += o.y;
        }
        out[(int64_t)b * ldout + i] = v;
    }
}

// ======================================================================================== host side
static size_t lds_bytes(int n, int ls = FFT_LS_YZ) { return (size_t)n * (ls + 1) * sizeof(cd); }

int fft_ensure_scratch(dftk_mi_basis* b, dftk_mi_kblock* kb, int nb) {
    const size_t t1 = (size_t)nb * kb->n_lines * b->nxp * sizeof(cd);
    const size_t t2 = (size_t)nb * kb->nzx * b->ny * b->nxp * sizeof(cd);
    if (t1 > b->T1_bytes) {
        if (b->T1) HIPCHK(hipFree(b->T1));
        b->T1 = nullptr;
        HIPCHK(dftk_scratch_malloc((void**)&b->T1, t1));
        b->T1_bytes = t1;
    }
    if (t2 > b->T2_bytes) {
        if (b->T2) HIPCHK(hipFree(b->T2));
        b->T2 = nullptr;
        HIPCHK(dftk_scratch_malloc((void**)&b->T2, t2));
        b->T2_bytes = t2;
    }
    return 0;
}

static int check_lds(dftk_mi_basis* b) {
    const int nmax = b->nx > b->ny ? (b->nx > b->nz ? b->nx : b->nz) : (b->ny > b->nz ? b->ny : b->nz);
    if (lds_bytes(nmax, FFT_LS_X) > 160 * 1024) {
        dftk_set_error("FFT axis length %d needs %zu B of LDS (> 160 KiB)", nmax, lds_bytes(nmax, FFT_LS_X));
        return DFTK_MI_EINVAL;
    }
    return 0;
}

template <typename K>
static int set_lds_attr(K kernel, size_t bytes) {
    if (bytes > 64 * 1024) {
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    }
    return 0;
}

static bool axis_generic(const FftAxis& ax) {
    for (int s = 0; s < ax.nrad; ++s)
        if (ax.rad[s] > 5 && ax.rad[s] != 6 && ax.rad[s] != 8) return true;
    return false;
}

// launch KERNEL<GEN> with GEN chosen from the axis plan (2-3-5-smooth => lean kernel)
#define LAUNCH_FFT(KERNEL, AX, GRID, LDS, STREAM, ...)                                               \
    do {                                                                                             \
        if (axis_generic(AX)) {                                                                      \
            CHK(set_lds_attr(KERNEL<true>, LDS));                                                    \
            hipLaunchKernelGGL((KERNEL<true>), GRID, dim3(FFT_THREADS), LDS, STREAM, __VA_ARGS__);   \
        } else {                                                                                     \
            CHK(set_lds_attr(KERNEL<false>, LDS));                                                   \
            hipLaunchKernelGGL((KERNEL<false>), GRID, dim3(FFT_THREADS), LDS, STREAM, __VA_ARGS__);  \
        }                                                                                            \
    } while (0)
#define LAUNCH_ZPASS(MODE, AX, GRID, LDS, STREAM, ...)                                                      \
    do {                                                                                                    \
        if (axis_generic(AX)) {                                                                             \
            CHK(set_lds_attr(k_zpass<MODE, true>, LDS));                                                    \
            hipLaunchKernelGGL((k_zpass<MODE, true>), GRID, dim3(FFT_THREADS), LDS, STREAM, __VA_ARGS__);   \
        } else {                                                                                            \
            CHK(set_lds_attr(k_zpass<MODE, false>, LDS));                                                   \
            hipLaunchKernelGGL((k_zpass<MODE, false>), GRID, dim3(FFT_THREADS), LDS, STREAM, __VA_ARGS__);  \
        }                                                                                                   \
    } while (0)

// Register-resident z kernels (FourStep): used for the axis lengths with an instantiated factorisation n = R1 R2
// (R1 = R1A R1B >= R2 = R2A R2B) when the sphere's z planes wrap around contiguously (kb->z_lo >= 0: always so for a sphere
// of G vectors).  The y passes keep the LDS-pass kernels: they are HBM bound (4.9 / 3.9 TB/s at 192^3) and the
// register-resident variants of stages B and D ran at exactly their speed (113.8 vs 112.9, 144.5 vs 141.4 us per launch).  DFTK_MI_FFT_REG=0 keeps the LDS-pass kernels everywhere, DFTK_MI_FFT_REG_MIN=n (default 64)
// is the shortest axis they take (below that a tile has fewer than 64 threads).  The launchers return 1 if not applicable.
#define REG_SIZES(X)                                                                                                      \
    X(24, 2, 2, 3, 2) X(27, 3, 3, 3, 1) X(30, 5, 1, 3, 2) X(32, 2, 2, 4, 2) X(36, 3, 2, 3, 2) X(40, 2, 2, 5, 2) X(48, 3, 2, 4, 2) X(54, 3, 2, 3, 3)           \
    X(45, 3, 3, 5, 1) X(50, 5, 2, 5, 1) X(60, 3, 2, 5, 2)                                                                 \
    X(64, 4, 2, 4, 2) X(72, 3, 3, 4, 2) X(80, 5, 2, 4, 2) X(90, 5, 2, 3, 3) X(96, 4, 3, 4, 2) X(100, 5, 2, 5, 2)          \
    X(108, 4, 3, 3, 3) X(120, 4, 3, 5, 2) X(128, 4, 4, 4, 2) X(144, 4, 3, 4, 3) X(150, 5, 3, 5, 2) X(160, 4, 4, 5, 2)     \
    X(180, 5, 3, 4, 3) X(192, 4, 4, 4, 3) X(200, 5, 4, 5, 2) X(216, 6, 3, 4, 3) X(240, 4, 4, 5, 3) X(256, 4, 4, 4, 4)
static bool fft_reg_on(int n) {
    static const bool off = getenv("DFTK_MI_FFT_REG") != nullptr && atoi(getenv("DFTK_MI_FFT_REG")) == 0;
    // (round 6: the factorisations of 24 ... 60 were added for the cubes of the k-point workloads -- 36^3 of the Al cell:
    //  stage C of 432 bands 280 -> ~150 us, 56 -> 63 SCF it/s; the default threshold was 64)
    static const int nmin = getenv("DFTK_MI_FFT_REG_MIN") ? atoi(getenv("DFTK_MI_FFT_REG_MIN")) : 24;
    return !off && n >= nmin;
}
// the z kernels address a band's T2 slab, the potential and the density cube with 32-bit byte offsets from a tile origin
static bool fft_reg_fits(const dftk_mi_basis* b) {
    return (uint64_t)b->nz * (uint64_t)b->ny * (uint64_t)b->nxp * sizeof(cd) < (1ull << 32);
}
struct RegZ {   // arguments of the z kernels
    dftk_mi_basis* b;
    hipStream_t stream;
    dim3 grid;
    int nzx, z_lo, nbands;
    const double* Vs;
    cd* T2;
    int64_t s2;
    const double *w, *wim;
    double* rho;
    const FftJob* jobs;
    double* part = nullptr;   // density: partial cubes of grid.z band groups (k_dens_reduce), or null
    double* rho2 = nullptr;   // density: second accumulated cube (weights FftJob::w2), or null
};
template <int A, int B, int C, int D>
static int reg_zpass_t(const RegZ& r) {
    typedef FourStep<A, B, C, D> FS;
    const size_t lds = (size_t)FS::LDS_ELEMS * sizeof(cd);
    CHK(set_lds_attr(k_zpass_reg<A, B, C, D>, lds));
    hipLaunchKernelGGL((k_zpass_reg<A, B, C, D>), r.grid, dim3(FS::THREADS), lds, r.stream, r.b->ax[2], r.b->nx, r.b->nxp, r.b->ny,
                       r.nzx, r.z_lo, r.nbands, r.Vs, r.T2, r.s2, r.jobs);
    return 0;
}
template <int A, int B, int C, int D>
static int reg_zdens_t(const RegZ& r) {
    typedef FourStep<A, B, C, D> FS;
    const size_t lds = (size_t)FS::LDS_ELEMS * sizeof(cd);
    CHK(set_lds_attr(k_zdensity_reg<A, B, C, D>, lds));
    hipLaunchKernelGGL((k_zdensity_reg<A, B, C, D>), r.grid, dim3(FS::THREADS), lds, r.stream, r.b->ax[2], r.b->nx, r.b->nxp,
                       r.b->ny, r.nzx, r.z_lo, r.nbands, r.w, r.wim, (const cd*)r.T2, r.s2, r.rho, r.jobs, r.part, r.rho2);
    return 0;
}
static int reg_zpass(const RegZ& r, bool tables_ok) {
    if (!tables_ok || !fft_reg_on(r.b->nz) || !fft_reg_fits(r.b)) return 1;
    switch (r.b->nz) {
#define X(NN, A, B, C, D) case NN: return reg_zpass_t<A, B, C, D>(r);
        REG_SIZES(X)
#undef X
        default: return 1;
    }
}
static int reg_zdens(const RegZ& r, bool tables_ok) {
    if (!tables_ok || !fft_reg_on(r.b->nz) || !fft_reg_fits(r.b)) return 1;
    switch (r.b->nz) {
#define X(NN, A, B, C, D) case NN: return reg_zdens_t<A, B, C, D>(r);
        REG_SIZES(X)
#undef X
        default: return 1;
    }
}

// 1-D grid of k_zpass: (x tile, y) columns rounded up to a multiple of the 8 XCDs, times the bands of the launch
static dim3 zpass_grid(const dftk_mi_basis* b, int nbands) {
    const int64_t groups = (int64_t)(b->nxp / FFT_L) * b->ny;
    return dim3((unsigned)(((groups + 7) / 8) * 8 * nbands));
}

struct Strides {
    int64_t s1, s2;
};
static Strides strides(dftk_mi_kblock* kb) {
    dftk_mi_basis* b = kb->basis;
    return Strides{(int64_t)kb->n_lines * b->nxp, (int64_t)kb->nzx * b->ny * b->nxp};
}

static int run_AB(dftk_mi_kblock* kb, int nbb, const cd* psi, int64_t ldpsi) {
    dftk_mi_basis* b = kb->basis;
    const Strides st = strides(kb);
    const int gl = (int)((kb->n_lines + FFT_L - 1) / FFT_L);
    const int nxt = b->nxp / FFT_L;
    const double Ncube = (double)b->nx * b->ny * b->nz;
    // algorithmic HBM bytes of the PRUNED pipeline (what a perfect implementation must move):
    //   T1 = n_lines x nxp, T2 = nzx x ny x nxp complex numbers per band
    const double t1b = 16.0 * (double)kb->n_lines * b->nxp, t2b = 16.0 * (double)kb->nzx * b->ny * b->nxp;
    (void)Ncube;
    int ps = prof_begin(b, PROF_FFT_A, (16.0 * kb->n_G + t1b) * nbb);
    LAUNCH_FFT(k_xbwd_scatter, b->ax[0], dim3(gl, nbb), lds_bytes(b->nx, FFT_LS_X), b->stream, b->ax[0],
                       b->nxp, (int)kb->n_lines, kb->d_line_start, kb->d_cpos, psi, ldpsi, b->T1, st.s1, (const FftJob*)nullptr);
    prof_end(b, ps);
    ps = prof_begin(b, PROF_FFT_B, (t1b + t2b) * nbb);
    LAUNCH_FFT(k_ybwd, b->ax[1], dim3(nxt, kb->nzx, nbb), lds_bytes(b->ny), b->stream, b->ax[1],
                       b->nxp, b->ny, kb->d_zls, kb->d_line_ypos, b->T1, st.s1, b->T2, st.s2, (const FftJob*)nullptr);
    prof_end(b, ps);
    return 0;
}

static int run_DE(dftk_mi_kblock* kb, int nbb, const double* kin, const cd* psi, int64_t ldpsi, cd* out,
                  int64_t ldout) {
    dftk_mi_basis* b = kb->basis;
    const Strides st = strides(kb);
    const int gl = (int)((kb->n_lines + FFT_L - 1) / FFT_L);
    const int nxt = b->nxp / FFT_L;
    const double Ncube = (double)b->nx * b->ny * b->nz;
    const double t1b = 16.0 * (double)kb->n_lines * b->nxp, t2b = 16.0 * (double)kb->nzx * b->ny * b->nxp;
    (void)Ncube;
    int ps = prof_begin(b, PROF_FFT_D, (t1b + t2b) * nbb);
    LAUNCH_FFT(k_yfwd, b->ax[1], dim3(nxt, kb->nzx, nbb), lds_bytes(b->ny), b->stream, b->ax[1],
                       b->nxp, b->ny, kb->d_zls, kb->d_line_ypos, b->T2, st.s2, b->T1, st.s1, (const FftJob*)nullptr);
    prof_end(b, ps);
    ps = prof_begin(b, PROF_FFT_E, (t1b + (kin ? 40.0 : 16.0) * kb->n_G) * nbb);
    LAUNCH_FFT(k_xfwd_gather, b->ax[0], dim3(gl, nbb), lds_bytes(b->nx, FFT_LS_X), b->stream, b->ax[0],
                       b->nxp, (int)kb->n_lines, kb->d_line_start, kb->d_cpos, b->T1, st.s1, kin, psi, ldpsi, out,
                       ldout, (const FftJob*)nullptr);
    prof_end(b, ps);
    return 0;
}

int launch_local_apply(dftk_mi_kblock* kb, int nb, const cd* psi, int64_t ldpsi, cd* out, int64_t ldout,
                       bool add_kinetic, bool have_local) {
    dftk_mi_basis* b = kb->basis;
    if (!have_local || kb->d_Vs == nullptr) {
        // no local potential: out = kinetic * psi (or zero)
        return launch_kinetic_only(kb, nb, psi, ldpsi, out, ldout, false, add_kinetic);
    }
    CHK(check_lds(b));
    const int batch = b->fft_batch;
    CHK(fft_ensure_scratch(b, kb, nb < batch ? nb : batch));
    const Strides st = strides(kb);
    const int nxt = b->nxp / FFT_L;
    for (int b0 = 0; b0 < nb; b0 += batch) {
        const int nbb = (nb - b0) < batch ? (nb - b0) : batch;
        const cd* p = psi + (int64_t)b0 * ldpsi;
        CHK(run_AB(kb, nbb, p, ldpsi));
        // stage C: T2 read + written per band, the potential once per launch
        const int pc = prof_begin(b, PROF_FFT_C, 2.0 * 16.0 * (double)kb->nzx * b->ny * b->nxp * nbb +
                                                     8.0 * (double)b->nz * b->ny * b->nxp);
        const RegZ rz{b, b->stream, zpass_grid(b, nbb), kb->nzx, kb->z_lo, nbb, kb->d_Vs, b->T2, st.s2, nullptr, nullptr, nullptr,
                      nullptr};
        const int reg_st = reg_zpass(rz, kb->z_lo >= 0);
        if (reg_st < 0) return reg_st;
        if (reg_st != 0)
        LAUNCH_ZPASS(0, b->ax[2], zpass_grid(b, nbb), lds_bytes(b->nz), b->stream, b->ax[2], b->nx, b->nxp, b->ny, kb->nzx, nbb, kb->d_zpos, kb->d_Vs, b->T2, st.s2,
                           (cd*)nullptr, (const FftJob*)nullptr);
        prof_end(b, pc);
        CHK(run_DE(kb, nbb, add_kinetic ? kb->d_kin : nullptr, p, ldpsi, out + (int64_t)b0 * ldout, ldout));
    }
    HIPCHK(hipGetLastError());
    return 0;
}

int launch_kinetic_only(dftk_mi_kblock* kb, int nb, const cd* psi, int64_t ldpsi, cd* out, int64_t ldout,
                        bool accumulate, bool use_kin) {
    dftk_mi_basis* b = kb->basis;
    const int threads = 256;
    const int blocks = (int)((kb->n_G + threads - 1) / threads);
    hipLaunchKernelGGL(k_kinetic, dim3(blocks), dim3(threads), 0, b->stream, kb->n_G, nb,
                       use_kin ? kb->d_kin : (const double*)nullptr, psi, ldpsi, out, ldout, accumulate ? 1 : 0);
    HIPCHK(hipGetLastError());
    return 0;
}

// nb coefficient vectors (n_G apart) <-> nb cubes (nx ny nz apart) through ONE pipeline; in place is fine (the destination
// is written by the last stage only, the source is consumed by the first)
int launch_ifft_to_cube(dftk_mi_kblock* kb, const cd* c, cd* cube, int nb) {
    dftk_mi_basis* b = kb->basis;
    CHK(check_lds(b));
    CHK(fft_ensure_scratch(b, kb, nb));
    const Strides st = strides(kb);
    CHK(run_AB(kb, nb, c, kb->n_G));
    LAUNCH_ZPASS(1, b->ax[2], zpass_grid(b, nb), lds_bytes(b->nz), b->stream, b->ax[2], b->nx, b->nxp, b->ny, kb->nzx, nb, kb->d_zpos, (const double*)nullptr, b->T2, st.s2,
                       cube, (const FftJob*)nullptr);
    HIPCHK(hipGetLastError());
    return 0;
}

int launch_fft_from_cube(dftk_mi_kblock* kb, const cd* cube, cd* c, int nb) {
    dftk_mi_basis* b = kb->basis;
    CHK(check_lds(b));
    CHK(fft_ensure_scratch(b, kb, nb));
    const Strides st = strides(kb);
    LAUNCH_ZPASS(2, b->ax[2], zpass_grid(b, nb), lds_bytes(b->nz), b->stream, b->ax[2], b->nx, b->nxp, b->ny, kb->nzx, nb, kb->d_zpos, (const double*)nullptr, b->T2, st.s2,
                       const_cast<cd*>(cube), (const FftJob*)nullptr);
    CHK(run_DE(kb, nb, nullptr, c, kb->n_G, c, kb->n_G));
    HIPCHK(hipGetLastError());
    return 0;
}

int launch_density(dftk_mi_kblock* kb, int nb, const cd* psi, int64_t ldpsi, const double* w_h, double* rho,
                   const double* w_im_h, const double* w2_h, double* rho2) {
    dftk_mi_basis* b = kb->basis;
    if (batching() && nb > 0) {   // part of a batched multi-k call: the bands of all k-blocks share one pipeline later
        // payload: [w | wim (flags & 1) | w2 (flags & 2)]; D = the second cube of a two-weight pass
        BOp o;
        o.b = b;
        o.kb = kb;
        o.type = BOP_DENSITY; o.m = nb; o.A = psi; o.lda = ldpsi; o.C = rho; o.D = w2_h ? rho2 : nullptr;
        o.flags = (w_im_h ? 1 : 0) | (w2_h ? 2 : 0);
        o.payload.resize((size_t)(1 + (w_im_h ? 1 : 0) + (w2_h ? 1 : 0)) * nb * sizeof(double));
        memcpy(o.payload.data(), w_h, (size_t)nb * sizeof(double));
        size_t at = (size_t)nb * sizeof(double);
        if (w_im_h) {
            memcpy(o.payload.data() + at, w_im_h, (size_t)nb * sizeof(double));
            at += (size_t)nb * sizeof(double);
        }
        if (w2_h) memcpy(o.payload.data() + at, w2_h, (size_t)nb * sizeof(double));
        return batch_record(std::move(o));
    }
    if (w2_h) {   // outside a batched call: two passes (the one-pass form exists for the merged multi-k pipeline only)
        CHK(launch_density(kb, nb, psi, ldpsi, w_h, rho, w_im_h, nullptr, nullptr));
        return launch_density(kb, nb, psi, ldpsi, w2_h, rho2, nullptr, nullptr, nullptr);
    }
    CHK(check_lds(b));
    if (b->nz > DENS_MAXACC * FFT_TPL) {
        dftk_set_error("density kernel supports nz <= %d", DENS_MAXACC * FFT_TPL);
        return DFTK_MI_EINVAL;
    }
    const int batch = b->fft_batch;
    CHK(fft_ensure_scratch(b, kb, nb < batch ? nb : batch));
    const Strides st = strides(kb);
    // all weights go to the device once (general workspace; the copy from the caller's pageable array is
    // staged by the runtime before the call returns) -- no host synchronisation between the band batches
    CHK(ensure_ws(b, 2 * (size_t)nb * sizeof(double)));
    double* w_d = reinterpret_cast<double*>(b->ws);
    double* wim_d = w_im_h ? w_d + nb : nullptr;
    HIPCHK(hipMemcpyAsync(w_d, w_h, (size_t)nb * sizeof(double), hipMemcpyHostToDevice, b->stream));
    if (w_im_h) HIPCHK(hipMemcpyAsync(wim_d, w_im_h, (size_t)nb * sizeof(double), hipMemcpyHostToDevice, b->stream));
    for (int b0 = 0; b0 < nb; b0 += batch) {
        const int nbb = (nb - b0) < batch ? (nb - b0) : batch;
        bool any = false;
        for (int i = 0; i < nbb; ++i) any = any || (w_h[b0 + i] != 0.0) || (w_im_h && w_im_h[b0 + i] != 0.0);
        if (!any) continue;
        CHK(run_AB(kb, nbb, psi + (int64_t)b0 * ldpsi, ldpsi));
        const int pz = prof_begin(b, PROF_DENS_Z, 16.0 * (double)kb->nzx * b->ny * b->nxp * nbb +
                                                      16.0 * (double)b->nx * b->ny * b->nz);   // T2 per band + rho read-modify-write
        const RegZ rz{b, b->stream, dim3(b->nxp / FFT_L, b->ny), kb->nzx, kb->z_lo, nbb, nullptr, b->T2, st.s2, w_d + b0,
                      wim_d ? wim_d + b0 : (const double*)nullptr, rho, nullptr};
        const int rs = reg_zdens(rz, kb->z_lo >= 0);
        if (rs < 0) return rs;
        if (rs == 1)
        LAUNCH_FFT(k_zdensity, b->ax[2], dim3(b->nxp / FFT_L, b->ny), lds_bytes(b->nz), b->stream, b->ax[2], b->nx, b->nxp, b->ny, kb->nzx, kb->d_zpos, nbb, w_d + b0,
                           wim_d ? wim_d + b0 : (const double*)nullptr, b->T2, st.s2, rho, (const FftJob*)nullptr, (double*)nullptr);
        prof_end(b, pz);
    }
    HIPCHK(hipGetLastError());
    return 0;
}

int launch_pad_potential(dftk_mi_kblock* kb, const double* V) {
    dftk_mi_basis* b = kb->basis;
    const int64_t rows = (int64_t)b->ny * b->nz;
    const int64_t total = rows * b->nxp;
    const double scale = 1.0 / ((double)b->nx * b->ny * b->nz);
    hipLaunchKernelGGL(k_pad_potential, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, b->stream, b->nx,
                       b->nxp, rows, scale, V, kb->d_Vs);
    HIPCHK(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------- multi-k executors (batch.h)
// One pipeline for the bands of MANY k-blocks: job table (one entry per band), scratch for all of them, the five
// stages launched once with grids sized for the largest k-block.  Bands are processed in chunks so that the scratch
// stays below ~2 GiB.
namespace {
struct MultiPlan {
    std::vector<FftJob> jobs;
    int max_lines = 1, max_nzx = 1;
    bool reg_z = true;   // every k-block's sphere planes wrap around contiguously (register-resident z kernels)
};
void add_jobs(MultiPlan& mp, const dftk_mi_kblock* kb, int nb, const cd* psi, int64_t ldpsi, cd* out, int64_t ldout,
              bool kinetic, const double* w, const double* wim, const double* w2 = nullptr) {
    for (int i = 0; i < nb; ++i) {
        FftJob j;
        j.line_start = kb->d_line_start;
        j.cpos = kb->d_cpos;
        j.line_ypos = kb->d_line_ypos;
        j.zls = kb->d_zls;
        j.zpos = kb->d_zpos;
        j.z_lo = kb->z_lo;
        j.kin = kinetic ? kb->d_kin : nullptr;
        j.Vs = kb->d_Vs;
        j.psi = psi + (int64_t)i * ldpsi;
        j.out = out ? out + (int64_t)i * ldout : nullptr;
        j.n_lines = (int)kb->n_lines;
        j.nzx = kb->nzx;
        j.w = w ? w[i] : 0.0;
        j.wim = wim ? wim[i] : j.w;
        j.w2 = w2 ? w2[i] : 0.0;
        mp.jobs.push_back(j);
    }
    mp.max_lines = std::max(mp.max_lines, (int)kb->n_lines);
    mp.max_nzx = std::max(mp.max_nzx, kb->nzx);
    mp.reg_z = mp.reg_z && kb->z_lo >= 0;
}
}  // namespace

int batch_exec_apply_H(BatchCtx* ctx, hipStream_t stream, std::vector<BOp*>& ops) {
    dftk_mi_basis* b = ops[0]->kb->basis;
    for (BOp* o : ops) {
        const dftk_mi_kblock* kb = o->kb;
        // the merged pipeline covers the full H psi of unsharded blocks with a local potential; anything else one by one
        if (o->flags != 7 || kb->basis != b || kb->sh_comm || kb->d_Vs == nullptr) return 1;
    }
    if (axis_generic(b->ax[0]) || axis_generic(b->ax[1]) || axis_generic(b->ax[2])) return 1;
    CHK(check_lds(b));
    MultiPlan mp;
    for (BOp* o : ops)
        add_jobs(mp, o->kb, o->m, reinterpret_cast<const cd*>(o->A), o->lda, reinterpret_cast<cd*>(o->C), o->ldc, true, nullptr,
                 nullptr);
    const int nxt = b->nxp / FFT_L;
    const int64_t s1 = (int64_t)mp.max_lines * b->nxp, s2 = (int64_t)mp.max_nzx * b->ny * b->nxp;
    const size_t per_band = (size_t)(s1 + s2) * sizeof(cd);
    int chunk = (int)std::max<size_t>(1, std::min<size_t>(mp.jobs.size(), ((size_t)2 << 30) / per_band));
    if (chunk > 4096) chunk = 4096;
    // one scratch request for the whole call: T1 / T2 of a chunk of bands, then the two projection panels of all k-blocks
    size_t np_total = 0;
    for (BOp* o : ops) np_total += (size_t)o->kb->n_p * o->m;
    BatchScratchScope scratch_scope(ctx);
    cd* T1 = reinterpret_cast<cd*>(batch_scratch(ctx, per_band * chunk + 2 * np_total * sizeof(cd)));
    if (!T1) return DFTK_MI_EHIP;
    cd* T2 = T1 + (size_t)s1 * chunk;
    cd* Pbuf = reinterpret_cast<cd*>(reinterpret_cast<char*>(T1) + per_band * chunk);
    const int gl = (mp.max_lines + FFT_L - 1) / FFT_L;
    for (size_t j0 = 0; j0 < mp.jobs.size(); j0 += chunk) {
        const int nb = (int)std::min<size_t>(chunk, mp.jobs.size() - j0);
        const FftJob* dj = reinterpret_cast<const FftJob*>(batch_stage(ctx, mp.jobs.data() + j0, nb * sizeof(FftJob)));
        if (!dj) return DFTK_MI_EHIP;
        CHK(set_lds_attr(k_xbwd_scatter<false>, lds_bytes(b->nx, FFT_LS_X)));
        CHK(set_lds_attr(k_ybwd<false>, lds_bytes(b->ny)));
        CHK(set_lds_attr(k_yfwd<false>, lds_bytes(b->ny)));
        CHK(set_lds_attr(k_xfwd_gather<false>, lds_bytes(b->nx, FFT_LS_X)));
        hipLaunchKernelGGL((k_xbwd_scatter<false>), dim3(gl, nb), dim3(FFT_THREADS), lds_bytes(b->nx, FFT_LS_X), stream, b->ax[0],
                           b->nxp, 0, (const int*)nullptr, (const int*)nullptr, (const cd*)nullptr, (int64_t)0, T1, s1, dj);
        hipLaunchKernelGGL((k_ybwd<false>), dim3(nxt, mp.max_nzx, nb), dim3(FFT_THREADS), lds_bytes(b->ny), stream, b->ax[1],
                           b->nxp, b->ny, (const int*)nullptr, (const int*)nullptr, (const cd*)T1, s1, T2, s2, dj);
        {
            const int64_t groups = (int64_t)nxt * b->ny;
            const dim3 grid((unsigned)(((groups + 7) / 8) * 8 * nb));
            const RegZ rz{b, stream, grid, 0, 0, nb, nullptr, T2, s2, nullptr, nullptr, nullptr, dj};
            const int rs = reg_zpass(rz, mp.reg_z);
            if (rs < 0) return rs;
            if (rs == 1) {
                CHK(set_lds_attr(k_zpass<0, false>, lds_bytes(b->nz)));
                hipLaunchKernelGGL((k_zpass<0, false>), grid, dim3(FFT_THREADS), lds_bytes(b->nz), stream, b->ax[2], b->nx, b->nxp,
                                   b->ny, 0, nb, (const int*)nullptr, (const double*)nullptr, T2, s2, (cd*)nullptr, dj);
            }
        }
        hipLaunchKernelGGL((k_yfwd<false>), dim3(nxt, mp.max_nzx, nb), dim3(FFT_THREADS), lds_bytes(b->ny), stream, b->ax[1],
                           b->nxp, b->ny, (const int*)nullptr, (const int*)nullptr, (const cd*)T2, s2, T1, s1, dj);
        hipLaunchKernelGGL((k_xfwd_gather<false>), dim3(gl, nb), dim3(FFT_THREADS), lds_bytes(b->nx, FFT_LS_X), stream, b->ax[0],
                           b->nxp, 0, (const int*)nullptr, (const int*)nullptr, (const cd*)T1, s1, (const double*)nullptr,
                           (const cd*)nullptr, (int64_t)0, (cd*)nullptr, (int64_t)0, dj);
    }
    HIPCHK(hipGetLastError());
    // nonlocal part: H psi += P (D (P' psi)) per k-block through the batched small products
    if (np_total == 0) return 0;
    std::vector<BOp> g1, g2, g3;
    size_t off = 0;
    for (BOp* o : ops) {
        const dftk_mi_kblock* kb = o->kb;
        if (kb->n_p == 0) continue;
        cd* Ppsi = Pbuf + off;
        cd* DPpsi = Pbuf + np_total + off;
        off += (size_t)kb->n_p * o->m;
        BOp a;
        a.type = BOP_ZGEMM; a.b = b; a.trans = 'C'; a.gm = kb->n_p; a.gn = o->m; a.gk = kb->n_G; a.alpha = make_double2(1.0, 0.0);
        a.A = kb->P; a.lda = kb->ldP; a.B = o->A; a.ldb = o->lda; a.beta = make_double2(0.0, 0.0); a.C = Ppsi; a.ldc = kb->n_p;
        g1.push_back(a);
        BOp d;
        d.type = BOP_APPLYD; d.b = b; d.kb = o->kb; d.m = o->m; d.A = Ppsi; d.C = DPpsi;
        g2.push_back(d);
        BOp c;
        c.type = BOP_ZGEMM; c.b = b; c.trans = 'N'; c.gm = kb->n_G; c.gn = o->m; c.gk = kb->n_p; c.alpha = make_double2(1.0, 0.0);
        c.A = kb->P; c.lda = kb->ldP; c.B = DPpsi; c.ldb = kb->n_p; c.beta = make_double2(1.0, 0.0); c.C = o->C; c.ldc = o->ldc;
        g3.push_back(c);
    }
    for (auto* g : {&g1, &g2, &g3}) {
        std::vector<BOp*> ptrs;
        for (auto& o : *g) ptrs.push_back(&o);
        if (ptrs.empty()) continue;
        const int st = batch_exec_group(ctx, stream, (*g)[0].type, ptrs);
        if (st < 0) return st;
        if (st == 1) {   // shapes outside the batched kernels: the recorded entry points (same stream, same order)
            for (auto& o : *g) {
                const int s1_ = o.type == BOP_APPLYD
                                    ? apply_D(o.kb, o.m, (const cd*)o.A, (cd*)o.C)
                                    : zgemm(b, o.trans, o.gm, o.gn, o.gk, o.alpha, (const cd*)o.A, o.lda, (const cd*)o.B, o.ldb, o.beta,
                                            (cd*)o.C, o.ldc, 0);
                if (s1_ != 0) return s1_;
            }
        }
    }
    return 0;
}

int batch_exec_density(BatchCtx* ctx, hipStream_t stream, std::vector<BOp*>& ops) {
    dftk_mi_basis* b = ops[0]->kb->basis;
    double* rho = reinterpret_cast<double*>(ops[0]->C);
    double* rho2 = reinterpret_cast<double*>(ops[0]->D);       // second cube of a two-weight pass (density + LDOS), or null
    for (BOp* o : ops)
        if (o->kb->basis != b || o->kb->sh_comm || o->C != rho || o->D != rho2) return 1;
    if (axis_generic(b->ax[0]) || axis_generic(b->ax[1]) || axis_generic(b->ax[2])) return 1;
    if (b->nz > DENS_MAXACC * FFT_TPL) return 1;
    CHK(check_lds(b));
    MultiPlan mp;
    for (BOp* o : ops) {
        const double* w = reinterpret_cast<const double*>(o->payload.data());
        // bands without weight never enter the pipeline (compute_density's occupation threshold, densities.jl:25-33)
        const double* wim = (o->flags & 1) ? w + o->m : nullptr;
        const double* w2 = (o->flags & 2) ? w + (size_t)((o->flags & 1) ? 2 : 1) * o->m : nullptr;
        for (int i = 0; i < o->m; ++i) {
            const double wi = wim ? wim[i] : w[i];
            if (w[i] == 0.0 && wi == 0.0 && (!w2 || w2[i] == 0.0)) continue;
            add_jobs(mp, o->kb, 1, reinterpret_cast<const cd*>(o->A) + (int64_t)i * o->lda, o->lda, nullptr, 0, false, w + i,
                     wim ? wim + i : nullptr, w2 ? w2 + i : nullptr);
        }
    }
    if (mp.jobs.empty()) return 0;
    const int nxt = b->nxp / FFT_L;
    const int64_t s1 = (int64_t)mp.max_lines * b->nxp, s2 = (int64_t)mp.max_nzx * b->ny * b->nxp;
    const size_t per_band = (size_t)(s1 + s2) * sizeof(cd);
    int chunk = (int)std::max<size_t>(1, std::min<size_t>(mp.jobs.size(), ((size_t)2 << 30) / per_band));
    if (chunk > 4096) chunk = 4096;
    BatchScratchScope scratch_scope(ctx);
    cd* T1 = reinterpret_cast<cd*>(batch_scratch(ctx, per_band * chunk));
    if (!T1) return DFTK_MI_EHIP;
    cd* T2 = T1 + (size_t)s1 * chunk;
    const int gl = (mp.max_lines + FFT_L - 1) / FFT_L;
    for (size_t j0 = 0; j0 < mp.jobs.size(); j0 += chunk) {
        const int nb = (int)std::min<size_t>(chunk, mp.jobs.size() - j0);
        const FftJob* dj = reinterpret_cast<const FftJob*>(batch_stage(ctx, mp.jobs.data() + j0, nb * sizeof(FftJob)));
        if (!dj) return DFTK_MI_EHIP;
        hipLaunchKernelGGL((k_xbwd_scatter<false>), dim3(gl, nb), dim3(FFT_THREADS), lds_bytes(b->nx, FFT_LS_X), stream, b->ax[0],
                           b->nxp, 0, (const int*)nullptr, (const int*)nullptr, (const cd*)nullptr, (int64_t)0, T1, s1, dj);
        hipLaunchKernelGGL((k_ybwd<false>), dim3(nxt, mp.max_nzx, nb), dim3(FFT_THREADS), lds_bytes(b->ny), stream, b->ax[1],
                           b->nxp, b->ny, (const int*)nullptr, (const int*)nullptr, (const cd*)T1, s1, T2, s2, dj);
        // small cubes: (x tile, y) columns alone are 100-200 workgroups walking ALL bands one after the other (36^3, 54 bands:
        // 178 us); the bands are dealt to up to 32 groups with a partial cube each, summed in group order by k_dens_reduce
        const int64_t cube = (int64_t)b->nx * b->ny * b->nz;
        int groups = 1;
        if ((int64_t)nxt * b->ny < 1024 && nb >= 16) groups = std::min(32, std::max(1, nb / 8));
        double* part = nullptr;
        if (groups > 1) {
            part = reinterpret_cast<double*>(batch_scratch(ctx, (size_t)(rho2 ? 2 : 1) * groups * cube * sizeof(double)));
            if (!part) return DFTK_MI_EHIP;
        }
        RegZ rz{b, stream, dim3(nxt, b->ny, groups), 0, 0, nb, nullptr, T2, s2, nullptr, nullptr, rho, dj};
        rz.part = part;
        rz.rho2 = rho2;
        const int rs = reg_zdens(rz, mp.reg_z);
        if (rs < 0) return rs;
        if (rs == 1) {
            CHK(set_lds_attr(k_zdensity<false>, lds_bytes(b->nz)));
            hipLaunchKernelGGL((k_zdensity<false>), dim3(nxt, b->ny, groups), dim3(FFT_THREADS), lds_bytes(b->nz), stream, b->ax[2], b->nx,
                               b->nxp, b->ny, 0, (const int*)nullptr, nb, (const double*)nullptr, (const double*)nullptr,
                               (const cd*)T2, s2, rho, dj, part, rho2);
        }
        if (part) {
            hipLaunchKernelGGL(k_dens_reduce, dim3((unsigned)((cube + 255) / 256)), dim3(256), 0, stream, cube, groups, (const double*)part, rho);
            if (rho2)
                hipLaunchKernelGGL(k_dens_reduce, dim3((unsigned)((cube + 255) / 256)), dim3(256), 0, stream, cube, groups,
                                   (const double*)(part + (size_t)groups * cube), rho2);
        }
    }
    HIPCHK(hipGetLastError());
    return 0;
}
