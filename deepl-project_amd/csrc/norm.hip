// HBM-bound normalisation kernels for gfx950 (bf16 data, fp32 statistics, 16-byte vector access).
//
//   GroupNorm(32)+SiLU  -- R/transvae/modules/blocks.py:33,36,60-65 ; decoder.py:93,128-129
//   RMSNorm / RMSNorm->LayerNorm-hat (token rows) -- blocks.py:179-194 ; attention.py:39-41,71-73
//
// GroupNorm statistics need a reduction over all pixels of an image, i.e. across workgroups:
// every block reduces a slab of pixels to per-channel partial sums (registers -> LDS) and writes
// them to its own row of a partials buffer; a finalize kernel adds the rows in a fixed order into
// the [B][C][2] fp32 result (no atomics: bit-reproducible); groups are folded from channels by
// the consumer.  The same reduction skeleton serves the backward sums.
#include "common.h"

namespace {

constexpr int GN_THREADS = 256;
constexpr int GN_PIX_PER_BLOCK = 512;

// Streaming accesses of the GroupNorm passes.  A ResBlock tensor is 0.4-1.6 GB at micro-batch 64, read or written once per pass
// and far beyond the 256 MiB Infinity Cache: with the NON-TEMPORAL policy on loads and stores the four kernels move 5.7-5.8
// TB/s instead of 5.1-5.4 (tools/probes/gn_bench.py, two interleaved repetitions: 6.167 -> 5.765 ms over the passes of one
// 256 x 256 and one 128 x 128 GroupNorm forward + backward; loads alone 6.04; a deeper unroll nothing).  NT is a kernel template
// parameter chosen per launch by the tensor's size (gn_nt): small tensors keep the default policy -- their consumer finds
// them in the cache.  TV_GN_NT_MIB (compile time): threshold in MiB, -1 = never.
#ifndef TV_GN_NT_MIB
#define TV_GN_NT_MIB 128
#endif
template <bool NT>
__device__ __forceinline__ bf16x8 gn_ld(const bf16* p) {
    if constexpr (NT) return __builtin_nontemporal_load((const bf16x8*)p);
    else return *(const bf16x8*)p;
}
template <bool NT>
__device__ __forceinline__ void gn_st(bf16* p, const bf16x8& v) {
    if constexpr (NT) __builtin_nontemporal_store(v, (bf16x8*)p);
    else *(bf16x8*)p = v;
}
static bool gn_nt(int batch, int hw, int C) { return TV_GN_NT_MIB >= 0 && (long long)batch * hw * C * 2 >= ((long long)TV_GN_NT_MIB << 20); }

// ---- GroupNorm: per-(b,c) sum / sum of squares ABOUT A PIVOT --------------------------------------
// The one-pass form var = E[x^2] - mean^2 cancels when |mean| >> std (a residual stream whose mean has drifted; up to
// ~400 K elements per group in fp32).  All sums are therefore taken about the pivot piv[b][c] = x[b, pixel 0, c] (the
// same for every block of the image, so the partials still add in a fixed order), and gn_prepare merges the channels
// of a group with Chan's parallel (mean, M2) update -- the result matches a two-pass / Welford computation like
// ATen's group_norm to fp32 rounding.
template <bool NT>
__global__ __launch_bounds__(GN_THREADS) void gn_stats_kernel(const bf16* __restrict__ x, float* __restrict__ part,
                                                              int hw, int C) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* s_acc = (float*)smem;  // [rows][C][2]
    const int nch = C >> 3;
    const int b = blockIdx.y;
    const int p0 = blockIdx.x * GN_PIX_PER_BLOCK;
    const int p1 = min(hw, p0 + GN_PIX_PER_BLOCK);
    const int rows = GN_THREADS / nch;  // pixel lanes per sweep
    const int chunk = threadIdx.x % nch, prow = threadIdx.x / nch;
    if (prow < rows) {
        float s[8], q[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) s[e] = q[e] = 0.f;
        const bf16* base = x + ((size_t)b * hw) * C + chunk * 8;
        const bf16x8 pv = *(const bf16x8*)base;   // pivot: pixel 0 of this image
#pragma unroll 2
        for (int p = p0 + prow; p < p1; p += rows) {
            const bf16x8 v = gn_ld<NT>(base + (size_t)p * C);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float f = (float)v[e] - (float)pv[e];
                s[e] += f;
                q[e] = fmaf(f, f, q[e]);
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            s_acc[(prow * C + chunk * 8 + e) * 2 + 0] = s[e];
            s_acc[(prow * C + chunk * 8 + e) * 2 + 1] = q[e];
        }
    }
    __syncthreads();
    // fixed-order sum over the pixel lanes -> this block's partial (no atomics: bit-reproducible)
    float* dst = part + ((size_t)b * gridDim.x + blockIdx.x) * C * 2;
    for (int i = threadIdx.x; i < 2 * C; i += GN_THREADS) {
        float a = 0.f;
        for (int r = 0; r < rows; ++r) a += s_acc[r * C * 2 + i];
        dst[i] = a;
    }
}

// out[b][i] = sum over blocks of part[b][blk][i], in block order (deterministic)
__global__ __launch_bounds__(256) void gn_finalize_kernel(const float* __restrict__ part, float* __restrict__ out, int nblk, int n) {
    const int b = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float* src = part + (size_t)b * nblk * n + i;
    float a = 0.f;
    for (int k = 0; k < nblk; ++k) a += src[(size_t)k * n];
    out[(size_t)b * n + i] = a;
}

// per-channel scale/shift of image b from the channel sums:  h = x*scale[c] + shift[c]
__device__ __forceinline__ void gn_prepare(const float* stats_b, const bf16* x_b, const float* gamma, const float* beta,
                                           float* s_scale, float* s_shift, float* s_mean, float* s_rstd, int hw, int C, int G,
                                           float eps) {
    const int cpg = C / G;
    const float inv_hw = 1.0f / (float)hw;
    for (int g = threadIdx.x; g < G; g += blockDim.x) {
        // channel c: mean_c = piv_c + S_c / hw,  M2_c = Q_c - S_c^2 / hw  (sums about the pivot: no large cancellation);
        // group: mean = avg_c mean_c,  M2 = sum_c M2_c + hw * sum_c (mean_c - mean)^2   (Chan et al., equal counts)
        float msum = 0.f, m2 = 0.f;
        for (int c = g * cpg; c < (g + 1) * cpg; ++c) {
            const float S = stats_b[c * 2], Q = stats_b[c * 2 + 1];
            msum += (float)x_b[c] + S * inv_hw;
            m2 += fmaxf(Q - S * S * inv_hw, 0.f);
        }
        const float mean = msum / (float)cpg;
        float spread = 0.f;
        for (int c = g * cpg; c < (g + 1) * cpg; ++c) {
            const float d = (float)x_b[c] + stats_b[c * 2] * inv_hw - mean;
            spread = fmaf(d, d, spread);
        }
        const float var = (m2 + (float)hw * spread) / ((float)cpg * (float)hw);
        s_mean[g] = mean;
        s_rstd[g] = rsqrtf(var + eps);
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const int g = c / cpg;
        const float sc = s_rstd[g] * gamma[c];
        s_scale[c] = sc;
        s_shift[c] = beta[c] - s_mean[g] * sc;
    }
    __syncthreads();
}

// FROM_MR: the statistics are READ from mr (the mean / rstd a forward call wrote) instead of being derived from the channel
// sums -- the recompute of silu(gn(x)) in the backward pass of a checkpointed ResBlock: same scale / shift bits, same output.
template <bool FROM_MR, bool NT>
__global__ __launch_bounds__(GN_THREADS) void gn_silu_fwd_kernel(const bf16* __restrict__ x, const float* __restrict__ stats,
                                                                 const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                 float* __restrict__ mr, bf16* __restrict__ y, int hw, int C, int G,
                                                                 float eps) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* s_scale = (float*)smem;
    float* s_shift = s_scale + C;
    float* s_mean = s_shift + C;
    float* s_rstd = s_mean + G;
    const int b = blockIdx.y;
    if constexpr (FROM_MR) {
        const int cpg = C / G;
        for (int c = threadIdx.x; c < C; c += GN_THREADS) {
            const int g = c / cpg;
            const float sc = mr[((size_t)b * G + g) * 2 + 1] * gamma[c];
            s_scale[c] = sc;
            s_shift[c] = beta[c] - mr[((size_t)b * G + g) * 2] * sc;
        }
        __syncthreads();
    } else {
        gn_prepare(stats + (size_t)b * C * 2, x + (size_t)b * hw * C, gamma, beta, s_scale, s_shift, s_mean, s_rstd, hw, C, G, eps);
        if (blockIdx.x == 0)
            for (int g = threadIdx.x; g < G; g += GN_THREADS) {
                mr[((size_t)b * G + g) * 2] = s_mean[g];
                mr[((size_t)b * G + g) * 2 + 1] = s_rstd[g];
            }
    }
    // one 16-byte channel chunk per thread for the whole slab: scale / shift in registers
    const int nch = C >> 3;
    const int rows = GN_THREADS / nch;
    const int chunk = threadIdx.x % nch, prow = threadIdx.x / nch;
    if (prow >= rows) return;
    float sc[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        sc[e] = s_scale[chunk * 8 + e];
        sh[e] = s_shift[chunk * 8 + e];
    }
    const int p0 = blockIdx.x * GN_PIX_PER_BLOCK;
    const int p1 = min(hw, p0 + GN_PIX_PER_BLOCK);
    const size_t img = (size_t)b * hw;
#pragma unroll 2
    for (int px = p0 + prow; px < p1; px += rows) {
        const size_t off = (img + px) * C + chunk * 8;
        const bf16x8 v = gn_ld<NT>(x + off);
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (bf16)tv_silu(fmaf((float)v[e], sc[e], sh[e]));
        gn_st<NT>(y + off, o);
    }
}

// backward pass 1: red[b][c] += (sum dh, sum dh*xhat),  dh = dy * silu'(h)
template <bool NT>
__global__ __launch_bounds__(GN_THREADS) void gn_silu_bwd_reduce_kernel(const bf16* __restrict__ x, const bf16* __restrict__ dy,
                                                                        const float* __restrict__ mr, const float* __restrict__ gamma,
                                                                        const float* __restrict__ beta, float* __restrict__ part, int hw,
                                                                        int C, int G) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* s_acc = (float*)smem;  // [rows][C][2]
    const int nch = C >> 3, cpg = C / G;
    const int b = blockIdx.y;
    const int p0 = blockIdx.x * GN_PIX_PER_BLOCK;
    const int p1 = min(hw, p0 + GN_PIX_PER_BLOCK);
    const int rows = GN_THREADS / nch;
    const int chunk = threadIdx.x % nch, prow = threadIdx.x / nch;
    if (prow < rows) {
        float mean[8], rstd[8], ga[8], be[8], s[8], q[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = chunk * 8 + e, g = c / cpg;
            mean[e] = mr[((size_t)b * G + g) * 2];
            rstd[e] = mr[((size_t)b * G + g) * 2 + 1];
            ga[e] = gamma[c];
            be[e] = beta[c];
            s[e] = q[e] = 0.f;
        }
        const size_t base = ((size_t)b * hw) * C + chunk * 8;
#pragma unroll 2
        for (int p = p0 + prow; p < p1; p += rows) {
            const bf16x8 xv = gn_ld<NT>(x + base + (size_t)p * C);
            const bf16x8 gv = gn_ld<NT>(dy + base + (size_t)p * C);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float xh = ((float)xv[e] - mean[e]) * rstd[e];
                const float h = fmaf(xh, ga[e], be[e]);
                const float dh = (float)gv[e] * tv_silu_grad(h);
                s[e] += dh;
                q[e] = fmaf(dh, xh, q[e]);
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            s_acc[(prow * C + chunk * 8 + e) * 2 + 0] = s[e];
            s_acc[(prow * C + chunk * 8 + e) * 2 + 1] = q[e];
        }
    }
    __syncthreads();
    // fixed-order sum over the pixel lanes -> this block's partial (no atomics: bit-reproducible)
    float* dst = part + ((size_t)b * gridDim.x + blockIdx.x) * C * 2;
    for (int i = threadIdx.x; i < 2 * C; i += GN_THREADS) {
        float a = 0.f;
        for (int r = 0; r < rows; ++r) a += s_acc[r * C * 2 + i];
        dst[i] = a;
    }
}

// backward pass 2: dx = rstd*(dh*gamma - (S1_g + xhat*S2_g)/n) (+ dres)
template <bool NT>
__global__ __launch_bounds__(GN_THREADS) void gn_silu_bwd_apply_kernel(const bf16* __restrict__ x, const bf16* __restrict__ dy,
                                                                       const bf16* __restrict__ dres, const float* __restrict__ mr,
                                                                       const float* __restrict__ red, const float* __restrict__ gamma,
                                                                       const float* __restrict__ beta, bf16* __restrict__ dx,
                                                                       float* __restrict__ dgamma, float* __restrict__ dbeta, int hw, int C,
                                                                       int G) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* s_mean = (float*)smem;   // [C] (expanded per channel)
    float* s_rstd = s_mean + C;     // [C]
    float* s_ga = s_rstd + C;       // [C]
    float* s_be = s_ga + C;         // [C]
    float* s_m1 = s_be + C;         // [C]  S1_g/n per channel
    float* s_m2 = s_m1 + C;         // [C]  S2_g/n per channel
    float* s_g1 = s_m2 + C;         // [G]
    float* s_g2 = s_g1 + G;         // [G]
    const int cpg = C / G;
    const int b = blockIdx.y;
    const float* red_b = red + (size_t)b * C * 2;
    for (int g = threadIdx.x; g < G; g += GN_THREADS) {
        float a1 = 0.f, a2 = 0.f;
        for (int c = g * cpg; c < (g + 1) * cpg; ++c) {
            a1 = fmaf(gamma[c], red_b[c * 2], a1);
            a2 = fmaf(gamma[c], red_b[c * 2 + 1], a2);
        }
        const float inv_n = 1.0f / ((float)cpg * (float)hw);
        s_g1[g] = a1 * inv_n;
        s_g2[g] = a2 * inv_n;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += GN_THREADS) {
        const int g = c / cpg;
        s_mean[c] = mr[((size_t)b * G + g) * 2];
        s_rstd[c] = mr[((size_t)b * G + g) * 2 + 1];
        s_ga[c] = gamma[c];
        s_be[c] = beta[c];
        s_m1[c] = s_g1[g];
        s_m2[c] = s_g2[g];
        if (blockIdx.x == 0) {  // parameter gradients: one block per image adds its sums
            atomicAdd(dbeta + c, red_b[c * 2]);
            atomicAdd(dgamma + c, red_b[c * 2 + 1]);
        }
    }
    __syncthreads();
    // Each thread keeps one 16-byte channel chunk for the whole slab (the mapping of gn_stats), so the per-channel terms
    // live in registers:  h = x*A + B,  dx = dh*A - x*Cc + D  with A = rstd*gamma, B = beta - mean*A,
    // Cc = rstd^2 * S2_g/n, D = mean*Cc - rstd * S1_g/n.  (Was: six LDS reads per element and a 64-bit modulo per vector;
    // 4.5 TB/s of the ~6.3 TB/s the other norm kernels reach.)
    const int nch = C >> 3;
    const int rows = GN_THREADS / nch;
    const int chunk = threadIdx.x % nch, prow = threadIdx.x / nch;
    if (prow >= rows) return;
    float cA[8], cB[8], cC[8], cD[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int c = chunk * 8 + e;
        const float rs = s_rstd[c], mu = s_mean[c];
        cA[e] = rs * s_ga[c];
        cB[e] = s_be[c] - mu * cA[e];
        cC[e] = rs * rs * s_m2[c];
        cD[e] = mu * cC[e] - rs * s_m1[c];
    }
    const int p0 = blockIdx.x * GN_PIX_PER_BLOCK;
    const int p1 = min(hw, p0 + GN_PIX_PER_BLOCK);
    const size_t img = (size_t)b * hw;
#pragma unroll 2
    for (int px = p0 + prow; px < p1; px += rows) {
        const size_t off = (img + px) * C + chunk * 8;
        const bf16x8 xv = gn_ld<NT>(x + off);
        const bf16x8 gv = gn_ld<NT>(dy + off);
        bf16x8 rv = {0, 0, 0, 0, 0, 0, 0, 0};
        if (dres) rv = gn_ld<NT>(dres + off);
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float xf = (float)xv[e];
            const float h = fmaf(xf, cA[e], cB[e]);
            const float dh = (float)gv[e] * tv_silu_grad(h);
            const float d = fmaf(dh, cA[e], fmaf(-xf, cC[e], cD[e]));
            o[e] = (bf16)(d + (float)rv[e]);
        }
        gn_st<NT>(dx + off, o);
    }
}

// ---- token-row norms -------------------------------------------------------------------------
constexpr int RN_MAXCH = 5;  // 16-byte chunks per lane: C <= 64*8*5 = 2560
// KCH = chunks per lane the row actually needs (1: C <= 512, 2: <= 1024, 3: <= 1536, 5: <= 2560).  Round 3: the loops used to run
// over all RN_MAXCH chunks whatever C was -- five times the arithmetic on a 384-channel row (stage 2 holds half of the row-norm
// bytes), which made these HBM-rate kernels ALU-bound: rownorm_bwd<1> moved 0.8 GB in 0.50 ms on the 384-channel rows.

template <int MODE, int KCH>
__global__ __launch_bounds__(256) void rownorm_fwd_kernel(const bf16* __restrict__ x, const float* __restrict__ w, bf16* __restrict__ y,
                                                          int T, int C, float eps_rms, float eps_ln) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int nch = C >> 3;
    const float inv_c = 1.0f / (float)C;
    float wv[KCH][8];
    if constexpr (MODE == 1) {
#pragma unroll
        for (int k = 0; k < KCH; ++k) {
            const int ch = lane + 64 * k;
#pragma unroll
            for (int e = 0; e < 8; ++e) wv[k][e] = (ch < nch) ? w[ch * 8 + e] : 0.f;
        }
    }
    for (int row = blockIdx.x * 4 + wave; row < T; row += gridDim.x * 4) {
        const bf16* xr = x + (size_t)row * C;
        float v[KCH][8];
        float ss = 0.f;
#pragma unroll
        for (int k = 0; k < KCH; ++k) {
            const int ch = lane + 64 * k;
            bf16x8 t = {0, 0, 0, 0, 0, 0, 0, 0};
            if (ch < nch) t = *(const bf16x8*)(xr + ch * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                v[k][e] = (float)t[e];
                ss = fmaf(v[k][e], v[k][e], ss);
            }
        }
        ss = tv_wave_sum(ss);
        const float r = rsqrtf(ss * inv_c + eps_rms);
        float mu = 0.f, s = 1.f;
        if constexpr (MODE == 1) {
            float su = 0.f;
#pragma unroll
            for (int k = 0; k < KCH; ++k)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    v[k][e] = v[k][e] * r * wv[k][e];
                    su += v[k][e];
                }
            mu = tv_wave_sum(su) * inv_c;
            float sv = 0.f;
#pragma unroll
            for (int k = 0; k < KCH; ++k) {
                const int ch = lane + 64 * k;
                if (ch < nch) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float d = v[k][e] - mu;
                        sv = fmaf(d, d, sv);
                    }
                }
            }
            s = rsqrtf(tv_wave_sum(sv) * inv_c + eps_ln);
        }
        bf16* yr = y + (size_t)row * C;
#pragma unroll
        for (int k = 0; k < KCH; ++k) {
            const int ch = lane + 64 * k;
            if (ch < nch) {
                bf16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (MODE == 1) ? (bf16)((v[k][e] - mu) * s) : (bf16)(v[k][e] * r);
                *(bf16x8*)(yr + ch * 8) = o;
            }
        }
    }
}

template <int MODE, int KCH>
__global__ __launch_bounds__(256) void rownorm_bwd_kernel(const bf16* __restrict__ x, const float* __restrict__ w, const bf16* __restrict__ dy,
                                                          const bf16* __restrict__ dres, bf16* __restrict__ dx, float* __restrict__ dw, int T,
                                                          int C, float eps_rms, float eps_ln) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* s_dw = (float*)smem;  // [C] (mode 1)
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int nch = C >> 3;
    const float inv_c = 1.0f / (float)C;
    float wv[KCH][8], dwv[KCH][8];
    if constexpr (MODE == 1) {
        for (int i = threadIdx.x; i < C; i += 256) s_dw[i] = 0.f;
#pragma unroll
        for (int k = 0; k < KCH; ++k) {
            const int ch = lane + 64 * k;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                wv[k][e] = (ch < nch) ? w[ch * 8 + e] : 0.f;
                dwv[k][e] = 0.f;
            }
        }
        __syncthreads();
    }
    for (int row = blockIdx.x * 4 + wave; row < T; row += gridDim.x * 4) {
        const bf16* xr = x + (size_t)row * C;
        const bf16* gr = dy + (size_t)row * C;
        float xh[KCH][8], g[KCH][8];
        float ss = 0.f;
#pragma unroll
        for (int k = 0; k < KCH; ++k) {
            const int ch = lane + 64 * k;
            bf16x8 t = {0, 0, 0, 0, 0, 0, 0, 0}, u = {0, 0, 0, 0, 0, 0, 0, 0};
            if (ch < nch) {
                t = *(const bf16x8*)(xr + ch * 8);
                u = *(const bf16x8*)(gr + ch * 8);
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                xh[k][e] = (float)t[e];
                g[k][e] = (float)u[e];
                ss = fmaf(xh[k][e], xh[k][e], ss);
            }
        }
        ss = tv_wave_sum(ss);
        const float r = rsqrtf(ss * inv_c + eps_rms);
#pragma unroll
        for (int k = 0; k < KCH; ++k)
#pragma unroll
            for (int e = 0; e < 8; ++e) xh[k][e] *= r;  // xhat = x * r
        if constexpr (MODE == 1) {
            // u = xhat*w ; y = (u-mu)*s ; du = s*(dy - mean(dy) - y*mean(dy*y))
            float su = 0.f;
#pragma unroll
            for (int k = 0; k < KCH; ++k)
#pragma unroll
                for (int e = 0; e < 8; ++e) su += xh[k][e] * wv[k][e];
            const float mu = tv_wave_sum(su) * inv_c;
            float sv = 0.f;
#pragma unroll
            for (int k = 0; k < KCH; ++k) {
                const int ch = lane + 64 * k;
                if (ch < nch) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float d = xh[k][e] * wv[k][e] - mu;
                        sv = fmaf(d, d, sv);
                    }
                }
            }
            const float s = rsqrtf(tv_wave_sum(sv) * inv_c + eps_ln);
            float a1 = 0.f, a2 = 0.f;
#pragma unroll
            for (int k = 0; k < KCH; ++k) {
                const int ch = lane + 64 * k;
                if (ch < nch) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float yv = (xh[k][e] * wv[k][e] - mu) * s;
                        a1 += g[k][e];
                        a2 = fmaf(g[k][e], yv, a2);
                    }
                }
            }
            a1 = tv_wave_sum(a1) * inv_c;
            a2 = tv_wave_sum(a2) * inv_c;
#pragma unroll
            for (int k = 0; k < KCH; ++k) {
                const int ch = lane + 64 * k;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float yv = (xh[k][e] * wv[k][e] - mu) * s;
                    const float du = (ch < nch) ? s * (g[k][e] - a1 - yv * a2) : 0.f;
                    dwv[k][e] = fmaf(du, xh[k][e], dwv[k][e]);
                    g[k][e] = du * wv[k][e];  // gradient w.r.t. xhat
                }
            }
        }
        // dx = r * (g - xhat * mean(g*xhat))
        float a3 = 0.f;
#pragma unroll
        for (int k = 0; k < KCH; ++k)
#pragma unroll
            for (int e = 0; e < 8; ++e) a3 = fmaf(g[k][e], xh[k][e], a3);
        a3 = tv_wave_sum(a3) * inv_c;
        bf16* dr = dx + (size_t)row * C;
#pragma unroll
        for (int k = 0; k < KCH; ++k) {
            const int ch = lane + 64 * k;
            if (ch < nch) {
                bf16x8 rv = {0, 0, 0, 0, 0, 0, 0, 0};
                if (dres) rv = *(const bf16x8*)(dres + (size_t)row * C + ch * 8);
                bf16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (bf16)(r * (g[k][e] - xh[k][e] * a3) + (float)rv[e]);
                *(bf16x8*)(dr + ch * 8) = o;
            }
        }
    }
    if constexpr (MODE == 1) {
#pragma unroll
        for (int k = 0; k < KCH; ++k) {
            const int ch = lane + 64 * k;
            if (ch < nch) {
#pragma unroll
                for (int e = 0; e < 8; ++e) atomicAdd(&s_dw[ch * 8 + e], dwv[k][e]);
            }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < C; i += 256) atomicAdd(dw + i, s_dw[i]);
    }
}

}  // namespace

// ---- C ABI ------------------------------------------------------------------------------------
static int gn_check(const char* name, int batch, int hw, int C, int G) {
    if (batch <= 0 || hw <= 0 || C <= 0 || C % 8 != 0 || G <= 0 || C % G != 0 || (C >> 3) > GN_THREADS) {
        tv_set_error("%s: bad shape batch=%d hw=%d C=%d G=%d (C %% 8 == 0, C %% G == 0, C <= 2048 required)", name, batch, hw, C, G);
        return TV_ERR_ARG;
    }
    return TV_OK;
}

static size_t gn_lds_bytes(int C) { return (size_t)(GN_THREADS / (C >> 3)) * C * 2 * sizeof(float); }

extern "C" long long tv_gn_partial_count(int batch, int hw, int C) {
    return (long long)batch * tv_cdiv(hw, GN_PIX_PER_BLOCK) * C * 2;
}

extern "C" int tv_gn_stats(const void* x, float* stats, float* partials, int batch, int hw, int C, void* stream) {
    if (gn_check("tv_gn_stats", batch, hw, C, 1)) return TV_ERR_ARG;
    TV_CHECK_ARG(x && stats && partials, "tv_gn_stats: null pointer");
    dim3 grid(tv_cdiv(hw, GN_PIX_PER_BLOCK), batch);
    if (gn_nt(batch, hw, C)) hipLaunchKernelGGL(gn_stats_kernel<true>, grid, dim3(GN_THREADS), gn_lds_bytes(C), (hipStream_t)stream, (const bf16*)x, partials, hw, C);
    else hipLaunchKernelGGL(gn_stats_kernel<false>, grid, dim3(GN_THREADS), gn_lds_bytes(C), (hipStream_t)stream, (const bf16*)x, partials, hw, C);
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(tv_cdiv(2 * C, 256), batch), dim3(256), 0, (hipStream_t)stream, (const float*)partials, stats,
                       (int)grid.x, 2 * C);
    TV_CHECK_LAUNCH("tv_gn_stats");
    return TV_OK;
}

extern "C" int tv_gn_silu_fwd(const void* x, const float* stats, const float* gamma, const float* beta, float* mr, void* y,
                              int batch, int hw, int C, int G, float eps, void* stream) {
    if (gn_check("tv_gn_silu_fwd", batch, hw, C, G)) return TV_ERR_ARG;
    TV_CHECK_ARG(x && stats && gamma && beta && mr && y, "tv_gn_silu_fwd: null pointer");
    dim3 grid(tv_cdiv(hw, GN_PIX_PER_BLOCK), batch);
    if (gn_nt(batch, hw, C))
        hipLaunchKernelGGL((gn_silu_fwd_kernel<false, true>), grid, dim3(GN_THREADS), (2 * C + 2 * G) * sizeof(float), (hipStream_t)stream,
                           (const bf16*)x, stats, gamma, beta, mr, (bf16*)y, hw, C, G, eps);
    else
        hipLaunchKernelGGL((gn_silu_fwd_kernel<false, false>), grid, dim3(GN_THREADS), (2 * C + 2 * G) * sizeof(float), (hipStream_t)stream,
                           (const bf16*)x, stats, gamma, beta, mr, (bf16*)y, hw, C, G, eps);
    TV_CHECK_LAUNCH("tv_gn_silu_fwd");
    return TV_OK;
}

extern "C" int tv_gn_silu_apply(const void* x, const float* mr, const float* gamma, const float* beta, void* y, int batch, int hw,
                                int C, int G, void* stream) {
    if (gn_check("tv_gn_silu_apply", batch, hw, C, G)) return TV_ERR_ARG;
    TV_CHECK_ARG(x && mr && gamma && beta && y, "tv_gn_silu_apply: null pointer");
    dim3 grid(tv_cdiv(hw, GN_PIX_PER_BLOCK), batch);
    if (gn_nt(batch, hw, C))
        hipLaunchKernelGGL((gn_silu_fwd_kernel<true, true>), grid, dim3(GN_THREADS), (2 * C + 2 * G) * sizeof(float), (hipStream_t)stream,
                           (const bf16*)x, nullptr, gamma, beta, const_cast<float*>(mr), (bf16*)y, hw, C, G, 0.f);
    else
        hipLaunchKernelGGL((gn_silu_fwd_kernel<true, false>), grid, dim3(GN_THREADS), (2 * C + 2 * G) * sizeof(float), (hipStream_t)stream,
                           (const bf16*)x, nullptr, gamma, beta, const_cast<float*>(mr), (bf16*)y, hw, C, G, 0.f);
    TV_CHECK_LAUNCH("tv_gn_silu_apply");
    return TV_OK;
}

extern "C" int tv_gn_silu_bwd_reduce(const void* x, const void* dy, const float* mr, const float* gamma, const float* beta,
                                     float* red, float* partials, int batch, int hw, int C, int G, void* stream) {
    if (gn_check("tv_gn_silu_bwd_reduce", batch, hw, C, G)) return TV_ERR_ARG;
    TV_CHECK_ARG(x && dy && mr && gamma && beta && red && partials, "tv_gn_silu_bwd_reduce: null pointer");
    dim3 grid(tv_cdiv(hw, GN_PIX_PER_BLOCK), batch);
    if (gn_nt(batch, hw, C))
        hipLaunchKernelGGL(gn_silu_bwd_reduce_kernel<true>, grid, dim3(GN_THREADS), gn_lds_bytes(C), (hipStream_t)stream, (const bf16*)x,
                           (const bf16*)dy, mr, gamma, beta, partials, hw, C, G);
    else
        hipLaunchKernelGGL(gn_silu_bwd_reduce_kernel<false>, grid, dim3(GN_THREADS), gn_lds_bytes(C), (hipStream_t)stream, (const bf16*)x,
                           (const bf16*)dy, mr, gamma, beta, partials, hw, C, G);
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(tv_cdiv(2 * C, 256), batch), dim3(256), 0, (hipStream_t)stream, (const float*)partials, red,
                       (int)grid.x, 2 * C);
    TV_CHECK_LAUNCH("tv_gn_silu_bwd_reduce");
    return TV_OK;
}

extern "C" int tv_gn_silu_bwd_apply(const void* x, const void* dy, const void* dres, const float* mr, const float* red,
                                    const float* gamma, const float* beta, void* dx, float* dgamma, float* dbeta, int batch, int hw,
                                    int C, int G, void* stream) {
    if (gn_check("tv_gn_silu_bwd_apply", batch, hw, C, G)) return TV_ERR_ARG;
    TV_CHECK_ARG(x && dy && mr && red && gamma && beta && dx && dgamma && dbeta, "tv_gn_silu_bwd_apply: null pointer");
    dim3 grid(tv_cdiv(hw, GN_PIX_PER_BLOCK), batch);
    if (gn_nt(batch, hw, C))
        hipLaunchKernelGGL(gn_silu_bwd_apply_kernel<true>, grid, dim3(GN_THREADS), (6 * C + 2 * G) * sizeof(float), (hipStream_t)stream,
                           (const bf16*)x, (const bf16*)dy, (const bf16*)dres, mr, red, gamma, beta, (bf16*)dx, dgamma, dbeta, hw, C, G);
    else
        hipLaunchKernelGGL(gn_silu_bwd_apply_kernel<false>, grid, dim3(GN_THREADS), (6 * C + 2 * G) * sizeof(float), (hipStream_t)stream,
                           (const bf16*)x, (const bf16*)dy, (const bf16*)dres, mr, red, gamma, beta, (bf16*)dx, dgamma, dbeta, hw, C, G);
    TV_CHECK_LAUNCH("tv_gn_silu_bwd_apply");
    return TV_OK;
}

extern "C" int tv_rownorm_fwd(const void* x, const float* w, void* y, int T, int C, int mode, float eps_rms, float eps_ln,
                              void* stream) {
    TV_CHECK_ARG(x && y && T > 0, "tv_rownorm_fwd: null pointer / empty");
    TV_CHECK_ARG(C > 0 && C % 8 == 0 && C <= 64 * 8 * RN_MAXCH, "tv_rownorm_fwd: C=%d must be a multiple of 8 and <= 2560", C);
    TV_CHECK_ARG(mode == 0 || (mode == 1 && w), "tv_rownorm_fwd: mode %d (mode 1 needs w)", mode);
    const int grid = min(tv_cdiv(T, 4), 256 * 8);
    const int kch = tv_cdiv(C >> 3, 64);      // 16-byte chunks per lane
#define TV_RN_FWD(M, K) hipLaunchKernelGGL((rownorm_fwd_kernel<M, K>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16*)x, w, (bf16*)y, T, C, eps_rms, eps_ln)
    if (mode == 0) {
        if (kch <= 1) TV_RN_FWD(0, 1); else if (kch == 2) TV_RN_FWD(0, 2); else if (kch == 3) TV_RN_FWD(0, 3); else TV_RN_FWD(0, 5);
    } else {
        if (kch <= 1) TV_RN_FWD(1, 1); else if (kch == 2) TV_RN_FWD(1, 2); else if (kch == 3) TV_RN_FWD(1, 3); else TV_RN_FWD(1, 5);
    }
#undef TV_RN_FWD
    TV_CHECK_LAUNCH("tv_rownorm_fwd");
    return TV_OK;
}

extern "C" int tv_rownorm_bwd(const void* x, const float* w, const void* dy, const void* dres, void* dx, float* dw, int T, int C,
                              int mode, float eps_rms, float eps_ln, void* stream) {
    TV_CHECK_ARG(x && dy && dx && T > 0, "tv_rownorm_bwd: null pointer / empty");
    TV_CHECK_ARG(C > 0 && C % 8 == 0 && C <= 64 * 8 * RN_MAXCH, "tv_rownorm_bwd: C=%d must be a multiple of 8 and <= 2560", C);
    TV_CHECK_ARG(mode == 0 || (mode == 1 && w && dw), "tv_rownorm_bwd: mode %d (mode 1 needs w and dw)", mode);
    const int grid = min(tv_cdiv(T, 4), 256 * 4);
    const int kch = tv_cdiv(C >> 3, 64);      // 16-byte chunks per lane
#define TV_RN_BWD(M, K) hipLaunchKernelGGL((rownorm_bwd_kernel<M, K>), dim3(grid), dim3(256), (M) ? C * sizeof(float) : 0, (hipStream_t)stream, \
                                           (const bf16*)x, w, (const bf16*)dy, (const bf16*)dres, (bf16*)dx, dw, T, C, eps_rms, eps_ln)
    if (mode == 0) {
        if (kch <= 1) TV_RN_BWD(0, 1); else if (kch == 2) TV_RN_BWD(0, 2); else if (kch == 3) TV_RN_BWD(0, 3); else TV_RN_BWD(0, 5);
    } else {
        if (kch <= 1) TV_RN_BWD(1, 1); else if (kch == 2) TV_RN_BWD(1, 2); else if (kch == 3) TV_RN_BWD(1, 3); else TV_RN_BWD(1, 5);
    }
#undef TV_RN_BWD
    TV_CHECK_LAUNCH("tv_rownorm_bwd");
    return TV_OK;
}
