// Parameter folds of the TransVAE block, forward and backward as one launch each (train-step glue).
//
//   Wf[r][c] = W[r][c] * gamma[c]          bf[r] = sum_c W[r][c] * beta[c]        (beta may be NULL: no bias term)
//
// is how a LayerNorm / RMSNorm affine in front of a Linear disappears into the projection:
//   Linear(LayerNorm-hat(x) * gamma + beta) = x-hat (W gamma)^T + W beta       (R/transvae/modules/attention.py:39-48,71-78)
//   Linear(RMSNorm-hat(x) * w_rms)          = x-hat (W w_rms)^T                (R/transvae/modules/blocks.py:146-149, conv.py:85)
// The PyTorch formulation costs ~40 launch-bound kernels per attention block and micro-batch (mul, gemv, cat and their
// autograd twins: 1.2 % of the train step, profiles/r02_rocprof_kernel_stats.csv of the previous build); here 2.
// Backward, one pass over the matrices:
//   dW[r][c] = dWf[r][c] gamma[c] + dbf[r] beta[c]     dgamma[c] = sum_r dWf[r][c] W[r][c]     dbeta[c] = sum_r dbf[r] W[r][c]
// Column sums are taken in a fixed order (one block owns 64 columns): bit-reproducible.
#include "common.h"

namespace {

// forward: one block per 4 rows, float4 along the columns (C % 4 == 0 on this path; scalar tail kernel otherwise)
__global__ __launch_bounds__(256) void fold_fwd_kernel(const float* __restrict__ W, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       float* __restrict__ Wf, float* __restrict__ bf, int R, int Cc) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = blockIdx.x * 4 + wave;          // one wave per row
    if (r >= R) return;
    const float* w = W + (size_t)r * Cc;
    float* o = Wf + (size_t)r * Cc;
    float acc = 0.f;
    if ((Cc & 3) == 0) {
        const int n4 = Cc >> 2;
        for (int c4 = lane; c4 < n4; c4 += 64) {
            const f32x4 v = *(const f32x4*)(w + c4 * 4), g = *(const f32x4*)(gamma + c4 * 4);
            *(f32x4*)(o + c4 * 4) = v * g;
            if (beta) {
                const f32x4 b = *(const f32x4*)(beta + c4 * 4);
                acc += v[0] * b[0] + v[1] * b[1] + v[2] * b[2] + v[3] * b[3];
            }
        }
    } else {
        for (int c = lane; c < Cc; c += 64) {
            const float v = w[c];
            o[c] = v * gamma[c];
            if (beta) acc = fmaf(v, beta[c], acc);
        }
    }
    if (bf) {
        acc = tv_wave_sum(acc);
        if (lane == 0) bf[r] = acc;
    }
}

// backward: block = 64 columns (16 threads x float4) x 16 row lanes over a chunk of FOLD_ROWS rows; partial column sums per
// row chunk, added in chunk order by fold_finalize_kernel (bit-reproducible; a [6144, 1536] projection is 24 x 96 blocks)
constexpr int FOLD_ROWS = 64;
__global__ __launch_bounds__(256) void fold_bwd_kernel(const float* __restrict__ dWf, const float* __restrict__ dbf, const float* __restrict__ W,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ dW,
                                                       float* __restrict__ part, int R, int Cc) {
    __shared__ float s_g[16][64], s_b[16][64];
    const int cg = threadIdx.x & 15, ry = threadIdx.x >> 4;
    const int c = blockIdx.x * 64 + cg * 4;
    const int r0 = blockIdx.y * FOLD_ROWS, r1 = min(R, r0 + FOLD_ROWS);
    f32x4 ag = {0.f, 0.f, 0.f, 0.f}, ab = {0.f, 0.f, 0.f, 0.f};
    const bool vec = (Cc & 3) == 0 && c + 3 < Cc;
    if (vec) {
        const f32x4 ga = *(const f32x4*)(gamma + c);
        const f32x4 be = beta ? *(const f32x4*)(beta + c) : f32x4{0.f, 0.f, 0.f, 0.f};
        for (int r = r0 + ry; r < r1; r += 16) {
            const f32x4 g = *(const f32x4*)(dWf + (size_t)r * Cc + c), w = *(const f32x4*)(W + (size_t)r * Cc + c);
            const float gb = dbf ? dbf[r] : 0.f;
            *(f32x4*)(dW + (size_t)r * Cc + c) = g * ga + be * gb;
            ag += g * w;
            ab += w * gb;
        }
    } else {
        for (int e = 0; e < 4; ++e) {
            if (c + e >= Cc) break;
            const float ga = gamma[c + e], be = beta ? beta[c + e] : 0.f;
            for (int r = r0 + ry; r < r1; r += 16) {
                const float g = dWf[(size_t)r * Cc + c + e], w = W[(size_t)r * Cc + c + e];
                const float gb = dbf ? dbf[r] : 0.f;
                dW[(size_t)r * Cc + c + e] = fmaf(g, ga, gb * be);
                ag[e] = fmaf(g, w, ag[e]);
                ab[e] = fmaf(gb, w, ab[e]);
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        s_g[ry][cg * 4 + e] = ag[e];
        s_b[ry][cg * 4 + e] = ab[e];
    }
    __syncthreads();
    if (threadIdx.x < 64 && blockIdx.x * 64 + threadIdx.x < Cc) {
        float g = 0.f, b = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) {        // fixed order
            g += s_g[k][threadIdx.x];
            b += s_b[k][threadIdx.x];
        }
        float* pr = part + (size_t)blockIdx.y * 2 * Cc;
        pr[blockIdx.x * 64 + threadIdx.x] = g;
        pr[Cc + blockIdx.x * 64 + threadIdx.x] = b;
    }
}

__global__ __launch_bounds__(256) void fold_finalize_kernel(const float* __restrict__ part, int nchunk, float* __restrict__ dgamma,
                                                            float* __restrict__ dbeta, int Cc) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= Cc) return;
    float g = 0.f, b = 0.f;
    for (int k = 0; k < nchunk; ++k) {
        g += part[(size_t)k * 2 * Cc + c];
        b += part[(size_t)k * 2 * Cc + Cc + c];
    }
    dgamma[c] = g;
    if (dbeta) dbeta[c] = b;
}

}  // namespace

extern "C" int tv_fold_cols(const float* W, const float* gamma, const float* beta, float* Wf, float* bf, int R, int C, void* stream) {
    TV_CHECK_ARG(W && gamma && Wf && R > 0 && C > 0 && (bf == nullptr) == (beta == nullptr), "tv_fold_cols: bad arguments");
    hipLaunchKernelGGL(fold_fwd_kernel, dim3(tv_cdiv(R, 4)), dim3(256), 0, (hipStream_t)stream, W, gamma, beta, Wf, bf, R, C);
    TV_CHECK_LAUNCH("tv_fold_cols");
    return TV_OK;
}

extern "C" long long tv_fold_partial_count(int R, int C) { return (long long)tv_cdiv(R, FOLD_ROWS) * 2 * C; }

extern "C" int tv_fold_cols_bwd(const float* dWf, const float* dbf, const float* W, const float* gamma, const float* beta, float* dW,
                                float* dgamma, float* dbeta, float* partials, int R, int C, void* stream) {
    TV_CHECK_ARG(dWf && W && gamma && dW && dgamma && partials && R > 0 && C > 0, "tv_fold_cols_bwd: bad arguments");
    TV_CHECK_ARG((beta == nullptr) == (dbeta == nullptr) && (beta == nullptr || dbf != nullptr), "tv_fold_cols_bwd: beta / dbeta / dbf must come together");
    const int nchunk = tv_cdiv(R, FOLD_ROWS);
    hipLaunchKernelGGL(fold_bwd_kernel, dim3(tv_cdiv(C, 64), nchunk), dim3(256), 0, (hipStream_t)stream, dWf, dbf, W, gamma, beta, dW, partials, R, C);
    hipLaunchKernelGGL(fold_finalize_kernel, dim3(tv_cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, (const float*)partials, nchunk, dgamma, dbeta, C);
    TV_CHECK_LAUNCH("tv_fold_cols_bwd");
    return TV_OK;
}
