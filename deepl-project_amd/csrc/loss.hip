// Closed-form terms of the reference loss on the path's outputs, forward AND gradient in one pass.
//
//   L1 : mean |recon - target|                                   R/transvae/losses/vae_loss.py:83-84
//        (sigmoid = 1: recon passes through a sigmoid first       P/.../vae_loss.py:80-84)
//   KL : -0.5 * sum(1 + logvar - mu^2 - exp(logvar)) / denom      R/...:94-96 (denom = B*H*W)   P/...:96-102 (mean: denom = numel)
//        with logvar clamped to [lo, hi] first when lo < hi       R/train_2.py:316-318, P/...:98
//
// One kernel reads recon / target (and mu / logvar) once, writes d(loss)/d(recon), d/d(mu), d/d(logvar) already scaled by
// the term weights, and one partial sum per block; a single-block kernel adds the partials in a fixed order (bit-reproducible)
// into out[0] = weighted L1, out[1] = weighted KL, out[2] = total.  Replaces ~12 elementwise / reduction launches of the
// torch formulation and their intermediate tensors.  HBM-bound: 4 B read + 4 B read + 4 B written per image element.
#include "common.h"

namespace {

constexpr int LOSS_THREADS = 256;
constexpr int LOSS_PER_BLOCK = 256 * 16;

__device__ __forceinline__ float loss_block_sum(float v, float* s_red) {
    v = tv_wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) s_red[wave] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < LOSS_THREADS / 64; ++i) t += s_red[i];
    __syncthreads();
    return t;
}

// blocks [0, nb_l1): L1 over n_img elements; blocks [nb_l1, nb_l1 + nb_kl): KL over n_lat elements
__global__ __launch_bounds__(LOSS_THREADS) void vae_loss_kernel(const float* __restrict__ recon, const float* __restrict__ target,
                                                                const float* __restrict__ mu, const float* __restrict__ logvar,
                                                                float* __restrict__ d_recon, float* __restrict__ d_mu,
                                                                float* __restrict__ d_logvar, float* __restrict__ partial, long long n_img,
                                                                long long n_lat, int nb_l1, float l1_scale, float kl_scale, int sigmoid,
                                                                float lv_lo, float lv_hi) {
    __shared__ float s_red[LOSS_THREADS / 64];
    float acc = 0.f;
    if ((int)blockIdx.x < nb_l1) {
        const long long start = (long long)blockIdx.x * LOSS_PER_BLOCK;
        const long long end = min(n_img, start + LOSS_PER_BLOCK);
        for (long long i = start + threadIdx.x; i < end; i += LOSS_THREADS) {
            float r = recon[i];
            float dr = 1.f;
            if (sigmoid) {
                r = 1.f / (1.f + __expf(-r));
                dr = r * (1.f - r);
            }
            const float d = r - target[i];
            acc += fabsf(d);
            if (d_recon) d_recon[i] = (d > 0.f ? l1_scale : (d < 0.f ? -l1_scale : 0.f)) * dr;
        }
        acc *= l1_scale;
    } else {
        const long long start = (long long)((int)blockIdx.x - nb_l1) * LOSS_PER_BLOCK;
        const long long end = min(n_lat, start + LOSS_PER_BLOCK);
        const bool clamp = lv_lo < lv_hi;
        for (long long i = start + threadIdx.x; i < end; i += LOSS_THREADS) {
            const float m = mu[i];
            float lv = logvar[i];
            bool inside = true;
            if (clamp) {
                inside = lv >= lv_lo && lv <= lv_hi;
                lv = fminf(fmaxf(lv, lv_lo), lv_hi);
            }
            const float e = __expf(lv);
            acc += -0.5f * (1.f + lv - m * m - e);
            if (d_mu) d_mu[i] = kl_scale * m;
            if (d_logvar) d_logvar[i] = inside ? kl_scale * -0.5f * (1.f - e) : 0.f;
        }
        acc *= kl_scale;
    }
    const float s = loss_block_sum(acc, s_red);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

__global__ __launch_bounds__(1024) void vae_loss_finalize_kernel(const float* __restrict__ partial, int nb_l1, int nb, float* __restrict__ out) {
    __shared__ double s_red[2][16];
    double a = 0.0, b = 0.0;
    for (int i = threadIdx.x; i < nb; i += 1024) {
        if (i < nb_l1) a += (double)partial[i];
        else b += (double)partial[i];
    }
    for (int off = 32; off > 0; off >>= 1) {
        a += __shfl_down(a, off, 64);
        b += __shfl_down(b, off, 64);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
        s_red[0][wave] = a;
        s_red[1][wave] = b;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double ta = 0.0, tb = 0.0;
        for (int i = 0; i < 16; ++i) {
            ta += s_red[0][i];
            tb += s_red[1][i];
        }
        out[0] = (float)ta;
        out[1] = (float)tb;
        out[2] = (float)(ta + tb);
    }
}

}  // namespace

extern "C" long long tv_vae_loss_partial_count(long long n_img, long long n_lat) {
    return (n_img + LOSS_PER_BLOCK - 1) / LOSS_PER_BLOCK + (n_lat + LOSS_PER_BLOCK - 1) / LOSS_PER_BLOCK;
}

extern "C" int tv_vae_loss_l1_kl(const float* recon, const float* target, const float* mu, const float* logvar, float* d_recon,
                                 float* d_mu, float* d_logvar, float* partials, float* out, long long n_img, long long n_lat,
                                 float l1_weight, float kl_weight, float kl_denom, int sigmoid, float logvar_lo, float logvar_hi,
                                 void* stream) {
    TV_CHECK_ARG(recon && target && mu && logvar && partials && out, "tv_vae_loss_l1_kl: null pointer");
    TV_CHECK_ARG(n_img > 0 && n_lat > 0 && kl_denom > 0.f, "tv_vae_loss_l1_kl: empty tensors");
    const long long nb_l1 = (n_img + LOSS_PER_BLOCK - 1) / LOSS_PER_BLOCK, nb_kl = (n_lat + LOSS_PER_BLOCK - 1) / LOSS_PER_BLOCK;
    TV_CHECK_ARG(nb_l1 + nb_kl < (1ll << 31), "tv_vae_loss_l1_kl: too many elements");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(vae_loss_kernel, dim3((unsigned)(nb_l1 + nb_kl)), dim3(LOSS_THREADS), 0, s, recon, target, mu, logvar, d_recon, d_mu,
                       d_logvar, partials, n_img, n_lat, (int)nb_l1, l1_weight / (float)n_img, kl_weight / kl_denom, sigmoid, logvar_lo,
                       logvar_hi);
    hipLaunchKernelGGL(vae_loss_finalize_kernel, dim3(1), dim3(1024), 0, s, partials, (int)nb_l1, (int)(nb_l1 + nb_kl), out);
    TV_CHECK_LAUNCH("tv_vae_loss_l1_kl");
    return TV_OK;
}
