// Train-step glue around the path, HBM-bound multi-tensor kernels for gfx950:
//
//   tv_opt_grad_norm   global L2 norm of all gradients -> clip coefficient, non-finite guard, step counter
//   tv_opt_adamw       AdamW over every parameter tensor in ONE launch; the clip is the un-scale of the gradient inside
//                      the update, a non-finite step is skipped on the device, and the bf16 copy of each weight (the
//                      forward operand of tv_igemm_nt) is written in the same pass
//   tv_pack_weight_multi   all transposed (data-gradient) operands refreshed from those bf16 copies in one launch
//
// Reference behaviour: torch.optim.AdamW(lr, betas, eps, weight_decay) -- R/train.py:681-687;
// clip_grad_norm_(max_norm) -- R/train.py:610-612; skip on non-finite -- R/train_2.py:328-338.
// Per element (the arithmetic of ATen's fused AdamW, fp32):
//     p  -= lr * wd * p
//     m   = m + (g - m) * (1 - beta1)
//     v   = beta2 * v + (1 - beta2) * g * g
//     p  -= (lr / (1 - beta1^t)) * m / (sqrt(v) / sqrt(1 - beta2^t) + eps)
// Algorithmic bytes per parameter element: norm 4 (read g) ; update 4 (g) + 3 * 8 (p, m, v read + write) + 2 (bf16 copy).
#include "common.h"

namespace {

constexpr int OPT_CHUNK = 65536;   // elements per workgroup
constexpr int OPT_THREADS = 256;

struct OptTensor {   // == struct tv_opt_tensor
    float* param;
    const float* grad;
    float* exp_avg;
    float* exp_avg_sq;
    bf16* shadow;
    long long numel;
};

__device__ __forceinline__ float block_sum(float v, float* s_red) {
    v = tv_wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) s_red[wave] = v;
    __syncthreads();
    float t = 0.f;
    const int nw = blockDim.x >> 6;
    for (int i = 0; i < nw; ++i) t += s_red[i];   // fixed order
    return t;
}

// chunk c = (tensor, offset): partial[c] = sum g^2 over that chunk
__global__ __launch_bounds__(OPT_THREADS) void opt_sqnorm_kernel(const OptTensor* __restrict__ tab, const int2* __restrict__ chunks,
                                                                 float* __restrict__ partial) {
    __shared__ float s_red[OPT_THREADS / 64];
    const int2 c = chunks[blockIdx.x];
    const OptTensor t = tab[c.x];
    const long long start = (long long)c.y * OPT_CHUNK;
    const long long end = min(t.numel, start + OPT_CHUNK);
    const float* g = t.grad;
    float acc = 0.f;
    if ((((uintptr_t)g) & 15) == 0) {
        const long long n4 = (end - start) >> 2;
        const f32x4* g4 = (const f32x4*)(g + start);
        for (long long i = threadIdx.x; i < n4; i += OPT_THREADS) {
            const f32x4 v = g4[i];
            acc = fmaf(v[0], v[0], acc);
            acc = fmaf(v[1], v[1], acc);
            acc = fmaf(v[2], v[2], acc);
            acc = fmaf(v[3], v[3], acc);
        }
        for (long long i = start + (n4 << 2) + threadIdx.x; i < end; i += OPT_THREADS) acc = fmaf(g[i], g[i], acc);
    } else {
        for (long long i = start + threadIdx.x; i < end; i += OPT_THREADS) acc = fmaf(g[i], g[i], acc);
    }
    const float s = block_sum(acc, s_red);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

// ctrl: [0] step t (float, persists)  [1] gradient norm  [2] clip coefficient  [3] 1 if this step is skipped
//       [4] number of skipped steps so far  [5] 1 - beta1^t  [6] sqrt(1 - beta2^t)
__global__ __launch_bounds__(1024) void opt_ctrl_kernel(const float* __restrict__ partial, int n, float* __restrict__ ctrl, float max_norm,
                                                        float beta1, float beta2, int have_norm) {
    __shared__ double s_red[16];
    double acc = 0.0;
    if (have_norm)
        for (int i = threadIdx.x; i < n; i += 1024) acc += (double)partial[i];   // fixed assignment, fixed order below
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) s_red[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double tot = 0.0;
        for (int i = 0; i < 16; ++i) tot += s_red[i];
        const float norm = (float)sqrt(tot);
        const bool bad = !(fabsf(norm) <= 3.0e38f);   // NaN or inf
        float coef = 1.0f;
        if (!bad && max_norm > 0.f) coef = fminf(1.0f, max_norm / (norm + 1e-6f));
        const float step = ctrl[0] + (bad ? 0.f : 1.f);
        ctrl[0] = step;
        ctrl[1] = norm;
        ctrl[2] = coef;
        ctrl[3] = bad ? 1.f : 0.f;
        ctrl[4] += bad ? 1.f : 0.f;
        ctrl[5] = (float)(1.0 - pow((double)beta1, (double)step));
        ctrl[6] = (float)sqrt(1.0 - pow((double)beta2, (double)step));
    }
}

__device__ __forceinline__ void adamw_elem(float& p, float g, float& m, float& v, float coef, float lr, float beta1, float beta2, float eps,
                                           float wd, float step_size, float bc2s_inv) {
    g *= coef;
    p -= lr * wd * p;
    m = m + (g - m) * (1.f - beta1);
    v = beta2 * v + (1.f - beta2) * g * g;
    const float denom = sqrtf(v) * bc2s_inv + eps;
    p -= step_size * (m / denom);
}

__global__ __launch_bounds__(OPT_THREADS) void opt_adamw_kernel(const OptTensor* __restrict__ tab, const int2* __restrict__ chunks,
                                                                const float* __restrict__ ctrl, float lr, float beta1, float beta2, float eps,
                                                                float wd) {
    if (ctrl[3] != 0.f) return;   // non-finite gradients: the whole step is skipped (uniform over the grid)
    const float coef = ctrl[2];
    const float step_size = lr / ctrl[5];
    const float bc2s_inv = 1.0f / ctrl[6];
    const int2 c = chunks[blockIdx.x];
    const OptTensor t = tab[c.x];
    const long long start = (long long)c.y * OPT_CHUNK;
    const long long end = min(t.numel, start + OPT_CHUNK);
    const bool vec = ((((uintptr_t)t.param) | ((uintptr_t)t.grad) | ((uintptr_t)t.exp_avg) | ((uintptr_t)t.exp_avg_sq)) & 15) == 0 &&
                     (t.shadow == nullptr || (((uintptr_t)t.shadow) & 7) == 0);
    long long tail = start;
    if (vec) {
        const long long n4 = (end - start) >> 2;
        f32x4* p4 = (f32x4*)(t.param + start);
        const f32x4* g4 = (const f32x4*)(t.grad + start);
        f32x4* m4 = (f32x4*)(t.exp_avg + start);
        f32x4* v4 = (f32x4*)(t.exp_avg_sq + start);
        bf16x4* s4 = t.shadow ? (bf16x4*)(t.shadow + start) : nullptr;
        for (long long i = threadIdx.x; i < n4; i += OPT_THREADS) {
            f32x4 p = p4[i], m = m4[i], v = v4[i];
            const f32x4 g = g4[i];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float pe = p[e], me = m[e], ve = v[e];
                adamw_elem(pe, g[e], me, ve, coef, lr, beta1, beta2, eps, wd, step_size, bc2s_inv);
                p[e] = pe;
                m[e] = me;
                v[e] = ve;
            }
            p4[i] = p;
            m4[i] = m;
            v4[i] = v;
            if (s4) {
                bf16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (bf16)p[e];
                s4[i] = o;
            }
        }
        tail = start + (n4 << 2);
    }
    for (long long i = tail + threadIdx.x; i < end; i += OPT_THREADS) {
        float p = t.param[i], m = t.exp_avg[i], v = t.exp_avg_sq[i];
        adamw_elem(p, t.grad[i], m, v, coef, lr, beta1, beta2, eps, wd, step_size, bc2s_inv);
        t.param[i] = p;
        t.exp_avg[i] = m;
        t.exp_avg_sq[i] = v;
        if (t.shadow) t.shadow[i] = (bf16)p;
    }
}

// fp32 -> bf16 copies of every tensor (first fill of the shadows, before any optimizer step)
__global__ __launch_bounds__(OPT_THREADS) void opt_cast_kernel(const OptTensor* __restrict__ tab, const int2* __restrict__ chunks) {
    const int2 c = chunks[blockIdx.x];
    const OptTensor t = tab[c.x];
    if (!t.shadow) return;
    const long long start = (long long)c.y * OPT_CHUNK;
    const long long end = min(t.numel, start + OPT_CHUNK);
    for (long long i = start + threadIdx.x; i < end; i += OPT_THREADS) t.shadow[i] = (bf16)t.param[i];
}

// ---- all transposed operands in one launch: bf16 [O][T][I] -> bf16 [I][T'][O] -----------------------------------------
struct PackForm {   // == struct tv_pack_form
    const bf16* src;
    bf16* dst_t;
    int O, T, I, flip;
    long long tile_start;   // first 64x64 tile of this form in the launch (exclusive prefix sum)
};

__global__ __launch_bounds__(256) void pack_multi_kernel(const PackForm* __restrict__ forms, int n_forms) {
    __shared__ bf16 tile[64][66];
    // binary search: last form with tile_start <= blockIdx.x
    int lo = 0, hi = n_forms - 1;
    const long long blk = blockIdx.x;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (forms[mid].tile_start <= blk) lo = mid;
        else hi = mid - 1;
    }
    const PackForm f = forms[lo];
    long long r = blk - f.tile_start;
    const int ti = (f.I + 63) >> 6, to = (f.O + 63) >> 6;
    const int bi = (int)(r % ti);
    r /= ti;
    const int bo = (int)(r % to);
    const int t = (int)(r / to);
    const int o0 = bo * 64, i0 = bi * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;   // 64 x 4
    for (int rr = ty; rr < 64; rr += 4) {
        const int o = o0 + rr, i = i0 + tx;
        tile[rr][tx] = (o < f.O && i < f.I) ? f.src[((size_t)o * f.T + t) * f.I + i] : (bf16)0.f;
    }
    __syncthreads();
    const int tt = f.flip ? (f.T - 1 - t) : t;
    for (int rr = ty; rr < 64; rr += 4) {
        const int i = i0 + rr, o = o0 + tx;
        if (o < f.O && i < f.I) f.dst_t[((size_t)i * f.T + tt) * f.O + o] = tile[tx][rr];
    }
}

}  // namespace

extern "C" int tv_opt_chunk_elems(void) { return OPT_CHUNK; }

extern "C" int tv_opt_grad_norm(const tv_opt_tensor* table_dev, const int* chunks_dev, int n_chunks, float* partials, float* ctrl,
                                float max_norm, float beta1, float beta2, int compute_norm, void* stream) {
    TV_CHECK_ARG(table_dev && chunks_dev && n_chunks > 0 && partials && ctrl, "tv_opt_grad_norm: bad arguments");
    TV_CHECK_ARG(beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f, "tv_opt_grad_norm: betas must be in [0, 1)");
    hipStream_t s = (hipStream_t)stream;
    if (compute_norm)
        hipLaunchKernelGGL(opt_sqnorm_kernel, dim3(n_chunks), dim3(OPT_THREADS), 0, s, (const OptTensor*)table_dev, (const int2*)chunks_dev, partials);
    hipLaunchKernelGGL(opt_ctrl_kernel, dim3(1), dim3(1024), 0, s, partials, n_chunks, ctrl, max_norm, beta1, beta2, compute_norm);
    TV_CHECK_LAUNCH("tv_opt_grad_norm");
    return TV_OK;
}

extern "C" int tv_opt_adamw(const tv_opt_tensor* table_dev, const int* chunks_dev, int n_chunks, const float* ctrl, float lr, float beta1,
                            float beta2, float eps, float weight_decay, void* stream) {
    TV_CHECK_ARG(table_dev && chunks_dev && n_chunks > 0 && ctrl, "tv_opt_adamw: bad arguments");
    TV_CHECK_ARG(lr >= 0.f && eps >= 0.f && weight_decay >= 0.f, "tv_opt_adamw: lr / eps / weight_decay must be non-negative");
    hipLaunchKernelGGL(opt_adamw_kernel, dim3(n_chunks), dim3(OPT_THREADS), 0, (hipStream_t)stream, (const OptTensor*)table_dev,
                       (const int2*)chunks_dev, ctrl, lr, beta1, beta2, eps, weight_decay);
    TV_CHECK_LAUNCH("tv_opt_adamw");
    return TV_OK;
}

extern "C" int tv_opt_cast_shadows(const tv_opt_tensor* table_dev, const int* chunks_dev, int n_chunks, void* stream) {
    TV_CHECK_ARG(table_dev && chunks_dev && n_chunks > 0, "tv_opt_cast_shadows: bad arguments");
    hipLaunchKernelGGL(opt_cast_kernel, dim3(n_chunks), dim3(OPT_THREADS), 0, (hipStream_t)stream, (const OptTensor*)table_dev,
                       (const int2*)chunks_dev);
    TV_CHECK_LAUNCH("tv_opt_cast_shadows");
    return TV_OK;
}

extern "C" int tv_pack_weight_multi(const tv_pack_form* forms_dev, int n_forms, long long total_tiles, void* stream) {
    TV_CHECK_ARG(forms_dev && n_forms > 0 && total_tiles > 0 && total_tiles < (1ll << 31), "tv_pack_weight_multi: bad arguments");
    hipLaunchKernelGGL(pack_multi_kernel, dim3((unsigned)total_tiles), dim3(256), 0, (hipStream_t)stream, (const PackForm*)forms_dev, n_forms);
    TV_CHECK_LAUNCH("tv_pack_weight_multi");
    return TV_OK;
}
