// Small HBM-bound helpers of the TransVAE path: weight repack, RoPE, activation backward,
// layout conversion at the NCHW API boundary, stem im2col, 2x2 sum pooling.
#include "common.h"

namespace {

// ---- fp32 [O][T][I] -> bf16 [O][T][I] and bf16 [I][T'][O] ------------------------------------
__global__ __launch_bounds__(256) void pack_weight_kernel(const float* __restrict__ src, bf16* __restrict__ dst, bf16* __restrict__ dst_t,
                                                          int O, int T, int I, int flip) {
    __shared__ float tile[32][33];
    const int t = blockIdx.z;
    const int o0 = blockIdx.y * 32, i0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
#pragma unroll
    for (int r = ty; r < 32; r += 8) {
        const int o = o0 + r, i = i0 + tx;
        float v = 0.f;
        if (o < O && i < I) {
            v = src[((size_t)o * T + t) * I + i];
            if (dst) dst[((size_t)o * T + t) * I + i] = (bf16)v;
        }
        tile[r][tx] = v;
    }
    if (!dst_t) return;
    __syncthreads();
    const int tt = flip ? (T - 1 - t) : t;
#pragma unroll
    for (int r = ty; r < 32; r += 8) {
        const int i = i0 + r, o = o0 + tx;
        if (o < O && i < I) dst_t[((size_t)i * T + tt) * O + o] = (bf16)tile[tx][r];
    }
}

// ---- RoPE on the q,k thirds of qkv [B,N,3,h,64]; tab [N][4][32] = cos1,sin1,cos2,sin2 --------
__global__ __launch_bounds__(256) void rope_qk_kernel(bf16* __restrict__ qkv, const float* __restrict__ tab, long long total, int N,
                                                      int heads, int transpose) {
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        long long r = idx;
        const int v = (int)(r & 7);  // pairs 4v..4v+3
        r >>= 3;
        const int head = (int)(r % heads);
        r /= heads;
        const int which = (int)(r & 1);
        r >>= 1;
        const int n = (int)(r % N);
        const long long b = r / N;
        bf16* ptr = qkv + ((((size_t)b * N + n) * 3 + which) * heads + head) * 64 + v * 8;
        bf16x8 t = *(const bf16x8*)ptr;
        const float* tb = tab + (size_t)n * 128 + v * 4;
        const f32x4 c1 = *(const f32x4*)(tb), s1 = *(const f32x4*)(tb + 32), c2 = *(const f32x4*)(tb + 64),
                    s2 = *(const f32x4*)(tb + 96);
        bf16x8 o;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const float a = (float)t[2 * p], bb = (float)t[2 * p + 1];
            float o1, o2;
            if (!transpose) {
                o1 = a * c1[p] - bb * s1[p];
                o2 = a * s2[p] + bb * c2[p];
            } else {  // adjoint of the (non-orthogonal) 2x2
                o1 = a * c1[p] + bb * s2[p];
                o2 = -a * s1[p] + bb * c2[p];
            }
            o[2 * p] = (bf16)o1;
            o[2 * p + 1] = (bf16)o2;
        }
        *(bf16x8*)ptr = o;
    }
}

__global__ __launch_bounds__(256) void act_bwd_kernel(const bf16* __restrict__ z, const bf16* __restrict__ dy, bf16* __restrict__ dz,
                                                      long long nvec, int act) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (long long)gridDim.x * 256) {
        const bf16x8 zv = *(const bf16x8*)(z + i * 8);
        const bf16x8 gv = *(const bf16x8*)(dy + i * 8);
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (bf16)((float)gv[e] * tv_act_grad_rt(act, (float)zv[e]));
        *(bf16x8*)(dz + i * 8) = o;
    }
}

__global__ __launch_bounds__(256) void add_kernel(bf16* __restrict__ a, const bf16* __restrict__ b, long long nvec) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (long long)gridDim.x * 256) {
        bf16x8 av = *(const bf16x8*)(a + i * 8);
        const bf16x8 bv = *(const bf16x8*)(b + i * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) av[e] = (bf16)((float)av[e] + (float)bv[e]);
        *(bf16x8*)(a + i * 8) = av;
    }
}

// dst[b][y][x][c] (bf16, Cpad channels, zero filled) <- src[b][c][y][x] (fp32)
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ src, bf16* __restrict__ dst, long long total, int C,
                                                           int HW, int Cpad) {
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int c = (int)(idx % Cpad);
        const long long pix = idx / Cpad;
        const long long b = pix / HW;
        const int p = (int)(pix - b * HW);
        dst[idx] = (c < C) ? (bf16)src[((size_t)b * C + c) * HW + p] : (bf16)0.f;
    }
}

// dst[b][c][y][x] (fp32) <- src[b][y][x][c] (bf16, Cpad channel stride)
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const bf16* __restrict__ src, float* __restrict__ dst, long long total, int C,
                                                           int HW, int Cpad) {
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int p = (int)(idx % HW);
        const long long bc = idx / HW;
        const int c = (int)(bc % C);
        const long long b = bc / C;
        dst[idx] = (float)src[((size_t)b * HW + p) * Cpad + c];
    }
}

// 3x3 / pad 1 patches of an NCHW fp32 image -> rows [ (ky,kx,c) ... zero pad ] of Kpad bf16
__global__ __launch_bounds__(256) void im2col3x3_kernel(const float* __restrict__ src, bf16* __restrict__ dst, long long total, int C,
                                                        int H, int W, int Kpad) {
    const int kv = Kpad >> 3;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int v = (int)(idx % kv);
        const long long pix = idx / kv;
        const int x = (int)(pix % W);
        const long long r = pix / W;
        const int y = (int)(r % H);
        const long long b = r / H;
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = v * 8 + e;
            float f = 0.f;
            if (k < 9 * C) {
                const int tap = k / C, c = k - tap * C;
                const int iy = y + tap / 3 - 1, ix = x + tap % 3 - 1;
                if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) f = src[(((size_t)b * C + c) * H + iy) * W + ix];
            }
            o[e] = (bf16)f;
        }
        *(bf16x8*)(dst + idx * 8) = o;
    }
}

// dst[b][y][x][c] = sum of src[b][2y+dy][2x+dx][c]
__global__ __launch_bounds__(256) void pool2x2_sum_kernel(const bf16* __restrict__ src, bf16* __restrict__ dst, long long total, int H, int W,
                                                          int C) {
    const int cv = C >> 3;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int v = (int)(idx % cv);
        const long long pix = idx / cv;
        const int x = (int)(pix % W);
        const long long r = pix / W;
        const int y = (int)(r % H);
        const long long b = r / H;
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = 0.f;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const size_t sp = ((size_t)b * (2 * H) + 2 * y + (d >> 1)) * (2 * W) + 2 * x + (d & 1);
            const bf16x8 t = *(const bf16x8*)(src + sp * C + v * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += (float)t[e];
        }
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (bf16)acc[e];
        *(bf16x8*)(dst + idx * 8) = o;
    }
}

inline int ew_grid(long long n) {
    long long g = (n + 255) / 256;
    if (g > 256 * 16) g = 256 * 16;
    if (g < 1) g = 1;
    return (int)g;
}


// ---- operands derived from a 3x3 weight by sums of taps (polyphase up-conv, its adjoint, the parity form of the stride-2 data
// gradient) and the fold of the 4x4 adjoint's weight gradient back onto the 3x3 taps: out[tap_o] = sum_j in[tap_j], optionally
// transposed ([A][Tin][B] -> [B][..][A]).  One launch instead of 20-60 slice adds / copies of torch (transvae/hip/ops.py: the
// polyphase algebra ran ~600 small kernels per optimizer step).  Sums run in the order of the table = the order of the torch
// expression they replace (ky outer, kx inner): bit-identical.
struct TapSumTable {
    int n[16];
    int taps[16][4];
    long long base[16];   // element offset of output tap t
};

template <typename OutT, bool TRANSPOSE, bool ACC>
__global__ __launch_bounds__(256) void tapsum_kernel(const float* __restrict__ src, OutT* __restrict__ dst, const TapSumTable tab, int A, int Tin,
                                                     int Bd, long long stride_out) {
    __shared__ float tile[32][33];
    const int t = blockIdx.z;
    const int a0 = blockIdx.y * 32, b0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    const int n = tab.n[t];
#pragma unroll
    for (int r = ty; r < 32; r += 8) {
        const int a = a0 + r, b = b0 + tx;
        float v = 0.f;
        if (a < A && b < Bd) {
            for (int j = 0; j < n; ++j) {
                const float u = src[((size_t)a * Tin + tab.taps[t][j]) * Bd + b];
                v = j == 0 ? u : v + u;
            }
            if constexpr (!TRANSPOSE) {
                OutT* q = dst + tab.base[t] + (long long)a * stride_out + b;
                if constexpr (ACC) *q = (OutT)((float)*q + v);
                else *q = (OutT)v;
            }
        }
        if constexpr (TRANSPOSE) tile[r][tx] = v;
    }
    if constexpr (TRANSPOSE) {
        __syncthreads();
#pragma unroll
        for (int r = ty; r < 32; r += 8) {
            const int b = b0 + r, a = a0 + tx;
            if (a < A && b < Bd) {
                OutT* q = dst + tab.base[t] + (long long)b * stride_out + a;
                if constexpr (ACC) *q = (OutT)((float)*q + tile[tx][r]);
                else *q = (OutT)tile[tx][r];
            }
        }
    }
}

}  // namespace

extern "C" int tv_pack_weight(const float* src, void* dst, void* dst_t, int O, int T, int I, int flip_taps, void* stream) {
    TV_CHECK_ARG(src && (dst || dst_t) && O > 0 && T > 0 && I > 0 && T <= 65535, "tv_pack_weight: bad arguments O=%d T=%d I=%d", O, T, I);
    dim3 grid(tv_cdiv(I, 32), tv_cdiv(O, 32), T);
    TV_CHECK_ARG(grid.y <= 65535, "tv_pack_weight: O too large");
    hipLaunchKernelGGL(pack_weight_kernel, grid, dim3(256), 0, (hipStream_t)stream, src, (bf16*)dst, (bf16*)dst_t, O, T, I, flip_taps);
    TV_CHECK_LAUNCH("tv_pack_weight");
    return TV_OK;
}

extern "C" int tv_conv3x3_derived(const float* src, void* dst, int form, int c_out, int c_in, int accumulate, void* stream) {
    TV_CHECK_ARG(src && dst && c_out > 0 && c_in > 0 && form >= TV_DERIVE_UP_FWD && form <= TV_DERIVE_S2_PARITY,
                 "tv_conv3x3_derived: bad arguments form=%d c_out=%d c_in=%d", form, c_out, c_in);
    TV_CHECK_ARG(!accumulate || form == TV_DERIVE_UP_WGRAD_FOLD, "tv_conv3x3_derived: only the weight-gradient fold accumulates");
    static const int up_sets[2][2][2] = {{{0, -1}, {1, 2}}, {{0, 1}, {2, -1}}};   // [phase][tap] -> ky (or kx) folded in (ops.py _UP_SETS)
    static const int up_adj[4][2] = {{2, -1}, {1, 2}, {0, 1}, {0, -1}};            // adjoint tap t -> ky (ops.py _UP_ADJ)
    static const int fold_taps[3][2] = {{2, 3}, {1, 2}, {0, 1}};                   // 3x3 tap k <- adjoint taps containing it
    TapSumTable tab = {};
    int nt = 0, A = c_out, Tin = 9, Bd = c_in;
    long long stride = 0;
    auto add = [&](int t, const int* ys, const int* xs, int tin_w) {
        int n = 0;
        for (int i = 0; i < 2; ++i)
            for (int j = 0; j < 2; ++j)
                if (ys[i] >= 0 && xs[j] >= 0) tab.taps[t][n++] = ys[i] * tin_w + xs[j];
        tab.n[t] = n;
    };
    if (form == TV_DERIVE_UP_FWD) {              // [Cout,3,3,Cin] -> bf16 [4*Cout,2,2,Cin]
        for (int py = 0; py < 2; ++py)
            for (int px = 0; px < 2; ++px)
                for (int ty = 0; ty < 2; ++ty)
                    for (int tx = 0; tx < 2; ++tx) {
                        add(nt, up_sets[py][ty], up_sets[px][tx], 3);
                        tab.base[nt++] = ((long long)(2 * py + px) * c_out * 4 + (ty * 2 + tx)) * c_in;
                    }
        stride = 4LL * c_in;
    } else if (form == TV_DERIVE_UP_DGRAD) {     // [Cout,3,3,Cin] -> bf16 [Cin,4,4,Cout]
        for (int ty = 0; ty < 4; ++ty)
            for (int tx = 0; tx < 4; ++tx) {
                add(nt, up_adj[ty], up_adj[tx], 3);
                tab.base[nt++] = (long long)(ty * 4 + tx) * c_out;
            }
        stride = 16LL * c_out;
    } else if (form == TV_DERIVE_UP_WGRAD_FOLD) {   // fp32 [Cin,4,4,Cout] -> fp32 [Cout,3,3,Cin]
        A = c_in;
        Tin = 16;
        Bd = c_out;
        for (int ky = 0; ky < 3; ++ky)
            for (int kx = 0; kx < 3; ++kx) {
                add(nt, fold_taps[ky], fold_taps[kx], 4);
                tab.base[nt++] = (long long)(ky * 3 + kx) * c_in;
            }
        stride = 9LL * c_in;
    } else {                                     // TV_DERIVE_S2_PARITY: [Cout,3,3,Cin] -> bf16 [4*Cin,2,2,Cout]
        for (int py = 0; py < 2; ++py)
            for (int px = 0; px < 2; ++px)
                for (int ty = 0; ty < 2; ++ty)
                    for (int tx = 0; tx < 2; ++tx) {
                        if (ty <= py && tx <= px) {
                            const int ky = py == 0 ? 1 : (ty == 0 ? 2 : 0), kx = px == 0 ? 1 : (tx == 0 ? 2 : 0);
                            tab.taps[nt][0] = ky * 3 + kx;
                            tab.n[nt] = 1;
                        } else {
                            tab.n[nt] = 0;       // outside the class's footprint: zeros
                        }
                        tab.base[nt++] = ((long long)(2 * py + px) * c_in * 4 + (ty * 2 + tx)) * c_out;
                    }
        stride = 4LL * c_out;
    }
    dim3 grid(tv_cdiv(Bd, 32), tv_cdiv(A, 32), nt);
    TV_CHECK_ARG(grid.y <= 65535, "tv_conv3x3_derived: channel count too large");
    hipStream_t s = (hipStream_t)stream;
    if (form == TV_DERIVE_UP_FWD)
        hipLaunchKernelGGL((tapsum_kernel<bf16, false, false>), grid, dim3(256), 0, s, src, (bf16*)dst, tab, A, Tin, Bd, stride);
    else if (form == TV_DERIVE_UP_WGRAD_FOLD && accumulate)
        hipLaunchKernelGGL((tapsum_kernel<float, true, true>), grid, dim3(256), 0, s, src, (float*)dst, tab, A, Tin, Bd, stride);
    else if (form == TV_DERIVE_UP_WGRAD_FOLD)
        hipLaunchKernelGGL((tapsum_kernel<float, true, false>), grid, dim3(256), 0, s, src, (float*)dst, tab, A, Tin, Bd, stride);
    else
        hipLaunchKernelGGL((tapsum_kernel<bf16, true, false>), grid, dim3(256), 0, s, src, (bf16*)dst, tab, A, Tin, Bd, stride);
    TV_CHECK_LAUNCH("tv_conv3x3_derived");
    return TV_OK;
}

extern "C" int tv_rope_qk(void* qkv, const float* tab, int B, int N, int heads, int transpose, void* stream) {
    TV_CHECK_ARG(qkv && tab && B > 0 && N > 0 && heads > 0, "tv_rope_qk: bad arguments");
    const long long total = (long long)B * N * 2 * heads * 8;
    hipLaunchKernelGGL(rope_qk_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, (bf16*)qkv, tab, total, N, heads, transpose);
    TV_CHECK_LAUNCH("tv_rope_qk");
    return TV_OK;
}

extern "C" int tv_act_bwd(const void* z, const void* dy, void* dz, long long n, int act, void* stream) {
    TV_CHECK_ARG(z && dy && dz && n > 0 && n % 8 == 0, "tv_act_bwd: n=%lld must be a positive multiple of 8", n);
    TV_CHECK_ARG(act >= 0 && act <= 2, "tv_act_bwd: unknown activation %d", act);
    hipLaunchKernelGGL(act_bwd_kernel, dim3(ew_grid(n / 8)), dim3(256), 0, (hipStream_t)stream, (const bf16*)z, (const bf16*)dy, (bf16*)dz,
                       n / 8, act);
    TV_CHECK_LAUNCH("tv_act_bwd");
    return TV_OK;
}

extern "C" int tv_add_(void* a, const void* b, long long n, void* stream) {
    TV_CHECK_ARG(a && b && n > 0 && n % 8 == 0, "tv_add_: n=%lld must be a positive multiple of 8", n);
    hipLaunchKernelGGL(add_kernel, dim3(ew_grid(n / 8)), dim3(256), 0, (hipStream_t)stream, (bf16*)a, (const bf16*)b, n / 8);
    TV_CHECK_LAUNCH("tv_add_");
    return TV_OK;
}

extern "C" int tv_nchw_to_nhwc(const float* src, void* dst, int B, int C, int H, int W, int Cpad, void* stream) {
    TV_CHECK_ARG(src && dst && B > 0 && C > 0 && H > 0 && W > 0 && Cpad >= C, "tv_nchw_to_nhwc: bad arguments");
    const long long total = (long long)B * H * W * Cpad;
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, src, (bf16*)dst, total, C, H * W, Cpad);
    TV_CHECK_LAUNCH("tv_nchw_to_nhwc");
    return TV_OK;
}

extern "C" int tv_nhwc_to_nchw(const void* src, float* dst, int B, int C, int H, int W, int Cpad, void* stream) {
    TV_CHECK_ARG(src && dst && B > 0 && C > 0 && H > 0 && W > 0 && Cpad >= C, "tv_nhwc_to_nchw: bad arguments");
    const long long total = (long long)B * C * H * W;
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, (const bf16*)src, dst, total, C, H * W, Cpad);
    TV_CHECK_LAUNCH("tv_nhwc_to_nchw");
    return TV_OK;
}

extern "C" int tv_im2col3x3(const float* src, void* dst, int B, int C, int H, int W, int Kpad, void* stream) {
    TV_CHECK_ARG(src && dst && B > 0 && C > 0 && H > 0 && W > 0 && Kpad >= 9 * C && Kpad % 8 == 0, "tv_im2col3x3: Kpad=%d must be >= 9*C and a multiple of 8", Kpad);
    const long long total = (long long)B * H * W * (Kpad / 8);
    hipLaunchKernelGGL(im2col3x3_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, src, (bf16*)dst, total, C, H, W, Kpad);
    TV_CHECK_LAUNCH("tv_im2col3x3");
    return TV_OK;
}

extern "C" int tv_pool2x2_sum(const void* src, void* dst, int B, int H, int W, int C, void* stream) {
    TV_CHECK_ARG(src && dst && B > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, "tv_pool2x2_sum: bad arguments");
    const long long total = (long long)B * H * W * (C / 8);
    hipLaunchKernelGGL(pool2x2_sum_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, (const bf16*)src, (bf16*)dst, total, H, W, C);
    TV_CHECK_LAUNCH("tv_pool2x2_sum");
    return TV_OK;
}
