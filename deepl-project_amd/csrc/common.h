// Shared device/host helpers for the TransVAE gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/transvae_hip.h"

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define TV_LDS(ptr) ((__attribute__((address_space(3))) void*)(ptr))
#define TV_GLB(ptr) ((const __attribute__((address_space(1))) void*)(ptr))

// ---------------------------------------------------------------------------
// error plumbing (host)
// ---------------------------------------------------------------------------
void tv_set_error(const char* fmt, ...);
const void* tv_zero_page();  // >= 4 KiB of device zeros (source for padded LDS-DMA lanes)

// Per-device "done once" flag for hipFuncSetAttribute (function attributes are per device: a second GPU in the same
// process needs its own > 64 KiB dynamic-LDS opt-in).  Usage:  static TvPerDeviceOnce once;  if (once.first()) { ... }
struct TvPerDeviceOnce {
    bool done[64] = {};
    bool first() {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return true;
        if (done[dev]) return false;
        done[dev] = true;
        return true;
    }
};

#define TV_CHECK_ARG(cond, ...)        \
    do {                               \
        if (!(cond)) {                 \
            tv_set_error(__VA_ARGS__); \
            return TV_ERR_ARG;         \
        }                              \
    } while (0)

#define TV_CHECK_LAUNCH(name)                                                  \
    do {                                                                       \
        hipError_t e__ = hipGetLastError();                                    \
        if (e__ != hipSuccess) {                                               \
            tv_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
            return TV_ERR_LAUNCH;                                              \
        }                                                                      \
    } while (0)

// ---------------------------------------------------------------------------
// activations (device).  erf by Abramowitz-Stegun 7.1.26 (|err| <= 1.5e-7),
// i.e. the exact (erf) GELU of the reference to fp32 rounding noise.
// ---------------------------------------------------------------------------
#define TV_ACT_NONE 0
#define TV_ACT_GELU 1
#define TV_ACT_SILU 2
#ifndef TV_ACT_DERIV
#define TV_ACT_DERIV 3        // (aux_act only) the saved tensor already holds act'(pre-activation)
#endif
#ifndef TV_ACT_ADD
#define TV_ACT_ADD 4          // (aux_act only) the second tensor is added: out = acc + residual + aux
#endif
#define TV_ACT_SAVE_DERIV 16  // (flag on desc.act) pre_act receives act'(pre-activation) instead of the pre-activation

__device__ __forceinline__ float tv_fast_exp(float x) { return __expf(x); }
// v_rcp_f32 (1 ulp).  `__frcp_rn` compiles to the IEEE-correct division sequence (v_div_scale x2, v_rcp, 4 FMAs, v_div_fmas,
// v_div_fixup: ten instructions) -- it was a third of the GELU / SiLU epilogue arithmetic, which runs with the matrix pipe idle.
__device__ __forceinline__ float tv_fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

// erf(x) with e = exp(-x^2) handed back (the Gaussian the GELU gradient needs as well)
__device__ __forceinline__ float tv_erf_e(float x, float& e) {
    const float ax = fabsf(x);
    const float t = tv_fast_rcp(fmaf(0.3275911f, ax, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    p *= t;
    e = tv_fast_exp(-ax * ax);
    const float r = fmaf(-p, e, 1.0f);
    return copysignf(r, x);
}
__device__ __forceinline__ float tv_erf(float x) {
    float e;
    return tv_erf_e(x, e);
}

__device__ __forceinline__ float tv_gelu(float z) { return 0.5f * z * (1.0f + tv_erf(z * 0.70710678118654752f)); }

// d/dz gelu(z) = Phi(z) + z*phi(z);  phi(z) = exp(-z^2/2)/sqrt(2 pi) is the exp(-x^2) of the erf at x = z/sqrt(2): one
// exponential per element (the activation-gradient epilogues are VALU-bound on exactly this arithmetic)
__device__ __forceinline__ float tv_gelu_grad(float z) {
    float e;
    const float cdf = fmaf(0.5f, tv_erf_e(z * 0.70710678118654752f, e), 0.5f);
    return fmaf(z * 0.3989422804014327f, e, cdf);
}

__device__ __forceinline__ float tv_sigmoid(float z) { return tv_fast_rcp(1.0f + tv_fast_exp(-z)); }
__device__ __forceinline__ float tv_silu(float z) { return z * tv_sigmoid(z); }
__device__ __forceinline__ float tv_silu_grad(float z) {
    const float s = tv_sigmoid(z);
    return s * fmaf(z, 1.0f - s, 1.0f);
}

template <int ACT>
__device__ __forceinline__ float tv_act(float z) {
    if constexpr (ACT == TV_ACT_GELU) return tv_gelu(z);
    if constexpr (ACT == TV_ACT_SILU) return tv_silu(z);
    return z;
}
__device__ __forceinline__ float tv_act_rt(int act, float z) {
    if (act == TV_ACT_GELU) return tv_gelu(z);
    if (act == TV_ACT_SILU) return tv_silu(z);
    return z;
}
__device__ __forceinline__ float tv_act_grad_rt(int act, float z) {
    if (act == TV_ACT_GELU) return tv_gelu_grad(z);
    if (act == TV_ACT_SILU) return tv_silu_grad(z);
    return 1.0f;
}
template <int ACT>
__device__ __forceinline__ float tv_act_with_grad(float z, float& g) {
    if constexpr (ACT == TV_ACT_GELU) {
        float e;
        const float cdf = fmaf(0.5f, tv_erf_e(z * 0.70710678118654752f, e), 0.5f);
        g = fmaf(z * 0.3989422804014327f, e, cdf);
        return z * cdf;
    } else if constexpr (ACT == TV_ACT_SILU) {
        const float s = tv_sigmoid(z);
        g = s * fmaf(z, 1.0f - s, 1.0f);
        return z * s;
    } else {
        g = 1.0f;
        return z;
    }
}
// act(z) with its derivative from the same erf / exponential / sigmoid (forward epilogue that saves the derivative)
__device__ __forceinline__ float tv_act_with_grad_rt(int act, float z, float& g) {
    if (act == TV_ACT_GELU) {
        float e;
        const float cdf = fmaf(0.5f, tv_erf_e(z * 0.70710678118654752f, e), 0.5f);
        g = fmaf(z * 0.3989422804014327f, e, cdf);
        return z * cdf;
    }
    if (act == TV_ACT_SILU) {
        const float s = tv_sigmoid(z);
        g = s * fmaf(z, 1.0f - s, 1.0f);
        return z * s;
    }
    g = 1.0f;
    return z;
}

// ---------------------------------------------------------------------------
// LDS-DMA through a buffer descriptor (SGPR base + 32-bit per-lane offset + scalar offset)
// ---------------------------------------------------------------------------
constexpr int OOB_OFFSET = (int)0x80000000u;  // beyond any < 2 GiB buffer: the LDS-DMA writes zeros (probed: tools/probes)

// lds[base + lane*16] = buf[voff + soff .. +16), or zeros if that range is outside [0, bytes).
// The descriptor type only exists in the device pass (a kernel body naming it loses its host stub), hence the guard.
__device__ __forceinline__ void buffer_load_lds16(const void* base, unsigned bytes, char* lds, int voff, int soff) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(__builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000),
                                             TV_LDS(lds), 16, voff, soff, 0, 0);
#endif
}
// the same with the non-temporal cache policy (aux bit 1): operand bytes a launch reads once
__device__ __forceinline__ void buffer_load_lds16_nt(const void* base, unsigned bytes, char* lds, int voff, int soff) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(__builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000),
                                             TV_LDS(lds), 16, voff, soff, 0, 2);
#endif
}

// ---------------------------------------------------------------------------
// wave / block reductions (wave = 64 lanes)
// ---------------------------------------------------------------------------
__device__ __forceinline__ float tv_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float tv_wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

static inline int tv_cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }
