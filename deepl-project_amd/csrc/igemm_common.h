// Shared by igemm_nt.hip (generic implicit-GEMM kernel, host dispatch, C entries) and conv3x3_halo.hip (the halo-tile kernel for
// 3x3 / stride-1 convolutions): launch arguments, fragment / swizzle helpers and the epilogues.  Two translation units
// because every kernel instantiation carries the unrolled register epilogues: they compile side by side.
#pragma once
// No implicit fma contraction in this file: every epilogue form (and every tile-shape instantiation of it) must round the
// same expression the same way -- a batch is required to equal its images run alone bit for bit, and the tile shape depends
// on the batch.  (With contraction left to the optimiser the RoPE rotation a*c - b*s came out as fma(a, c, -(b*s)) in one
// instantiation and fma(-b, s, a*c) in another: bf16 flips in 1 of ~10^4 elements.)  Explicit fmaf() calls are kept.
#pragma clang fp contract(off)
#include "common.h"

#include <type_traits>

namespace tvi {   // (named: the argument block and the host-side tuning state cross the two translation units)

struct IgemmArgs {
    const bf16* x;
    const bf16* w;
    const float* bias;
    const bf16* res;
    const bf16* aux;   // backward fusion: out = (acc + res) * act'(aux)
    bf16* pre;
    bf16* out;
    const char* zeros;
    int M, N, K;
    int batch, h_in, w_in, c_in, ldx;
    int h_out, w_out, ldo;
    int kh, kw, stride, pad, up_shift, dil_mask;
    int tiles_n, tiles_m, xcd_order;
    int sup_r, sup_c;  // xcd_order 2: the blocks an XCD runs side by side form sup_r x sup_c super-tiles (row tiles x column tiles)
    int tpb, nchunks;  // persistent column loop: a block walks `tpb` consecutive column tiles of its row tile (nchunks = tiles_n / tpb)
    int shuffle;
    int act;
    int aux_act;
    int pre_deriv;     // pre receives act'(pre-activation) (TV_ACT_SAVE_DERIV)
    int form;          // register-epilogue form of an EPI 0 launch (EF_*), 0 = generic LDS loop
    const float* rope; // QKV projection: RoPE table [tokens][4][32] applied to output columns < rope_cols (q and k thirds)
    int rope_tokens, rope_cols;
    int hw_shift, w_shift;  // log2 of h_out*w_out / w_out when both are powers of two, else -1
    // shuffled stores (store_shuffle 1 / 2): x / d for d = h_out*w_out, w_out, N/4 as  umulhi(x, m) >> s  (x < 2^31; m = 0: d = 1).
    // The LDS epilogue split every 16-byte chunk's (pixel, channel) with three integer divisions (~40 instructions each):
    // the pixel-shuffle / polyphase / parity layers spent up to 45 % of their time there (profiles/r03_kernel_experiments.txt item 19)
    unsigned dv_hw_m, dv_w_m, dv_cq_m;
    int dv_hw_s, dv_w_s, dv_cq_s;
    int store_nt;         // register epilogues: non-temporal stores (outputs beyond the Infinity Cache; tv_set_igemm_nt_threshold)
    unsigned out_bytes;   // extent of out (and of pre / res / aux: same shape) when below 2 GiB, else 0: the register epilogues
                          // address them through buffer descriptors with 32-bit offsets
    unsigned x_bytes, w_bytes;  // MODE 2: extents of the two buffers (< 2 GiB)
    // two-source rows (tv_igemm_nt_cat2, eight-phase loop only): K-steps t >= k1_steps come from x2 (row pitch ldx2), i.e. the
    // GEMM runs over the K-concatenation [x | x2] without the concatenated tensor existing
    const bf16* x2;
    unsigned x2_bytes;
    int ldx2, k1_steps;
};

constexpr int LDS_MAX = 160 * 1024;
extern int g_cfg_bm, g_cfg_bn, g_halo_ring;
extern bool g_halo_w4, g_xcd_order, g_epi_modes;
int epilogue_mode(const IgemmArgs& a);
int pick_tile(long long M, int N, bool allow256, bool allow192_256rows, int* bm);
int launch_halo(IgemmArgs& a, hipStream_t s);   // conv3x3_halo.hip; -1: the shape does not qualify
#ifdef TV_PROBE
int set_halo_probe(void* dev_buf);
#endif

}  // namespace tvi

namespace {
using namespace tvi;

#ifndef TV_NO_PIPE2
#define TV_NO_PIPE2 0
#endif
#ifndef TV_NO_LOADER_SPLIT
#define TV_NO_LOADER_SPLIT 1   // DMA from waves 0-3 only (8-wave halo tiles): measured slower, kept for A/B
#endif
#ifndef TV_HALO_VOFF_REGS
#define TV_HALO_VOFF_REGS 1
#endif
#ifndef TV_HALO_BURST
#define TV_HALO_BURST 1       // 256x256 halo tile (single fragment set): DMA burst after the barrier, measured +5 % over threading
#endif
#ifndef TV_GENERIC_BURST
#define TV_GENERIC_BURST 1    // same for the generic 256x256 tile (A/B on the 768-channel linear layers: +2-5 %)
#endif
#ifndef TV_PIPE_ALL_MAX
#define TV_PIPE_ALL_MAX 80   // fragment registers (both halves) up to which the generic kernel runs the pipelined loop
#endif
#ifndef TV_DMA_STAGGER
#define TV_DMA_STAGGER 0    // two code copies with shifted DMA slots for waves 0-3 / 4-7: measured -3 % (register pressure)
#endif
#ifndef TV_RD_THREAD
#define TV_RD_THREAD 0
#endif
#ifndef TV_SETPRIO
#define TV_SETPRIO 1           // waves 4-7 (the arbitration losers on every SIMD) run at priority 1
#endif
#ifndef TV_HALO_PP
#define TV_HALO_PP 1       // wave-group ping-pong main loop of the 8-wave halo tiles with a 3-deep weight ring (see conv3x3_halo_kernel):
#endif                     // +8-10 % on the 192-channel 3x3 layers over the lockstep pipelined loop (tools/probes/ab_lib.py)
#ifndef TV_P8_B0PF
#define TV_P8_B0PF 1       // eight-phase GEMM loop: B0 fragments resident for both of their quadrants, the next K-step's read in phase 4
#endif
#ifndef TV_HALO_A_NT
#define TV_HALO_A_NT 0     // halo pieces (activations, read ~1.1 times per launch) with the non-temporal policy: A/B in item 24
#endif
#ifndef TV_HALO_P8
#define TV_HALO_P8 0       // halo ping-pong loop in four phases of 12 MFMAs per tap step (one DMA piece per load section) instead of two: measured 8-11 % SLOWER (profiles/r03_kernel_experiments.txt item 15), kept for A/B
#endif
#ifndef TV_PP_NM
#define TV_PP_NM 0         // ping-pong: weight-slab DMA pieces (of B_IT per wave and step) issued from the MFMA phase
#endif
#ifndef TV_PP_DMA_FIRST
#define TV_PP_DMA_FIRST 0  // ping-pong: DMA pieces of a load phase before (1) or after (0) its fragment reads; 2 = threaded between them
#endif
#ifndef TV_GENERIC_DPHASE
#define TV_GENERIC_DPHASE 1   // generic pipelined loop: DMA slots staggered by wave through a run-time phase (scalar branches in the MFMA stream)
#endif
#ifndef TV_PERSIST_DRAIN
#define TV_PERSIST_DRAIN 1    // persistent column loop (off by default, tv_set_igemm_persist): 1 = open every tile with vmcnt(0) -- drains the previous tile's stores too, no hand-counted wait; 0 = the counted vmcnt(NST) form, for A/B
#endif
#ifndef TV_NO_PINGPONG
#define TV_NO_PINGPONG 1   // ping-pong main loop of the 8-wave tiles: measured, not (yet) a win -- see DESIGN.md
#endif

// Order of the output channels inside a wave's WTN-wide weight slab.  The MFMA is issued as D' = W_frag x A_frag^T: lane
// (fq, fi) ends up with 4 channels (operand rows 4 fq .. 4 fq + 3) of pixel fi per fragment.  Fragments come in pairs
// (2c, 2c+1) over a 32-channel block c; operand row r of fragment j is local channel  (j/2)*32 + (r/4)*8 + (j%2)*4 + r%4,
// so a lane's pair is 8 CONSECUTIVE channels (one 16-byte bf16 chunk) and the four lanes of a pixel cover the block's 64
// contiguous bytes: the epilogue stores straight from the registers (epilogue_direct).
__device__ __forceinline__ constexpr int bfrag_off(int j) { return (j >> 1) * 32 + (j & 1) * 4; }   // fragment j, operand row 0
__device__ __forceinline__ int bfrag_lane_row(int fi) { return (fi >> 2) * 8 + (fi & 3); }           // operand row fi of fragment 0
__device__ __forceinline__ int bfrag_reader(int rl) { return (((rl & 31) >> 3) << 2) | (rl & 3); }    // operand row that reads local channel rl

// f(integral_constant<int, I>) for I = I0 .. N-1: an unrolled loop whose index is usable as a template argument
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

__device__ __forceinline__ int fast_div(int x, unsigned m, int sh) {   // x / d for 0 <= x < 2^31 (host: igemm_fast_div)
    return m ? (int)(__umulhi((unsigned)x, m) >> sh) : x;
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// In-kernel phase timer (diagnostic builds only: -DTV_PROBE, tools/probes/igemm_phase_probe.py).  TV_T(i) adds the shader
// cycles since the previous mark to counter i; one block in 509 dumps its per-wave counters at the end.
#ifdef TV_PROBE
__device__ unsigned long long* g_probe_dev = nullptr;
#define TV_PROBE_DECL unsigned long long pr_c[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long pr_t = __builtin_amdgcn_s_memtime();
#define TV_T(i) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
                     const unsigned long long n__ = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
                     __builtin_amdgcn_sched_barrier(0); pr_c[i] += n__ - pr_t; pr_t = n__; } while (0)
#define TV_PROBE_DUMP(wave, lane) do { if (g_probe_dev && blockIdx.x % 509 == 0 && blockIdx.x / 509 < 16 && (lane) == 0) \
        for (int i__ = 0; i__ < 8; ++i__) g_probe_dev[((blockIdx.x / 509) * 8 + (wave)) * 8 + i__] = pr_c[i__]; } while (0)
#else
#define TV_PROBE_DECL
#define TV_T(i) do { } while (0)
#define TV_PROBE_DUMP(wave, lane) do { } while (0)
#endif

template <int BK>
__device__ __forceinline__ int swz_of(int i) {  // i: row index inside a 16-row fragment
    if constexpr (BK == 64)
        return (i >> 1) & 7;
    else
        return (0x78 >> (2 * ((i >> 2) & 3))) & 3;
}

// Epilogue shared by the tile kernels.  The accumulator layout scatters a row over lanes (8-byte pieces); stored
// directly the tile costs ~25 % of a K=1728 convolution (measured: K=64 launch 0.44 ms of 1.56 ms).  Instead every wave
// parks its tile (+bias) in its own slice of the now idle stage buffers and streams it out row by row, 16 bytes of bf16
// per lane: pre-activation store, activation, residual add and the output store are all full-line accesses.
// The tile is parked in FP32, half of its rows at a time: activation and residual add see the unrounded accumulator and
// every output is rounded to bf16 exactly once (a bf16 park rounded the pre-activation first: +20-40 % rel-L2 error on
// whole-model outputs, tests/precision_report.py).
// m_of_row(r) = output pixel index (b, oy, ox linearised) of wave-tile row r; the caller has synchronised the block.
#ifndef TV_EPI_LDS
#define TV_EPI_LDS 0   // 1: the round-1 epilogue (tile parked in LDS as fp32, streamed out row by row), kept for A/B timing
#endif

template <int WTM, int WTN>
constexpr int epilogue_lds_bytes(int nwaves) {
    return nwaves * (WTM >= 32 ? WTM / 2 : WTM) * (WTN * 4 + 16);
}

// EPI = 0: every option decided at run time inside the row loop (shuffled stores, saved pre-activation, activation,
//          activation gradient, ...).
// EPI = 1: plain row store + residual add, no activation (ffn_out / proj / ResBlock conv2 and every data gradient that
//          adds a second gradient of the same tensor).  ALL of a pass's residual loads are issued before the tile is
//          parked, so their latency runs under the LDS round trip instead of once per pair of row chunks: +3...37 % on
//          these layers (tools/probes/ab_epilogue.py).
// EPI = 2: activation gradient from a saved DERIVATIVE (aux_act == TV_ACT_DERIV), no residual: out = acc * aux, loads
//          issued early as in EPI 1: +2...22 %.  (With a residual as well -- two batches of loads in flight -- the form
//          measured -0...5 %, and with the erf / exp arithmetic of act'(pre-activation) inside -13...+5 %: EPI 0.)
// The lane's bias values (4 consecutive output channels per fragment column j), loaded BEFORE the block barrier that opens
// the epilogue: their latency then runs under the barrier wait instead of once per pass inside the park loop (a forward
// layer with a bias ran 5 % slower than the same kernel without: 2.28-2.33 vs 2.15-2.19 ms on the dominant shape).
template <int WTN>
__device__ __forceinline__ void load_bias(const IgemmArgs& p, int lane, int nw0, f32x4 (&bv)[WTN / 16]) {
    constexpr int NF = WTN / 16;
    const int fq = lane >> 4;
#pragma unroll
    for (int j = 0; j < NF; ++j) {
        const int n = nw0 + bfrag_off(j) + fq * 8;
        bv[j] = (p.bias && n < p.N) ? *(const f32x4*)(p.bias + n) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
}

// Register-layout epilogues.  A lane's accumulators of fragment row i are, per 32-channel block c, 8 CONSECUTIVE output
// channels of ONE pixel (bfrag_off): 16 bytes of bf16, and the common epilogues are elementwise on those chunks -- bias,
// activation (+ saved derivative), residual add, saved-derivative multiply -- so their arithmetic needs no row-major
// view.  Memory accesses do: the finished bf16 chunks (and the loaded residual / saved chunks, the other way) go through
// a cross-lane transposition (epi_to_lines / epi_from_lines) so that every load and store instruction covers 8 whole
// 128-byte lines with consecutive lanes on consecutive addresses.  A block without a line partner (WTN = 96: one of
// three) moves as 64-byte halves.
// Measured (tools/probes/k1_probe.py, K = 384 -> N = 1536, 24 tiles per CU): the LDS epilogue below (tile parked as fp32,
// streamed out by a row loop) costs 6.5-7.7 us per 256x256 tile, 6.1 us of it without any global store, against 5.2 us of
// MFMA time.  The register form is fully unrolled (accumulator indices must be static), so it exists only as COMPACT
// compile-time forms, one per common (activation, residual) combination: a single body with run-time flags unrolls to
// ~20 000 instructions, runs out of the instruction cache and is 20-35 % SLOWER than the LDS loop, which therefore stays
// as the generic form (RoPE, shuffled stores, saved pre-activations, ...).  Same-box A/B of the plain form against the
// LDS loop: K = 384 layers 1.22x, the dominant 3x3 convolution 1.07x forward / 1.10x data gradient (2.21 -> 2.01 ms =
// 1385 TFLOP/s); results bit-identical (same fp32 arithmetic, one rounding).
enum { EF_GENERIC = 0, EF_PLAIN = 1, EF_GELU_D = 2, EF_SILU_D = 3, EF_GELU = 4, EF_SILU = 5, EF_RES = 6, EF_DERIV = 7,
       EF_RES_DERIV = 8,   // (acc + residual) * saved derivative: the data gradient that joins two branches
       EF_ROPE = 9,        // QKV projection with the RoPE rotation of its q / k columns
       // register forms with a SHUFFLED store (round 3): pixel-shuffle (store_shuffle 1: DC paths, parity data gradient of the
       // stride-2 convolutions) and the phase-shuffled store of the polyphase upsampling convolution (store_shuffle 2)
       EF_PLAIN_S1 = 10, EF_RES_S1 = 11, EF_DERIV_S1 = 12, EF_SILU_D_S2 = 13, EF_PLAIN_S2 = 14,
       EF_RES2 = 15 };     // acc + residual + second residual (TV_ACT_ADD): two branch values join the stream in one fp32 sum

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void epi_pair_exchange(bf16x8& a, bf16x8& b) {   // an involution: lanes fi < 8 keep a, lanes fi >= 8 keep b
    const u32x4 x = __builtin_bit_cast(u32x4, a), y = __builtin_bit_cast(u32x4, b);
    u32x4 s, t;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        s[e] = __builtin_amdgcn_update_dpp(x[e], y[e], 0x128 /* row_ror:8 */, 0xF, 0xC, false);
        t[e] = __builtin_amdgcn_update_dpp(y[e], x[e], 0x128, 0xF, 0x3, false);
    }
    a = __builtin_bit_cast(bf16x8, s);
    b = __builtin_bit_cast(bf16x8, t);
}
// register layout <-> line layout.  Register layout: lane (fq, fi) holds blocks C0 (a) and C0+1 (b) of pixel fi, 16 bytes
// each at byte fq*16 of the block.  Line layout: lane l holds bytes (l & 7) * 16 of the 128-byte line of pixel l >> 3 (a)
// and of pixel 8 + (l >> 3) (b) -- consecutive lanes are consecutive addresses, which is what the memory pipeline
// coalesces (it merges neighbouring lanes only: with the 16-byte pieces of a line on lanes 8 or 16 apart the stores ran
// 30-40 % slower than through LDS).  Two steps: the fi ^ 8 exchange, then one ds_bpermute per dword (the LDS crossbar, no
// LDS memory).  `idx` = epi_line_index(lane).
__device__ __forceinline__ int epi_line_index(int lane) { return (((lane & 3) << 4) + (lane >> 3) + (((lane >> 2) & 1) << 3)) << 2; }
__device__ __forceinline__ void epi_to_lines(bf16x8& a, bf16x8& b, int idx) {
    epi_pair_exchange(a, b);
    u32x4 x = __builtin_bit_cast(u32x4, a), y = __builtin_bit_cast(u32x4, b);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        x[e] = __builtin_amdgcn_ds_bpermute(idx, x[e]);
        y[e] = __builtin_amdgcn_ds_bpermute(idx, y[e]);
    }
    a = __builtin_bit_cast(bf16x8, x);
    b = __builtin_bit_cast(bf16x8, y);
}
__device__ __forceinline__ void epi_from_lines(bf16x8& a, bf16x8& b, int idx) {
    u32x4 x = __builtin_bit_cast(u32x4, a), y = __builtin_bit_cast(u32x4, b);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        x[e] = __builtin_amdgcn_ds_permute(idx, x[e]);
        y[e] = __builtin_amdgcn_ds_permute(idx, y[e]);
    }
    a = __builtin_bit_cast(bf16x8, x);
    b = __builtin_bit_cast(bf16x8, y);
    epi_pair_exchange(a, b);
}
// one block alone: lane l holds bytes (l & 3) * 16 of the 64-byte half line of pixel l >> 2
__device__ __forceinline__ int epi_half_index(int lane) { return (((lane & 3) << 4) + (lane >> 2)) << 2; }
__device__ __forceinline__ void epi_to_half(bf16x8& a, int idx) {
    u32x4 x = __builtin_bit_cast(u32x4, a);
#pragma unroll
    for (int e = 0; e < 4; ++e) x[e] = __builtin_amdgcn_ds_bpermute(idx, x[e]);
    a = __builtin_bit_cast(bf16x8, x);
}
__device__ __forceinline__ void epi_from_half(bf16x8& a, int idx) {
    u32x4 x = __builtin_bit_cast(u32x4, a);
#pragma unroll
    for (int e = 0; e < 4; ++e) x[e] = __builtin_amdgcn_ds_permute(idx, x[e]);
    a = __builtin_bit_cast(bf16x8, x);
}

// RoPE2D of the reference on the unrounded projection (R/transvae/modules/attention.py:156-197): the chunk's 8 channels are
// pairs 4v .. 4v+3 of one head; out[2p] = a cos1 - b sin1, out[2p+1] = a sin2 + b cos2.  tb = table row of the token + pair.
__device__ __forceinline__ void epi_rope(float (&v)[8], const float* tb) {
    const f32x4 c1 = *(const f32x4*)(tb), s1 = *(const f32x4*)(tb + 32), c2 = *(const f32x4*)(tb + 64), s2 = *(const f32x4*)(tb + 96);
#pragma unroll
    for (int pr = 0; pr < 4; ++pr) {
        const float a = v[2 * pr], bb = v[2 * pr + 1];
        v[2 * pr] = a * c1[pr] - bb * s1[pr];
        v[2 * pr + 1] = a * s2[pr] + bb * c2[pr];
    }
}

// the elementwise part on one 8-channel chunk (v: accumulator + bias, fp32): the output chunk; `zd` the saved derivative.
// `ld`: the residual (EF_RES) or the saved derivative (EF_DERIV) of the same elements.
template <int FORM>
__device__ __forceinline__ bf16x8 epi_math(float (&v)[8], const bf16x8& ld, const bf16x8& ld2, bf16x8& zd) {
    if constexpr (FORM == EF_RES_DERIV) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (v[e] + (float)ld[e]) * (float)ld2[e];
    } else if constexpr (FORM == EF_RES2) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = v[e] + (float)ld2[e] + (float)ld[e];     // (branch values first, the stream last)
    } else if constexpr (FORM == EF_RES) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += (float)ld[e];
    } else if constexpr (FORM == EF_DERIV) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] *= (float)ld[e];
    } else if constexpr (FORM == EF_GELU_D || FORM == EF_SILU_D) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float g;
            v[e] = tv_act_with_grad<FORM == EF_GELU_D ? TV_ACT_GELU : TV_ACT_SILU>(v[e], g);
            zd[e] = (bf16)g;
        }
    } else if constexpr (FORM == EF_GELU || FORM == EF_SILU) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = tv_act<FORM == EF_GELU ? TV_ACT_GELU : TV_ACT_SILU>(v[e]);
    }
    bf16x8 z;
#pragma unroll
    for (int e = 0; e < 8; ++e) z[e] = (bf16)v[e];
    return z;
}

// SH = store_shuffle of the launch (compile time): where element (GEMM row m, GEMM column nx .. nx+7) lives in the output
// tensor -- the address arithmetic of the LDS loop, per 16-byte piece of a line (a 64-channel pair of blocks never straddles
// a phase: the host sends only N/4 % 64 == 0 here).  Residual / saved tensors have the output's layout: same offsets.
// TV_EPI_BUF (round 3): every load / store of the register epilogue goes through a buffer descriptor with a 32-bit byte offset
// per lane, masked pieces as OUT-OF-RANGE offsets (a store is dropped, a load returns zeros).  The pointer form predicated each
// access (`if (ok) store`): one basic block and an s_cbranch_execz per store, 64-bit address arithmetic per piece -- the rows
// of a tile went out strictly one after the other, every cross-lane transposition exposed in front of its store.  Without
// branches the body is one block: the scheduler runs the transpositions of later rows under earlier stores.  The host sends
// only tensors below 2 GiB here (out_bytes; larger ones keep the LDS loop).
#ifndef TV_EPI_BUF
#define TV_EPI_BUF 1
#endif
// Cache policy of the epilogue's stores (gfx950 aux bits: 1 sc0, 2 nt, 16 sc1).  Non-temporal: the output stream of a tile is written
// once and not read by this launch -- kept out of the way of the operands the launch re-reads through L2 / the Infinity Cache:
// linear layers -4...-7 % in isolation, bench +0.5-0.6 % (item 23; sc1: no change).  Chosen per launch (IgemmArgs.store_nt): an
// output small enough to stay in the 256 MiB Infinity Cache for its consumer keeps the default policy.
#ifndef TV_EPI_LOAD_AUX
#define TV_EPI_LOAD_AUX 0    // ... of the epilogue's residual / saved-derivative loads (read once)
#endif
typedef unsigned int epi_u32x4 __attribute__((__vector_size__(16)));
#ifdef TV_EXP_GN_EPI
__device__ float* g_gn_epi_buf = nullptr;   // timing experiment: per-wave GroupNorm partials [block][wave][256] (tv_set_gn_epi_probe)
#endif
__device__ __forceinline__ bf16x8 epi_bload(const void* base, unsigned bytes, int off) {
#if defined(__HIP_DEVICE_COMPILE__)
    const epi_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(__builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000), off, 0, TV_EPI_LOAD_AUX);
    return __builtin_bit_cast(bf16x8, v);
#else
    return bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
#endif
}
__device__ __forceinline__ void epi_bstore(void* base, unsigned bytes, int off, const bf16x8& z) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(epi_u32x4, z), __builtin_amdgcn_make_buffer_rsrc(base, 0, bytes, 0x00020000), off, 0, 0);
#endif
}
__device__ __forceinline__ void epi_bstore_nt(void* base, unsigned bytes, int off, const bf16x8& z) {   // non-temporal (aux bit 1)
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(epi_u32x4, z), __builtin_amdgcn_make_buffer_rsrc(base, 0, bytes, 0x00020000), off, 0, 2);
#endif
}

template <int WTM, int WTN, int FORM, int SH = 0, class RowMap>
__device__ __forceinline__ void epilogue_direct(const IgemmArgs& p, const f32x4 (&acc)[WTM / 16][WTN / 16], const f32x4 (&bv)[WTN / 16],
                                                int lane, int nw0, RowMap m_of_row) {
    constexpr int MF = WTM / 16, NF = WTN / 16, NC = NF / 2;
    static_assert(NF % 2 == 0, "a lane's channels must come in whole 8-channel chunks");
    constexpr bool LOADS = FORM == EF_RES || FORM == EF_DERIV || FORM == EF_RES_DERIV || FORM == EF_RES2;
    constexpr bool LOADS2 = FORM == EF_RES_DERIV || FORM == EF_RES2;
    constexpr bool SAVES = FORM == EF_GELU_D || FORM == EF_SILU_D;
    // fragment rows per batch: the loads of a whole batch are issued before its arithmetic (their latency runs once per
    // batch, not once per line; one load per line in flight measured 0.96-0.99x of the LDS form, which batches them)
    constexpr int IB = !LOADS ? 1 : (MF % 4 == 0 && !LOADS2 ? 4 : (MF % 2 == 0 ? 2 : 1));
    const int fq = lane >> 4, fi = lane & 15;
    const bf16* __restrict__ lsrc = FORM == EF_DERIV ? p.aux : p.res;
    const bf16* __restrict__ lsrc2 = p.aux;
    const int lidx = epi_line_index(lane), hidx = epi_half_index(lane);
    const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    // BYTE offset of element (GEMM row m, GEMM column nx) in the output tensor (residual / saved tensors share its layout), or
    // OOB_OFFSET for a masked piece
    auto out_off = [&](int m, int nx) -> int {
        const bool ok = m < p.M && nx < p.N;
        if constexpr (SH == 0) {
            return ok ? (m * p.ldo + nx) * 2 : OOB_OFFSET;
        } else {
            if (!ok) return OOB_OFFSET;
            const int hw = p.h_out * p.w_out;
            const int sb = fast_div(m, p.dv_hw_m, p.dv_hw_s);
            const int rr = m - sb * hw;
            const int sy = fast_div(rr, p.dv_w_m, p.dv_w_s), sx = rr - sy * p.w_out;
            const int cq = p.N >> 2;
            const int qs = fast_div(nx, p.dv_cq_m, p.dv_cq_s);
            const int c = nx - qs * cq;
            if constexpr (SH == 1) {
                const int pix = (sb * (2 * p.h_out) + 2 * sy + (qs >> 1)) * (2 * p.w_out) + 2 * sx + (qs & 1);
                return (pix * p.ldo + c) * 2;
            } else {   // polyphase: phase (py, px) of cell (sy, sx) is pixel (2 sy - py, 2 sx - px) of the [2(H-1), 2(W-1)] output grid
                const int Y = 2 * sy - (qs >> 1), X = 2 * sx - (qs & 1);
                const int H2 = 2 * (p.h_out - 1), W2 = 2 * (p.w_out - 1);
                if ((unsigned)Y >= (unsigned)H2 || (unsigned)X >= (unsigned)W2) return OOB_OFFSET;
                return (((sb * H2 + Y) * W2 + X) * p.ldo + c) * 2;
            }
        }
    };
    // (the pointer form of an access, kept behind TV_EPI_BUF = 0 for A/B: same offsets, predicated)
    auto ld = [&](const bf16* base, int off) -> bf16x8 {
#if TV_EPI_BUF
        return epi_bload(base, p.out_bytes, off);
#else
        return off != OOB_OFFSET ? *(const bf16x8*)((const char*)base + (unsigned)off) : zero8;
#endif
    };
    auto st = [&](bf16* base, int off, const bf16x8& z) {
#if TV_EPI_BUF
        if (p.store_nt) epi_bstore_nt(base, p.out_bytes, off, z);   // (wave-uniform)
        else epi_bstore(base, p.out_bytes, off, z);
#else
        if (off != OOB_OFFSET) *(bf16x8*)((char*)base + (unsigned)off) = z;
#endif
    };
#ifdef TV_EXP_GN_EPI
    constexpr bool GNX = SH == 0 && (FORM == EF_PLAIN || FORM == EF_RES) && NC == 3;
    [[maybe_unused]] float gs1[NC][8], gs2[NC][8];
    [[maybe_unused]] const float gn_piv = p.bias ? p.bias[0] : 0.f;   // (stand-in pivot: a per-(image, channel) value in the real thing)
    if constexpr (GNX) {
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int e = 0; e < 8; ++e) gs1[c][e] = gs2[c][e] = 0.f;
    }
#endif
    auto chunk_values = [&](int i, int c, float (&v)[8]) {
        const f32x4 lo = acc[i][2 * c] + bv[2 * c], hi = acc[i][2 * c + 1] + bv[2 * c + 1];
        v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3]; v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
    };
    // which blocks pair up into lines depends on where the wave's slab starts inside a 128-byte line (wave-uniform)
    auto run = [&](auto odd_c) {
        constexpr bool ODD = decltype(odd_c)::value;     // slab starts in the second half of a line
        constexpr int P0 = ODD ? 1 : 0, NP = (NC - P0) / 2;
        constexpr int S0 = ODD ? 0 : NC - 1;             // the block without a partner, if any
        constexpr bool SINGLE = P0 + 2 * NP < NC || ODD;
        static_for<0, MF / IB>([&](auto b_c) {
            constexpr int i0 = decltype(b_c)::value * IB;
            int o1[IB][NP > 0 ? NP : 1], o2[IB][NP > 0 ? NP : 1], os[IB];
            [[maybe_unused]] bf16x8 la[IB][NP > 0 ? NP : 1], lb[IB][NP > 0 ? NP : 1], ls[IB];
            [[maybe_unused]] bf16x8 la2[IB][NP > 0 ? NP : 1], lb2[IB][NP > 0 ? NP : 1], ls2[IB];
#pragma unroll
            for (int ii = 0; ii < IB; ++ii) {
                const int r1 = (i0 + ii) * 16 + (lane >> 3);               // line layout: my pixel in the first / second access
                const int m1 = m_of_row(r1), m2 = m_of_row(r1 + 8);
#pragma unroll
                for (int u = 0; u < NP; ++u) {
                    const int nx = nw0 + (P0 + 2 * u) * 32 + (lane & 7) * 8;   //          my channels
                    o1[ii][u] = out_off(m1, nx);
                    o2[ii][u] = out_off(m2, nx);
                    if constexpr (LOADS) {
                        la[ii][u] = ld(lsrc, o1[ii][u]);
                        lb[ii][u] = ld(lsrc, o2[ii][u]);
                    }
                    if constexpr (LOADS2) {
                        la2[ii][u] = ld(lsrc2, o1[ii][u]);
                        lb2[ii][u] = ld(lsrc2, o2[ii][u]);
                    }
                }
                if constexpr (SINGLE) {
                    const int ms = m_of_row((i0 + ii) * 16 + (lane >> 2));
                    const int nx = nw0 + S0 * 32 + (lane & 3) * 8;
                    os[ii] = out_off(ms, nx);
                    if constexpr (LOADS) ls[ii] = ld(lsrc, os[ii]);
                    if constexpr (LOADS2) ls2[ii] = ld(lsrc2, os[ii]);
                }
            }
#pragma unroll
            for (int ii = 0; ii < IB; ++ii) {
                const int i = i0 + ii;
#pragma unroll
                for (int u = 0; u < NP; ++u) {
                    const int C0 = P0 + 2 * u;
                    bf16x8 ra = zero8, rb = zero8, ra2 = zero8, rb2 = zero8, da = zero8, db = zero8;
                    if constexpr (LOADS) {
                        ra = la[ii][u];
                        rb = lb[ii][u];
                        epi_from_lines(ra, rb, lidx);    // -> blocks C0 / C0+1 of MY pixel (register layout)
                    }
                    if constexpr (LOADS2) {
                        ra2 = la2[ii][u];
                        rb2 = lb2[ii][u];
                        epi_from_lines(ra2, rb2, lidx);
                    }
                    float va[8], vb[8];
                    chunk_values(i, C0, va);
                    chunk_values(i, C0 + 1, vb);
                    if constexpr (FORM == EF_ROPE) {   // (whole 32-channel blocks are inside or outside the rotated columns)
                        const float* tb = p.rope + (size_t)(m_of_row(i * 16 + fi) % p.rope_tokens) * 128;
                        const int na = nw0 + C0 * 32 + fq * 8;
                        if (nw0 + C0 * 32 < p.rope_cols) epi_rope(va, tb + ((na & 63) >> 1));
                        if (nw0 + C0 * 32 + 32 < p.rope_cols) epi_rope(vb, tb + (((na + 32) & 63) >> 1));
                    }
                    bf16x8 za = epi_math<FORM>(va, ra, ra2, da);
                    bf16x8 zb = epi_math<FORM>(vb, rb, rb2, db);
                    if constexpr (SAVES) {
                        epi_to_lines(da, db, lidx);
                        st(p.pre, o1[ii][u], da);
                        st(p.pre, o2[ii][u], db);
                    }
#ifdef TV_EXP_GN_EPI   // (timing experiment: GroupNorm statistics of the OUTPUT in the producing convolution's epilogue)
                    if constexpr (GNX) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float xa = (float)za[e] - gn_piv, xb = (float)zb[e] - gn_piv;
                            gs1[C0][e] += xa; gs2[C0][e] = fmaf(xa, xa, gs2[C0][e]);
                            gs1[C0 + 1][e] += xb; gs2[C0 + 1][e] = fmaf(xb, xb, gs2[C0 + 1][e]);
                        }
                    }
#endif
                    epi_to_lines(za, zb, lidx);
#ifdef TV_ABL_NO_STORE
                    if (za[0] == (bf16)123.0f)   // (keeps the values live; practically never true)
#endif
                    {
                        st(p.out, o1[ii][u], za);
                        st(p.out, o2[ii][u], zb);
                    }
                }
                if constexpr (SINGLE) {
                    bf16x8 r = zero8, r2 = zero8, d = zero8;
                    if constexpr (LOADS) {
                        r = ls[ii];
                        epi_from_half(r, hidx);
                    }
                    if constexpr (LOADS2) {
                        r2 = ls2[ii];
                        epi_from_half(r2, hidx);
                    }
                    float v[8];
                    chunk_values(i, S0, v);
                    if constexpr (FORM == EF_ROPE) {
                        const int na = nw0 + S0 * 32 + fq * 8;
                        if (nw0 + S0 * 32 < p.rope_cols)
                            epi_rope(v, p.rope + (size_t)(m_of_row(i * 16 + fi) % p.rope_tokens) * 128 + ((na & 63) >> 1));
                    }
                    bf16x8 z = epi_math<FORM>(v, r, r2, d);
                    if constexpr (SAVES) {
                        epi_to_half(d, hidx);
                        st(p.pre, os[ii], d);
                    }
#ifdef TV_EXP_GN_EPI
                    if constexpr (GNX) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float xa = (float)z[e] - gn_piv;
                            gs1[S0][e] += xa; gs2[S0][e] = fmaf(xa, xa, gs2[S0][e]);
                        }
                    }
#endif
                    epi_to_half(z, hidx);
#ifdef TV_ABL_NO_STORE
                    if (z[0] == (bf16)123.0f)
#endif
                    st(p.out, os[ii], z);
                }
            }
        });
    };
    if constexpr (WTN % 64 == 0) {
        run(std::false_type{});
    } else {
        if (nw0 & 32) run(std::true_type{});
        else run(std::false_type{});
    }
#ifdef TV_EXP_GN_EPI
    if constexpr (GNX) {
        // the wave's 64 pixels per channel: the 16 lanes of a DPP row (same fq, pixels fi = 0..15) hold 48 partial values each
        // (3 blocks x 8 channels x {sum, sum of squares}); reduce-scatter over the row in four halving steps (a lane keeps the
        // half of the values its bit selects and adds the partner's copy of that half): 3 finished values per lane.
        float v[48];
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int e = 0; e < 8; ++e) { v[c * 16 + e] = gs1[c][e]; v[c * 16 + 8 + e] = gs2[c][e]; }
        int n = 48;
#pragma unroll
        for (int bit = 8; bit >= 1; bit >>= 1) {
            const bool up = (lane & bit) != 0;
            const int h = n / 2;
#pragma unroll
            for (int k = 0; k < 24; ++k) {
                if (k < h) {
                    const float keep = up ? v[h + k] : v[k], send = up ? v[k] : v[h + k];
                    v[k] = keep + __shfl_xor(send, bit, 64);
                }
            }
            n = h;
        }
        // 3 values per lane; 64 lanes x 3 = the wave's 192 statistics -> its row of the partials buffer (16-byte aligned rows)
        if (g_gn_epi_buf) {
            float* dst = g_gn_epi_buf + ((size_t)blockIdx.x * 8 + (threadIdx.x >> 6)) * 256 + lane * 4;
            *(f32x4*)dst = f32x4{v[0], v[1], v[2], 0.f};
        }
    }
#endif
}

template <int WTM, int WTN, int EPI, class RowMap>
__device__ __forceinline__ void epilogue_lds(const IgemmArgs& p, const f32x4 (&acc)[WTM / 16][WTN / 16], const f32x4 (&bv)[WTN / 16], char* smem,
                                             int wave, int lane, int nw0, RowMap m_of_row) {
    constexpr int MF = WTM / 16, NF = WTN / 16;
    constexpr int PASSES = MF >= 2 ? 2 : 1, MFP = MF / PASSES, RH = MFP * 16;   // rows per pass
    static_assert(MF % PASSES == 0, "wave tile rows");
    constexpr int ERS = WTN * 4 + 16;          // LDS row stride of the parked fp32 rows (16 B pad: bank spread)
    constexpr int EB = RH * ERS;               // bytes per wave
    constexpr int CPW = WTN / 8;               // 16-byte OUTPUT chunks (8 channels) per tile row
    const int fi = lane & 15, fq = lane >> 4;
    char* ebuf = smem + wave * EB;
    const int hw = p.h_out * p.w_out;
    const int cq = p.N >> 2;
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
        constexpr int ITER = (RH * CPW + 63) / 64;
        [[maybe_unused]] unsigned eoff[EPI ? ITER : 1];   // in 16-byte units (host checks the range); ~0u: outside the tensor
        [[maybe_unused]] bf16x8 erv[EPI ? ITER : 1];   // residual (EPI 1) or saved derivative (EPI 2)
        if constexpr (EPI != 0) {
#pragma unroll
            for (int k = 0; k < ITER; ++k) {
                const int idx = lane + 64 * k;
                const int rl = idx / CPW, c8 = idx - rl * CPW;
                const int m = m_of_row(ps * RH + rl);
                const int n = nw0 + c8 * 8;
                const bool ok = idx < RH * CPW && m < p.M && n < p.N;
                eoff[k] = ok ? (unsigned)(((long long)m * p.ldo + n) >> 3) : ~0u;
            }
            const bf16* __restrict__ esrc = EPI == 2 ? p.aux : p.res;
#pragma unroll
            for (int k = 0; k < ITER; ++k)
                if (eoff[k] != ~0u) erv[k] = *(const bf16x8*)(esrc + (size_t)eoff[k] * 8);
        }
#pragma unroll
        for (int ii = 0; ii < MFP; ++ii) {
            const int i = ps * MFP + ii;
#pragma unroll
            for (int j = 0; j < NF; ++j) {
                const int nl = bfrag_off(j) + fq * 8;
                const f32x4 v = acc[i][j] + bv[j];
                *(f32x4*)(ebuf + (ii * 16 + fi) * ERS + nl * 4) = v;
            }
        }
        if constexpr (EPI != 0) {
#pragma unroll
            for (int k = 0; k < ITER; ++k) {
                if (eoff[k] == ~0u) continue;
                const int idx = lane + 64 * k;
                const int rl = idx / CPW, c8 = idx - rl * CPW;
                const f32x4 v0 = *(const f32x4*)(ebuf + rl * ERS + c8 * 32);
                const f32x4 v1 = *(const f32x4*)(ebuf + rl * ERS + c8 * 32 + 16);
                float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = EPI == 2 ? v[e] * (float)erv[k][e] : v[e] + (float)erv[k][e];
                bf16x8 z;
#pragma unroll
                for (int e = 0; e < 8; ++e) z[e] = (bf16)v[e];
                *(bf16x8*)(p.out + (size_t)eoff[k] * 8) = z;
            }
            continue;
        }
        // (same wave writes and reads: LDS executes a wave's accesses in order, no barrier needed)
#pragma unroll 2
        for (int idx = lane; idx < RH * CPW; idx += 64) {
            const int rl = idx / CPW, c8 = idx - rl * CPW;
            const int r = ps * RH + rl;
            const int m = m_of_row(r);
            const int n = nw0 + c8 * 8;
            if (m >= p.M || n >= p.N) continue;
            size_t off;
            if (p.shuffle == 2) {   // polyphase upsampling conv: grid (H+1) x (W+1), phase (py, px) of cell (sy, sx) is pixel
                const int sb = fast_div(m, p.dv_hw_m, p.dv_hw_s);   //   (2*sy - py, 2*sx - px) of the [2H, 2W] output; cells on the rim
                const int rr = m - sb * hw;                         //   have phases that fall outside
                const int sy = fast_div(rr, p.dv_w_m, p.dv_w_s), sx = rr - sy * p.w_out;
                const int qs = fast_div(n, p.dv_cq_m, p.dv_cq_s);
                const int c = n - qs * cq;
                const int Y = 2 * sy - (qs >> 1), X = 2 * sx - (qs & 1);
                const int H2 = 2 * (p.h_out - 1), W2 = 2 * (p.w_out - 1);
                if ((unsigned)Y >= (unsigned)H2 || (unsigned)X >= (unsigned)W2) continue;
                off = (((size_t)sb * H2 + Y) * W2 + X) * p.ldo + c;
            } else if (p.shuffle) {
                const int sb = fast_div(m, p.dv_hw_m, p.dv_hw_s);
                const int rr = m - sb * hw;
                const int sy = fast_div(rr, p.dv_w_m, p.dv_w_s), sx = rr - sy * p.w_out;
                const int qs = fast_div(n, p.dv_cq_m, p.dv_cq_s);
                const int c = n - qs * cq;
                const size_t pix = ((size_t)sb * (2 * p.h_out) + 2 * sy + (qs >> 1)) * (2 * p.w_out) + 2 * sx + (qs & 1);
                off = pix * p.ldo + c;
            } else {
                off = (size_t)m * p.ldo + n;
            }
            const f32x4 v0 = *(const f32x4*)(ebuf + rl * ERS + c8 * 32);
            const f32x4 v1 = *(const f32x4*)(ebuf + rl * ERS + c8 * 32 + 16);
            float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
            if (p.rope && n < p.rope_cols) epi_rope(v, p.rope + (size_t)(m % p.rope_tokens) * 128 + ((n & 63) >> 1));
            bf16x8 z;
            const bool save_deriv = p.pre && p.pre_deriv;
            if (save_deriv) {   // save act'(pre-activation): the backward epilogue then is one multiply per element
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float g;
                    v[e] = tv_act_with_grad_rt(p.act, v[e], g);
                    z[e] = (bf16)g;
                }
                *(bf16x8*)(p.pre + off) = z;
            } else if (p.pre) {
#pragma unroll
                for (int e = 0; e < 8; ++e) z[e] = (bf16)v[e];
                *(bf16x8*)(p.pre + off) = z;
            }
            if (p.aux) {  // gradient w.r.t. a pre-activation: (acc + residual gradient) * act'(saved pre-activation)
                const bf16x8 av = *(const bf16x8*)(p.aux + off);
                if (p.res) {
                    const bf16x8 rv = *(const bf16x8*)(p.res + off);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += (float)rv[e];
                }
                if (p.aux_act == TV_ACT_ADD) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += (float)av[e];
                } else if (p.aux_act == TV_ACT_DERIV) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] *= (float)av[e];
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] *= tv_act_grad_rt(p.aux_act, (float)av[e]);
                }
            } else if (p.act != TV_ACT_NONE || p.res) {
                if (!save_deriv) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = tv_act_rt(p.act, v[e]);
                }
                if (p.res) {
                    const bf16x8 rv = *(const bf16x8*)(p.res + off);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += (float)rv[e];
                }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) z[e] = (bf16)v[e];
#ifdef TV_ABL_NO_STORE
            if (z[0] == (bf16)123.0f)   // (keeps the value live; practically never true)
#endif
            *(bf16x8*)(p.out + off) = z;
        }
    }
}

// EPI (compile time) 1 / 2: the residual-add / derivative-multiply launches; EPI 0: p.form picks a compact register form
// or the generic LDS loop.  The caller has NOT synchronised the block: only the LDS form needs every wave to be done with
// the stage buffers (it parks the tile in them), the register forms let early waves start storing.
// REGFORMS: instantiate the register forms (the production kernels: buffer-descriptor DMA; the bring-up / >= 2 GiB staging
// modes keep the one LDS loop -- every register form is a fully unrolled body per kernel, i.e. compile time)
template <int WTM, int WTN, int EPI, bool REGFORMS, class RowMap>
__device__ __forceinline__ void epilogue(const IgemmArgs& p, const f32x4 (&acc)[WTM / 16][WTN / 16], const f32x4 (&bv)[WTN / 16], char* smem,
                                         int wave, int lane, int nw0, RowMap m_of_row) {
    constexpr bool REG = REGFORMS && !TV_EPI_LDS && (WTN / 16) % 2 == 0;
    if constexpr (REG && EPI == 1) {
        epilogue_direct<WTM, WTN, EF_RES>(p, acc, bv, lane, nw0, m_of_row);
    } else if constexpr (REG && EPI == 2) {
        epilogue_direct<WTM, WTN, EF_DERIV>(p, acc, bv, lane, nw0, m_of_row);
    } else {
        if constexpr (REG && EPI == 0) {
            switch (p.form) {   // (wave-uniform)
                case EF_PLAIN: epilogue_direct<WTM, WTN, EF_PLAIN>(p, acc, bv, lane, nw0, m_of_row); return;
                case EF_GELU_D: epilogue_direct<WTM, WTN, EF_GELU_D>(p, acc, bv, lane, nw0, m_of_row); return;
                case EF_SILU_D: epilogue_direct<WTM, WTN, EF_SILU_D>(p, acc, bv, lane, nw0, m_of_row); return;
                case EF_GELU: epilogue_direct<WTM, WTN, EF_GELU>(p, acc, bv, lane, nw0, m_of_row); return;
                case EF_SILU: epilogue_direct<WTM, WTN, EF_SILU>(p, acc, bv, lane, nw0, m_of_row); return;
                case EF_RES_DERIV: epilogue_direct<WTM, WTN, EF_RES_DERIV>(p, acc, bv, lane, nw0, m_of_row); return;
                case EF_RES2: epilogue_direct<WTM, WTN, EF_RES2>(p, acc, bv, lane, nw0, m_of_row); return;
                case EF_ROPE: epilogue_direct<WTM, WTN, EF_ROPE>(p, acc, bv, lane, nw0, m_of_row); return;
                case EF_PLAIN_S1: epilogue_direct<WTM, WTN, EF_PLAIN, 1>(p, acc, bv, lane, nw0, m_of_row); return;
                case EF_RES_S1: epilogue_direct<WTM, WTN, EF_RES, 1>(p, acc, bv, lane, nw0, m_of_row); return;
                case EF_DERIV_S1: epilogue_direct<WTM, WTN, EF_DERIV, 1>(p, acc, bv, lane, nw0, m_of_row); return;
                case EF_SILU_D_S2: epilogue_direct<WTM, WTN, EF_SILU_D, 2>(p, acc, bv, lane, nw0, m_of_row); return;
                case EF_PLAIN_S2: epilogue_direct<WTM, WTN, EF_PLAIN, 2>(p, acc, bv, lane, nw0, m_of_row); return;
                default: break;
            }
        }
        __syncthreads();                           // every wave is done reading the stage buffers
        epilogue_lds<WTM, WTN, EPI>(p, acc, bv, smem, wave, lane, nw0, m_of_row);
    }
}

}  // namespace
