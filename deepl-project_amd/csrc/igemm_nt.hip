// Implicit-GEMM "NT" kernel for gfx950: out[m][n] = sum_k A[m][k] * W[n][k]
//
//   m : output pixel (b, oy, ox)           M = batch*h_out*w_out
//   n : output channel                      N = c_out
//   k : (ky, kx, ci)                        K = kh*kw*c_in, both operands K-contiguous
//
// A rows are gathered on the fly from the NHWC activation tensor (3x3 / 1x1 / 2x2-stride-2
// taps, stride, nearest-x2 upsample folded into the index, zero-dilation for the data
// gradient of a stride-2 conv), so one kernel serves nn.Linear, every Conv2d of the path
// and their data gradients.
//
// Structure: 128- or 256-row tiles, 4 or 8 waves, bf16 v_mfma_f32_16x16x32, operands staged
// global->LDS with 16-byte LDS-DMA (buffer_load ... lds; global_load_lds for tensors >= 2 GiB) into a ring
// of STAGES buffers, counted s_waitcnt vmcnt(N) and raw s_barriers so the DMA spans barriers
// (cdna_hip_programming.md section 5, "Pipelining across barriers", T3/T4).  The main loop is register
// pipelined: two half-step fragment sets per wave, the barrier between the halves of a K-step (see PIPE2).
// The LDS image is lane-linear (DMA constraint); bank conflicts are removed by XOR-swizzling the 16-byte
// chunk index on the SOURCE address and on the ds_read_b128 address (rule 21).
// 3x3 stride-1 convolutions take conv3x3_halo_kernel below (spatial tiles, halo staged once per chunk).
//
// The MFMA is issued as D' = W_frag x A_frag^T so that every lane ends up with 4*NF
// consecutive output channels of one pixel: bias / activation / residual / store work on
// contiguous channel runs.
#include "common.h"

#include <type_traits>

namespace {

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
    int shuffle;
    int act;
    int aux_act;
    int pre_deriv;     // pre receives act'(pre-activation) (TV_ACT_SAVE_DERIV)
    int form;          // register-epilogue form of an EPI 0 launch (EF_*), 0 = generic LDS loop
    const float* rope; // QKV projection: RoPE table [tokens][4][32] applied to output columns < rope_cols (q and k thirds)
    int rope_tokens, rope_cols;
    int hw_shift, w_shift;  // log2 of h_out*w_out / w_out when both are powers of two, else -1
    unsigned x_bytes, w_bytes;  // MODE 2: extents of the two buffers (< 2 GiB)
};

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
#ifndef TV_PP_NM
#define TV_PP_NM 0         // ping-pong: weight-slab DMA pieces (of B_IT per wave and step) issued from the MFMA phase
#endif
#ifndef TV_PP_DMA_FIRST
#define TV_PP_DMA_FIRST 0  // ping-pong: DMA pieces of a load phase before (1) or after (0) its fragment reads; 2 = threaded between them
#endif
#ifndef TV_GENERIC_DPHASE
#define TV_GENERIC_DPHASE 1   // generic pipelined loop: DMA slots staggered by wave through a run-time phase (scalar branches in the MFMA stream)
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
enum { EF_GENERIC = 0, EF_PLAIN = 1, EF_GELU_D = 2, EF_SILU_D = 3, EF_GELU = 4, EF_SILU = 5, EF_RES = 6, EF_DERIV = 7 };

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

// the elementwise part on one 8-channel chunk (v: accumulator + bias, fp32): the output chunk; `zd` the saved derivative.
// `ld`: the residual (EF_RES) or the saved derivative (EF_DERIV) of the same elements.
template <int FORM>
__device__ __forceinline__ bf16x8 epi_math(float (&v)[8], const bf16x8& ld, bf16x8& zd) {
    if constexpr (FORM == EF_RES) {
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

template <int WTM, int WTN, int FORM, class RowMap>
__device__ __forceinline__ void epilogue_direct(const IgemmArgs& p, const f32x4 (&acc)[WTM / 16][WTN / 16], const f32x4 (&bv)[WTN / 16],
                                                int lane, int nw0, RowMap m_of_row) {
    constexpr int MF = WTM / 16, NF = WTN / 16, NC = NF / 2;
    static_assert(NF % 2 == 0, "a lane's channels must come in whole 8-channel chunks");
    constexpr bool LOADS = FORM == EF_RES || FORM == EF_DERIV;
    constexpr bool SAVES = FORM == EF_GELU_D || FORM == EF_SILU_D;
    // fragment rows per batch: the loads of a whole batch are issued before its arithmetic (their latency runs once per
    // batch, not once per line; one load per line in flight measured 0.96-0.99x of the LDS form, which batches them)
    constexpr int IB = !LOADS ? 1 : (MF % 4 == 0 ? 4 : (MF % 2 == 0 ? 2 : 1));
    const int fq = lane >> 4;
    const bf16* __restrict__ lsrc = FORM == EF_DERIV ? p.aux : p.res;
    const int lidx = epi_line_index(lane), hidx = epi_half_index(lane);
    const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
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
            size_t o1[IB][NP > 0 ? NP : 1], o2[IB][NP > 0 ? NP : 1], os[IB];
            bool k1[IB][NP > 0 ? NP : 1], k2[IB][NP > 0 ? NP : 1], ks[IB];
            [[maybe_unused]] bf16x8 la[IB][NP > 0 ? NP : 1], lb[IB][NP > 0 ? NP : 1], ls[IB];
#pragma unroll
            for (int ii = 0; ii < IB; ++ii) {
                const int r1 = (i0 + ii) * 16 + (lane >> 3);               // line layout: my pixel in the first / second access
                const int m1 = m_of_row(r1), m2 = m_of_row(r1 + 8);
#pragma unroll
                for (int u = 0; u < NP; ++u) {
                    const int nx = nw0 + (P0 + 2 * u) * 32 + (lane & 7) * 8;   //          my channels
                    k1[ii][u] = m1 < p.M && nx < p.N;
                    k2[ii][u] = m2 < p.M && nx < p.N;
                    o1[ii][u] = k1[ii][u] ? (size_t)m1 * p.ldo + nx : 0;
                    o2[ii][u] = k2[ii][u] ? (size_t)m2 * p.ldo + nx : 0;
                    if constexpr (LOADS) {
                        la[ii][u] = k1[ii][u] ? *(const bf16x8*)(lsrc + o1[ii][u]) : zero8;
                        lb[ii][u] = k2[ii][u] ? *(const bf16x8*)(lsrc + o2[ii][u]) : zero8;
                    }
                }
                if constexpr (SINGLE) {
                    const int ms = m_of_row((i0 + ii) * 16 + (lane >> 2));
                    const int nx = nw0 + S0 * 32 + (lane & 3) * 8;
                    ks[ii] = ms < p.M && nx < p.N;
                    os[ii] = ks[ii] ? (size_t)ms * p.ldo + nx : 0;
                    if constexpr (LOADS) ls[ii] = ks[ii] ? *(const bf16x8*)(lsrc + os[ii]) : zero8;
                }
            }
#pragma unroll
            for (int ii = 0; ii < IB; ++ii) {
                const int i = i0 + ii;
#pragma unroll
                for (int u = 0; u < NP; ++u) {
                    const int C0 = P0 + 2 * u;
                    bf16x8 ra = zero8, rb = zero8, da = zero8, db = zero8;
                    if constexpr (LOADS) {
                        ra = la[ii][u];
                        rb = lb[ii][u];
                        epi_from_lines(ra, rb, lidx);    // -> blocks C0 / C0+1 of MY pixel (register layout)
                    }
                    float va[8], vb[8];
                    chunk_values(i, C0, va);
                    chunk_values(i, C0 + 1, vb);
                    bf16x8 za = epi_math<FORM>(va, ra, da);
                    bf16x8 zb = epi_math<FORM>(vb, rb, db);
                    if constexpr (SAVES) {
                        epi_to_lines(da, db, lidx);
                        if (k1[ii][u]) *(bf16x8*)(p.pre + o1[ii][u]) = da;
                        if (k2[ii][u]) *(bf16x8*)(p.pre + o2[ii][u]) = db;
                    }
                    epi_to_lines(za, zb, lidx);
#ifdef TV_ABL_NO_STORE
                    if (za[0] == (bf16)123.0f)   // (keeps the values live; practically never true)
#endif
                    {
                        if (k1[ii][u]) *(bf16x8*)(p.out + o1[ii][u]) = za;
                        if (k2[ii][u]) *(bf16x8*)(p.out + o2[ii][u]) = zb;
                    }
                }
                if constexpr (SINGLE) {
                    bf16x8 r = zero8, d = zero8;
                    if constexpr (LOADS) {
                        r = ls[ii];
                        epi_from_half(r, hidx);
                    }
                    float v[8];
                    chunk_values(i, S0, v);
                    bf16x8 z = epi_math<FORM>(v, r, d);
                    if constexpr (SAVES) {
                        epi_to_half(d, hidx);
                        if (ks[ii]) *(bf16x8*)(p.pre + os[ii]) = d;
                    }
                    epi_to_half(z, hidx);
#ifdef TV_ABL_NO_STORE
                    if (z[0] == (bf16)123.0f)
#endif
                    if (ks[ii]) *(bf16x8*)(p.out + os[ii]) = z;
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
                const int sb = m / hw;                      //   (2*sy - py, 2*sx - px) of the [2H, 2W] output; cells on the rim
                const int rr = m - sb * hw;                 //   have phases that fall outside
                const int sy = rr / p.w_out, sx = rr - sy * p.w_out;
                const int qs = n / cq;
                const int c = n - qs * cq;
                const int Y = 2 * sy - (qs >> 1), X = 2 * sx - (qs & 1);
                const int H2 = 2 * (p.h_out - 1), W2 = 2 * (p.w_out - 1);
                if ((unsigned)Y >= (unsigned)H2 || (unsigned)X >= (unsigned)W2) continue;
                off = (((size_t)sb * H2 + Y) * W2 + X) * p.ldo + c;
            } else if (p.shuffle) {
                const int sb = m / hw;
                const int rr = m - sb * hw;
                const int sy = rr / p.w_out, sx = rr - sy * p.w_out;
                const int qs = n / cq;
                const int c = n - qs * cq;
                const size_t pix = ((size_t)sb * (2 * p.h_out) + 2 * sy + (qs >> 1)) * (2 * p.w_out) + 2 * sx + (qs & 1);
                off = pix * p.ldo + c;
            } else {
                off = (size_t)m * p.ldo + n;
            }
            const f32x4 v0 = *(const f32x4*)(ebuf + rl * ERS + c8 * 32);
            const f32x4 v1 = *(const f32x4*)(ebuf + rl * ERS + c8 * 32 + 16);
            float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
            if (p.rope && n < p.rope_cols) {
                // RoPE2D of the reference on the unrounded projection (R/transvae/modules/attention.py:156-197): the lane's 8
                // channels are pairs 4v .. 4v+3 of one head; out[2p] = a cos1 - b sin1, out[2p+1] = a sin2 + b cos2
                const float* tb = p.rope + (size_t)(m % p.rope_tokens) * 128 + ((n & 63) >> 1);
                const f32x4 c1 = *(const f32x4*)(tb), s1 = *(const f32x4*)(tb + 32), c2 = *(const f32x4*)(tb + 64), s2 = *(const f32x4*)(tb + 96);
#pragma unroll
                for (int pr = 0; pr < 4; ++pr) {
                    const float a = v[2 * pr], bb = v[2 * pr + 1];
                    v[2 * pr] = a * c1[pr] - bb * s1[pr];
                    v[2 * pr + 1] = a * s2[pr] + bb * c2[pr];
                }
            }
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
                if (p.aux_act == TV_ACT_DERIV) {
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
template <int WTM, int WTN, int EPI, class RowMap>
__device__ __forceinline__ void epilogue(const IgemmArgs& p, const f32x4 (&acc)[WTM / 16][WTN / 16], const f32x4 (&bv)[WTN / 16], char* smem,
                                         int wave, int lane, int nw0, RowMap m_of_row) {
    constexpr bool REG = !TV_EPI_LDS && (WTN / 16) % 2 == 0;
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
                default: break;
            }
        }
        __syncthreads();                           // every wave is done reading the stage buffers
        epilogue_lds<WTM, WTN, EPI>(p, acc, bv, smem, wave, lane, nw0, m_of_row);
    }
}

// MODE 0: register-staged loads + ds_write (bring-up / debugging)
// MODE 1: global_load_lds with 64-bit per-lane addresses (tensors >= 2 GiB)
// MODE 2: buffer_load ... lds: SGPR descriptor + 32-bit per-lane offset fixed per tap + scalar K offset; padding
//         = out-of-range offset (the hardware writes zeros) -- no vector ALU work per DMA in the steady state
// waves per SIMD the register allocation must leave room for: what the LDS footprint allows, at most 2 (the second
// argument of __launch_bounds__; without it the compiler spends registers freely and the 4-wave tiles lose a block per CU)
template <int BM, int BN, int NW, int BK, int STAGES>
constexpr int igemm_min_waves() {
    const int ring = STAGES * (BM + BN) * BK * 2 + (STAGES > 2 ? 1024 : 0);
    const int epi = BM / 2 * (BN * 4 + 64);   // (fp32 park, half the rows at a time; upper bound over the wave grids)
    const int lds = ring > epi ? ring : epi;
    const int blocks = 160 * 1024 / lds;
    const int w = blocks * NW / 4;
    return w >= 2 ? 2 : 1;
}

template <int BM, int BN, int WGM, int WGN, int BK, int STAGES, int MODE, int EPI>
__global__ __launch_bounds__(WGM* WGN * 64, (igemm_min_waves<BM, BN, WGM * WGN, BK, STAGES>())) void igemm_nt_kernel(const IgemmArgs p) {
    constexpr bool DMA = MODE != 0, BUF = MODE == 2;
    constexpr int NW = WGM * WGN;
    constexpr int CPR = BK / 8;     // 16-byte chunks per tile row
    constexpr int RPI = 64 / CPR;   // tile rows covered by one wave-wide 1 KiB piece
    constexpr int A_INSTR = BM / RPI, B_INSTR = BN / RPI;
    constexpr int A_IT = (A_INSTR + NW - 1) / NW, B_IT = (B_INSTR + NW - 1) / NW;
    constexpr int WTM = BM / WGM, WTN = BN / WGN, MF = WTM / 16, NF = WTN / 16;
    constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, STAGE = A_BYTES + B_BYTES;
    static_assert(WTM % 16 == 0 && WTN % 16 == 0, "wave tile must be a multiple of 16");
    static_assert(BM % RPI == 0 && BN % RPI == 0, "tile rows must fill whole DMA pieces");
    static_assert(STAGES == 2 || DMA, "deep rings are DMA only");
    // deep rings count DMA instructions per wave (vmcnt): waves that own no piece in the last round issue a
    // dummy 1 KiB DMA from the zero page into a scratch slot behind the ring, so every wave issues NI per K-step
    constexpr bool PADDED = STAGES > 2 && (A_INSTR % NW != 0 || B_INSTR % NW != 0);
    constexpr int NI = A_IT + B_IT;  // DMA instructions per thread and K-step

    extern __shared__ __attribute__((aligned(16))) char smem[];
    TV_PROBE_DECL

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave % WGN;
    // Block order.  The tiles_n blocks of one row tile read the same activation rows; on ONE XCD they share them through
    // its L2 instead of fetching them tiles_n times from HBM / Infinity Cache (a K = 384, N = 1536 linear layer moved
    // 2.4 GB in 0.47 ms that way: memory bound).  Workgroups go to XCDs round-robin by linear id, so row tile m takes
    // the ids congruent to m mod 8, its column tiles consecutive within that XCD's sequence.
    int tile_n, tile_m;
    if (p.xcd_order) {
        const int lin = blockIdx.x, j = lin >> 3;
        tile_n = j % p.tiles_n;
        tile_m = (j / p.tiles_n) * 8 + (lin & 7);
        if (tile_m >= p.tiles_m) return;
    } else {
        tile_n = blockIdx.x % p.tiles_n;
        tile_m = blockIdx.x / p.tiles_n;
    }
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    // ---- per-thread staging bookkeeping ------------------------------------------------
    const int srow = lane / CPR;   // row inside a DMA piece
    const int sslot = lane % CPR;  // physical 16-byte slot inside the row
    int a_oy[A_IT], a_ox[A_IT], a_pix[A_IT], a_chunk[A_IT];
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
        const int j = it * NW + wave;
        const int row = j * RPI + srow;
        const int m = m0 + row;
        a_chunk[it] = (sslot ^ swz_of<BK>(row & 15)) * 8;  // logical channel offset of my slot
        if (j < A_INSTR && m < p.M) {
            const int hw = p.h_out * p.w_out;
            int b, oy, ox;
            if (p.w_shift >= 0) {  // power-of-two output grid: no integer division
                b = m >> p.hw_shift;
                const int r = m & (hw - 1);
                oy = r >> p.w_shift;
                ox = r & (p.w_out - 1);
            } else {
                b = m / hw;
                const int r = m - b * hw;
                oy = r / p.w_out;
                ox = r - oy * p.w_out;
            }
            a_oy[it] = oy * p.stride - p.pad;
            a_ox[it] = ox * p.stride - p.pad;
            a_pix[it] = b * p.h_in * p.w_in;
        } else {
            a_oy[it] = -(1 << 28);  // never valid
            a_ox[it] = 0;
            a_pix[it] = 0;
        }
    }
    const bf16* b_src[B_IT];
    bool b_ok[B_IT];
    int b_voff[B_IT];
#pragma unroll
    for (int it = 0; it < B_IT; ++it) {
        const int j = it * NW + wave;
        const int row = j * RPI + srow;
        const int rl = row % WTN;
        const int fi = bfrag_reader(rl);
        const int c = (sslot ^ swz_of<BK>(fi)) * 8;
        const int n = n0 + row;
        b_ok[it] = (j < B_INSTR) && (n < p.N);
        b_src[it] = p.w + (size_t)(b_ok[it] ? n : 0) * p.K + c;
        b_voff[it] = b_ok[it] ? (n * p.K + c) * 2 : OOB_OFFSET;
    }

    // running state of the "next K-step to stage"
    const int cch = p.c_in / BK;  // channel chunks per tap
#ifdef TV_ABL_K1
    const int nk = 1;
#else
    const int nk = p.kh * p.kw * cch;
#endif
    int st_ky = 0, st_kx = 0, st_ch = 0, st_t = 0;
    const bf16* a_src[A_IT];
    bool a_ok[A_IT];
    int a_voff[A_IT];
    const int hv = p.h_in << p.up_shift, wv = p.w_in << p.up_shift;

    auto tap_setup = [&]() {
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            const int uy = a_oy[it] + st_ky, ux = a_ox[it] + st_kx;
            const bool ok = ((unsigned)uy < (unsigned)hv) && ((unsigned)ux < (unsigned)wv) &&
                            (((uy | ux) & p.dil_mask) == 0);
            const int iy = uy >> p.up_shift, ix = ux >> p.up_shift;
            const int pix = a_pix[it] + iy * p.w_in + ix;
            a_ok[it] = ok;
            if constexpr (BUF) a_voff[it] = ok ? (pix * p.ldx + a_chunk[it]) * 2 : OOB_OFFSET;
            else a_src[it] = p.x + (size_t)(ok ? pix : 0) * p.ldx + a_chunk[it];
        }
    };
    tap_setup();

    bf16x8 a_reg[DMA ? 1 : A_IT], b_reg[DMA ? 1 : B_IT];

    // issue piece q (0 .. NI-1: A pieces, then W pieces) of the global loads of K-step st_t (DMA: straight into LDS stage
    // `sbase`).  Pieces are separate so that the main loop can thread them between the MFMAs: issued as one burst right
    // after the barrier, the 7 DMAs of a wave wait ~120 cycles each for the address pipe (in-kernel timers,
    // tools/probes/igemm_phase_probe.py) and hold up the wave's fragment reads and MFMAs behind them.
    auto issue_piece = [&](char* sbase, int q) {
        if (q < A_IT) {
            const int it = q, j = it * NW + wave;
            const int koff = st_ch * BK;
            if (A_INSTR % NW != 0 && j >= A_INSTR) {
                if constexpr (PADDED && BUF) buffer_load_lds16(p.x, p.x_bytes, smem + STAGES * STAGE, OOB_OFFSET, 0);
                else if constexpr (PADDED) __builtin_amdgcn_global_load_lds(TV_GLB(p.zeros + lane * 16), TV_LDS(smem + STAGES * STAGE), 16, 0, 0);
                return;
            }
            if constexpr (BUF) {
                buffer_load_lds16(p.x, p.x_bytes, sbase + j * 1024, a_voff[it], koff * 2);
            } else if constexpr (DMA) {
                const void* src = a_ok[it] ? (const void*)(a_src[it] + koff) : (const void*)(p.zeros + lane * 16);
                __builtin_amdgcn_global_load_lds(TV_GLB(src), TV_LDS(sbase + j * 1024), 16, 0, 0);
            } else {
                bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
                if (a_ok[it]) v = *(const bf16x8*)(a_src[it] + koff);
                a_reg[it] = v;
            }
        } else {
            const int it = q - A_IT, j = it * NW + wave;
            const int kb = st_t * BK;
            if (B_INSTR % NW != 0 && j >= B_INSTR) {
                if constexpr (PADDED && BUF) buffer_load_lds16(p.w, p.w_bytes, smem + STAGES * STAGE, OOB_OFFSET, 0);
                else if constexpr (PADDED) __builtin_amdgcn_global_load_lds(TV_GLB(p.zeros + lane * 16), TV_LDS(smem + STAGES * STAGE), 16, 0, 0);
                return;
            }
            if constexpr (BUF) {
                buffer_load_lds16(p.w, p.w_bytes, sbase + A_BYTES + j * 1024, b_voff[it], kb * 2);
            } else if constexpr (DMA) {
                const void* src = b_ok[it] ? (const void*)(b_src[it] + kb) : (const void*)(p.zeros + lane * 16);
                __builtin_amdgcn_global_load_lds(TV_GLB(src), TV_LDS(sbase + A_BYTES + j * 1024), 16, 0, 0);
            } else {
                bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
                if (b_ok[it]) v = *(const bf16x8*)(b_src[it] + kb);
                b_reg[it] = v;
            }
        }
    };
    auto stage_advance = [&]() {   // move the staging state to the following K-step
        ++st_t;
        if (++st_ch == cch) {
            st_ch = 0;
            if (++st_kx == p.kw) {
                st_kx = 0;
                ++st_ky;
            }
            if (st_t < nk) tap_setup();
        }
    };
    auto stage_issue = [&](char* sbase) {
#pragma unroll
        for (int q = 0; q < NI; ++q) issue_piece(sbase, q);
        stage_advance();
    };
    // register-staged variant only: park the loaded registers in LDS stage `sbase`
    auto stage_write = [&](char* sbase) {
        if constexpr (!DMA) {
#pragma unroll
            for (int it = 0; it < A_IT; ++it) {
                const int j = it * NW + wave;
                if (A_INSTR % NW != 0 && j >= A_INSTR) break;
                *(bf16x8*)(sbase + j * 1024 + lane * 16) = a_reg[it];
            }
#pragma unroll
            for (int it = 0; it < B_IT; ++it) {
                const int j = it * NW + wave;
                if (B_INSTR % NW != 0 && j >= B_INSTR) break;
                *(bf16x8*)(sbase + A_BYTES + j * 1024 + lane * 16) = b_reg[it];
            }
        }
    };

    // ---- fragment addressing -------------------------------------------------------------
    const int fi = lane & 15, fq = lane >> 4;
    const int sw = swz_of<BK>(fi);
    const int a_row_off = (wm * WTM + fi) * (BK * 2);
    const int b_row_off = A_BYTES + (wn * WTN + bfrag_lane_row(fi)) * (BK * 2);

    f32x4 acc[MF][NF];
#pragma unroll
    for (int i = 0; i < MF; ++i)
#pragma unroll
        for (int j = 0; j < NF; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // Fragment reads are software-pipelined against the MFMAs: where the registers allow (<= 80 fragment VGPRs) ALL reads
    // of the K-step (both 32-deep halves) are issued up front, so the second half's ds_read_b128s fly under the first
    // half's MFMAs instead of exposing their latency a second time; the 256x256 tile keeps one half in flight at a time.
    constexpr bool PIPE_ALL = (MF + NF) * 4 * (BK / 32) <= TV_PIPE_ALL_MAX;
    // `nbase` != nullptr: the DMA pieces of the K-step being staged go out between the MFMAs, one every GAP of them
    constexpr int NMF = MF * NF * (BK / 32), GAP = NMF / NI > 0 ? NMF / NI : 1;
    auto compute = [&](const char* sbase, char* nbase, auto issue_c) {
        constexpr bool ISSUE = DMA && decltype(issue_c)::value;
        if constexpr (PIPE_ALL) {
            bf16x8 af[BK / 32][MF], bfr[BK / 32][NF];
#pragma unroll
            for (int kk = 0; kk < BK / 32; ++kk) {
                const int coff = ((kk * 4 + fq) ^ sw) * 16;
#pragma unroll
#ifdef TV_ABL_NO_LDSREAD
                for (int i = 0; i < MF; ++i) { af[kk][i] = bf16x8{1, 1, 1, 1, 1, 1, 1, 1}; asm volatile("" : "+v"(af[kk][i])); }
#pragma unroll
                for (int j = 0; j < NF; ++j) { bfr[kk][j] = bf16x8{1, 1, 1, 1, 1, 1, 1, 1}; asm volatile("" : "+v"(bfr[kk][j])); }
#else
                for (int i = 0; i < MF; ++i) af[kk][i] = *(const bf16x8*)(sbase + a_row_off + i * 16 * (BK * 2) + coff);
#pragma unroll
                for (int j = 0; j < NF; ++j) bfr[kk][j] = *(const bf16x8*)(sbase + b_row_off + bfrag_off(j) * (BK * 2) + coff);
#endif
            }
            __builtin_amdgcn_sched_barrier(0);   // keep every read ahead of the MFMAs (the scheduler would sink them again)
            TV_T(3);
#pragma unroll
            for (int kk = 0; kk < BK / 32; ++kk)
#pragma unroll
                for (int i = 0; i < MF; ++i)
#pragma unroll
                    for (int j = 0; j < NF; ++j) {
#ifdef TV_ABL_NO_MFMA
                        asm volatile("" ::"v"(bfr[kk][j]), "v"(af[kk][i]));
#else
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[kk][j], af[kk][i], acc[i][j], 0, 0, 0);
#endif
                        const int idx = (kk * MF + i) * NF + j;
                        if (ISSUE && idx % GAP == GAP - 1 && idx / GAP < NI) {
                            __builtin_amdgcn_sched_barrier(0);
                            issue_piece(nbase, idx / GAP);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
            if constexpr (ISSUE) {
#pragma unroll
                for (int q = NMF / GAP; q < NI; ++q) issue_piece(nbase, q);
                stage_advance();
            }
        } else {
#pragma unroll
            for (int kk = 0; kk < BK / 32; ++kk) {
                const int coff = ((kk * 4 + fq) ^ sw) * 16;
                bf16x8 af[MF], bfr[NF];
#pragma unroll
                for (int i = 0; i < MF; ++i) af[i] = *(const bf16x8*)(sbase + a_row_off + i * 16 * (BK * 2) + coff);
#pragma unroll
                for (int j = 0; j < NF; ++j) bfr[j] = *(const bf16x8*)(sbase + b_row_off + bfrag_off(j) * (BK * 2) + coff);
#pragma unroll
                for (int i = 0; i < MF; ++i)
#pragma unroll
                    for (int j = 0; j < NF; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
                        const int idx = (kk * MF + i) * NF + j;
                        if (ISSUE && idx % GAP == GAP - 1 && idx / GAP < NI) {
                            __builtin_amdgcn_sched_barrier(0);
                            issue_piece(nbase, idx / GAP);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
            }
            if constexpr (ISSUE) {
#pragma unroll
                for (int qq = NMF / GAP; qq < NI; ++qq) issue_piece(nbase, qq);
                stage_advance();
            }
        }
    };

    // ---- main loop -------------------------------------------------------------------------
    // 8-wave tiles: waves w and w+4 share a SIMD.  With one barrier per K-step all eight issue their fragment reads
    // together and every MFMA pipe idles until the LDS has served them (ablation: the reads cost 0.25-0.4 ms of 2.7).
    // Ping-pong instead: the block runs in HALF-steps, group 0 (waves 0-3) reads the fragments of step t while group 1
    // multiplies step t-1, then they swap, so each SIMD always has one wave in its MFMA phase.
    constexpr bool PINGPONG = DMA && STAGES == 2 && NW == 8 && PIPE_ALL && !TV_NO_PINGPONG;
    // Register-pipelined loop (in-kernel timers: after a K-step barrier all waves read their fragments at once, ~640 LDS
    // cycles during which no MFMA issues).  Each wave keeps two half-step fragment sets: while the MFMAs of one half run,
    // the reads of the next half are in flight, ACROSS the stage boundary.  The block barrier therefore sits in the middle
    // of a K-step: every wave has read all of stage t (second half issued before the first half's MFMAs), stage t+1 has
    // landed, and stage t's buffer is refilled with stage t+STAGES between the MFMAs that follow.
    constexpr bool PIPE2 = DMA && PIPE_ALL && BK == 64 && !PINGPONG && !TV_NO_PIPE2;
    if constexpr (PIPE2) {
        bf16x8 f0a[MF], f0b[NF], f1a[MF], f1b[NF];
        auto read_half = [&](const char* sbase, int kk, bf16x8 (&fa)[MF], bf16x8 (&fb)[NF]) {
            const int coff = ((kk * 4 + fq) ^ sw) * 16;
#pragma unroll
            for (int i = 0; i < MF; ++i) fa[i] = *(const bf16x8*)(sbase + a_row_off + i * 16 * (BK * 2) + coff);
#pragma unroll
            for (int j = 0; j < NF; ++j) fb[j] = *(const bf16x8*)(sbase + b_row_off + bfrag_off(j) * (BK * 2) + coff);
        };
        constexpr int HMF = MF * NF, HGAP = HMF / NI > 0 ? HMF / NI : 1;
        const int dphase = __builtin_amdgcn_readfirstlane(wave % HGAP);
        auto mfma_half = [&](const bf16x8 (&fa)[MF], const bf16x8 (&fb)[NF], char* nbase, auto issue_c) {
            constexpr bool ISSUE = decltype(issue_c)::value;
#pragma unroll
            for (int i = 0; i < MF; ++i)
#pragma unroll
                for (int j = 0; j < NF; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
                    const int idx = i * NF + j;   // DMA slots staggered by wave (see conv3x3_halo_kernel)
                    if (ISSUE && idx / HGAP < NI && (TV_GENERIC_DPHASE ? idx % HGAP == dphase : idx % HGAP == HGAP - 1)) {
                        __builtin_amdgcn_sched_barrier(0);
                        issue_piece(nbase, idx / HGAP);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            if constexpr (ISSUE) {
#pragma unroll
                for (int q = HMF / HGAP; q < NI; ++q) issue_piece(nbase, q);
                stage_advance();
            }
        };
#pragma unroll
        for (int s = 0; s < STAGES; ++s)
            if (s < nk) stage_issue(smem + s * STAGE);
        if (nk >= STAGES) wait_vmcnt<(STAGES - 1) * NI>();   // stage 0 landed
        else wait_vmcnt<0>();
        if (TV_SETPRIO && NW == 8 && wave >= 4) __builtin_amdgcn_s_setprio(1);
        __builtin_amdgcn_s_barrier();
        read_half(smem, 0, f0a, f0b);
        int cur = 0, t = 0;
        auto step = [&](auto issue_c, auto last_c) {
            char* const sb = smem + cur * STAGE;
            char* const sn = smem + ((cur + 1 == STAGES) ? 0 : cur + 1) * STAGE;
            read_half(sb, 1, f1a, f1b);
            __builtin_amdgcn_sched_barrier(0);
            mfma_half(f0a, f0b, nullptr, std::false_type{});
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (!decltype(last_c)::value) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // my reads of stage t are done: its buffer may be refilled
                // stage t+1 landed; stages t+2 .. t+STAGES-1 may still be in flight (fewer in the drain)
                if (STAGES >= 3 && t + 2 < nk) wait_vmcnt<(STAGES - 2) * NI>();
                else wait_vmcnt<0>();
                __builtin_amdgcn_s_barrier();
                read_half(sn, 0, f0a, f0b);
                __builtin_amdgcn_sched_barrier(0);
            }
            mfma_half(f1a, f1b, sb, issue_c);
            __builtin_amdgcn_sched_barrier(0);
            cur = (cur + 1 == STAGES) ? 0 : cur + 1;
        };
        for (; t + STAGES < nk; ++t) step(std::true_type{}, std::false_type{});
        for (; t + 1 < nk; ++t) step(std::false_type{}, std::false_type{});
        step(std::false_type{}, std::true_type{});
    } else if constexpr (PINGPONG) {
        // Both groups run the same code, group 1 one barrier behind.  Phase k lies between block barriers k and k+1:
        //   group 0: reads step t in phase 2t,   multiplies it in phase 2t+1
        //   group 1: reads step t in phase 2t+1, multiplies it in phase 2t+2
        // The buffer of step t+1 is last read in phase 2t-1 and first read in phase 2t+2: every wave issues its share of
        // that DMA in phase 2t and waits for it at the end of phase 2t+1.
        const int grp = wave >> 2;
        stage_issue(smem);
        wait_vmcnt<0>();
        if (grp == 1) {
            __builtin_amdgcn_s_barrier();
            if (1 < nk) stage_issue(smem + STAGE);
        }
        for (int t = 0; t < nk; ++t) {
            __builtin_amdgcn_s_barrier();
            if (grp == 0 && t + 1 < nk) stage_issue(smem + ((t + 1) & 1) * STAGE);
            const char* sbase = smem + (t & 1) * STAGE;
            bf16x8 af[BK / 32][MF], bfr[BK / 32][NF];
#pragma unroll
            for (int kk = 0; kk < BK / 32; ++kk) {
                const int coff = ((kk * 4 + fq) ^ sw) * 16;
#pragma unroll
                for (int i = 0; i < MF; ++i) af[kk][i] = *(const bf16x8*)(sbase + a_row_off + i * 16 * (BK * 2) + coff);
#pragma unroll
                for (int j = 0; j < NF; ++j) bfr[kk][j] = *(const bf16x8*)(sbase + b_row_off + bfrag_off(j) * (BK * 2) + coff);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the buffer may be refilled after the next barrier
            if (grp == 1) wait_vmcnt<0>();
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            if (grp == 1 && t + 2 < nk) stage_issue(smem + (t & 1) * STAGE);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kk = 0; kk < BK / 32; ++kk)
#pragma unroll
                for (int i = 0; i < MF; ++i)
#pragma unroll
                    for (int j = 0; j < NF; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[kk][j], af[kk][i], acc[i][j], 0, 0, 0);
            if (grp == 0) wait_vmcnt<0>();
        }
        if (grp == 0) __builtin_amdgcn_s_barrier();
    } else if constexpr (DMA) {
        constexpr int LA = STAGES - 1;  // K-steps of lookahead
#pragma unroll
        for (int s = 0; s < LA; ++s)
            if (s < nk) stage_issue(smem + s * STAGE);
        int cur = 0, nxt = LA % STAGES, t = 0;
        // steady state: K-step t must have landed, the LA-1 younger ones may stay in flight; the pieces of K-step t+LA go
        // out between the MFMAs of step t (into the buffer every wave left before this barrier)
        for (; t + LA < nk; ++t) {
            wait_vmcnt<(LA - 1) * NI>();
            TV_T(0);
#ifndef TV_ABL_NO_BARRIER
            __builtin_amdgcn_s_barrier();
#endif
            TV_T(1);
#if TV_GENERIC_BURST
            stage_issue(smem + nxt * STAGE);
            compute(smem + cur * STAGE, nullptr, std::false_type{});
#elif !defined(TV_ABL_NO_DMA)
            compute(smem + cur * STAGE, smem + nxt * STAGE, std::true_type{});
#else
            compute(smem + cur * STAGE, nullptr, std::false_type{});
#endif
            TV_T(4);
            cur = (cur + 1 == STAGES) ? 0 : cur + 1;
            nxt = (nxt + 1 == STAGES) ? 0 : nxt + 1;
        }
        for (; t < nk; ++t) {   // drain: nothing left to issue
            if (LA >= 2 && nk - 1 - t == 1) wait_vmcnt<NI>();
            else wait_vmcnt<0>();
            TV_T(0);
#ifndef TV_ABL_NO_BARRIER
            __builtin_amdgcn_s_barrier();
#endif
            TV_T(1);
            compute(smem + cur * STAGE, nullptr, std::false_type{});
            TV_T(4);
            cur = (cur + 1 == STAGES) ? 0 : cur + 1;
        }
    } else {
        stage_issue(smem);
        stage_write(smem);
        for (int t = 0; t < nk; ++t) {
            __syncthreads();
            if (t + 1 < nk) stage_issue(nullptr);
            compute(smem + (t & 1) * STAGE, nullptr, std::false_type{});
            if (t + 1 < nk) stage_write(smem + ((t + 1) & 1) * STAGE);
        }
    }

    // ---- epilogue ---------------------------------------------------------------------------------
    if (TV_SETPRIO) __builtin_amdgcn_s_setprio(0);
    TV_T(5);
#ifdef TV_ABL_NO_EPI
    {
        float chk = 0.f;
#pragma unroll
        for (int i = 0; i < MF; ++i)
#pragma unroll
            for (int j = 0; j < NF; ++j) chk += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
        if (chk != 123.456f) return;   // (keeps the accumulators live; practically always taken)
    }
#endif
    f32x4 bvals[NF];
    load_bias<WTN>(p, lane, n0 + wn * WTN, bvals);
    const int mrow0 = m0 + wm * WTM;
    epilogue<WTM, WTN, EPI>(p, acc, bvals, smem, wave, lane, n0 + wn * WTN, [&](int r) { return mrow0 + r; });
    TV_T(6);
    TV_PROBE_DUMP(wave, lane);
}

// ---------------------------------------------------------------------------------------------------------------
// 3x3 / stride 1 / pad 1 convolutions (and their data gradients): halo-tile variant.
//
// Ablation of the generic kernel on its dominant shape (tools/probes/build_ablations.sh: 2.69 ms as is, 1.95 ms without
// the DMA, 1.65 ms with MFMA + epilogue only) shows the L2 -> LDS fill rate, not the MFMA or the LDS reads, to be the
// bound: the generic kernel fetches every activation row once per tap.  Here a block owns a TH x 16 SPATIAL tile and
// stages, per 64-channel chunk, the (TH+2) x 18 halo once (1.27x / 1.41x the tile instead of 9x); the nine taps are
// nine shifted fragment views of that one LDS image.  Weights stream as before, one [BN][64] slab per tap.
//
//   K order: channel chunk outer, tap inner.  Halo chunk c+1 is fetched piecewise under the taps of chunk c.
//   LDS image of the halo: pixel-major rows of 128 B, 16-byte chunk index XORed with (halo pixel & 7): every
//   16-pixel run of a halo row is conflict-free for ds_read_b128 whatever the tap shift.
//   Zero padding: halo pixels outside the image use an out-of-range buffer offset (the DMA writes zeros).
// ---------------------------------------------------------------------------------------------------------------
template <int BM, int BN, int WGM, int WGN, int BST, int EPI>
__global__ __launch_bounds__(WGM* WGN * 64) void conv3x3_halo_kernel(const IgemmArgs p) {
    constexpr int BK = 64, NW = WGM * WGN, TW = 16, TH = BM / TW, HWD = TW + 2, HP = (TH + 2) * HWD;
    constexpr int A_PIECES = (HP * 8 + 63) / 64;             // 1 KiB DMA pieces per halo chunk (8 pixels each)
    constexpr int WTM = BM / WGM, WTN = BN / WGN, MF = WTM / 16, NF = WTN / 16;
    constexpr bool PIPE_ALL = (MF + NF) * 4 * (BK / 32) <= (NW == 4 && BM == 256 ? 128 : 80);   // (one wave per SIMD owns 512 registers)
    // Loader waves.  Waves w and w+4 of an 8-wave block share a SIMD and run in lockstep between barriers; a DMA issued by
    // all eight at the same point of the MFMA stream queues ~120 cycles at the address pipe (64 B/clk per CU) and stalls
    // BOTH waves of every SIMD.  In the pipelined loop only waves 0 .. NWL-1 issue DMAs: while one of them waits at the
    // address pipe its partner keeps the MFMA pipe busy.
    constexpr int NWL = (NW == 8 && PIPE_ALL && !TV_NO_PIPE2 && !TV_NO_LOADER_SPLIT) ? 4 : NW;
    constexpr int A_IT = (A_PIECES + NWL - 1) / NWL;         // halo pieces per loader wave and chunk
    constexpr int A_BYTES = A_PIECES * 1024, B_BYTES = BN * BK * 2;
    constexpr int B_INSTR = BN / 8, B_IT = B_INSTR / NWL;
    static_assert(BST == 2 || BST == 3, "weight ring depth");
    static_assert(B_INSTR % NWL == 0 && WTM % 16 == 0 && WTN % 16 == 0, "tile shape");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    TV_PROBE_DECL
    char* const a_buf = smem;                  // [2][A_BYTES]
    char* const b_buf = smem + 2 * A_BYTES;    // [BST][B_BYTES]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave % WGN;
    // Block order.  The tiles_n blocks of one row tile read the same activation rows; on ONE XCD they share them through
    // its L2 instead of fetching them tiles_n times from HBM / Infinity Cache (a K = 384, N = 1536 linear layer moved
    // 2.4 GB in 0.47 ms that way: memory bound).  Workgroups go to XCDs round-robin by linear id, so row tile m takes
    // the ids congruent to m mod 8, its column tiles consecutive within that XCD's sequence.
    int tile_n, tile_m;
    if (p.xcd_order) {
        const int lin = blockIdx.x, j = lin >> 3;
        tile_n = j % p.tiles_n;
        tile_m = (j / p.tiles_n) * 8 + (lin & 7);
        if (tile_m >= p.tiles_m) return;
    } else {
        tile_n = blockIdx.x % p.tiles_n;
        tile_m = blockIdx.x / p.tiles_n;
    }
    const int n0 = tile_n * BN;
    const int tiles_x = p.w_out / TW, tiles_y = p.h_out / TH;
    const int b = tile_m / (tiles_x * tiles_y);
    const int trem = tile_m - b * (tiles_x * tiles_y);
    const int y0 = (trem / tiles_x) * TH, x0 = (trem % tiles_x) * TW;

    // ---- staging bookkeeping ---------------------------------------------------------------------------------------------
    // halo piece j covers halo pixels 8j .. 8j+7 (lane / 8) x 8 chunks (lane % 8); the per-lane source offset is rebuilt for
    // each piece (a dozen VALU operations per KiB) instead of parking A_IT registers for the whole loop
    const int a_c16 = ((lane & 7) ^ ((lane >> 3) & 7)) * 16;   // (8j + lane/8) & 7 == (lane/8) & 7
    auto a_voff_of = [&](int j) {
        int l8 = lane >> 3;
        asm volatile("" : "+v"(l8));   // loop-invariant otherwise: the compiler would hoist all A_IT offsets and spill
        const int hp = j * 8 + l8;
        const int hy = hp / HWD, hx = hp - hy * HWD;
        const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
        const bool ok = hp < HP && (unsigned)iy < (unsigned)p.h_in && (unsigned)ix < (unsigned)p.w_in;
        return ok ? ((b * p.h_in + iy) * p.w_in + ix) * p.ldx * 2 + a_c16 : OOB_OFFSET;
    };
    // weight piece j covers rows 8j .. 8j+7 of the [BN][64] slab; its source offset is rebuilt per piece as well
    auto b_voff_of = [&](int j) {
        int l8 = lane >> 3;
        asm volatile("" : "+v"(l8));
        const int row = j * 8 + l8;
        const int rl = row % WTN;
        const int fr = bfrag_reader(rl);
        const int c = ((lane & 7) ^ swz_of<BK>(fr)) * 8;
        const int n = n0 + row;
        return (n < p.N) ? (n * p.K + c) * 2 : OOB_OFFSET;
    };
    const int cch = p.c_in / BK;

    // where the registers allow, the offsets are computed once (the recomputation is ~12 VALU operations per piece in the
    // middle of the MFMA stream)
    constexpr bool VOFF_REGS = TV_HALO_VOFF_REGS && (MF * NF * 4 + (MF + NF) * 8 <= 160 || BST == 3 || !PIPE_ALL);   // (256x192, ring 2 would spill)
    int a_voff_r[VOFF_REGS ? A_IT : 1], b_voff_r[VOFF_REGS ? B_IT : 1];
    if constexpr (VOFF_REGS) {
#pragma unroll
        for (int it = 0; it < A_IT; ++it) a_voff_r[it] = a_voff_of(it * NWL + wave);
#pragma unroll
        for (int it = 0; it < B_IT; ++it) b_voff_r[it] = b_voff_of(it * NWL + wave);
    }
    auto issue_a = [&](char* dst, int it, int ch) {   // piece `it` of this (loader) wave, channel chunk ch
#ifdef TV_ABL_NO_DMA
        return;
#endif
        const int j = it * NWL + wave;
        if (j < A_PIECES) buffer_load_lds16(p.x, p.x_bytes, dst + j * 1024, VOFF_REGS ? a_voff_r[VOFF_REGS ? it : 0] : a_voff_of(j), ch * (BK * 2));
    };
    auto issue_b_piece = [&](char* dst, int it, int koff) {
#ifdef TV_ABL_NO_DMA
        return;
#endif
        buffer_load_lds16(p.w, p.w_bytes, dst + (it * NWL + wave) * 1024, VOFF_REGS ? b_voff_r[VOFF_REGS ? it : 0] : b_voff_of(it * NWL + wave), koff);
    };
    auto issue_b = [&](char* dst, int tap, int ch) {
        const int koff = (tap * p.c_in + ch * BK) * 2;
#pragma unroll
        for (int it = 0; it < B_IT; ++it) issue_b_piece(dst, it, koff);
    };

    // ---- fragment addressing -------------------------------------------------------------------------------------------
    const int fi = lane & 15, fq = lane >> 4;
    const int sw = swz_of<BK>(fi);
    const int hp_base = (wm * MF) * HWD + fi;     // halo pixel of (fragment 0, tap (0,0)); fragment i adds i*HWD, tap adds dy*HWD+dx
    const int b_row_off = (wn * WTN + bfrag_lane_row(fi)) * (BK * 2);

    f32x4 acc[MF][NF];
#pragma unroll
    for (int i = 0; i < MF; ++i)
#pragma unroll
        for (int j = 0; j < NF; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // issue(q), q = 0 .. B_IT: the DMA pieces that go out during this step (halo piece first, then the weight slab), one
    // every GAP MFMAs -- a burst after the barrier would hold the wave's own reads and MFMAs behind the address pipe
    constexpr int NMF = MF * NF * (BK / 32), NIH = B_IT + 1, GAP = NMF / NIH;
    auto compute = [&](const char* abase, const char* bbase, int toff, auto issue) {
        // the fragment addresses of a tap are cheap to rebuild and loop-invariant: left alone, the compiler hoists all
        // 9 x MF x 2 of them out of the chunk loop and spills; the opaque copy pins the arithmetic to this tap
        int hpb = hp_base;
        asm volatile("" : "+v"(hpb));
        int a_off[MF], a_sw[MF];
#pragma unroll
        for (int i = 0; i < MF; ++i) {
            const int hp = hpb + i * HWD + toff;
            a_off[i] = hp * (BK * 2);
            a_sw[i] = hp & 7;
        }
        if constexpr (PIPE_ALL) {
            bf16x8 af[BK / 32][MF], bfr[BK / 32][NF];
#pragma unroll
            for (int kk = 0; kk < BK / 32; ++kk) {
#pragma unroll
                for (int i = 0; i < MF; ++i) af[kk][i] = *(const bf16x8*)(abase + a_off[i] + (((kk * 4 + fq) ^ a_sw[i]) << 4));
                const int coff = ((kk * 4 + fq) ^ sw) * 16;
#pragma unroll
                for (int j = 0; j < NF; ++j) bfr[kk][j] = *(const bf16x8*)(bbase + b_row_off + bfrag_off(j) * (BK * 2) + coff);
            }
            __builtin_amdgcn_sched_barrier(0);
            TV_T(3);
#pragma unroll
            for (int kk = 0; kk < BK / 32; ++kk)
#pragma unroll
                for (int i = 0; i < MF; ++i)
#pragma unroll
                    for (int j = 0; j < NF; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[kk][j], af[kk][i], acc[i][j], 0, 0, 0);
                        const int idx = (kk * MF + i) * NF + j;
                        if (idx % GAP == GAP - 1 && idx / GAP < NIH) {
                            __builtin_amdgcn_sched_barrier(0);
                            issue(idx / GAP);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
        } else {
#pragma unroll
            for (int kk = 0; kk < BK / 32; ++kk) {
                bf16x8 af[MF], bfr[NF];
#pragma unroll
                for (int i = 0; i < MF; ++i) af[i] = *(const bf16x8*)(abase + a_off[i] + (((kk * 4 + fq) ^ a_sw[i]) << 4));
                const int coff = ((kk * 4 + fq) ^ sw) * 16;
#pragma unroll
                for (int j = 0; j < NF; ++j) bfr[j] = *(const bf16x8*)(bbase + b_row_off + bfrag_off(j) * (BK * 2) + coff);
#pragma unroll
                for (int i = 0; i < MF; ++i)
#pragma unroll
                    for (int j = 0; j < NF; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
                        const int idx = (kk * MF + i) * NF + j;
                        if (idx % GAP == GAP - 1 && idx / GAP < NIH) {
                            __builtin_amdgcn_sched_barrier(0);
                            issue(idx / GAP);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
            }
        }
    };

    // ---- main loop ------------------------------------------------------------------------------------------------------
    // Wave-group ping-pong (8-wave tiles, 3-deep weight ring).  Waves w and w+4 share a SIMD.  In the lockstep loops below
    // both of them read fragments, issue DMAs and multiply at the same moments, so nothing covers the non-matrix work
    // (ablation: DMA issue 0.5 ms + fragment reads 0.25 ms of a 2.5 ms launch are fully exposed).  Here the block runs in
    // PHASES (one block barrier each); group 0 (waves 0-3) reads the fragments of step t and issues its DMAs in phase 2t and
    // multiplies in phase 2t+1, group 1 (waves 4-7) does the same one phase later -- every SIMD always has one wave in its
    // MFMA phase (at priority 1) and the other in its load phase.
    //   phase 2t   : g0 reads step t,  issues its share of slab t+2 (+ a halo piece of the next chunk) | g1 multiplies step t-1
    //   phase 2t+1 : g0 multiplies step t                                                              | g1 reads step t, issues its share
    // Weight slot (t+2) % 3 held slab t-1: last read by g0 in phase 2t-2 and by g1 in phase 2t-1, so it is free in both load
    // phases of step t.  A wave waits (counted vmcnt) at the END of each load phase for everything it issued in EARLIER load
    // phases: slab t+2 has landed and is visible (next barrier) one full step before g0 reads it.  The halo pieces of chunk
    // c+1 go out during taps 0-5 of chunk c into the buffer last read at the final tap of chunk c-1.
    constexpr bool PP = TV_HALO_PP && NW == 8 && PIPE_ALL && BST == 3;
    if constexpr (PP) {
        static_assert(NWL == NW, "ping-pong: every wave loads");
        constexpr int ATAPS = 6, A_PT = (A_IT + ATAPS - 1) / ATAPS;
        static_assert(A_PT == 1, "one halo piece per wave and tap");
        constexpr auto nsure = [](int tap) { return (tap >= 0 && tap < ATAPS && tap < A_IT && (tap + 1) * NW <= A_PIECES) ? 1 : 0; };
        const int grp = wave >> 2;
        bf16x8 fa[2][MF], fb[2][NF];
#pragma unroll
        for (int it = 0; it < A_IT; ++it) issue_a(a_buf, it, 0);
        issue_b(b_buf, 0, 0);
        issue_b(b_buf + B_BYTES, 1, 0);
        wait_vmcnt<0>();
        if (grp == 1) __builtin_amdgcn_s_barrier();   // the stagger: group 1 runs one phase behind
        int bcur = 0;
        for (int ch = 0; ch < cch; ++ch) {
            const char* acur = a_buf + (ch & 1) * A_BYTES;
            char* anxt = a_buf + ((ch + 1) & 1) * A_BYTES;
            const bool more = ch + 1 < cch;
            static_for<0, 9>([&](auto tap_c) {
                constexpr int tap = decltype(tap_c)::value;
                constexpr int toff = (tap / 3) * HWD + (tap % 3);
                const char* const bslot = b_buf + bcur * B_BYTES;
                char* const bfill = b_buf + ((bcur + 2 >= BST) ? bcur + 2 - BST : bcur + 2) * B_BYTES;
                // ---- load phase --------------------------------------------------------------------------------------------
                constexpr int NM = TV_PP_NM < B_IT ? TV_PP_NM : B_IT, NL = B_IT - NM;   // slab pieces from the MFMA / load phase
                const bool b_go = (tap + 2 < 9) || more;
                const int b_koff = ((tap + 2 < 9) ? (tap + 2) * p.c_in + ch * BK : (tap + 2 - 9) * p.c_in + (ch + 1) * BK) * 2;
                // The weight pieces go out FIRST, the halo piece (HBM latency) last: vmcnt retires in issue order, so a slow halo
                // piece in front of them would hold back the wait for the (L2-resident) weight slab behind it.
                auto load_dma = [&]() {
                    if (b_go) {
#pragma unroll
                        for (int it = 0; it < NL; ++it) issue_b_piece(bfill, it, b_koff);
                    }
                    if (tap < ATAPS && tap < A_IT && more) issue_a(anxt, tap, ch + 1);
                };
                __builtin_amdgcn_s_barrier();
                TV_T(1);
                if constexpr (TV_PP_DMA_FIRST == 1) {
                    load_dma();
                    __builtin_amdgcn_sched_barrier(0);
                }
                int hpb = hp_base;
                asm volatile("" : "+v"(hpb));   // pin the address arithmetic to this tap
#ifndef TV_ABL_NO_LDSREAD
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
                    for (int i = 0; i < MF; ++i) {
                        const int hp = hpb + i * HWD + toff;
                        fa[kk][i] = *(const bf16x8*)(acur + hp * (BK * 2) + (((kk * 4 + fq) ^ (hp & 7)) << 4));
                    }
                    if constexpr (TV_PP_DMA_FIRST == 2) {   // threaded: a DMA piece after each group of reads
                        __builtin_amdgcn_sched_barrier(0);
                        if (b_go && kk < NL) issue_b_piece(bfill, kk, b_koff);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    const int coff = ((kk * 4 + fq) ^ sw) * 16;
#pragma unroll
                    for (int j = 0; j < NF; ++j) fb[kk][j] = *(const bf16x8*)(bslot + b_row_off + bfrag_off(j) * (BK * 2) + coff);
                    if constexpr (TV_PP_DMA_FIRST == 2) {
                        __builtin_amdgcn_sched_barrier(0);
                        if (kk == 0) {
                            if (b_go && 2 < NL) issue_b_piece(bfill, 2, b_koff);
                        } else if (tap < ATAPS && tap < A_IT && more) {
                            issue_a(anxt, tap, ch + 1);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
#else
                if (tap == 0 && ch == 0) {
                    for (int kk = 0; kk < 2; ++kk) {
                        for (int i = 0; i < MF; ++i) { fa[kk][i] = bf16x8{1, 1, 1, 1, 1, 1, 1, 1}; asm volatile("" : "+v"(fa[kk][i])); }
                        for (int j = 0; j < NF; ++j) { fb[kk][j] = bf16x8{1, 1, 1, 1, 1, 1, 1, 1}; asm volatile("" : "+v"(fb[kk][j])); }
                    }
                }
#endif
                __builtin_amdgcn_sched_barrier(0);
                TV_T(3);
                if constexpr (TV_PP_DMA_FIRST == 0) {
                    load_dma();
                    __builtin_amdgcn_sched_barrier(0);
                }
                TV_T(2);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // fragments in registers: the slot may be refilled two barriers on
                // The weight pieces of the PREVIOUS load phase (and of the MFMA phase after it) have landed; still in flight may be,
                // in issue order: [that phase's halo piece] [this phase's weight pieces] [this phase's halo piece].  The halo
                // pieces of a chunk are all retired by tap 7 (taps 6-8 issue none).
                if (more) wait_vmcnt<nsure(tap - 1) + NL + nsure(tap)>();
                else if (tap + 2 < 9) wait_vmcnt<NL>();
                else wait_vmcnt<0>();
                TV_T(0);
                // ---- MFMA phase --------------------------------------------------------------------------------------------
                __builtin_amdgcn_s_barrier();
                TV_T(5);
#ifndef TV_PP_NOPRIO
                __builtin_amdgcn_s_setprio(1);
#endif
                constexpr int NMF2 = 2 * MF * NF, MGAP = NM > 0 ? NMF2 / (NM + 1) : NMF2 + 1;
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                    for (int i = 0; i < MF; ++i)
#pragma unroll
                        for (int j = 0; j < NF; ++j) {
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[kk][j], fa[kk][i], acc[i][j], 0, 0, 0);
                            const int idx = (kk * MF + i) * NF + j;
                            if (NM > 0 && idx % MGAP == MGAP - 1 && idx / MGAP < NM) {
                                __builtin_amdgcn_sched_barrier(0);
                                if (b_go) issue_b_piece(bfill, NL + idx / MGAP, b_koff);
                                __builtin_amdgcn_sched_barrier(0);
                            }
                        }
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
                TV_T(4);
                bcur = (bcur + 1 == BST) ? 0 : bcur + 1;
            });
        }
        if (grp == 0) __builtin_amdgcn_s_barrier();   // (both groups have passed the same number of barriers)
    } else if constexpr (PIPE_ALL && !TV_NO_PIPE2) {
        // Register-pipelined like the generic kernel: two half-step (32-deep) fragment sets per wave; the block barrier sits
        // between the halves of a step, when every wave has read all of step t.  After it the weight slot of step t is
        // refilled with step t+BST and the halo pieces of the next chunk go out (taps 0-5), threaded between the MFMAs of
        // the loader waves.
        //   DMA order per step: [slab pieces x B_IT, halo pieces x <= A_PT].  At the barrier of step t slab t+1 must have
        //   landed: BST 2: only the halo pieces of step t-1 are younger;  BST 3: halo(t-2), slab(t+2), halo(t-1) are.
        constexpr int ATAPS = 6, A_PT = (A_IT + ATAPS - 1) / ATAPS;
        constexpr auto nsure = [](int tap) {   // halo pieces of a tap that every loader wave issues
            int n = 0;
            for (int it = tap * A_PT; it < tap * A_PT + A_PT; ++it)
                if (tap >= 0 && tap < ATAPS && it < A_IT && (it + 1) * NWL <= A_PIECES) ++n;
            return n;
        };
        constexpr int HMF = MF * NF, NIS = B_IT + A_PT, HGAP = HMF / NIS;
        static_assert(HGAP >= 1, "more DMA pieces than MFMAs in a half-step");
        bf16x8 f0a[MF], f0b[NF], f1a[MF], f1b[NF];
#ifdef TV_ABL_CHEAP_ADDR
        const int abl_a_off = (hp_base & ~7) * (BK * 2) + fi * (BK * 2) + ((fq ^ (fi & 7)) << 4);   // conflict-free, tap-invariant
#endif
#ifdef TV_ABL_NO_LDSREAD
        for (int i = 0; i < MF; ++i) { f0a[i] = bf16x8{1, 1, 1, 1, 1, 1, 1, 1}; f1a[i] = f0a[i]; asm volatile("" : "+v"(f0a[i]), "+v"(f1a[i])); }
        for (int j = 0; j < NF; ++j) { f0b[j] = bf16x8{1, 1, 1, 1, 1, 1, 1, 1}; f1b[j] = f0b[j]; asm volatile("" : "+v"(f0b[j]), "+v"(f1b[j])); }
#endif
        auto read_half = [&](const char* abase, const char* bbase, int toff, int kk, bf16x8 (&fa)[MF], bf16x8 (&fb)[NF]) {
#ifdef TV_ABL_NO_LDSREAD
            return;
#endif
            int hpb = hp_base;
            asm volatile("" : "+v"(hpb));   // pin the address arithmetic to this tap (see compute)
#pragma unroll
            for (int i = 0; i < MF; ++i) {
#ifdef TV_ABL_CHEAP_ADDR
                fa[i] = *(const bf16x8*)(abase + abl_a_off + i * HWD * (BK * 2) + kk * 64);
#else
                const int hp = hpb + i * HWD + toff;
                fa[i] = *(const bf16x8*)(abase + hp * (BK * 2) + (((kk * 4 + fq) ^ (hp & 7)) << 4));
#endif
            }
            const int coff = ((kk * 4 + fq) ^ sw) * 16;
#pragma unroll
            for (int j = 0; j < NF; ++j) fb[j] = *(const bf16x8*)(bbase + b_row_off + bfrag_off(j) * (BK * 2) + coff);
        };
        // One wave per SIMD (4-wave 256-row tile) has no partner to cover a block of fragment reads: there the reads of the
        // next half-step are threaded between the MFMAs as well (RD_THREAD), in the order the MFMAs will want them.
        constexpr bool RD_THREAD = (NW == 4 && BM == 256) || TV_RD_THREAD;
        constexpr int NRD = MF + NF, RGAP = HMF / NRD > 0 ? HMF / NRD : 1;
        auto read_piece = [&](const char* abase, const char* bbase, int hpb, int toff, int kk, bf16x8 (&fa)[MF], bf16x8 (&fb)[NF], int k) {
#ifdef TV_ABL_NO_LDSREAD
            return;
#endif
            if (k >= 1 && k <= NF) {
                const int j = k - 1;
                fb[j] = *(const bf16x8*)(bbase + b_row_off + bfrag_off(j) * (BK * 2) + ((kk * 4 + fq) ^ sw) * 16);
            } else {
                const int i = k == 0 ? 0 : k - NF;
#ifdef TV_ABL_CHEAP_ADDR
                fa[i] = *(const bf16x8*)(abase + abl_a_off + i * HWD * (BK * 2) + kk * 64);
#else
                const int hp = hpb + i * HWD + toff;
                fa[i] = *(const bf16x8*)(abase + hp * (BK * 2) + (((kk * 4 + fq) ^ (hp & 7)) << 4));
#endif
            }
        };
        auto mfma_half = [&](const bf16x8 (&fa)[MF], const bf16x8 (&fb)[NF], auto issue, auto rd, auto phase_c) {
            // DMA slots: waves 4-7 (PH = 1) issue half a gap before waves 0-3, so that the eight waves of a block do not
            // queue at the address pipe at the same MFMA index (two code copies; a per-wave branch costs more than it saves)
            constexpr int PH = decltype(phase_c)::value;
            constexpr int SLOT = PH ? (HGAP - 1) / 2 : HGAP - 1;
#pragma unroll
            for (int i = 0; i < MF; ++i)
#pragma unroll
                for (int j = 0; j < NF; ++j) {
#ifdef TV_ABL_NO_MFMA
                    asm volatile("" ::"v"(fb[j]), "v"(fa[i]));
#else
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
#endif
                    const int idx = i * NF + j;
                    if (RD_THREAD && idx % RGAP == 0 && idx / RGAP < NRD) {
                        __builtin_amdgcn_sched_barrier(0);
                        rd(idx / RGAP);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if (idx % HGAP == SLOT && idx / HGAP < NIS) {
                        __builtin_amdgcn_sched_barrier(0);
                        issue(idx / HGAP);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
        };
        auto run = [&](auto loader_c, auto phase_c) {
            constexpr bool LOADER = decltype(loader_c)::value;
            if constexpr (LOADER) {
#pragma unroll
                for (int it = 0; it < A_IT; ++it) issue_a(a_buf, it, 0);
#pragma unroll
                for (int sl = 0; sl < BST; ++sl) issue_b(b_buf + sl * B_BYTES, sl, 0);
                wait_vmcnt<(BST - 1) * B_IT>();
            }
            if (TV_SETPRIO && NW == 8 && wave >= 4) __builtin_amdgcn_s_setprio(1);
            __builtin_amdgcn_s_barrier();
            read_half(a_buf, b_buf, 0, 0, f0a, f0b);
            int bcur = 0;
            for (int ch = 0; ch < cch; ++ch) {
                const char* acur = a_buf + (ch & 1) * A_BYTES;
                char* anxt = a_buf + ((ch + 1) & 1) * A_BYTES;
                const bool more = ch + 1 < cch;
                static_for<0, 9>([&](auto tap_c) {
                    constexpr int tap = decltype(tap_c)::value;
                    char* const bslot = b_buf + bcur * B_BYTES;
                    const char* const bnext = b_buf + ((bcur + 1 == BST) ? 0 : bcur + 1) * B_BYTES;
                    constexpr int toff = (tap / 3) * HWD + (tap % 3), toff_n = tap < 8 ? ((tap + 1) / 3) * HWD + ((tap + 1) % 3) : 0;
                    int hpb = hp_base;
                    if constexpr (RD_THREAD) asm volatile("" : "+v"(hpb));   // pin the address arithmetic to this tap (see compute)
                    if constexpr (!RD_THREAD) read_half(acur, bslot, toff, 1, f1a, f1b);
                    __builtin_amdgcn_sched_barrier(0);
                    mfma_half(f0a, f0b, [](int) {}, [&](int k) { read_piece(acur, bslot, hpb, toff, 1, f1a, f1b, k); }, phase_c);
                    __builtin_amdgcn_sched_barrier(0);
                    TV_T(3);
                    const bool go_on = tap < 8 || more;
                    if (go_on) {
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // my reads of step t are done: its slab slot may be refilled
                        if constexpr (LOADER) {
                            if (more) wait_vmcnt<(BST == 3 ? nsure(tap - 2) + B_IT : 0) + nsure(tap - 1)>();
                            else wait_vmcnt<(BST == 3 && tap <= 6) ? B_IT : 0>();
                        }
                        TV_T(0);
#ifndef TV_ABL_NO_BARRIER
                        __builtin_amdgcn_s_barrier();
#endif
                        TV_T(1);
                        if constexpr (!RD_THREAD) {
                            if (tap < 8) read_half(acur, bnext, toff_n, 0, f0a, f0b);
                            else read_half(anxt, bnext, 0, 0, f0a, f0b);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    const char* const a_n = tap < 8 ? acur : (const char*)anxt;
                    auto rd_next = [&](int k) {
                        if (go_on) read_piece(a_n, bnext, hpb, toff_n, 0, f0a, f0b, k);
                    };
                    if constexpr (LOADER) {
                        const bool b_go = (tap + BST < 9) || more;
                        const int b_koff = ((tap + BST < 9) ? (tap + BST) * p.c_in + ch * BK : (tap + BST - 9) * p.c_in + (ch + 1) * BK) * 2;
                        mfma_half(f1a, f1b, [&](int q) {
                            if (q < B_IT) {
                                if (b_go) issue_b_piece(bslot, q, b_koff);
                            } else if (tap < ATAPS && tap * A_PT + (q - B_IT) < A_IT && more) {
                                issue_a(anxt, tap * A_PT + (q - B_IT), ch + 1);
                            }
                        }, rd_next, phase_c);
                    } else {
                        mfma_half(f1a, f1b, [](int) {}, rd_next, phase_c);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    TV_T(4);
                    bcur = (bcur + 1 == BST) ? 0 : bcur + 1;
                });
            }
        };
        if constexpr (NWL == NW) {
            if constexpr (NW == 8 && HGAP >= 4 && TV_DMA_STAGGER) {
                if (wave < 4) run(std::true_type{}, std::integral_constant<int, 0>{});
                else run(std::true_type{}, std::integral_constant<int, 1>{});
            } else {
                run(std::true_type{}, std::integral_constant<int, 0>{});
            }
        } else {
            if (wave < NWL) run(std::true_type{}, std::integral_constant<int, 0>{});
            else run(std::false_type{}, std::integral_constant<int, 0>{});
        }
    } else {
        // ---- main loop: one barrier per (chunk, tap).  The weight slab of step t + BST - 1 and one halo piece of the next
        // chunk are issued at step t, the halo piece FIRST: vmcnt counts in order, so "at most B_IT outstanding" (BST = 3)
        // means everything but the youngest weight slab -- in particular step t's slab and every older halo piece -- has landed.
        constexpr int LA = BST - 1;
    #pragma unroll
        for (int it = 0; it < A_IT; ++it) issue_a(a_buf, it, 0);
        issue_b(b_buf, 0, 0);
        if constexpr (LA == 2) issue_b(b_buf + B_BYTES, 1, 0);
        int bcur = 0, bnxt = LA;   // ring slots of step t and of step t + LA
        for (int ch = 0; ch < cch; ++ch) {
            const char* acur = a_buf + (ch & 1) * A_BYTES;
            char* anxt = a_buf + ((ch + 1) & 1) * A_BYTES;
            const bool more = ch + 1 < cch;
    #pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                if (LA == 2 && (tap < 8 || more)) wait_vmcnt<B_IT>();
                else wait_vmcnt<0>();
                TV_T(0);
                __builtin_amdgcn_s_barrier();
                TV_T(1);
                const bool b_go = (tap + LA < 9) || more;
                const int b_koff = ((tap + LA < 9) ? (tap + LA) * p.c_in + ch * BK : (tap + LA - 9) * p.c_in + (ch + 1) * BK) * 2;
                char* const b_dst = b_buf + bnxt * B_BYTES;
                TV_T(2);
    #if TV_HALO_BURST
                if (tap < A_IT && more) issue_a(anxt, tap, ch + 1);
                if (b_go) {
#pragma unroll
                    for (int it = 0; it < B_IT; ++it) issue_b_piece(b_dst, it, b_koff);
                }
                compute(acur, b_buf + bcur * B_BYTES, (tap / 3) * HWD + (tap % 3), [](int) {});
#else
                compute(acur, b_buf + bcur * B_BYTES, (tap / 3) * HWD + (tap % 3), [&](int q) {
                    if (q == 0) {
                        if (tap < A_IT && more) issue_a(anxt, tap, ch + 1);
                    } else if (b_go) {
                        issue_b_piece(b_dst, q - 1, b_koff);
                    }
                });
#endif
                TV_T(4);
                bcur = (bcur + 1 == BST) ? 0 : bcur + 1;
                bnxt = (bnxt + 1 == BST) ? 0 : bnxt + 1;
            }
        }

    }

    if (TV_SETPRIO) __builtin_amdgcn_s_setprio(0);
    TV_T(5);
    f32x4 bvals[NF];
    load_bias<WTN>(p, lane, n0 + wn * WTN, bvals);
    // wave-tile row r -> output pixel: fragment row i = r / 16 is tile row wm*MF + i, r % 16 the column
    const int pix0 = (b * p.h_out + y0 + wm * MF) * p.w_out + x0;
    epilogue<WTM, WTN, EPI>(p, acc, bvals, smem, wave, lane, n0 + wn * WTN, [&](int r) { return pix0 + (r >> 4) * p.w_out + (r & 15); });
    TV_T(6);
    TV_PROBE_DUMP(wave, lane);
}

bool g_use_dma = true;
int g_cfg_bm = 0;      // 0 = heuristic, else 128 / 256
int g_cfg_stages = 0;  // 0 = heuristic, else 2 / 3 / 4
int g_cfg_bk = 0;      // 0 = largest that divides c_in, else 32 / 64
int g_cfg_bn = 0;      // 0 = heuristic, 256 = 256-wide N tiles whenever c_out % 256 == 0, 128 = never
int g_addr_mode = 0;   // 0 = buffer DMA when the tensors are < 2 GiB, 1 = force 64-bit global DMA
int g_halo_ring = 3;    // weight ring depth of the halo kernel (2 / 3; 3 falls back to 2 where the LDS is too small)
bool g_halo_w4 = false;   // experiment: 256x192 halo tile with 4 waves (one per SIMD, 128x96 wave tiles)
bool g_xcd_order = true;  // column tiles of a row tile on one XCD (block order in the kernels)
bool g_use_halo = true;  // 3x3 stride-1 convolutions through conv3x3_halo_kernel when the shape qualifies

constexpr int LDS_MAX = 160 * 1024;

bool g_epi_modes = true;   // compile-time epilogue forms (tv_set_igemm_epilogue(0): the generic one everywhere, for A/B timing)

// which epilogue form a call takes (see epilogue<>): 1 = residual add only, 2 = saved-derivative multiply, 0 = the rest
int epilogue_mode(const IgemmArgs& a) {
    if (a.rope) return 0;
    if (!g_epi_modes || a.shuffle || a.pre || ((long long)a.M * a.ldo >> 3) >= 0xffffffffll) return 0;
    if (a.aux) return (a.aux_act == TV_ACT_DERIV && !a.res) ? 2 : 0;
    return (a.res && a.act == TV_ACT_NONE) ? 1 : 0;
}

// register form of an EPI 0 launch (see epilogue<>): the common elementwise combinations; everything else -> LDS loop
int epilogue_form(const IgemmArgs& a) {
    if (!g_epi_modes || a.rope || a.shuffle || a.aux || a.res) return EF_GENERIC;
    if (a.pre) {
        if (!a.pre_deriv) return EF_GENERIC;
        return a.act == TV_ACT_GELU ? EF_GELU_D : (a.act == TV_ACT_SILU ? EF_SILU_D : EF_GENERIC);
    }
    return a.act == TV_ACT_NONE ? EF_PLAIN : (a.act == TV_ACT_GELU ? EF_GELU : (a.act == TV_ACT_SILU ? EF_SILU : EF_GENERIC));
}

template <int BM, int BN, int WGM, int WGN, int BK, int STAGES, int MODE>
int launch_one(const IgemmArgs& a_in, hipStream_t s) {
    constexpr int RING = STAGES * (BM + BN) * BK * 2 + (STAGES > 2 ? 1024 : 0);  // + dummy-DMA scratch slot
    constexpr int EPI = epilogue_lds_bytes<BM / WGM, BN / WGN>(WGM * WGN);   // parked output tile (epilogue)
    constexpr int BYTES = RING > EPI ? RING : EPI;
    if constexpr (BYTES > LDS_MAX) {
        return -1;
    } else {
        const int tiles_m = (a_in.M + BM - 1) / BM;
        IgemmArgs a = a_in;
        a.tiles_m = tiles_m;
        a.xcd_order = (g_xcd_order && a.tiles_n > 1) ? 1 : 0;
        dim3 grid((unsigned)(a.xcd_order ? 8 * a.tiles_n * ((tiles_m + 7) / 8) : tiles_m * a.tiles_n)), block(WGM * WGN * 64);
        auto go = [&](auto epi) {
            constexpr int EPI_MODE = decltype(epi)::value;
            static TvPerDeviceOnce attr_once;
            if (attr_once.first()) {  // > 64 KiB of dynamic LDS needs the opt-in
                (void)hipFuncSetAttribute((const void*)igemm_nt_kernel<BM, BN, WGM, WGN, BK, STAGES, MODE, EPI_MODE>,
                                          hipFuncAttributeMaxDynamicSharedMemorySize, BYTES);
            }
            hipLaunchKernelGGL((igemm_nt_kernel<BM, BN, WGM, WGN, BK, STAGES, MODE, EPI_MODE>), grid, block, BYTES, s, a);
        };
        const int epi = epilogue_mode(a);
        if constexpr (MODE == 2 && BM * BN >= 128 * 128) {   // (the bring-up modes and the narrow tiles keep the one generic epilogue)
            if (epi == 1) { go(std::integral_constant<int, 1>{}); return 0; }
            if (epi == 2) { go(std::integral_constant<int, 2>{}); return 0; }
        }
        go(std::integral_constant<int, 0>{});
        return 0;
    }
}

// big-N tiles: BN in {128, 192, 256}; BM 128 (4 waves) or 256 (8 waves); ring depth 2..3
template <int BN, int BK, int MODE>
int launch_big(const IgemmArgs& a, int bm, int stages, hipStream_t s) {
    constexpr int WGM8 = (BN == 256) ? 2 : 4, WGN8 = (BN == 256) ? 4 : 2;  // 8-wave grids: wave tile 128x64 / 64x64 / 64x96
    if constexpr (MODE == 0) {
        return launch_one<128, BN, 2, 2, BK, 2, 0>(a, s);
    } else {
        if (bm == 256) {
            if (stages >= 3 && launch_one<256, BN, WGM8, WGN8, BK, 3, MODE>(a, s) == 0) return 0;
            return launch_one<256, BN, WGM8, WGN8, BK, 2, MODE>(a, s);
        }
        if (stages >= 3 && launch_one<128, BN, 2, 2, BK, 3, MODE>(a, s) == 0) return 0;
        return launch_one<128, BN, 2, 2, BK, 2, MODE>(a, s);
    }
}

// Tile choice by a wave-quantisation model: time ~ rounds over the chip's block slots x work of one tile / efficiency
// of the tile shape.  256x256 and 256x192 hold one block per CU, 128x128 two; the 128-row tile pays ~15 % in L2->LDS
// traffic per FLOP.  (Library GEMMs at these shapes run 1000-1290 TFLOP/s, tools/probes/mm_bench.py; the fixed rule
// "256 wide when >= 512 tiles, else 128" left M = 16 K, N = 1536 layers on 128x128 tiles at 865.)
// Returns BN (256 / 192 / 128) and sets bm; tuning hooks override.
int pick_tile(long long M, int N, bool allow256, bool allow192_256rows, int* bm) {
    const long long m256 = (M + 255) / 256, m128 = (M + 127) / 128;
    double best = 1e30;
    int bn = 128;
    *bm = 128;
    auto consider = [&](int cbn, int cbm, long long tiles, double slots, double eff) {
        const double rounds = (double)((long long)((tiles + slots - 1) / slots));
        const double t = rounds * (double)cbm * cbn * (slots / 256.0) / eff;
        if (t < best) { best = t; bn = cbn; *bm = cbm; }
    };
    consider(128, 128, m128 * ((N + 127) / 128), 512.0, 0.85);
    if (N % 192 == 0 && allow192_256rows) consider(192, 256, m256 * (N / 192), 256.0, 0.97);
    if (N % 192 == 0 && N % 128 != 0) consider(192, 128, m128 * (N / 192), 256.0, 0.80);
    if (N % 256 == 0 && allow256) consider(256, 256, m256 * (N / 256), 256.0, 1.0);
    return bn;
}

template <int BK, int MODE>
int launch_mode(IgemmArgs& a, hipStream_t s) {
    const int N = a.N;
    const int stages = g_cfg_stages ? g_cfg_stages : 2;
    if (N > 64) {
        int bm = 128;
        int bn = pick_tile(a.M, N, true, true, &bm);
        if (g_cfg_bn == 256 && N % 256 == 0) { bn = 256; bm = 256; }
        else if (g_cfg_bn == 192 && N % 192 == 0) { bn = 192; bm = 256; }
        else if (g_cfg_bn == 128) { bn = (N % 192 == 0 && N % 128 != 0) ? 192 : 128; bm = 128; }
        if (g_cfg_bm) bm = g_cfg_bm;
        if (bn == 256) {
            a.tiles_n = N / 256;
            return launch_big<256, BK, MODE>(a, bm, stages, s);
        }
        if (bn == 192) {
            a.tiles_n = N / 192;
            return launch_big<192, BK, MODE>(a, bm, stages, s);
        }
        a.tiles_n = (N + 127) / 128;
        return launch_big<128, BK, MODE>(a, bm, stages, s);
    }
    a.tiles_n = 1;
    if (N > 32) return launch_one<128, 64, 2, 2, BK, 2, MODE>(a, s);
    return launch_one<128, 32, 4, 1, BK, 2, MODE>(a, s);
}

template <int BM, int BN, int WGM, int WGN, int BST>
int launch_halo_one(const IgemmArgs& a_in, hipStream_t s) {
    constexpr int NW = WGM * WGN, HP = (BM / 16 + 2) * 18;
    constexpr int RING = 2 * (((HP * 8 + 63) / 64) * 1024) + BST * BN * 64 * 2;
    constexpr int EPI = epilogue_lds_bytes<BM / WGM, BN / WGN>(NW);
    constexpr int BYTES = RING > EPI ? RING : EPI;
    if constexpr (BYTES > LDS_MAX) {
        return -1;
    } else {
        const int tiles_m = a_in.batch * (a_in.h_out / (BM / 16)) * (a_in.w_out / 16);
        IgemmArgs a = a_in;
        a.tiles_m = tiles_m;
        a.xcd_order = (g_xcd_order && a.tiles_n > 1) ? 1 : 0;
        dim3 grid((unsigned)(a.xcd_order ? 8 * a.tiles_n * ((tiles_m + 7) / 8) : tiles_m * a.tiles_n)), block(NW * 64);
        auto go = [&](auto epi) {
            constexpr int EPI_MODE = decltype(epi)::value;
            static TvPerDeviceOnce attr_once;
            if (attr_once.first()) {
                (void)hipFuncSetAttribute((const void*)conv3x3_halo_kernel<BM, BN, WGM, WGN, BST, EPI_MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, BYTES);
            }
            hipLaunchKernelGGL((conv3x3_halo_kernel<BM, BN, WGM, WGN, BST, EPI_MODE>), grid, block, BYTES, s, a);
        };
        const int epi = epilogue_mode(a);
        if (epi == 1) go(std::integral_constant<int, 1>{});
        else if (epi == 2) go(std::integral_constant<int, 2>{});
        else go(std::integral_constant<int, 0>{});
        return 0;
    }
}

template <int BM, int BN, int WGM, int WGN>
int launch_halo_ring(const IgemmArgs& a, int ring, hipStream_t s) {
    if (ring >= 3 && launch_halo_one<BM, BN, WGM, WGN, 3>(a, s) == 0) return 0;
    return launch_halo_one<BM, BN, WGM, WGN, 2>(a, s);
}

// same tile heuristic as launch_mode; returns -1 when the shape does not qualify (the generic kernel takes it)
int launch_halo(IgemmArgs& a, hipStream_t s) {
    if (!(a.kh == 3 && a.kw == 3 && a.stride == 1 && a.pad == 1 && a.up_shift == 0 && a.dil_mask == 0 && !a.shuffle)) return -1;
    if (a.c_in % 64 != 0 || a.N <= 64 || a.x_bytes == 0 || a.w_bytes == 0) return -1;
    if (a.h_in != a.h_out || a.w_in != a.w_out || a.w_out % 16 != 0 || a.h_out % 8 != 0) return -1;
    const int N = a.N;
    const bool h16 = a.h_out % 16 == 0;
    int bm = 128;
    // (256x192 halo tiles with several N tiles run the 2-deep weight ring, which recomputes its DMA offsets: slower than
    //  128x128 there -- only the one-N-tile case takes 192 by choice)
    int bn = pick_tile(a.M, N, h16, h16 && (N == 192 || N % 128 != 0), &bm);
    // 256x192 tiles on the ping-pong loop with the 3-deep weight ring beat every other choice wherever N is a multiple of
    // 192 (tools/probes/ab_lib.py, 64 images): 384@64 1072 -> 1273 TFLOP/s over 128x128 tiles, 768@32 1394 -> 1498 and
    // 1536@16 1220 -> 1532 over 256x256 tiles
    if (TV_HALO_PP && N % 192 == 0 && h16 && g_cfg_bn == 0 && g_halo_ring == 3 && !g_halo_w4) { bn = 192; bm = 256; }
    if (g_cfg_bn == 256 && N % 256 == 0 && h16) { bn = 256; bm = 256; }
    else if (g_cfg_bn == 192 && N % 192 == 0 && h16) { bn = 192; bm = 256; }
    else if (g_cfg_bn == 128) { bn = (N % 192 == 0 && N % 128 != 0) ? 192 : 128; bm = 128; }
    if (g_cfg_bm) bm = g_cfg_bm;
    if (bm == 256 && !h16) bm = 128;
    if (bn == 192) {
        a.tiles_n = N / 192;
        // weight ring 3 deep only where measured faster (one N tile: res192@256/@128); 2 everywhere else
        if (bm == 256 && g_halo_w4) return launch_halo_ring<256, 192, 2, 2>(a, (g_halo_ring == 3 && a.tiles_n == 1) || g_halo_ring == 4 ? 3 : 2, s);
        if (bm == 256) return launch_halo_ring<256, 192, 4, 2>(a, (g_halo_ring == 3 && (a.tiles_n == 1 || TV_HALO_PP)) || g_halo_ring == 4 ? 3 : 2, s);
        return launch_halo_ring<128, 192, 2, 2>(a, g_halo_ring == 4 ? 3 : 2, s);
    }
    if (bn == 256 && bm == 256) {
        a.tiles_n = N / 256;
        return launch_halo_ring<256, 256, 2, 4>(a, 2, s);
    }
    a.tiles_n = (N + 127) / 128;
    if (bm == 256) return launch_halo_ring<256, 128, 4, 2>(a, g_halo_ring == 4 ? 3 : 2, s);
    return launch_halo_ring<128, 128, 2, 2>(a, g_halo_ring == 4 ? 3 : 2, s);
}

template <int BK>
int launch_bk(IgemmArgs& a, hipStream_t s) {
    if (!g_use_dma) return launch_mode<BK, 0>(a, s);
    if (g_addr_mode != 1 && a.x_bytes != 0 && a.w_bytes != 0) return launch_mode<BK, 2>(a, s);
    return launch_mode<BK, 1>(a, s);
}

}  // namespace

extern "C" int tv_set_dma(int on) {   // 0: register staging, 1: LDS-DMA (buffer form when possible), 2: LDS-DMA, global form only
    g_use_dma = on != 0;
    g_addr_mode = (on == 2) ? 1 : 0;
    return 0;
}

extern "C" int tv_set_igemm_halo(int on) {   // 0: 3x3 stride-1 convolutions through the generic kernel (tests, A/B timing)
    g_halo_w4 = on >= 100;  // +100: 4-wave 256x192 halo tile
    on %= 100;
    g_xcd_order = on < 10;  // +10: plain (row-tile-major) block order, for A/B timing
    on %= 10;
    g_use_halo = on != 0;   // 1: heuristic ring depth, 2: ring 2 everywhere, 4: ring 3 wherever it fits
    g_halo_ring = (on == 2) ? 2 : (on == 4 ? 4 : 3);
    return 0;
}

extern "C" int tv_set_igemm_epilogue(int on) {   // 0: generic (run-time) epilogue everywhere, for A/B timing and tests
    g_epi_modes = on != 0;
    return 0;
}

#ifdef TV_PROBE
extern "C" int tv_set_igemm_probe(void* dev_buf) {   // 16 blocks x 8 waves x 8 counters (u64), or null
    return hipMemcpyToSymbol(HIP_SYMBOL(g_probe_dev), &dev_buf, sizeof(void*)) == hipSuccess ? 0 : 1;
}
#endif

// tuning hook (tools/gemm_sweep.py): 0 restores the built-in heuristic
extern "C" int tv_set_igemm_config(int bm, int bn, int stages, int bk) {
    g_cfg_bm = bm;
    g_cfg_bn = bn;
    g_cfg_stages = stages;
    g_cfg_bk = bk;
    return 0;
}

struct RopeSpec {
    const float* tab;
    int tokens, cols;
};
static int igemm_nt_impl(const tv_conv_desc* d, const void* x, const void* w, const float* bias, const void* residual,
                         void* pre_act, void* out, const void* aux, int aux_act, void* stream, RopeSpec rope = RopeSpec{nullptr, 0, 0});

extern "C" int tv_igemm_nt_rope(const tv_conv_desc* d, const void* x, const void* w, const float* bias, void* out,
                                const float* rope_tab, int tokens_per_image, int rope_cols, void* stream) {
    TV_CHECK_ARG(d && rope_tab && tokens_per_image > 0 && rope_cols > 0 && rope_cols % 64 == 0 && rope_cols <= d->c_out,
                 "tv_igemm_nt_rope: rope_cols must be a positive multiple of 64 (whole heads) within c_out");
    TV_CHECK_ARG(d->store_shuffle == 0 && (d->act & ~TV_ACT_SAVE_DERIV) == TV_ACT_NONE, "tv_igemm_nt_rope: plain projection only");
    TV_CHECK_ARG(((long long)d->batch * d->h_out * d->w_out) % tokens_per_image == 0, "tv_igemm_nt_rope: rows must be whole images");
    return igemm_nt_impl(d, x, w, bias, nullptr, nullptr, out, nullptr, TV_ACT_NONE, stream, RopeSpec{rope_tab, tokens_per_image, rope_cols});
}

extern "C" int tv_igemm_nt(const tv_conv_desc* d, const void* x, const void* w, const float* bias,
                           const void* residual, void* pre_act, void* out, void* stream) {
    return igemm_nt_impl(d, x, w, bias, residual, pre_act, out, nullptr, TV_ACT_NONE, stream);
}

extern "C" int tv_igemm_nt_actgrad(const tv_conv_desc* d, const void* x, const void* w, const void* residual,
                                   const void* aux_pre_act, int aux_act, void* out, void* stream) {
    TV_CHECK_ARG(aux_pre_act && aux_act >= 0 && aux_act <= TV_ACT_DERIV && d && d->act == TV_ACT_NONE,
                 "tv_igemm_nt_actgrad: needs the saved pre-activation, a valid activation id and desc.act == NONE");
    return igemm_nt_impl(d, x, w, nullptr, residual, nullptr, out, aux_pre_act, aux_act, stream);
}

static int igemm_nt_impl(const tv_conv_desc* d, const void* x, const void* w, const float* bias, const void* residual,
                         void* pre_act, void* out, const void* aux, int aux_act, void* stream, RopeSpec rope) {
    TV_CHECK_ARG(d && x && w && out, "tv_igemm_nt: null pointer");
    TV_CHECK_ARG(d->c_in > 0 && d->c_in % 32 == 0, "tv_igemm_nt: c_in=%d must be a multiple of 32", d->c_in);
    TV_CHECK_ARG(d->c_out > 0 && d->c_out % 8 == 0, "tv_igemm_nt: c_out=%d must be a multiple of 8", d->c_out);
    TV_CHECK_ARG(d->ldx >= d->c_in && d->ldx % 8 == 0, "tv_igemm_nt: ldx=%d (c_in=%d) must be >= c_in and a multiple of 8", d->ldx, d->c_in);
    TV_CHECK_ARG(d->ldo % 8 == 0, "tv_igemm_nt: ldo=%d must be a multiple of 8", d->ldo);
    TV_CHECK_ARG(d->batch > 0 && d->h_in > 0 && d->w_in > 0 && d->h_out > 0 && d->w_out > 0, "tv_igemm_nt: empty geometry");
    TV_CHECK_ARG(d->kh > 0 && d->kw > 0 && d->stride > 0 && d->pad >= 0, "tv_igemm_nt: bad taps");
    TV_CHECK_ARG((d->up_shift | 1) == 1 && (d->dil_mask | 1) == 1, "tv_igemm_nt: up_shift/dil_mask must be 0 or 1");
    TV_CHECK_ARG(d->act >= 0 && (d->act & ~TV_ACT_SAVE_DERIV) <= 2, "tv_igemm_nt: unknown activation %d", d->act);
    TV_CHECK_ARG(!(d->act & TV_ACT_SAVE_DERIV) || (pre_act && (d->act & ~TV_ACT_SAVE_DERIV) != TV_ACT_NONE),
                 "tv_igemm_nt: TV_ACT_SAVE_DERIV needs an activation and a pre_act buffer");
    TV_CHECK_ARG(d->store_shuffle >= 0 && d->store_shuffle <= 2, "tv_igemm_nt: store_shuffle must be 0, 1 or 2");
    const long long M = (long long)d->batch * d->h_out * d->w_out;
    TV_CHECK_ARG(M < (1ll << 31) && (long long)d->batch * d->h_in * d->w_in < (1ll << 31), "tv_igemm_nt: too many pixels");
    if (d->store_shuffle) {
        TV_CHECK_ARG(d->c_out % 32 == 0 && d->ldo >= d->c_out / 4, "tv_igemm_nt: shuffle store needs c_out %% 32 == 0");
    } else {
        TV_CHECK_ARG(d->ldo >= d->c_out, "tv_igemm_nt: ldo < c_out");
    }
    if (tv_init() != TV_OK) return TV_ERR_INIT;

    IgemmArgs a;
    a.x = (const bf16*)x;
    a.w = (const bf16*)w;
    a.bias = bias;
    a.res = (const bf16*)residual;
    a.aux = (const bf16*)aux;
    a.aux_act = aux_act;
    a.pre = (bf16*)pre_act;
    a.out = (bf16*)out;
    a.zeros = (const char*)tv_zero_page();
    a.M = (int)M;
    a.N = d->c_out;
    a.K = d->kh * d->kw * d->c_in;
    a.batch = d->batch; a.h_in = d->h_in; a.w_in = d->w_in; a.c_in = d->c_in; a.ldx = d->ldx;
    a.h_out = d->h_out; a.w_out = d->w_out; a.ldo = d->ldo;
    a.kh = d->kh; a.kw = d->kw; a.stride = d->stride; a.pad = d->pad;
    a.up_shift = d->up_shift; a.dil_mask = d->dil_mask;
    a.tiles_n = 1;
    a.shuffle = d->store_shuffle;
    a.act = d->act & ~TV_ACT_SAVE_DERIV;
    a.pre_deriv = (d->act & TV_ACT_SAVE_DERIV) ? 1 : 0;
    a.rope = rope.tab;
    a.rope_tokens = rope.tokens;
    a.rope_cols = rope.cols;
    a.form = epilogue_form(a);
    {
        auto lg = [](int v) { int s = 0; while ((1 << s) < v) ++s; return ((1 << s) == v) ? s : -1; };
        a.w_shift = lg(d->w_out);
        a.hw_shift = lg(d->h_out * d->w_out);
        if (a.hw_shift < 0 || a.w_shift < 0) a.hw_shift = a.w_shift = -1;
    }
    {   // buffer-descriptor extents (0 = too large for 32-bit offsets -> global-address DMA)
        const long long xb = ((long long)d->batch * d->h_in * d->w_in - 1) * d->ldx * 2 + (long long)d->c_in * 2;
        const long long wb = (long long)a.N * a.K * 2;
        a.x_bytes = xb < (1ll << 31) ? (unsigned)xb : 0u;
        a.w_bytes = wb < (1ll << 31) ? (unsigned)wb : 0u;
    }
    hipStream_t s = (hipStream_t)stream;
    const bool bk64 = (d->c_in % 64 == 0) && g_cfg_bk != 32;
    int rc = (g_use_dma && g_use_halo && g_addr_mode != 1 && bk64) ? launch_halo(a, s) : -1;
    if (rc != 0) rc = bk64 ? launch_bk<64>(a, s) : launch_bk<32>(a, s);
    if (rc != 0) {
        tv_set_error("tv_igemm_nt: no kernel for this configuration");
        return TV_ERR_ARG;
    }
    TV_CHECK_LAUNCH("tv_igemm_nt");
    return TV_OK;
}
