// Flash attention for the TransVAE block on gfx950: softmax(q k^T * scale) v, non-causal,
// head_dim 64, bf16 in/out, fp32 statistics (R/transvae/modules/attention.py:88-92).
//
// Data: qkv [B, N, 3, heads, 64] bf16 (the fused QKV projection output, RoPE already applied),
//       o   [B, N, heads, 64] bf16, lse [B, heads, N] fp32.
//
// All three kernels use v_mfma_f32_32x32x16_bf16 and keep the softmax row on the MFMA *lane*:
//   forward  : S^T = K Q^T (key rows, query lanes)  ->  per-lane online softmax  ->  the
//              exponentiated accumulator is, as it stands, the B operand of O^T = V^T P^T
//              (guide section 3, "An accumulator tile as the next MFMA's operand"); V is
//              read from its row-major LDS tile with ds_read_b64_tr_b16.
//   dq       : same shape of loop with dS^T in place of P^T and K^T in place of V^T.
//   dk / dv  : S = Q K^T with the key on the lane; P and dS accumulators are the B operands of
//              dV^T = dO^T P and dK^T = Q^T dS; Q / dO tiles are read by rows and transposed
//              from one LDS image.  -LSE and -delta enter as initial accumulators.
// No atomics and no LDS round trip for P / dS; dq costs a second pass over S (3.5x forward
// MFMA work for the whole backward instead of 2.5x) -- the trade is recorded in DESIGN.md.
#include "common.h"

#include <type_traits>
#include <utility>

namespace {

// Diagnostic builds only (tools/probes/build_variant.sh ... -DTV_ATTN_ABL=<mask>; results are wrong, times are what is read):
// 1 no exponentials, 2 no second-phase MFMAs (P V / dS K / dV, dK), 4 no first-phase MFMAs (S, dP), 8 no DMA after the
// prologue, 16 no barrier and no DMA wait, 32 no LDS fragment reads.
#ifndef TV_ATTN_ABL
#define TV_ATTN_ABL 0
#endif
__device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) {
#if TV_ATTN_ABL & 4
    c[0] += (float)a[0] * (float)b[0];
    return c;
#else
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
#endif
}
__device__ __forceinline__ f32x16 mfma32b(bf16x8 a, bf16x8 b, f32x16 c) {   // second phase of a tile
#if TV_ATTN_ABL & 2
    c[0] += (float)a[0] * (float)b[0];
    return c;
#else
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
#endif
}
// ds_read_b64_tr_b16 via inline asm: the builtin makes hipcc (ROCm 7.2) drain vmcnt(0) before each transposed read
// while an LDS-DMA is in flight, serialising the K/V DMA with compute.  Callers wait with lds_wait_all() before use.
__device__ __forceinline__ bf16x4 lds_tr16(const char* p) {
    bf16x4 r;
    const unsigned a = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)p;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(r) : "v"(a) : "memory");
    return r;
}
__device__ __forceinline__ void lds_wait_all() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}
#if TV_ATTN_ABL & 1
__device__ __forceinline__ float fexp2(float x) { return x; }
#else
__device__ __forceinline__ float fexp2(float x) { return __builtin_amdgcn_exp2f(x); }
#endif

// output rows (o, dq, dk, dv): non-temporal stores measured 0.7 % SLOWER in the bench (the consumer GEMM finds a small o / dqkv in the
// Infinity Cache with the default policy; item 23) -- off, kept for A/B
#ifndef TV_ATTN_NT_STORE
#define TV_ATTN_NT_STORE 0
#endif
__device__ __forceinline__ void attn_store4(bf16* ptr, const bf16x4& v) {
#if TV_ATTN_NT_STORE
    __builtin_nontemporal_store(v, (bf16x4*)ptr);
#else
    *(bf16x4*)ptr = v;
#endif
}

// ---- ring of K / V (Q / dO) stages -------------------------------------------------------------------------------
// Round 3: the loops staged tile t+1 while computing tile t and opened every tile with `vmcnt(0)` + barrier: one tile of
// matrix work (512 - 1024 cycles per wave) is about the latency of an LDS-DMA piece under load, and 0.24-0.28 of the
// wave-cycles were parked (profiles/r03_attention_pmc.json).  The ring now holds NS stages, the DMA runs NS-1 tiles ahead and
// a tile opens with a COUNTED wait: stage t must have landed, the NS-2 younger stages (PIECES DMA instructions each, per
// wave, in issue order) may still be in flight.  Near the end of the sequence fewer stages are in flight: wait for all.
// The barrier is the bare s_barrier (`__syncthreads()` adds a fence, for which hipcc drains vmcnt(0)).
#ifndef TV_ATTN_NS
#define TV_ATTN_NS 3        // forward, dq: stages of K 8 KiB + V 8 KiB (3 blocks of 48 KiB per CU)
#endif
#ifndef TV_ATTN_NS_DKV
#define TV_ATTN_NS_DKV 2    // dk / dv: ring stages of TV_ATTN_DKV_QS x (Q 4 KiB + dO 4 KiB + 256 B); with two tiles per stage a ring of 2 measured best (3.17 -> 3.09-3.12 ms at N = 4096)
#endif
#ifndef TV_ATTN_SUM_MFMA
#define TV_ATTN_SUM_MFMA 0  // forward: softmax row sums on the matrix pipe (an all-ones A operand) instead of 32 v_add_f32 per key block: measured 2.7 % SLOWER, off
#endif
#ifndef TV_ATTN_LAZY
#define TV_ATTN_LAZY 8      // forward: rescale the running sums only when some row's maximum grew by more than this (log2 units); 0 = always
#endif
#ifndef TV_ATTN_DMA_LATE
#define TV_ATTN_DMA_LATE 1    // next stage's DMA pieces after the first-phase MFMAs of a tile instead of right behind the barrier
#endif
#ifndef TV_ATTN_V_LATE
#define TV_ATTN_V_LATE 0
#endif
#ifndef TV_ATTN_FWD64
#define TV_ATTN_FWD64 2048  // forward: sequences of at least this many tokens take the 64-queries-per-wave kernel (0 = never): N = 4096 1.745 -> 1.693 ms, N = 1024 equal
#endif
#ifndef TV_ATTN_DELTA_MFMA
#define TV_ATTN_DELTA_MFMA 1  // dq: -delta enters dP through one extra k-step (three bf16 pieces = fp32), not 16 v_add_f32 per tile
#endif
#ifndef TV_ATTN_DKV_ACCINIT
#define TV_ATTN_DKV_ACCINIT 1 // dk / dv: -delta read from LDS straight into dP's initial accumulator
#endif
template <int N>
__device__ __forceinline__ void vm_wait() {
#if !(TV_ATTN_ABL & 16)
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
#endif
}
__device__ __forceinline__ void block_sync() {
    __builtin_amdgcn_sched_barrier(0);
#if !(TV_ATTN_ABL & 16)
    __builtin_amdgcn_s_barrier();
#endif
    __builtin_amdgcn_sched_barrier(0);
}
template <int I>
using ic = std::integral_constant<int, I>;
// f(t, ic<t % NS>) for t = 0 .. n-1: the ring stage is a compile-time constant inside f (immediate LDS offsets)
template <int NS, class F>
__device__ __forceinline__ void ring_for(int n, F&& f) {
    static_assert(NS >= 2 && NS <= 4, "ring of 2 to 4 stages");
    for (int t = 0; t < n; t += NS) {
        f(t, ic<0>{});
        if (t + 1 < n) f(t + 1, ic<1>{});
        if constexpr (NS > 2) {
            if (t + 2 < n) f(t + 2, ic<2 % NS>{});
        }
        if constexpr (NS > 3) {
            if (t + 3 < n) f(t + 3, ic<3 % NS>{});
        }
    }
}
// max over the two lane halves (lane i <-> lane i + 32) without LDS: v_permlane32_swap exchanges the upper half of its first
// operand with the lower half of its second (the ds_bpermute of __shfl_xor cost an LDS round trip per key block)
__device__ __forceinline__ float xhalf_max(float x) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float xhalf_sum(float x) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// LDS tiles are [rows][64] bf16 (128-byte rows, 8 chunks of 16 bytes); physical chunk = c ^ swz(row)
__device__ __forceinline__ int swz_row(int r) { return (r >> 1) & 7; }          // conflict-free ds_read_b128 rows
__device__ __forceinline__ int swz_tr(int r) { return ((r >> 1) & 1) << 2; }    // conflict-free transposed reads

// DMA `nrows` rows (multiple of 8) of a [*, ld] matrix starting at row g0 into a swizzled LDS tile.
template <bool TR>
__device__ __forceinline__ void stage_rows(char* lds, const bf16* base, int g0, int nvalid, size_t ld, int nrows,
                                           int wave, int lane, const char* zeros) {
    const int rsub = lane >> 3, slot = lane & 7;
    for (int j = wave; j < nrows / 8; j += 4) {
        const int row = j * 8 + rsub;
        const int c = slot ^ (TR ? swz_tr(row) : swz_row(row));
        const int gr = g0 + row;
        const void* src = (gr < nvalid) ? (const void*)(base + (size_t)gr * ld + c * 8) : (const void*)(zeros + lane * 16);
        __builtin_amdgcn_global_load_lds(TV_GLB(src), TV_LDS(lds + j * 1024), 16, 0, 0);
    }
}

// The same staging through a buffer descriptor (tensors < 2 GiB per image): the per-lane offset of a piece inside the tile
// (row, swizzled chunk) is fixed for the whole kernel, the tile's first row enters as the SCALAR offset and rows beyond
// `nvalid` fall outside the descriptor's extent (the DMA writes zeros) -- no vector ALU work per tile.  The per-tile address
// arithmetic of stage_rows (64-bit multiply-add, bound check, select: ~15 operations per KiB piece) was a sixth of the
// instructions of loops that are bound by their vector ALU work (profiles/r02_attention.json).
template <bool TR, int NP>
__device__ __forceinline__ void stage_offsets(int (&voff)[NP], int ld, int wave, int lane) {
    const int rsub = lane >> 3, slot = lane & 7;
#pragma unroll
    for (int q = 0; q < NP; ++q) {
        const int row = (wave + 4 * q) * 8 + rsub;
        const int c = slot ^ (TR ? swz_tr(row) : swz_row(row));
        voff[q] = (row * ld + c * 8) * 2;
    }
}
template <int NP>
__device__ __forceinline__ void stage_rows_buf(char* lds, const bf16* base, unsigned bytes, int g0, int ld, const int (&voff)[NP], int wave) {
#pragma unroll
    for (int q = 0; q < NP; ++q) buffer_load_lds16(base, bytes, lds + (wave + 4 * q) * 1024, voff[q], g0 * ld * 2);
}
// extent of a [nvalid, 64] head slice of a matrix with row pitch ld (elements), from the slice's first element
__device__ __forceinline__ unsigned stage_extent(int nvalid, int ld) { return (unsigned)(((long long)(nvalid - 1) * ld + 64) * 2); }

// A operand (32 rows x 16 k) read by rows: lane (row = lane&31, half h) takes 16 bytes at d = 16*st + 8*h
__device__ __forceinline__ bf16x8 read_rows(const char* tile, int row0, int st, int lane) {
    const int row = row0 + (lane & 31);
    const int c = (2 * st + (lane >> 5)) ^ swz_row(row);
    return *(const bf16x8*)(tile + row * 128 + c * 16);
}

// A operand = transpose of a row-major tile: rows of the operand are columns d = dblk*32 + (lane&31),
// element j of lane half h is tile row  r0 + 8*(j>>2) + 4*h + (j&3)   (the k order of an accumulator
// tile used as B operand).  TR selects the tile's swizzle.
template <bool TR>
__device__ __forceinline__ bf16x8 read_cols(const char* tile, int r0, int dblk, int lane) {
    const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    const int h = g >> 1;
    const int chunk = dblk * 4 + (g & 1) * 2 + (pp >> 1);
    const int ra = r0 + 4 * h + q, rb = ra + 8;
    const int ca = chunk ^ (TR ? swz_tr(ra) : swz_row(ra));
    const int cb = chunk ^ (TR ? swz_tr(rb) : swz_row(rb));
    const bf16x4 lo = lds_tr16(tile + ra * 128 + ca * 16 + (pp & 1) * 8);
    const bf16x4 hi = lds_tr16(tile + rb * 128 + cb * 16 + (pp & 1) * 8);
    return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

// The same two readers with the per-lane part of the address precomputed (it does not depend on the key block): inside
// the loops only compile-time byte offsets remain (row0 * 128: the swizzles repeat every 16 rows), so a fragment read
// costs no vector ALU work -- the forward loop spent 62 of its ~190 non-transcendental VALU instructions on addresses.
//   rows: off_rows[st] = (lane&31)*128 + ((2*st + lane>>5) ^ swz_row(lane&31))*16         + row0*128 (row0 % 16 == 0)
//   cols: off_cols[db] = ra*128 + ((db*4 + (g&1)*2 + pp>>1) ^ swz(ra))*16 + (pp&1)*8,  ra = 4h+q;  second half: ra + 8
__device__ __forceinline__ void rows_offsets(int lane, int (&off)[4]) {
    const int row = lane & 31;
#pragma unroll
    for (int st = 0; st < 4; ++st) off[st] = row * 128 + (((2 * st + (lane >> 5)) ^ swz_row(row)) << 4);
}
template <bool TR>
__device__ __forceinline__ void cols_offsets(int lane, int (&off)[2], int (&offh)[2]) {
    const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    const int ra = 4 * (g >> 1) + q, rb = ra + 8;   // (the row swizzle of the second half differs: its own offset)
#pragma unroll
    for (int db = 0; db < 2; ++db) {
        const int chunk = db * 4 + (g & 1) * 2 + (pp >> 1);
        off[db] = ra * 128 + ((chunk ^ (TR ? swz_tr(ra) : swz_row(ra))) << 4) + (pp & 1) * 8;
        offh[db] = rb * 128 + ((chunk ^ (TR ? swz_tr(rb) : swz_row(rb))) << 4) + (pp & 1) * 8;
    }
}
__device__ __forceinline__ bf16x8 read_rows_at(const char* tile, int off, int row0) {
    return *(const bf16x8*)(tile + off + row0 * 128);
}
// (the asm read has no compiler-visible addressing mode: the constant part goes into its offset field by hand)
template <int BYTES>
__device__ __forceinline__ bf16x4 lds_tr16_imm(unsigned a) {
    bf16x4 r;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(a), "n"(BYTES) : "memory");
    return r;
}
// Round 3: the ring stage is a compile-time constant too (the key / query loops are unrolled over their two stages), so the
// stage base joins the immediate and a fragment read is `base register + offset field`, nothing else: the forward loop
// spent 41 of its ~230 vector-ALU issue slots per 64-key block on v_add_u32 of the stage base (the loops are bound by
// their vector ALU work: profiles/r02_attention.json).  abs_* = LDS byte address of (tile 0 of stage 0) + the lane's part.
#if TV_ATTN_ABL & 32
template <int IMM>
__device__ __forceinline__ bf16x8 read_rows_imm(const char* abs_row) {
    bf16x8 r;
    asm volatile("" : "=v"(r) : "v"(abs_row));
    return r;
}
template <int IMM>
__device__ __forceinline__ bf16x8 read_cols_imm(unsigned abs_lo, unsigned abs_hi) {
    bf16x8 r;
    asm volatile("" : "=v"(r) : "v"(abs_lo), "v"(abs_hi));
    return r;
}
#else
template <int IMM>
__device__ __forceinline__ bf16x8 read_rows_imm(const char* abs_row) {
    return *(const bf16x8*)(abs_row + IMM);
}
template <int IMM>
__device__ __forceinline__ bf16x8 read_cols_imm(unsigned abs_lo, unsigned abs_hi) {
    const bf16x4 lo = lds_tr16_imm<IMM>(abs_lo);
    const bf16x4 hi = lds_tr16_imm<IMM>(abs_hi);
    return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}
#endif
template <int R0>
__device__ __forceinline__ bf16x8 read_cols_at(const char* tile, int off, int offh) {
    const unsigned a = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)tile;
    const bf16x4 lo = lds_tr16_imm<R0 * 128>(a + off);
    const bf16x4 hi = lds_tr16_imm<R0 * 128>(a + offh);
    return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

// registers 8s..8s+7 of an accumulator tile -> bf16 fragment (B operand of the next MFMA)
__device__ __forceinline__ bf16x8 pack_acc(const f32x16& a, int s) {
    bf16x8 r;
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = (bf16)a[8 * s + j];
    return r;
}

struct AttnArgs {
    const bf16* qkv;
    const bf16* o;
    const bf16* d_o;
    bf16* out;    // fwd: o ; bwd: dqkv
    float* lse;
    float* delta;
    const char* zeros;
    const float* rope;   // backward only: table [N][4][32]; dq / dk are stored as gradients w.r.t. the UN-rotated projections
    int B, N, heads;
    float scale;
};

// adjoint of the reference's (non-orthogonal) RoPE 2x2 on four consecutive channels ch0 .. ch0+3 of a head (two pairs):
//   forward  o[2p] = a cos1 - b sin1,  o[2p+1] = a sin2 + b cos2      (R/transvae/modules/attention.py:156-197)
//   adjoint  a' = ga cos1 + gb sin2,   b' = -ga sin1 + gb cos2
__device__ __forceinline__ void rope_adjoint4(float (&v)[4], const float* tab_row, int ch0) {
    const float* tb = tab_row + (ch0 >> 1);
    const float c1a = tb[0], c1b = tb[1], s1a = tb[32], s1b = tb[33], c2a = tb[64], c2b = tb[65], s2a = tb[96], s2b = tb[97];
    const float a0 = v[0], b0 = v[1], a1 = v[2], b1 = v[3];
    v[0] = a0 * c1a + b0 * s2a;
    v[1] = -a0 * s1a + b0 * c2a;
    v[2] = a1 * c1b + b1 * s2b;
    v[3] = -a1 * s1b + b1 * c2b;
}

constexpr int KV_TILE = 64 * 128;  // bytes of one [64][64] bf16 tile

// Block -> (sequence tile, head, image).  All sequence tiles of one (image, head) stream the same K / V (forward, dq)
// or Q / dO (dk/dv) rows; placed on ONE XCD they share them through its L2 (workgroups go to XCDs round-robin by linear
// id, so head-image hb takes the ids congruent to hb mod 8).  Returns false for the padding ids of the last group.
__device__ __forceinline__ bool attn_block(const AttnArgs& p, int& tile, int& head, int& b) {
    const int tiles = (p.N + 127) / 128;
    const int lin = blockIdx.x, j = lin >> 3;
    tile = j % tiles;
    const int hb = (j / tiles) * 8 + (lin & 7);
    if (hb >= p.heads * p.B) return false;
    head = hb % p.heads;
    b = hb / p.heads;
    return true;
}

// ------------------------------------------------------------------------------------------------
// forward: block = 4 waves x 32 queries; loops over 64-key blocks (K: row tile, V: tr tile)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attn_fwd_kernel(const AttnArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];  // TV_ATTN_NS stages x (K 8 KiB + V 8 KiB)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int tile_x, head, b;
    if (!attn_block(p, tile_x, head, b)) return;
    const int C = p.heads * 64;
    const size_t ld = (size_t)3 * C;
    const bf16* qbase = p.qkv + (size_t)b * p.N * ld + head * 64;
    const bf16* kbase = qbase + C;
    const bf16* vbase = qbase + 2 * C;
    const int q0 = tile_x * 128 + wave * 32;
    const int qi = q0 + (lane & 31);
    const int h = lane >> 5;
    const bool q_ok = qi < p.N;

    bf16x8 qf[4];
#pragma unroll
    for (int st = 0; st < 4; ++st) {
        bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
        if (q_ok) v = *(const bf16x8*)(qbase + (size_t)qi * ld + 16 * st + 8 * h);
        qf[st] = v;
    }
    f32x16 ot[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) ot[0][i] = ot[1][i] = 0.f;
    float m = -INFINITY, l = 0.f;
    const float c2 = p.scale * 1.4426950408889634f;
    const int nblk = (p.N + 63) / 64;

    int off_k[4], off_v[2], off_vh[2];
    rows_offsets(lane, off_k);
    cols_offsets<true>(lane, off_v, off_vh);
    const unsigned smem_addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const char* abs_k[4];
    unsigned abs_v[2], abs_vh[2];
#pragma unroll
    for (int st = 0; st < 4; ++st) abs_k[st] = smem + off_k[st];
#pragma unroll
    for (int db = 0; db < 2; ++db) {
        abs_v[db] = smem_addr + off_v[db];
        abs_vh[db] = smem_addr + off_vh[db];
    }
    int vo_k[2], vo_v[2];
    stage_offsets<false>(vo_k, (int)ld, wave, lane);
    stage_offsets<true>(vo_v, (int)ld, wave, lane);
    const unsigned kv_bytes = stage_extent(p.N, (int)ld);
    constexpr int NS = TV_ATTN_NS, D = NS - 1;     // ring stages, DMA distance (tiles ahead)
    constexpr int PIECES = 4;                      // LDS-DMA instructions per wave and stage (K 2 + V 2)
    auto stage = [&](int t, char* sb) {
        stage_rows_buf(sb, kbase, kv_bytes, t * 64, (int)ld, vo_k, wave);
        stage_rows_buf(sb + KV_TILE, vbase, kv_bytes, t * 64, (int)ld, vo_v, wave);
    };
#pragma unroll
    for (int i = 0; i < D; ++i)
        if (i < nblk) stage(i, smem + i * 2 * KV_TILE);
#if TV_ATTN_SUM_MFMA
    // row sums of P on the matrix pipe: an A operand of ones makes every row of  ones x P^T  the column sums of P^T, i.e.
    // each lane receives sum_k P[q][k] over ALL 16 k slots (both lane halves) in every register of `lt` (only lt[0] is
    // used).  4 more MFMAs per 64-key block against 32 v_add_f32 + a cross-half exchange: the loop is bound by its vector
    // ALU work (~920 issue cycles per block beside 512 cycles of MFMA), the matrix pipe has the room.  The sum is taken over
    // the bf16-rounded P that V is multiplied with (a normaliser consistent with the numerator).
    const bf16 one_b = (bf16)1.0f;
    const bf16x8 ones = {one_b, one_b, one_b, one_b, one_b, one_b, one_b, one_b};
    f32x16 lt;
#pragma unroll
    for (int i = 0; i < 16; ++i) lt[i] = 0.f;
#endif
    // one 64-key block out of ring stage S (compile-time)
    auto key_block = [&](int t, auto stage_c) {
        constexpr int S = decltype(stage_c)::value;
        constexpr int KB = S * 2 * KV_TILE, VB = KB + KV_TILE;
        if (t + D - 1 < nblk) vm_wait<PIECES * (D - 1)>();
        else vm_wait<0>();
        block_sync();
#if !TV_ATTN_DMA_LATE
        if (!(TV_ATTN_ABL & 8) && t + D < nblk) stage(t + D, smem + ((S + D) % NS) * 2 * KV_TILE);
#endif
        f32x16 s[2];
#pragma unroll
        for (int i = 0; i < 16; ++i) s[0][i] = s[1][i] = 0.f;
        s[0] = mfma32(read_rows_imm<KB>(abs_k[0]), qf[0], s[0]);
        s[0] = mfma32(read_rows_imm<KB>(abs_k[1]), qf[1], s[0]);
        s[0] = mfma32(read_rows_imm<KB>(abs_k[2]), qf[2], s[0]);
        s[0] = mfma32(read_rows_imm<KB>(abs_k[3]), qf[3], s[0]);
        s[1] = mfma32(read_rows_imm<KB + 32 * 128>(abs_k[0]), qf[0], s[1]);
        s[1] = mfma32(read_rows_imm<KB + 32 * 128>(abs_k[1]), qf[1], s[1]);
        s[1] = mfma32(read_rows_imm<KB + 32 * 128>(abs_k[2]), qf[2], s[1]);
        s[1] = mfma32(read_rows_imm<KB + 32 * 128>(abs_k[3]), qf[3], s[1]);
#if TV_ATTN_DMA_LATE
        // the next stage's DMA pieces issue in the shadow of the eight S MFMAs (an LDS-DMA instruction holds its wave for
        // 60-180 cycles; right after the barrier all four waves paid that with the matrix pipe idle)
        __builtin_amdgcn_sched_barrier(0);
        if (!(TV_ATTN_ABL & 8) && t + D < nblk) stage(t + D, smem + ((S + D) % NS) * 2 * KV_TILE);
        __builtin_amdgcn_sched_barrier(0);
#endif
        bf16x8 vfr[2][2][2];
        auto read_v = [&]() {
#pragma unroll
            for (int db = 0; db < 2; ++db) {
                vfr[0][0][db] = read_cols_imm<VB + 0 * 128>(abs_v[db], abs_vh[db]);
                vfr[0][1][db] = read_cols_imm<VB + 16 * 128>(abs_v[db], abs_vh[db]);
                vfr[1][0][db] = read_cols_imm<VB + 32 * 128>(abs_v[db], abs_vh[db]);
                vfr[1][1][db] = read_cols_imm<VB + 48 * 128>(abs_v[db], abs_vh[db]);
            }
        };
#if !TV_ATTN_V_LATE
        // V^T fragments of the whole 64-key block (transposed reads, asm): issued now, consumed after the softmax
        read_v();
#endif
        // online softmax over this lane's 32 keys (+ the other half-wave's 32); scores stay unscaled, the scale
        // rides in the exp2 FMA:  p = exp2(s*c2 - m*c2)
        const int kv0 = t * 64;
        const bool partial = kv0 + 64 > p.N;
        if (partial) {   // only the last key block of a ragged sequence: mask keys >= N (wave-uniform branch)
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kv0 + kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    if (key >= p.N) s[kt][r] = -INFINITY;
                }
        }
        float mloc = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) mloc = fmaxf(mloc, s[kt][r]);
        mloc = xhalf_max(mloc);
        // Some row's maximum moved (TV_ATTN_LAZY = 0) / moved by more than 2^LAZY in the exponent: rescale the running sums
        // (wave-uniform branch).  With LAZY > 0 a row may keep a maximum that is up to LAZY below its true one: p <= 2^LAZY,
        // the same RELATIVE precision in bf16, sums in fp32; the statistic that leaves the kernel (lse) does not depend on it.
#if TV_ATTN_LAZY > 0
        const bool need = __any((mloc - m) * c2 > (float)TV_ATTN_LAZY);
#else
        const bool need = __any(mloc > m);
#endif
        if (need) {
            const float mnew = fmaxf(m, mloc);
            const float alpha = fexp2((m - mnew) * c2);
            m = mnew;
#if TV_ATTN_SUM_MFMA
            lt[0] *= alpha;
#else
            l *= alpha;
#endif
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                ot[0][i] *= alpha;
                ot[1][i] *= alpha;
            }
        }
        const float mc = m * c2;
#if !TV_ATTN_SUM_MFMA
        float rs = 0.f;
#endif
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pv = fexp2(fmaf(s[kt][r], c2, -mc));
                s[kt][r] = pv;
#if !TV_ATTN_SUM_MFMA
                rs += pv;
#endif
            }
#if !TV_ATTN_SUM_MFMA
        l += rs;           // this lane half's keys only: the halves are added once, after the loop
#endif
#if TV_ATTN_V_LATE == 1
        __builtin_amdgcn_sched_barrier(0);
        read_v();          // 32 registers less across the softmax
#endif
#if TV_ATTN_V_LATE == 2
        // V^T fragments 32 keys at a time, right before their MFMAs: 16 live registers instead of 32 held across the softmax
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            bf16x8 vh[2][2];
#pragma unroll
            for (int db = 0; db < 2; ++db) {
                if (kt == 0) {
                    vh[0][db] = read_cols_imm<VB + 0 * 128>(abs_v[db], abs_vh[db]);
                    vh[1][db] = read_cols_imm<VB + 16 * 128>(abs_v[db], abs_vh[db]);
                } else {
                    vh[0][db] = read_cols_imm<VB + 32 * 128>(abs_v[db], abs_vh[db]);
                    vh[1][db] = read_cols_imm<VB + 48 * 128>(abs_v[db], abs_vh[db]);
                }
            }
            bf16x8 pf[2] = {pack_acc(s[kt], 0), pack_acc(s[kt], 1)};
            lds_wait_all();
#pragma unroll
            for (int ss = 0; ss < 2; ++ss)
#pragma unroll
                for (int db = 0; db < 2; ++db) ot[db] = mfma32b(vh[ss][db], pf[ss], ot[db]);
        }
#else
        lds_wait_all();
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int ss = 0; ss < 2; ++ss) {
                const bf16x8 pf = pack_acc(s[kt], ss);
#pragma unroll
                for (int db = 0; db < 2; ++db) ot[db] = mfma32b(vfr[kt][ss][db], pf, ot[db]);
#if TV_ATTN_SUM_MFMA
                lt = mfma32b(ones, pf, lt);
#endif
            }
#endif
    };
    ring_for<NS>(nblk, key_block);
#if TV_ATTN_SUM_MFMA
    l = lt[0];
#else
    l = xhalf_sum(l);
#endif
    if (q_ok) {
        const float inv = 1.0f / l;
        bf16* orow = p.out + ((size_t)b * p.N + qi) * C + head * 64;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                bf16x4 v = {(bf16)(ot[db][4 * g] * inv), (bf16)(ot[db][4 * g + 1] * inv), (bf16)(ot[db][4 * g + 2] * inv),
                            (bf16)(ot[db][4 * g + 3] * inv)};
                attn_store4(orow + db * 32 + 8 * g + 4 * h, v);
            }
        if (h == 0) p.lse[((size_t)b * p.heads + head) * p.N + qi] = (m * c2 + __log2f(l)) * 0.6931471805599453f;
    }
}

// ------------------------------------------------------------------------------------------------
// forward, 64 queries per wave (TV_ATTN_FWD64): block = 4 waves x 64 queries = 256 queries.
// Ablation of the 32-query kernel (profiles/r03_kernel_experiments.txt item 11): without its LDS fragment reads it runs 31 %
// faster, without the DMA issue 15 % -- a wave with 32 queries needs one K / V fragment from LDS per MFMA.  Here a wave holds
// TWO query tiles (a, b): a K fragment (and a V^T fragment) read once feeds two MFMAs, and a block stages the same K / V
// bytes for twice the matrix work.  The 64-key stage is walked as two 32-key halves so that only two S tiles are alive
// (one per query tile): registers ~190, two blocks per CU.
// ------------------------------------------------------------------------------------------------
template <int BLOCK_Q>
__device__ __forceinline__ bool attn_block_q(const AttnArgs& p, int& tile, int& head, int& b) {
    const int tiles = (p.N + BLOCK_Q - 1) / BLOCK_Q;
    const int lin = blockIdx.x, j = lin >> 3;
    tile = j % tiles;
    const int hb = (j / tiles) * 8 + (lin & 7);
    if (hb >= p.heads * p.B) return false;
    head = hb % p.heads;
    b = hb / p.heads;
    return true;
}

#ifndef TV_ATTN_FWD64_OCC
#define TV_ATTN_FWD64_OCC 3   // waves per SIMD the register allocation must leave room for (176 registers / 2 waves: 1.813 ms; capped at 168 / 3 waves, 7 dwords spilled: 1.693 ms)
#endif
__global__ __launch_bounds__(256, TV_ATTN_FWD64_OCC) void attn_fwd64_kernel(const AttnArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];  // TV_ATTN_NS stages x (K 8 KiB + V 8 KiB)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int tile_x, head, b;
    if (!attn_block_q<256>(p, tile_x, head, b)) return;
    const int C = p.heads * 64;
    const size_t ld = (size_t)3 * C;
    const bf16* qbase = p.qkv + (size_t)b * p.N * ld + head * 64;
    const bf16* kbase = qbase + C;
    const bf16* vbase = qbase + 2 * C;
    const int q0 = tile_x * 256 + wave * 64;
    const int h = lane >> 5;
    int qi[2];
    bool q_ok[2];
    bf16x8 qf[2][4];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        qi[u] = q0 + 32 * u + (lane & 31);
        q_ok[u] = qi[u] < p.N;
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
            if (q_ok[u]) v = *(const bf16x8*)(qbase + (size_t)qi[u] * ld + 16 * st + 8 * h);
            qf[u][st] = v;
        }
    }
    f32x16 ot[2][2];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int i = 0; i < 16; ++i) ot[u][0][i] = ot[u][1][i] = 0.f;
    float m[2] = {-INFINITY, -INFINITY}, l[2] = {0.f, 0.f};
    const float c2 = p.scale * 1.4426950408889634f;
    const int nblk = (p.N + 63) / 64;

    int off_k[4], off_v[2], off_vh[2];
    rows_offsets(lane, off_k);
    cols_offsets<true>(lane, off_v, off_vh);
    const unsigned smem_addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const char* abs_k[4];
    unsigned abs_v[2], abs_vh[2];
#pragma unroll
    for (int st = 0; st < 4; ++st) abs_k[st] = smem + off_k[st];
#pragma unroll
    for (int db = 0; db < 2; ++db) {
        abs_v[db] = smem_addr + off_v[db];
        abs_vh[db] = smem_addr + off_vh[db];
    }
    int vo_k[2], vo_v[2];
    stage_offsets<false>(vo_k, (int)ld, wave, lane);
    stage_offsets<true>(vo_v, (int)ld, wave, lane);
    const unsigned kv_bytes = stage_extent(p.N, (int)ld);
    constexpr int NS = TV_ATTN_NS, D = NS - 1;
    constexpr int PIECES = 4;
    auto stage = [&](int t, char* sb) {
        stage_rows_buf(sb, kbase, kv_bytes, t * 64, (int)ld, vo_k, wave);
        stage_rows_buf(sb + KV_TILE, vbase, kv_bytes, t * 64, (int)ld, vo_v, wave);
    };
#pragma unroll
    for (int i = 0; i < D; ++i)
        if (i < nblk) stage(i, smem + i * 2 * KV_TILE);
    // 32 keys (half KT of the stage's 64) against both query tiles
    auto half_block = [&](int t, auto stage_c, auto kt_c) {
        constexpr int S = decltype(stage_c)::value, KT = decltype(kt_c)::value;
        constexpr int KB = S * 2 * KV_TILE + KT * 32 * 128, VB = S * 2 * KV_TILE + KV_TILE + KT * 32 * 128;
        f32x16 s[2];
#pragma unroll
        for (int i = 0; i < 16; ++i) s[0][i] = s[1][i] = 0.f;
        {
            const bf16x8 k0 = read_rows_imm<KB>(abs_k[0]), k1 = read_rows_imm<KB>(abs_k[1]);
            const bf16x8 k2 = read_rows_imm<KB>(abs_k[2]), k3 = read_rows_imm<KB>(abs_k[3]);
            s[0] = mfma32(k0, qf[0][0], s[0]);
            s[1] = mfma32(k0, qf[1][0], s[1]);
            s[0] = mfma32(k1, qf[0][1], s[0]);
            s[1] = mfma32(k1, qf[1][1], s[1]);
            s[0] = mfma32(k2, qf[0][2], s[0]);
            s[1] = mfma32(k2, qf[1][2], s[1]);
            s[0] = mfma32(k3, qf[0][3], s[0]);
            s[1] = mfma32(k3, qf[1][3], s[1]);
        }
        if constexpr (KT == 0) {   // the next stage's DMA pieces in the shadow of the first S MFMAs
            __builtin_amdgcn_sched_barrier(0);
            if (!(TV_ATTN_ABL & 8) && t + D < nblk) stage(t + D, smem + ((S + D) % NS) * 2 * KV_TILE);
            __builtin_amdgcn_sched_barrier(0);
        }
        bf16x8 vfr[2][2];   // V^T fragments of these 32 keys: [16-key half][d block]
#pragma unroll
        for (int db = 0; db < 2; ++db) {
            vfr[0][db] = read_cols_imm<VB + 0 * 128>(abs_v[db], abs_vh[db]);
            vfr[1][db] = read_cols_imm<VB + 16 * 128>(abs_v[db], abs_vh[db]);
        }
        const int kv0 = t * 64 + KT * 32;
        if (kv0 + 32 > p.N) {   // ragged end of the sequence: mask keys >= N (wave-uniform branch)
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kv0 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    if (key >= p.N) s[u][r] = -INFINITY;
                }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            float mloc = -INFINITY;
#pragma unroll
            for (int r = 0; r < 16; ++r) mloc = fmaxf(mloc, s[u][r]);
            mloc = xhalf_max(mloc);
            // (a half-block whose 32 keys are all masked leaves mloc = -inf: no rescale, p = exp2(-inf) = 0)
#if TV_ATTN_LAZY > 0
            const bool need = __any((mloc - m[u]) * c2 > (float)TV_ATTN_LAZY);
#else
            const bool need = __any(mloc > m[u]);
#endif
            if (need) {
                const float mnew = fmaxf(m[u], mloc);
                const float alpha = fexp2((m[u] - mnew) * c2);
                m[u] = mnew;
                l[u] *= alpha;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    ot[u][0][i] *= alpha;
                    ot[u][1][i] *= alpha;
                }
            }
            const float mc = m[u] * c2;
            float rs = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pv = fexp2(fmaf(s[u][r], c2, -mc));
                s[u][r] = pv;
                rs += pv;
            }
            l[u] += rs;
        }
        lds_wait_all();
#pragma unroll
        for (int ss = 0; ss < 2; ++ss) {
            const bf16x8 pa = pack_acc(s[0], ss), pb = pack_acc(s[1], ss);
#pragma unroll
            for (int db = 0; db < 2; ++db) {
                ot[0][db] = mfma32b(vfr[ss][db], pa, ot[0][db]);
                ot[1][db] = mfma32b(vfr[ss][db], pb, ot[1][db]);
            }
        }
    };
    auto key_block = [&](int t, auto stage_c) {
        if (t + D - 1 < nblk) vm_wait<PIECES * (D - 1)>();
        else vm_wait<0>();
        block_sync();
        half_block(t, stage_c, ic<0>{});
        if (t * 64 + 32 < p.N) half_block(t, stage_c, ic<1>{});
    };
    ring_for<NS>(nblk, key_block);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const float lt = xhalf_sum(l[u]);
        if (q_ok[u]) {
            const float inv = 1.0f / lt;
            bf16* orow = p.out + ((size_t)b * p.N + qi[u]) * C + head * 64;
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    bf16x4 v = {(bf16)(ot[u][db][4 * g] * inv), (bf16)(ot[u][db][4 * g + 1] * inv), (bf16)(ot[u][db][4 * g + 2] * inv),
                                (bf16)(ot[u][db][4 * g + 3] * inv)};
                    attn_store4(orow + db * 32 + 8 * g + 4 * h, v);
                }
            if (h == 0) p.lse[((size_t)b * p.heads + head) * p.N + qi[u]] = (m[u] * c2 + __log2f(lt)) * 0.6931471805599453f;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// delta[b][head][q] = sum_d dO[q][d] * O[q][d]
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attn_delta_kernel(const AttnArgs p) {
    const long long total = (long long)p.B * p.N * p.heads * 8;  // 8 lanes per (token, head)
    const int C = p.heads * 64;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int v = (int)(idx & 7);
        const long long th = idx >> 3;
        const int head = (int)(th % p.heads);
        const long long tok = th / p.heads;  // b*N + n
        const size_t off = (size_t)tok * C + head * 64 + v * 8;
        const bf16x8 a = *(const bf16x8*)(p.o + off);
        const bf16x8 g = *(const bf16x8*)(p.d_o + off);
        float acc = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) acc = fmaf((float)a[e], (float)g[e], acc);
        acc += __shfl_xor(acc, 1, 64);
        acc += __shfl_xor(acc, 2, 64);
        acc += __shfl_xor(acc, 4, 64);
        if (v == 0) {   // stored NEGATED, beside -lse log2(e): both enter the backward kernels as addends (see attn_bwd_dq_kernel)
            const long long bb = tok / p.N;
            const int n = (int)(tok - bb * p.N);
            const size_t at = ((size_t)bb * p.heads + head) * p.N + n;
            p.delta[at] = -acc;
            p.delta[(size_t)p.B * p.heads * p.N + at] = -p.lse[at] * 1.4426950408889634f;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// dq: block = 4 waves x 32 queries, loops over 64-key blocks.  K tile: rows + transposed reads,
// V tile: rows.   dS^T = P^T o (dP^T - delta) ;  dQ^T = K^T dS^T
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(const AttnArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int tile_x, head, b;
    if (!attn_block(p, tile_x, head, b)) return;
    const int C = p.heads * 64;
    const size_t ld = (size_t)3 * C;
    const bf16* qbase = p.qkv + (size_t)b * p.N * ld + head * 64;
    const bf16* kbase = qbase + C;
    const bf16* vbase = qbase + 2 * C;
    const int q0 = tile_x * 128 + wave * 32;
    const int qi = q0 + (lane & 31);
    const int h = lane >> 5;
    const bool q_ok = qi < p.N;

    bf16x8 qf[4], gf[4];
#pragma unroll
    for (int st = 0; st < 4; ++st) {
        bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0}, g = {0, 0, 0, 0, 0, 0, 0, 0};
        if (q_ok) {
            v = *(const bf16x8*)(qbase + (size_t)qi * ld + 16 * st + 8 * h);
            g = *(const bf16x8*)(p.d_o + ((size_t)b * p.N + qi) * C + head * 64 + 16 * st + 8 * h);
        }
        qf[st] = v;
        gf[st] = g;
    }
    // P = exp2(S c2 + nl), dS = P (dP + nd) with nl = -lse log2(e), nd = -delta (attn_delta_kernel).  The accumulators start
    // from the inline constant 0: starting them from -lse / scale and -delta instead (so that the MFMA chain delivers S - lse
    // and dP - delta) costs 32 register broadcasts per 32-key tile in a loop that is bound by its vector ALU work
    // (MFMA pipe 0.42-0.45 busy, profiles/r02_attention.json).
    float nl = 0.f, nd = 0.f;
    if (q_ok) {
        nd = p.delta[((size_t)b * p.heads + head) * p.N + qi];
        nl = p.delta[((size_t)p.B + b) * p.heads * p.N + (size_t)head * p.N + qi];
    }
    const float c2 = p.scale * 1.4426950408889634f;
    f32x16 dqt[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) dqt[0][i] = dqt[1][i] = 0.f;
    const int nblk = (p.N + 63) / 64;
    int off_r[4], off_c[2], off_ch[2];
    rows_offsets(lane, off_r);
    cols_offsets<false>(lane, off_c, off_ch);
    const unsigned smem_addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const char* abs_r[4];
    unsigned abs_c[2], abs_ch[2];
#pragma unroll
    for (int st = 0; st < 4; ++st) abs_r[st] = smem + off_r[st];
#pragma unroll
    for (int db = 0; db < 2; ++db) {
        abs_c[db] = smem_addr + off_c[db];
        abs_ch[db] = smem_addr + off_ch[db];
    }

    int vo_kv[2];
    stage_offsets<false>(vo_kv, (int)ld, wave, lane);
    const unsigned kv_bytes = stage_extent(p.N, (int)ld);
    constexpr int NS = TV_ATTN_NS, D = NS - 1;
    constexpr int PIECES = 4;
    auto stage = [&](int t, char* sb) {
        stage_rows_buf(sb, kbase, kv_bytes, t * 64, (int)ld, vo_kv, wave);
        stage_rows_buf(sb + KV_TILE, vbase, kv_bytes, t * 64, (int)ld, vo_kv, wave);
    };
#pragma unroll
    for (int i = 0; i < D; ++i)
        if (i < nblk) stage(i, smem + i * 2 * KV_TILE);
#if TV_ATTN_DELTA_MFMA
    // dP - delta out of the MFMA chain: one more k-step whose A operand holds 1 in k slots 0..2 (every key row) and whose B
    // operand holds -delta of the lane's query as three bf16 pieces (hi + mid + lo = the fp32 value to 2^-24) in the same
    // slots.  The query is on the LANE here, so starting the accumulator from -delta would take 16 register broadcasts per
    // tile -- as many vector instructions as the 16 v_add_f32 it removes; the extra MFMA takes none (the loop is bound by
    // its vector ALU work, the matrix pipe is ~0.58 busy).
    bf16x8 dl_a = {0, 0, 0, 0, 0, 0, 0, 0}, dl_b = {0, 0, 0, 0, 0, 0, 0, 0};
    {
        const bf16 d0 = (bf16)nd;
        const float r1 = nd - (float)d0;
        const bf16 d1 = (bf16)r1;
        const bf16 d2 = (bf16)(r1 - (float)d1);
        if (h == 0) {
            const bf16 one_b = (bf16)1.0f;
            dl_a[0] = dl_a[1] = dl_a[2] = one_b;
            dl_b[0] = d0;
            dl_b[1] = d1;
            dl_b[2] = d2;
        }
    }
#endif
    // 32 keys (half KT of a 64-key block) out of ring stage S, both compile-time
    auto half_block = [&](int t, auto stage_c, auto kt_c) {
        constexpr int S = decltype(stage_c)::value, KT = decltype(kt_c)::value;
        constexpr int KB = S * 2 * KV_TILE, VB = KB + KV_TILE, RO = KT * 32 * 128;
        f32x16 s, dp;
#pragma unroll
        for (int i = 0; i < 16; ++i) s[i] = dp[i] = 0.f;
#if TV_ATTN_DELTA_MFMA
        dp = mfma32(dl_a, dl_b, dp);
#endif
        s = mfma32(read_rows_imm<KB + RO>(abs_r[0]), qf[0], s);
        dp = mfma32(read_rows_imm<VB + RO>(abs_r[0]), gf[0], dp);
        s = mfma32(read_rows_imm<KB + RO>(abs_r[1]), qf[1], s);
        dp = mfma32(read_rows_imm<VB + RO>(abs_r[1]), gf[1], dp);
        s = mfma32(read_rows_imm<KB + RO>(abs_r[2]), qf[2], s);
        dp = mfma32(read_rows_imm<VB + RO>(abs_r[2]), gf[2], dp);
        s = mfma32(read_rows_imm<KB + RO>(abs_r[3]), qf[3], s);
        dp = mfma32(read_rows_imm<VB + RO>(abs_r[3]), gf[3], dp);
#if TV_ATTN_DMA_LATE
        if constexpr (KT == 0) {   // the next stage's DMA pieces in the shadow of the first S / dP MFMAs (see attn_fwd_kernel)
            __builtin_amdgcn_sched_barrier(0);
            if (!(TV_ATTN_ABL & 8) && t + D < nblk) stage(t + D, smem + ((S + D) % NS) * 2 * KV_TILE);
            __builtin_amdgcn_sched_barrier(0);
        }
#endif
        bf16x8 kfr[2][2];   // K^T fragments (transposed asm reads): in flight during the exponentials
#pragma unroll
        for (int db = 0; db < 2; ++db) {
            kfr[0][db] = read_cols_imm<KB + RO>(abs_c[db], abs_ch[db]);
            kfr[1][db] = read_cols_imm<KB + RO + 16 * 128>(abs_c[db], abs_ch[db]);
        }
        const int kv0 = t * 64;
        if (kv0 + 64 > p.N) {   // ragged last block: keys >= N must not contribute (P = 0): push their score to -inf
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kv0 + KT * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (key >= p.N) s[r] = -INFINITY;
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {   // dS^T (without the factor `scale`)
#if TV_ATTN_DELTA_MFMA
            s[r] = fexp2(fmaf(s[r], c2, nl)) * dp[r];
#else
            s[r] = fexp2(fmaf(s[r], c2, nl)) * (dp[r] + nd);
#endif
        }
        lds_wait_all();
#pragma unroll
        for (int ss = 0; ss < 2; ++ss) {
            const bf16x8 df = pack_acc(s, ss);
#pragma unroll
            for (int db = 0; db < 2; ++db) dqt[db] = mfma32b(kfr[ss][db], df, dqt[db]);
        }
    };
    auto key_block = [&](int t, auto stage_c) {
        constexpr int S = decltype(stage_c)::value;
        if (t + D - 1 < nblk) vm_wait<PIECES * (D - 1)>();
        else vm_wait<0>();
        block_sync();
#if !TV_ATTN_DMA_LATE
        if (!(TV_ATTN_ABL & 8) && t + D < nblk) stage(t + D, smem + ((S + D) % NS) * 2 * KV_TILE);
#endif
        half_block(t, stage_c, ic<0>{});
        half_block(t, stage_c, ic<1>{});
    };
    ring_for<NS>(nblk, key_block);
    if (q_ok) {
        bf16* row = p.out + ((size_t)b * p.N + qi) * ld + head * 64;  // q third of dqkv
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float f[4] = {dqt[db][4 * g] * p.scale, dqt[db][4 * g + 1] * p.scale, dqt[db][4 * g + 2] * p.scale, dqt[db][4 * g + 3] * p.scale};
                if (p.rope) rope_adjoint4(f, p.rope + (size_t)qi * 128, db * 32 + 8 * g + 4 * h);
                bf16x4 v = {(bf16)f[0], (bf16)f[1], (bf16)f[2], (bf16)f[3]};
                attn_store4(row + db * 32 + 8 * g + 4 * h, v);
            }
    }
}

// ------------------------------------------------------------------------------------------------
// dk / dv: block = 4 waves x 32 keys (key on the lane), loops over 32-query tiles.
// LDS stage: Q tile [32][64] (4 KiB), dO tile [32][64] (4 KiB), lse[32] + delta[32] fp32 (256 B)
// ------------------------------------------------------------------------------------------------
#ifndef TV_ATTN_DKV_QS
#define TV_ATTN_DKV_QS 2     // 32-query tiles per ring stage of the dk / dv kernel (one barrier and one DMA wait per stage)
#endif
constexpr int DKV_QS = TV_ATTN_DKV_QS;
constexpr int QT_TILE = DKV_QS * 32 * 128;
constexpr int DKV_STAGE = 2 * QT_TILE + DKV_QS * 256;

__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(const AttnArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int tile_x, head, b;
    if (!attn_block(p, tile_x, head, b)) return;
    const int C = p.heads * 64;
    const size_t ld = (size_t)3 * C;
    const bf16* qbase = p.qkv + (size_t)b * p.N * ld + head * 64;
    const bf16* kbase = qbase + C;
    const bf16* vbase = qbase + 2 * C;
    const bf16* gbase = p.d_o + (size_t)b * p.N * C + head * 64;
    const float* del_b = p.delta + ((size_t)b * p.heads + head) * p.N;                 // -delta
    const float* lse_b = del_b + (size_t)p.B * p.heads * p.N;                          // -lse log2(e)
    const int k0 = tile_x * 128 + wave * 32;
    const int ki = k0 + (lane & 31);
    const int h = lane >> 5;
    const bool k_ok = ki < p.N;

    bf16x8 kf[4], vf[4];
#pragma unroll
    for (int st = 0; st < 4; ++st) {
        bf16x8 a = {0, 0, 0, 0, 0, 0, 0, 0}, v = {0, 0, 0, 0, 0, 0, 0, 0};
        if (k_ok) {
            a = *(const bf16x8*)(kbase + (size_t)ki * ld + 16 * st + 8 * h);
            v = *(const bf16x8*)(vbase + (size_t)ki * ld + 16 * st + 8 * h);
        }
        kf[st] = a;
        vf[st] = v;
    }
    f32x16 dkt[2], dvt[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) dkt[0][i] = dkt[1][i] = dvt[0][i] = dvt[1][i] = 0.f;
    const float c2 = p.scale * 1.4426950408889634f;
    const int ntile = (p.N + 32 * DKV_QS - 1) / (32 * DKV_QS);   // ring stages of DKV_QS 32-query tiles
    int off_r[4], off_c[2], off_ch[2];
    rows_offsets(lane, off_r);
    cols_offsets<false>(lane, off_c, off_ch);

    int vo_q[DKV_QS], vo_g[DKV_QS];
    stage_offsets<false>(vo_q, (int)ld, wave, lane);
    stage_offsets<false>(vo_g, C, wave, lane);
    const unsigned q_bytes = stage_extent(p.N, (int)ld), g_bytes = stage_extent(p.N, C);
    auto stage = [&](int t, char* sb) {
        stage_rows_buf(sb, qbase, q_bytes, t * 32 * DKV_QS, (int)ld, vo_q, wave);
        stage_rows_buf(sb + QT_TILE, gbase, g_bytes, t * 32 * DKV_QS, C, vo_g, wave);
        if (wave < DKV_QS) {  // wave u: statistics of the stage's tile u -- lanes 0-31: lse, lanes 32-63: delta (4-byte LDS-DMA)
            const int q = (t * DKV_QS + wave) * 32 + (lane & 31);
            const float* src = (q < p.N) ? ((lane < 32 ? lse_b : del_b) + q) : (const float*)(p.zeros + lane * 4);
            __builtin_amdgcn_global_load_lds(TV_GLB(src), TV_LDS(sb + 2 * QT_TILE + wave * 256), 4, 0, 0);
        }
    };

    const unsigned smem_addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const char* abs_r[4];
    unsigned abs_c[2], abs_ch[2];
#pragma unroll
    for (int st = 0; st < 4; ++st) abs_r[st] = smem + off_r[st];
#pragma unroll
    for (int db = 0; db < 2; ++db) {
        abs_c[db] = smem_addr + off_c[db];
        abs_ch[db] = smem_addr + off_ch[db];
    }
    const char* abs_st = smem + 16 * h;      // this lane half's part of the lse / delta rows (the stage and tile offsets are immediates)
    constexpr int NS = TV_ATTN_NS_DKV, D = NS - 1;
#pragma unroll
    for (int i = 0; i < D; ++i)
        if (i < ntile) stage(i, smem + i * DKV_STAGE);
    // 32-query tile U of ring stage S (both compile-time)
    auto query_tile = [&](int t, auto stage_c, auto sub_c) {
        constexpr int S = decltype(stage_c)::value, U = decltype(sub_c)::value;
        constexpr int QB = S * DKV_STAGE + U * 32 * 128, GB = QB + QT_TILE;
        constexpr int STB = S * DKV_STAGE + 2 * QT_TILE + U * 256;   // the tile's lse / delta rows
        f32x16 s, dp;
#pragma unroll
        for (int i = 0; i < 16; ++i) s[i] = 0.f;   // (inline-constant accumulator start: see attn_bwd_dq_kernel)
        f32x4 nl[4], nd[4];                        // accumulator rows (queries) 8g+4h+{0..3}
        nl[0] = *(const f32x4*)(abs_st + STB + 0);
        nl[1] = *(const f32x4*)(abs_st + STB + 32);
        nl[2] = *(const f32x4*)(abs_st + STB + 64);
        nl[3] = *(const f32x4*)(abs_st + STB + 96);
        nd[0] = *(const f32x4*)(abs_st + STB + 128 + 0);
        nd[1] = *(const f32x4*)(abs_st + STB + 128 + 32);
        nd[2] = *(const f32x4*)(abs_st + STB + 128 + 64);
        nd[3] = *(const f32x4*)(abs_st + STB + 128 + 96);
#if TV_ATTN_DKV_ACCINIT
        // the query is on the accumulator ROW here: -delta of rows 8g+4h+{0..3} is one broadcast 16-byte LDS read per g and
        // IS dP's initial accumulator as it arrives (no 16 v_add_f32 per tile, no second copy in registers)
#pragma unroll
        for (int r = 0; r < 16; ++r) dp[r] = nd[r >> 2][r & 3];
#else
#pragma unroll
        for (int i = 0; i < 16; ++i) dp[i] = 0.f;
#endif
        s = mfma32(read_rows_imm<QB>(abs_r[0]), kf[0], s);
        dp = mfma32(read_rows_imm<GB>(abs_r[0]), vf[0], dp);
        s = mfma32(read_rows_imm<QB>(abs_r[1]), kf[1], s);
        dp = mfma32(read_rows_imm<GB>(abs_r[1]), vf[1], dp);
        s = mfma32(read_rows_imm<QB>(abs_r[2]), kf[2], s);
        dp = mfma32(read_rows_imm<GB>(abs_r[2]), vf[2], dp);
        s = mfma32(read_rows_imm<QB>(abs_r[3]), kf[3], s);
        dp = mfma32(read_rows_imm<GB>(abs_r[3]), vf[3], dp);
#if TV_ATTN_DMA_LATE
        if constexpr (U == 0) {   // the next stage's pieces in the shadow of the stage's first S / dP MFMAs
            __builtin_amdgcn_sched_barrier(0);
            if (!(TV_ATTN_ABL & 8) && t + D < ntile) stage(t + D, smem + ((S + D) % NS) * DKV_STAGE);
            __builtin_amdgcn_sched_barrier(0);
        }
#endif
        bf16x8 gfr[2][2], qfr[2][2];   // dO^T and Q^T fragments (transposed asm reads)
#pragma unroll
        for (int db = 0; db < 2; ++db) {
            gfr[0][db] = read_cols_imm<GB>(abs_c[db], abs_ch[db]);
            gfr[1][db] = read_cols_imm<GB + 16 * 128>(abs_c[db], abs_ch[db]);
            qfr[0][db] = read_cols_imm<QB>(abs_c[db], abs_ch[db]);
            qfr[1][db] = read_cols_imm<QB + 16 * 128>(abs_c[db], abs_ch[db]);
        }
        // queries beyond N have Q = dO = 0, lse = delta = 0  =>  P = 1, dS = 0, and dO^T P adds 0
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float pv = fexp2(fmaf(s[r], c2, nl[r >> 2][r & 3]));
            s[r] = pv;
#if TV_ATTN_DKV_ACCINIT
            dp[r] = pv * dp[r];
#else
            dp[r] = pv * (dp[r] + nd[r >> 2][r & 3]);
#endif
        }
        lds_wait_all();
#pragma unroll
        for (int ss = 0; ss < 2; ++ss) {
            const bf16x8 pf = pack_acc(s, ss);
            const bf16x8 df = pack_acc(dp, ss);
#pragma unroll
            for (int db = 0; db < 2; ++db) {
                dvt[db] = mfma32b(gfr[ss][db], pf, dvt[db]);
                dkt[db] = mfma32b(qfr[ss][db], df, dkt[db]);
            }
        }
    };
    // a ring stage: one counted wait and one barrier, then its DKV_QS tiles
    auto query_stage = [&](int t, auto stage_c) {
        constexpr int S = decltype(stage_c)::value;
        if (t + D - 1 < ntile) {       // per-wave DMA instructions of a stage: Q and dO DKV_QS each, waves < DKV_QS a statistics row too
            if (wave < DKV_QS) vm_wait<(2 * DKV_QS + 1) * (D - 1)>();
            else vm_wait<2 * DKV_QS * (D - 1)>();
        } else {
            vm_wait<0>();
        }
        block_sync();
#if !TV_ATTN_DMA_LATE
        if (!(TV_ATTN_ABL & 8) && t + D < ntile) stage(t + D, smem + ((S + D) % NS) * DKV_STAGE);
#endif
        query_tile(t, stage_c, ic<0>{});
        if constexpr (DKV_QS > 1) {
            if ((t * DKV_QS + 1) * 32 < p.N) query_tile(t, stage_c, ic<1>{});
        }
    };
    ring_for<NS>(ntile, query_stage);
    if (k_ok) {
        bf16* krow = p.out + ((size_t)b * p.N + ki) * ld + C + head * 64;
        bf16* vrow = krow + C;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float f[4] = {dkt[db][4 * g] * p.scale, dkt[db][4 * g + 1] * p.scale, dkt[db][4 * g + 2] * p.scale, dkt[db][4 * g + 3] * p.scale};
                if (p.rope) rope_adjoint4(f, p.rope + (size_t)ki * 128, db * 32 + 8 * g + 4 * h);
                bf16x4 a = {(bf16)f[0], (bf16)f[1], (bf16)f[2], (bf16)f[3]};
                bf16x4 v = {(bf16)dvt[db][4 * g], (bf16)dvt[db][4 * g + 1], (bf16)dvt[db][4 * g + 2], (bf16)dvt[db][4 * g + 3]};
                attn_store4(krow + db * 32 + 8 * g + 4 * h, a);
                attn_store4(vrow + db * 32 + 8 * g + 4 * h, v);
            }
    }
}

int attn_check(const char* name, int B, int N, int heads, float scale) {
    if (B <= 0 || N <= 0 || heads <= 0 || !(scale > 0.f) || B > 65535 || heads > 65535) {
        tv_set_error("%s: bad shape B=%d N=%d heads=%d scale=%g", name, B, N, heads, (double)scale);
        return TV_ERR_ARG;
    }
    return TV_OK;
}

}  // namespace

static int g_attn_bwd_mask = 7;   // timing hook (tools/probes): 1 delta, 2 dq, 4 dk/dv; not part of the public ABI
extern "C" int tv_set_attn_bwd_mask(int mask) {
    g_attn_bwd_mask = mask;
    return TV_OK;
}

extern "C" int tv_attn_fwd(const void* qkv, void* o, float* lse, int B, int N, int heads, float scale, void* stream) {
    if (attn_check("tv_attn_fwd", B, N, heads, scale)) return TV_ERR_ARG;
    TV_CHECK_ARG(qkv && o && lse, "tv_attn_fwd: null pointer");
    TV_CHECK_ARG((long long)N * heads * 64 * 3 * 2 < (1ll << 31), "tv_attn_fwd: one image's qkv must stay below 2 GiB (32-bit DMA offsets)");
    if (tv_init() != TV_OK) return TV_ERR_INIT;
    AttnArgs a{};
    a.qkv = (const bf16*)qkv; a.out = (bf16*)o; a.lse = lse; a.zeros = (const char*)tv_zero_page();
    a.B = B; a.N = N; a.heads = heads; a.scale = scale;
#if TV_ATTN_FWD64
    if (N >= TV_ATTN_FWD64) {   // long sequences: 64 queries per wave (half the LDS fragment reads and DMA pieces per MFMA)
        dim3 grid64((unsigned)(8 * tv_cdiv(N, 256) * tv_cdiv((long long)heads * B, 8)));
        hipLaunchKernelGGL(attn_fwd64_kernel, grid64, dim3(256), TV_ATTN_NS * 2 * KV_TILE, (hipStream_t)stream, a);
        TV_CHECK_LAUNCH("tv_attn_fwd");
        return TV_OK;
    }
#endif
    dim3 grid((unsigned)(8 * tv_cdiv(N, 128) * tv_cdiv((long long)heads * B, 8)));
    hipLaunchKernelGGL(attn_fwd_kernel, grid, dim3(256), TV_ATTN_NS * 2 * KV_TILE, (hipStream_t)stream, a);
    TV_CHECK_LAUNCH("tv_attn_fwd");
    return TV_OK;
}

extern "C" int tv_attn_bwd(const void* qkv, const void* o, const void* d_o, const float* lse, float* delta, const float* rope_tab,
                           void* dqkv, int B, int N, int heads, float scale, void* stream) {
    if (attn_check("tv_attn_bwd", B, N, heads, scale)) return TV_ERR_ARG;
    TV_CHECK_ARG(qkv && o && d_o && lse && delta && dqkv, "tv_attn_bwd: null pointer");
    TV_CHECK_ARG((long long)N * heads * 64 * 3 * 2 < (1ll << 31), "tv_attn_bwd: one image's qkv must stay below 2 GiB (32-bit DMA offsets)");
    if (tv_init() != TV_OK) return TV_ERR_INIT;
    AttnArgs a{};
    a.qkv = (const bf16*)qkv; a.o = (const bf16*)o; a.d_o = (const bf16*)d_o; a.out = (bf16*)dqkv;
    a.lse = const_cast<float*>(lse); a.delta = delta; a.zeros = (const char*)tv_zero_page();
    a.rope = rope_tab;
    a.B = B; a.N = N; a.heads = heads; a.scale = scale;
    hipStream_t s = (hipStream_t)stream;
    const long long tot = (long long)B * N * heads * 8;
    long long g = (tot + 255) / 256;
    if (g > 4096) g = 4096;
    if (g_attn_bwd_mask & 1) hipLaunchKernelGGL(attn_delta_kernel, dim3((unsigned)g), dim3(256), 0, s, a);
    dim3 grid((unsigned)(8 * tv_cdiv(N, 128) * tv_cdiv((long long)heads * B, 8)));
    if (g_attn_bwd_mask & 2) hipLaunchKernelGGL(attn_bwd_dq_kernel, grid, dim3(256), TV_ATTN_NS * 2 * KV_TILE, s, a);
    if (g_attn_bwd_mask & 4) hipLaunchKernelGGL(attn_bwd_dkv_kernel, grid, dim3(256), TV_ATTN_NS_DKV * DKV_STAGE, s, a);
    TV_CHECK_LAUNCH("tv_attn_bwd");
    return TV_OK;
}
