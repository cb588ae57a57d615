// Weight-gradient ("TN") kernel for gfx950:
//
//     dw[co][tap][ci] += sum_p gy[p][co] * x_gathered[p][tap][ci]          (fp32)
//     dbias[co]       += sum_p gy[p][co]
//
// The reduction runs over output pixels p, the dimension along which neither operand is
// contiguous in NHWC.  Tiles are therefore staged pixel-major ([32 pixels][TW channels], straight
// 16-byte LDS-DMA copies of NHWC rows) and the MFMA fragments, which need 8 consecutive k (=pixels)
// per lane, are pulled out with the hardware transposing read ds_read_b64_tr_b16 (guide T10).
// Bank conflicts of those reads are removed by an XOR swizzle of the 16-byte chunk index that is
// applied on the DMA source address and on the read address.
//
// Work split: grid.x = (co tiles) x (taps) x (ci tiles), grid.y = pixel chunks (split-K); every
// block adds its fp32 tile into dw with global atomics (MI355X_MICROARCH "Global float atomics":
// 64-byte contiguous runs per wave instruction).  The bias gradient rides along as one extra MFMA
// against an all-ones operand in the blocks that own (tap 0, ci tile 0).
#include "common.h"

// the kx-triple kernel for 3x3 / stride-1 layers (wgrad_kx3.hip); wgrad_impl asks it first
int tv_wgrad_kx3_try(const tv_conv_desc* d, const void* x, const void* gy, float* dw, float* dbias, hipStream_t s, bool plan_only, int accum);

#ifndef TV_WGRAD_NO_PIPE
#define TV_WGRAD_NO_PIPE 0
#endif

namespace {

struct WgradArgs {
    const bf16* x;
    const bf16* gy;
    float* dw;
    float* dbias;
    const char* zeros;
    int M;
    int batch, h_in, w_in, c_in, ldx;
    int h_out, w_out, c_out, ldo;
    int kh, kw, stride, pad, up_shift, dil_mask;
    int tiles_ci, chunk_px, hw_shift, w_shift, plain;
    int xcd_order, base, ny;   // block order: see the kernel
    int accum;                 // 1: add into dw / dbias even when one block owns the tile (in-place gradient accumulation)
    unsigned x_bytes;   // FAST path: extent of x in bytes
};

// physical 16-byte slot of logical chunk c in pixel-row r of a [32][TW] tile
template <int TW>
__device__ __forceinline__ int tn_swz(int c, int r) {
    if constexpr (TW % 128 == 0) {
        const int s = (((r >> 3) & 1) << 3) | ((r & 3) << 1);
        return (c & ~15) | ((c & 15) ^ s);
    } else {
        const int s = (((r >> 3) & 1) << 2) | (((r >> 1) & 1) << 1);
        return (c & ~7) | ((c & 7) ^ s);
    }
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// wait until at most `younger` K-steps (NI DMA instructions each) are still in flight
template <int NI, int K>
struct WaitSel {
    static __device__ __forceinline__ void run(int younger) {
        if (younger == K) wait_vmcnt<K * NI>();
        else WaitSel<NI, K - 1>::run(younger);
    }
};
template <int NI>
struct WaitSel<NI, 0> {
    static __device__ __forceinline__ void run(int) { wait_vmcnt<0>(); }
};

// ds_read_b64_tr_b16 through inline asm: with the builtin, hipcc (ROCm 7.2) puts an s_waitcnt vmcnt(0) in
// front of every transposed read while an LDS-DMA is in flight (it does not for plain ds_read_b128), which
// drains the ring each K-step.  An asm read is invisible to that pass; its completion is waited for by the
// explicit lgkmcnt(0) + sched_barrier pairs below (cdna_hip_programming.md section 5.7, rule 18).
__device__ __forceinline__ bf16x4 lds_tr16(unsigned lds_addr) {
    bf16x4 r;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(r) : "v"(lds_addr) : "memory");
    return r;
}
__device__ __forceinline__ void lds_wait_all() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

// BKP = pixels per K-step (32 or 64): 64 halves the barriers / index arithmetic per MFMA
// min waves/SIMD asked of the register allocator: 3 blocks/CU for the 4-wave tiles with <= 16 accumulator
// fragments per wave (one register over the 170 limit costs a third of the latency hiding), 1 block/CU (2 waves/SIMD)
// for the 8-wave tiles
template <int TG, int TX, int NWN>
constexpr int wgrad_min_waves() {
    return NWN == 4 ? 2 : (((TG / 32) * (TX / NWN / 16) <= 16) ? 3 : 1);
}

// FAST: stride-1 "same" convolution (or linear layer) on a power-of-two grid with both tensors < 2 GiB: staging by
// buffer_load...lds with per-thread fixed offsets, the pixel advance in the scalar offset, the tap shift folded into
// the descriptor base and padding / tails as out-of-range offsets (zeros) -- ~1/3 of the generic path's vector ALU work
// register-pipelined main loop where two fragment sets + accumulators + addressing fit the wave's register budget
template <int TG, int TX, int NWN>
constexpr bool wgrad_pipe() {
    const int MF = TG / 2 / 16, NF = TX / NWN / 16;
    const int est = 2 * (MF + NF) * 4 + MF * NF * 4 + MF * 4 + 40;
    const int budget = wgrad_min_waves<TG, TX, NWN>() == 3 ? 150 : (NWN == 4 ? 256 : 512);   // (3 waves/SIMD: 168 registers, minus slack)
    return est <= budget && est <= 216 && !TV_WGRAD_NO_PIPE;
}

template <int TG, int TX, int BKP, int NWN, bool FAST, int STAGES = 2>
__global__ __launch_bounds__(128 * NWN, (wgrad_min_waves<TG, TX, NWN>())) void wgrad_tn_kernel(const WgradArgs p) {
    constexpr int NW = 2 * NWN;                           // waves: 2 over co x NWN over ci
    constexpr int CG = TG / 8, CX = TX / 8;               // chunks per tile row
    constexpr int G_INSTR = BKP * CG / 64, X_INSTR = BKP * CX / 64;   // 1 KiB pieces per K-step tile
    constexpr int G_IT = (G_INSTR + NW - 1) / NW, X_IT = (X_INSTR + NW - 1) / NW;
    constexpr int G_BYTES = BKP * TG * 2, X_BYTES = BKP * TX * 2, STAGE = G_BYTES + X_BYTES;
    constexpr int WTG = TG / 2, WTX = TX / NWN, MF = WTG / 16, NF = WTX / 16;
    static_assert(WTX % 16 == 0, "wave tile must be a multiple of 16");
    static_assert(BKP == 32 || BKP == 64, "K-step is 32 or 64 pixels");

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / NWN, wn = wave % NWN;

    const int taps = p.kh * p.kw;
    // Block order.  All tiles (co, tap, ci) of one pixel chunk read the same gy / x rows; run side by side on ONE XCD they
    // share them through its L2 (they advance at the same pace; a block that runs ahead misses and falls back in step).
    // Workgroups go to XCDs round-robin by linear id, so chunk c takes the ids congruent to c mod 8.  Measured on the
    // dominant 3x3 layer before this: 27.3 GB fetched per launch (FETCH_SIZE) for 3.2 GB of operands, 7.8 TB/s.
    int bx, chunk_id;
    if (p.xcd_order) {
        const int lin = blockIdx.x;
        const int xcd = lin & 7, j = lin >> 3;
        bx = j % p.base;
        chunk_id = (j / p.base) * 8 + xcd;
        if (chunk_id >= p.ny) return;
    } else {
        bx = blockIdx.x;
        chunk_id = blockIdx.y;
    }
    const int ci_tile = bx % p.tiles_ci;
    bx /= p.tiles_ci;
    const int tap = bx % taps;
    const int co_tile = bx / taps;
    const int ky = tap / p.kw, kx = tap - ky * p.kw;
    const int co0 = co_tile * TG, ci0 = ci_tile * TX;

    const int p_begin = chunk_id * p.chunk_px;
    const int p_end = min(p.M, p_begin + p.chunk_px);
    const int nsteps = (p_end - p_begin + BKP - 1) / BKP;
    if (nsteps <= 0) return;

    // ---- staging bookkeeping (fixed per thread: tile row and channel chunk) ------------------
    int g_row[G_IT], g_col[G_IT];
    bool g_cok[G_IT];
#pragma unroll
    for (int it = 0; it < G_IT; ++it) {
        const int id = (it * NW + wave) * 64 + lane;
        const int r = id / CG, s = id - r * CG;
        // physical slot s of row r holds the logical chunk c with tn_swz(c, r) == s (involution)
        const int c = tn_swz<TG>(s, r);
        g_row[it] = r;
        g_col[it] = co0 + c * 8;
        g_cok[it] = g_col[it] < p.c_out;
    }
    int x_row[X_IT], x_col[X_IT];
    bool x_cok[X_IT];
#pragma unroll
    for (int it = 0; it < X_IT; ++it) {
        const int id = (it * NW + wave) * 64 + lane;
        const int r = id / CX, s = id - r * CX;
        const int c = tn_swz<TX>(s, r);
        x_row[it] = r;
        x_col[it] = ci0 + c * 8;
        x_cok[it] = x_col[it] < p.c_in;
    }
    const int hw = p.h_out * p.w_out;
    const int hv = p.h_in << p.up_shift, wv = p.w_in << p.up_shift;

    // FAST-path constants
    int g_voff[G_IT], x_voff[X_IT];
#pragma unroll
    for (int it = 0; it < G_IT; ++it) g_voff[it] = g_cok[it] ? (g_row[it] * p.ldo + g_col[it]) * 2 : OOB_OFFSET;
#pragma unroll
    for (int it = 0; it < X_IT; ++it) x_voff[it] = (x_row[it] * p.ldx + x_col[it]) * 2;
    const int dty = ky - p.pad, dtx = kx - p.pad;
    const long long dtap = ((long long)dty * p.w_in + dtx) * p.ldx;                 // tap shift in elements
    const bf16* xb = p.x + dtap;
    const unsigned xb_bytes = (unsigned)((long long)p.x_bytes - dtap * 2);
    const unsigned g_bytes = (unsigned)((long long)p_end * p.ldo * 2);              // rows >= p_end read as zeros

    auto stage_issue = [&](int step, char* sbase) {
        const int pz = p_begin + step * BKP;
        if constexpr (FAST) {
#pragma unroll
            for (int it = 0; it < G_IT; ++it) {
                if (G_INSTR % NW != 0 && it * NW + wave >= G_INSTR) break;
                buffer_load_lds16(p.gy, g_bytes, sbase + (it * NW + wave) * 1024, g_voff[it], pz * p.ldo * 2);
            }
#pragma unroll
            for (int it = 0; it < X_IT; ++it) {
                if (X_INSTR % NW != 0 && it * NW + wave >= X_INSTR) break;
                const int px = pz + x_row[it];
                const int r = px & (hw - 1);
                const int uy = (r >> p.w_shift) + dty, ux = (r & (p.w_out - 1)) + dtx;
                const bool ok = x_cok[it] && px < p_end && ((unsigned)uy < (unsigned)p.h_in) && ((unsigned)ux < (unsigned)p.w_in);
                buffer_load_lds16(xb, xb_bytes, sbase + G_BYTES + (it * NW + wave) * 1024, ok ? x_voff[it] : OOB_OFFSET, pz * p.ldx * 2);
            }
            return;
        }
#pragma unroll
        for (int it = 0; it < G_IT; ++it) {
            if (G_INSTR % NW != 0 && it * NW + wave >= G_INSTR) break;
            const int px = pz + g_row[it];
            const bool ok = g_cok[it] && px < p_end;
            const void* src = ok ? (const void*)(p.gy + (size_t)px * p.ldo + g_col[it])
                                 : (const void*)(p.zeros + lane * 16);
            __builtin_amdgcn_global_load_lds(TV_GLB(src), TV_LDS(sbase + (it * NW + wave) * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int it = 0; it < X_IT; ++it) {
            if (X_INSTR % NW != 0 && it * NW + wave >= X_INSTR) break;
            const int px = pz + x_row[it];
            bool ok = x_cok[it] && px < p_end;
            int b, oy, ox;
            if (p.w_shift >= 0) {  // power-of-two grid (every real TransVAE stage): shifts, no division
                b = px >> p.hw_shift;
                const int r = px & (hw - 1);
                oy = r >> p.w_shift;
                ox = r & (p.w_out - 1);
            } else {
                b = px / hw;
                const int r = px - b * hw;
                oy = r / p.w_out;
                ox = r - oy * p.w_out;
            }
            const int uy = oy * p.stride + ky - p.pad, ux = ox * p.stride + kx - p.pad;
            ok = ok && ((unsigned)uy < (unsigned)hv) && ((unsigned)ux < (unsigned)wv) &&
                 (((uy | ux) & p.dil_mask) == 0);
            const int pix = (b * p.h_in + (uy >> p.up_shift)) * p.w_in + (ux >> p.up_shift);
            const void* src = ok ? (const void*)(p.x + (size_t)pix * p.ldx + x_col[it])
                                 : (const void*)(p.zeros + lane * 16);
            __builtin_amdgcn_global_load_lds(TV_GLB(src), TV_LDS(sbase + G_BYTES + (it * NW + wave) * 1024), 16, 0, 0);
        }
    };

    // ---- fragment addressing ---------------------------------------------------------------
    // lane (g = lane>>4, q = (lane>>2)&3, pp = lane&3) supplies row 8g+4h+q, columns base+4pp..+3
    const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    int a_off[2][MF], b_off[2][NF];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int r = 8 * g + 4 * h + q;
#pragma unroll
        for (int i = 0; i < MF; ++i) {
            const int col = wm * WTG + i * 16 + 4 * pp;  // element column inside the G tile
            a_off[h][i] = r * (TG * 2) + tn_swz<TG>(col >> 3, r) * 16 + (pp & 1) * 8;
        }
#pragma unroll
        for (int j = 0; j < NF; ++j) {
            const int col = wn * WTX + j * 16 + 4 * pp;
            b_off[h][j] = G_BYTES + r * (TX * 2) + tn_swz<TX>(col >> 3, r) * 16 + (pp & 1) * 8;
        }
    }

    f32x4 acc[MF][NF];
#pragma unroll
    for (int i = 0; i < MF; ++i) {
#pragma unroll
        for (int j = 0; j < NF; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // bias gradient: one extra MFMA per co fragment against an all-ones operand.  Every wave of a row (and every tap's
    // block) holds the same gy fragments, so the fragments are dealt out: loading ONE wave of the tap-0 block with all MF
    // of them (+33 % MFMAs on the 192x192 tile) made it, and with it the block and the launch, wait for that wave.
    // Fragment f = wm*MF + i of this co tile is summed by the wave in slot f mod (taps*NWN), slot = tap + taps*wn, of the
    // blocks with ci tile 0: every tap's block carries its share (at most one extra MFMA per wave on the 3x3 layers).
    // (A single-branch form -- one owned fragment per wave, selected outside the fragment loop -- measured slower.)
    // (round 2: the extra MFMA per owned fragment sat behind a scalar branch per fragment in every wave's MFMA stream; the
    //  owners now add the gy fragments up with vector ALU work -- lane L of an A fragment holds 8 pixels of output channel
    //  L & 15 -- behind ONE wave-uniform branch per sub-step.  Owners: the blocks of (tap 0, ci tile 0); fragment i of a
    //  wave row belongs to the wave with wn == i % NWN.)
    // (round 3: with ONE owner block per co tile that block carried the sums in every K-step and ran ~25 % longer than the
    //  others -- and with one block per CU the launch with it: 2.75 -> 3.39 ms on the 192-channel 3x3 layers.  The taps x
    //  tiles_ci blocks of a pixel chunk that share a co tile hold the same gy fragments, so they take the K-steps in turn.
    //  Every block therefore ADDS its share into dbias with atomics, single-chunk launches included: the caller passes a
    //  zeroed dbias, or one that holds a running sum.)
#ifdef TV_BIAS_ONE_OWNER      // (A/B build: the round-2 scheme)
    const bool one_owner = true;
#else
    const bool one_owner = false;
#endif
    const bool do_bias = (p.dbias != nullptr) && (!one_owner || (ci_tile == 0 && tap == 0));
    const int nshare = one_owner ? 1 : taps * p.tiles_ci;
    int bias_ctr = one_owner ? 0 : tap * p.tiles_ci + ci_tile;     // K-steps until this block's next turn
    bool bias_now = false;
    auto bias_turn = [&]() {     // call once per K-step, before its fragments are summed
        bias_now = do_bias && bias_ctr == 0;
        bias_ctr = bias_ctr == 0 ? nshare - 1 : bias_ctr - 1;
    };
    float bsum[(MF + NWN - 1) / NWN];
#pragma unroll
    for (int k = 0; k < (MF + NWN - 1) / NWN; ++k) bsum[k] = 0.f;
    auto bias_add = [&](const bf16x4 (&alo)[MF], const bf16x4 (&ahi)[MF]) {
        if (!bias_now) return;
#pragma unroll
        for (int i = 0; i < MF; ++i)
            if (i % NWN == wn) {
                float a = 0.f;
#pragma unroll
                for (int e = 0; e < 4; ++e) a += (float)alo[i][e] + (float)ahi[i][e];
                bsum[i / NWN] += a;
            }
    };

    const unsigned smem_addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    auto join = [](bf16x4 lo, bf16x4 hi) { return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]}; };
    auto compute = [&](int stage) {
        constexpr int H = (NF + 1) / 2;
#pragma unroll
        for (int kk = 0; kk < BKP / 32; ++kk) {
            const unsigned sa = smem_addr + stage * STAGE + kk * 32 * (TG * 2);
            const unsigned sb = smem_addr + stage * STAGE + kk * 32 * (TX * 2);
            bf16x4 alo[MF], ahi[MF], blo[NF], bhi[NF];
#pragma unroll
            for (int i = 0; i < MF; ++i) {
                alo[i] = lds_tr16(sa + a_off[0][i]);
                ahi[i] = lds_tr16(sa + a_off[1][i]);
            }
#pragma unroll
            for (int j = 0; j < H; ++j) {
                blo[j] = lds_tr16(sb + b_off[0][j]);
                bhi[j] = lds_tr16(sb + b_off[1][j]);
            }
            lds_wait_all();
#pragma unroll
            for (int j = H; j < NF; ++j) {  // second batch lands while the first half multiplies
                blo[j] = lds_tr16(sb + b_off[0][j]);
                bhi[j] = lds_tr16(sb + b_off[1][j]);
            }
            bf16x8 af[MF];
#pragma unroll
            for (int i = 0; i < MF; ++i) af[i] = join(alo[i], ahi[i]);
#pragma unroll
            for (int i = 0; i < MF; ++i)
#pragma unroll
                for (int j = 0; j < H; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], join(blo[j], bhi[j]), acc[i][j], 0, 0, 0);
            bias_add(alo, ahi);
            lds_wait_all();
#pragma unroll
            for (int i = 0; i < MF; ++i)
#pragma unroll
                for (int j = H; j < NF; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], join(blo[j], bhi[j]), acc[i][j], 0, 0, 0);
        }
    };

    // Register-pipelined loop (same idea as igemm_nt.hip, PIPE2): two 32-pixel fragment sets per wave.  While the MFMAs of
    // sub-step u run, the transposing reads of sub-step u+1 are in flight -- across the stage boundary too, so the block
    // barrier sits where every wave has read all of a stage: the stage after it has landed, and the one just left is
    // refilled.  (The reads are inline asm, invisible to the compiler's waitcnt pass: each set is waited for with an
    // explicit lgkmcnt(0) BEFORE the next set is issued, when it has had a whole sub-step to land.)
    constexpr int KH = BKP / 32;
    constexpr bool PIPE = wgrad_pipe<TG, TX, NWN>();
    static_assert(STAGES == 2 || PIPE, "the 3-deep ring exists for the pipelined loop only");
    if constexpr (PIPE) {
        bf16x4 f0alo[MF], f0ahi[MF], f0blo[NF], f0bhi[NF], f1alo[MF], f1ahi[MF], f1blo[NF], f1bhi[NF];
        auto read_sub = [&](int u, bf16x4 (&alo)[MF], bf16x4 (&ahi)[MF], bf16x4 (&blo)[NF], bf16x4 (&bhi)[NF]) {
            const int stage = (u / KH) % STAGES, kk = u % KH;
            const unsigned sa = smem_addr + stage * STAGE + kk * 32 * (TG * 2);
            const unsigned sb = smem_addr + stage * STAGE + kk * 32 * (TX * 2);
#ifdef TV_ABL_WG_NOLDS   // (ablation builds, tools/probes/build_variant.sh: the main loop without its transposing reads)
#pragma unroll
            for (int i = 0; i < MF; ++i) { alo[i] = bf16x4{1, 1, 1, 1}; ahi[i] = bf16x4{1, 1, 1, 1}; asm volatile("" : "+v"(alo[i]), "+v"(ahi[i])); }
#pragma unroll
            for (int j = 0; j < NF; ++j) { blo[j] = bf16x4{1, 1, 1, 1}; bhi[j] = bf16x4{1, 1, 1, 1}; asm volatile("" : "+v"(blo[j]), "+v"(bhi[j])); }
            (void)sa; (void)sb;
#else
#pragma unroll
            for (int i = 0; i < MF; ++i) {
                alo[i] = lds_tr16(sa + a_off[0][i]);
                ahi[i] = lds_tr16(sa + a_off[1][i]);
            }
#pragma unroll
            for (int j = 0; j < NF; ++j) {
                blo[j] = lds_tr16(sb + b_off[0][j]);
                bhi[j] = lds_tr16(sb + b_off[1][j]);
            }
#endif
        };
        auto mfma_sub = [&](const bf16x4 (&alo)[MF], const bf16x4 (&ahi)[MF], const bf16x4 (&blo)[NF], const bf16x4 (&bhi)[NF]) {
#pragma unroll
            for (int i = 0; i < MF; ++i) {
                const bf16x8 af = join(alo[i], ahi[i]);
#pragma unroll
                for (int j = 0; j < NF; ++j) {
#ifdef TV_ABL_WG_NOMFMA
                    const bf16x8 bfv = join(blo[j], bhi[j]);
                    asm volatile("" ::"v"(af), "v"(bfv));
#else
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, join(blo[j], bhi[j]), acc[i][j], 0, 0, 0);
#endif
                }
            }
            bias_add(alo, ahi);
        };
        const int U = nsteps * KH;
        // sub-step u: wait for its fragments, start the reads of u+1 (after the stage hand-over if u+1 opens a stage), multiply
        auto sub = [&](int u, auto& calo, auto& cahi, auto& cblo, auto& cbhi, auto& nalo, auto& nahi, auto& nblo, auto& nbhi) {
            lds_wait_all();
            if (u + 1 < U) {
                if ((u + 1) % KH == 0) {
                    const int s1 = (u + 1) / KH;           // stage about to be read; stage s1-1 is now free
                    // stage s1 landed; with a 3-deep ring stage s1+1 may stay in flight (parked-wave counters showed the
                    // 2-deep ring waiting on the DMA for ~half of every wave's lifetime)
                    if (STAGES == 3 && s1 + 1 < nsteps) wait_vmcnt<G_IT + X_IT>();
                    else wait_vmcnt<0>();
#ifndef TV_ABL_WG_NOBAR
                    __builtin_amdgcn_s_barrier();
#endif
#ifndef TV_ABL_WG_NODMA
                    // (as one burst: threading the pieces between the MFMAs that follow, as the igemm kernels do, measured
                    //  0.96x here -- tools/gemm_sweep.py, same box)
                    if (s1 + STAGES - 1 < nsteps) stage_issue(s1 + STAGES - 1, smem + ((s1 + STAGES - 1) % STAGES) * STAGE);
#endif
                }
                read_sub(u + 1, nalo, nahi, nblo, nbhi);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (u % KH == 0) bias_turn();
            mfma_sub(calo, cahi, cblo, cbhi);
            __builtin_amdgcn_sched_barrier(0);
        };
        static_assert(STAGES == 2 || (G_INSTR % NW == 0 && X_INSTR % NW == 0), "counted waits need the same DMA count in every wave");
#pragma unroll
        for (int st = 0; st < STAGES; ++st)
            if (st < nsteps) stage_issue(st, smem + st * STAGE);
        constexpr bool UNIFORM = G_INSTR % NW == 0 && X_INSTR % NW == 0;   // else: waves with fewer pieces must not under-wait
        if (UNIFORM && STAGES == 3 && nsteps >= 3) wait_vmcnt<2 * (G_IT + X_IT)>();
        else if (UNIFORM && nsteps >= 2) wait_vmcnt<G_IT + X_IT>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        read_sub(0, f0alo, f0ahi, f0blo, f0bhi);
        int u = 0;
        for (; u + 1 < U; u += 2) {
            sub(u, f0alo, f0ahi, f0blo, f0bhi, f1alo, f1ahi, f1blo, f1bhi);
            sub(u + 1, f1alo, f1ahi, f1blo, f1bhi, f0alo, f0ahi, f0blo, f0bhi);
        }
        if (u < U) sub(u, f0alo, f0ahi, f0blo, f0bhi, f1alo, f1ahi, f1blo, f1bhi);
    } else {
        // double buffer: the DMA of K-step t+1 is in flight while step t is multiplied
        stage_issue(0, smem);
        for (int t = 0; t < nsteps; ++t) {
            wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            if (t + 1 < nsteps) stage_issue(t + 1, smem + ((t + 1) & 1) * STAGE);
            bias_turn();
            compute(t & 1);
        }
    }

    // ---- fp32 atomics into dw[co][tap][ci]; D layout: row = (lane>>4)*4+reg (co), col = lane&15 (ci)
    const size_t ldw = (size_t)taps * p.c_in;
#pragma unroll
    for (int i = 0; i < MF; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int co = co0 + wm * WTG + i * 16 + g * 4 + r;
            if (co >= p.c_out) continue;
            float* rowp = p.dw + (size_t)co * ldw + (size_t)tap * p.c_in;
#pragma unroll
            for (int j = 0; j < NF; ++j) {
                const int ci = ci0 + wn * WTX + j * 16 + (lane & 15);
                if (ci < p.c_in) {
                    if (p.plain && !p.accum) rowp[ci] = acc[i][j][r];   // one pixel chunk: this block owns the tile
                    else if (p.plain) rowp[ci] += acc[i][j][r];         // ... and adds to what an earlier micro-batch left there
                    else atomicAdd(rowp + ci, acc[i][j][r]);
                }
            }
        }
    }
    if (do_bias) {   // lanes L, L+16, L+32, L+48 hold partial sums of the same output channel
#pragma unroll
        for (int i = 0; i < MF; ++i)
            if (i % NWN == wn) {
                float v = bsum[i / NWN];
                v += __shfl_xor(v, 16, 64);
                v += __shfl_xor(v, 32, 64);
                const int co = co0 + wm * WTG + i * 16 + (lane & 15);
                if (lane < 16 && co < p.c_out) atomicAdd(p.dbias + co, v);
            }
    }
}

// (The kx-triple kernel for 3x3 / stride-1 layers lives in wgrad_kx3.hip; wgrad_impl asks it first.)
extern int g_wgrad_xcd, g_wgrad_blocks, g_wgrad_generic;

int g_wgrad_bkp = 0;     // 0 = heuristic (64), else 32 / 64 pixels per K-step
int g_wgrad_blocks = 0;  // 0 = heuristic: target number of blocks for the split-K choice
int g_wgrad_tile256 = 0; // 0 = heuristic (linear layers only), 1 = always when both channel counts are multiples of 256, 2 = never
int g_wgrad_prefer192 = 0;   // experiment: 192-wide tiles wherever the channel count divides
int g_wgrad_xcd = 1;     // 1 = tiles of one pixel chunk share an XCD (block order in the kernel)
int g_wgrad_generic = 0; // 1 = never use the FAST staging path (A/B)
int g_wgrad_waves = 0;   // 0 = heuristic (8 waves for the 192-wide co tile), 4 / 8 = force

template <int TG, int TX, int BKP, int NWN, bool FAST, int STAGES>
int launch_st(const WgradArgs& a, dim3 grid, hipStream_t s) {
    constexpr int BYTES = STAGES * BKP * (TG + TX) * 2;
    constexpr int NW = 2 * NWN;
    constexpr bool uniform = (BKP * (TG / 8) / 64) % NW == 0 && (BKP * (TX / 8) / 64) % NW == 0;
    if constexpr (BYTES > 160 * 1024 || (TX / NWN) % 16 != 0 || (STAGES == 3 && (!uniform || NWN != 4 || !wgrad_pipe<TG, TX, NWN>()))) {
        return -1;
    } else {
        static TvPerDeviceOnce attr_once;
        if (attr_once.first()) {
            (void)hipFuncSetAttribute((const void*)wgrad_tn_kernel<TG, TX, BKP, NWN, FAST, STAGES>, hipFuncAttributeMaxDynamicSharedMemorySize, BYTES);
        }
        hipLaunchKernelGGL((wgrad_tn_kernel<TG, TX, BKP, NWN, FAST, STAGES>), grid, dim3(128 * NWN), BYTES, s, a);
        return 0;
    }
}

int g_wgrad_stages = 0;  // 0 / 2 = two-stage ring, 3 = three stages where the tile leaves the LDS room (A/B)

template <int TG, int TX, int BKP, int NWN, bool FAST>
int launch_f(const WgradArgs& a, dim3 grid, hipStream_t s) {
    if (g_wgrad_stages == 3 && launch_st<TG, TX, BKP, NWN, FAST, 3>(a, grid, s) == 0) return 0;   // (measured 7 % slower: A/B only)
    return launch_st<TG, TX, BKP, NWN, FAST, 2>(a, grid, s);
}

template <int TG, int TX, int BKP, int NWN>
int launch_s(const WgradArgs& a, dim3 grid, hipStream_t s) {
    return (a.x_bytes != 0 && !g_wgrad_generic) ? launch_f<TG, TX, BKP, NWN, true>(a, grid, s) : launch_f<TG, TX, BKP, NWN, false>(a, grid, s);
}

template <int TG, int TX>
int launch(const WgradArgs& a0, hipStream_t s, bool plan_only, double* t_model = nullptr) {
    WgradArgs a = a0;
    const int bkp = g_wgrad_bkp ? g_wgrad_bkp : (TG == 192 ? 64 : 32);   // measured: 64 pays only where 8 waves share the tile
    const int tiles_co = (a.c_out + TG - 1) / TG;
    a.tiles_ci = (a.c_in + TX - 1) / TX;
    const long long base = (long long)tiles_co * a.tiles_ci * a.kh * a.kw;
    // split-K over pixel chunks.  Every partition re-adds the whole [Cout, taps*Cin] gradient with fp32 atomics
    // (1.3 TB/s chip wide, MI355X_MICROARCH "Global float atomics"), and blocks run in rounds of `slots` resident
    // blocks, so pick the split that minimises   rounds * (pixels per block) * t_pixel  +  atomic bytes / 1.3 TB/s.
    const bool w8 = g_wgrad_waves != 4 && (TG >= 192 || g_wgrad_waves == 8) && TX >= 128;
    const int per_cu = w8 ? 1 : (((TG / 32) * (TX / 32) <= 16) ? 3 : 2);
    const double slots = 256.0 * per_cu;
    // sustained rate of the tile shape (calibrated on the linear layers: 256x256 ~10 % over the pipelined 192-wide tiles,
    // those ~13 % over 128x128) -- only the ratios matter, for comparing candidate tiles
    const double rate = (TG == 256 && TX == 256) ? 990e12 : ((TG >= 192 && TX >= 192) ? 900e12 : ((TG >= 192 || TX >= 192) ? 850e12 : 800e12));
    const double t_px = 2.0 * TG * TX * slots / rate;                  // seconds per pixel for one resident block
    const double tile_bytes = 4.0 * TG * TX;
    long long split = 1;
    bool xcd = false;
    double best = 1e30;
    if (g_wgrad_blocks) {
        split = (g_wgrad_blocks + base - 1) / base;
        xcd = g_wgrad_xcd != 0;
    } else {
        const long long smax = a.M / 512 > 0 ? a.M / 512 : 1;
        for (long long sp = 1; sp <= smax && sp <= 4096; ++sp) {
            const double rounds = (double)((long long)((base * sp + slots - 1) / slots));
            const double t = rounds * ((double)a.M / sp) * t_px + (sp > 1 ? base * sp * tile_bytes / 1.3e12 : 0.0);
            if (t < best) { best = t; split = sp; xcd = false; }
        }
        // XCD-grouped order (every tile of a pixel chunk re-reads the same rows; one chunk per XCD at a time shares them
        // through that L2: measured +35 % per block on the 192-channel 3x3 layers, +15-45 % on the 384-channel linear ones): chunk c runs on XCD c % 8, so the load
        // is per XCD: base * ceil(split / 8) blocks on 32 * per_cu slots.
        const double xslots = 32.0 * per_cu;
        if (g_wgrad_xcd && base <= xslots) {   // (a chunk's tiles must be co-resident on the XCD; else measured slower)
            for (long long sp = 8; sp <= smax && sp <= 4096; sp += 8) {
                const double rounds = (double)((long long)((base * (sp / 8) + xslots - 1) / xslots));
                const double t = rounds * ((double)a.M / sp) * (t_px * 0.75) + base * sp * tile_bytes / 1.3e12;
                if (t < best) { best = t; split = sp; xcd = true; }
            }
        }
    }
    long long chunk = (a.M + split - 1) / split;
    if (chunk < 512) chunk = 512;
    chunk = (chunk + bkp - 1) / bkp * bkp;
    a.chunk_px = (int)chunk;
    const int ny = (int)((a.M + chunk - 1) / chunk);
    a.plain = (ny == 1) ? 1 : 0;
    if (t_model) *t_model = best;
    if (plan_only) return 100 + a.plain;   // (distinct from the TV_ERR_* codes)
    a.base = (int)base;
    a.ny = ny;
    a.xcd_order = (ny > 1 && xcd) ? 1 : 0;
    dim3 grid((unsigned)base, (unsigned)ny);
    if (a.xcd_order) grid = dim3((unsigned)(8 * base * ((ny + 7) / 8)), 1);
    auto log2_exact = [](int v) { int s = 0; while ((1 << s) < v) ++s; return ((1 << s) == v) ? s : -1; };
    a.w_shift = log2_exact(a.w_out);
    a.hw_shift = log2_exact(a.h_out * a.w_out);
    if (a.hw_shift < 0) a.w_shift = -1;
    {   // FAST staging path: stride-1 "same" conv / linear, power-of-two grid, 32-bit offsets
        const long long xbytes = (long long)a.batch * a.h_in * a.w_in * a.ldx * 2;
        const long long gbytes = (long long)a.M * a.ldo * 2;
        const long long shift = ((long long)a.pad * a.w_in + a.pad) * a.ldx * 2;
        const bool same = a.stride == 1 && a.up_shift == 0 && a.dil_mask == 0 && a.kh == a.kw && (a.kh == 1 || a.kh == 3) &&
                          a.pad == a.kh / 2 && a.h_in == a.h_out && a.w_in == a.w_out && a.w_shift >= 0;
        a.x_bytes = (same && xbytes + shift < (1ll << 31) && gbytes < (1ll << 31)) ? (unsigned)xbytes : 0u;
    }
    // (w8: 8 waves, 2 per SIMD, for the 192-wide co tile: one wave's transposed-read phase overlaps the other's MFMAs)
    if (bkp == 64) {
        if (w8 && launch_s<TG, TX, 64, 4>(a, grid, s) == 0) return 0;
        if (launch_s<TG, TX, 64, 2>(a, grid, s) == 0) return 0;
    }
    if (w8 && launch_s<TG, TX, 32, 4>(a, grid, s) == 0) return 0;
    return launch_s<TG, TX, 32, 2>(a, grid, s);
}

}  // namespace

extern "C" int tv_set_wgrad_stages(int stages) {   // 0 = heuristic, 2 / 3; +10: plain (chunk-major) block order
    g_wgrad_xcd = stages >= 10 ? 0 : 1;
    g_wgrad_stages = stages % 10;
    return 0;
}

extern "C" int tv_set_wgrad_config(int bkp, int waves, int blocks) {
    g_wgrad_generic = (blocks == -1) ? 1 : 0;
    g_wgrad_tile256 = (blocks == -2) ? 1 : ((blocks == -3 || blocks == -4) ? 2 : 0);
    g_wgrad_prefer192 = (blocks == -4) ? 1 : 0;
    if (blocks < 0) blocks = 0;
    g_wgrad_bkp = bkp;
    g_wgrad_waves = waves;
    g_wgrad_blocks = blocks;
    return 0;
}

// plan_only: no launch, returns 1 if the call would overwrite dw / dbias (single pixel chunk), 0 if it accumulates
static int wgrad_impl(const tv_conv_desc* d, const void* x, const void* gy, float* dw, float* dbias, void* stream, bool plan_only, int accum = 0) {
    TV_CHECK_ARG(d && (plan_only || (x && gy && dw)), "tv_wgrad_tn: null pointer");
    TV_CHECK_ARG(d->c_in > 0 && d->c_in % 8 == 0, "tv_wgrad_tn: c_in=%d must be a multiple of 8", d->c_in);
    TV_CHECK_ARG(d->c_out > 0 && d->c_out % 8 == 0, "tv_wgrad_tn: c_out=%d must be a multiple of 8", d->c_out);
    TV_CHECK_ARG(d->ldx >= d->c_in && d->ldx % 8 == 0 && d->ldo >= d->c_out && d->ldo % 8 == 0,
                 "tv_wgrad_tn: ldx/ldo must cover the channels and be multiples of 8");
    TV_CHECK_ARG(d->store_shuffle == 0, "tv_wgrad_tn: call with the transposed problem for shuffle-store layers");
    TV_CHECK_ARG(d->batch > 0 && d->h_in > 0 && d->w_in > 0 && d->h_out > 0 && d->w_out > 0, "tv_wgrad_tn: empty geometry");
    TV_CHECK_ARG((d->up_shift | 1) == 1 && (d->dil_mask | 1) == 1, "tv_wgrad_tn: up_shift/dil_mask must be 0 or 1");
    const long long M = (long long)d->batch * d->h_out * d->w_out;
    TV_CHECK_ARG(M < (1ll << 31) && (long long)d->batch * d->h_in * d->w_in < (1ll << 31), "tv_wgrad_tn: too many pixels");
    if (!plan_only && tv_init() != TV_OK) return TV_ERR_INIT;

    WgradArgs a;
    a.x = (const bf16*)x;
    a.gy = (const bf16*)gy;
    a.dw = dw;
    a.dbias = dbias;
    a.zeros = (const char*)tv_zero_page();
    a.M = (int)M;
    a.batch = d->batch; a.h_in = d->h_in; a.w_in = d->w_in; a.c_in = d->c_in; a.ldx = d->ldx;
    a.h_out = d->h_out; a.w_out = d->w_out; a.c_out = d->c_out; a.ldo = d->ldo;
    a.kh = d->kh; a.kw = d->kw; a.stride = d->stride; a.pad = d->pad;
    a.up_shift = d->up_shift; a.dil_mask = d->dil_mask;
    a.tiles_ci = 1; a.chunk_px = 0; a.hw_shift = -1; a.w_shift = -1; a.plain = 0; a.x_bytes = 0;
    a.xcd_order = 0; a.base = 1; a.ny = 1;
    a.accum = accum;
    hipStream_t s = (hipStream_t)stream;
    if (!g_wgrad_generic && !g_wgrad_blocks) {   // 3x3 / stride-1 / pad-1 on image rows of >= 64 pixels: the kx-triple kernel (wgrad_kx3.hip)
        const int r = tv_wgrad_kx3_try(d, x, gy, dw, dbias, s, plan_only, accum);
        if (r >= 0) {
            if (plan_only) return r;
            TV_CHECK_LAUNCH("tv_wgrad_tn (kx3)");
            return TV_OK;
        }
    }
    // 256x256 tiles (8 waves, 128x64 per wave): +5..20 % on linear layers, -9 % on 9-tap convolutions (gemm_sweep, mb 64)
    // single-tap layers: 256x256 or 192x192 tiles by the modelled time of their best split (wave quantisation decides:
    // 1536 -> 6144 at 16 K pixels is exactly 256 tiles of 192x192 with plain stores, while qkv widths fill better at 256)
    bool use256 = (g_wgrad_tile256 == 1 || (g_wgrad_tile256 == 0 && d->kh * d->kw == 1)) && d->c_out % 256 == 0 && d->c_in % 256 == 0;
    bool both192 = d->kh * d->kw == 1 && d->c_out % 192 == 0 && d->c_in % 192 == 0 && g_wgrad_tile256 == 0 && !g_wgrad_blocks;
    if (both192) {
        double t192 = 1e30, t_alt = 1e30;
        launch<192, 192>(a, s, true, &t192);
        if (use256) launch<256, 256>(a, s, true, &t_alt);
        else launch<128, 128>(a, s, true, &t_alt);
        if (t192 < t_alt) use256 = false;
        else both192 = false;
    }
    if (use256) {
        const int r = launch<256, 256>(a, s, plan_only);
        if (plan_only) return r;
        if (r != 0) { tv_set_error("tv_wgrad_tn: no kernel for this configuration"); return TV_ERR_ARG; }
        TV_CHECK_LAUNCH("tv_wgrad_tn");
        return TV_OK;
    }
    const bool g192 = (d->c_out % 192 == 0) && (d->c_out % 128 != 0 || g_wgrad_prefer192 || both192);
    const bool x192 = (d->c_in % 192 == 0) && (d->c_in % 128 != 0 || g_wgrad_prefer192 || both192);
    const bool g64 = d->c_out <= 64, x64 = d->c_in <= 64;
    int r;
    if (g192 && x192) r = launch<192, 192>(a, s, plan_only);
    else if (g192) { if (x64) r = launch<192, 64>(a, s, plan_only); else r = launch<192, 128>(a, s, plan_only); }
    else if (x192) { if (g64) r = launch<64, 192>(a, s, plan_only); else r = launch<128, 192>(a, s, plan_only); }
    else if (g64 && x64) r = launch<64, 64>(a, s, plan_only);
    else if (g64) r = launch<64, 128>(a, s, plan_only);
    else if (x64) r = launch<128, 64>(a, s, plan_only);
    else r = launch<128, 128>(a, s, plan_only);
    if (plan_only) return r;
    if (r != 0) { tv_set_error("tv_wgrad_tn: no kernel for this configuration"); return TV_ERR_ARG; }
    TV_CHECK_LAUNCH("tv_wgrad_tn");
    return TV_OK;
}

extern "C" int tv_wgrad_tn(const tv_conv_desc* d, const void* x, const void* gy, float* dw, float* dbias,
                           void* stream) {
    return wgrad_impl(d, x, gy, dw, dbias, stream, false);
}

extern "C" int tv_wgrad_tn_acc(const tv_conv_desc* d, const void* x, const void* gy, float* dw, float* dbias, void* stream) {
    return wgrad_impl(d, x, gy, dw, dbias, stream, false, 1);
}

extern "C" int tv_wgrad_tn_overwrites(const tv_conv_desc* d) {
    const int r = wgrad_impl(d, nullptr, nullptr, nullptr, nullptr, nullptr, true);
    return r >= 100 ? r - 100 : -1;
}
