// Weight gradient of 3x3 / stride-1 / pad-1 convolutions on image rows of >= 16 pixels ("kx-triple", round 3), and -- as the
// one-tap instantiation of the same loop (TAPS = 1) -- of the linear / 1x1 layers:
//
//     dw[co][ky][kx][ci] += sum_p gy[p][co] * x[p + (ky-1) w + (kx-1)][ci]          dbias[co] += sum_p gy[p][co]
//
// (R/transvae/modules/blocks.py:34,37: autograd of the two ResBlock convolutions; R/transvae/modules/conv.py:58: the 3x3
// of the Conv-FFN; R/transvae/modules/upsample.py:33,97: the stride-1 convolutions of Down/Upsample.)
//
// Why a second kernel.  The single-tap kernel (wgrad_tn.hip) stages gy and x once PER TAP: 48 KiB of LDS-DMA per 4.7 MFLOP
// K-step on its 192x192 tile = 96 FLOP per staged byte.  A CU sustains ~40 GB/s of LDS-DMA next to its matrix work
// (profiles/r02_gemm_sweep_mb64.txt: 999 TFLOP/s = 40 GB/s x 96 FLOP/B x 256 CUs), so that kernel runs AT its fill rate.
// Here a block owns (co tile, ky, ci tile) for all THREE kx: a K-step stages 64 output pixels of gy (one piece of an image
// row) and the matching 64 + 2 input pixels of x ONCE; tap kx reads its x fragments from the same LDS image shifted by kx
// rows.  With a 192 (co) x 96 (ci) x 3 (kx) tile that is 37 KiB per 7.1 MFLOP = 187 FLOP per staged byte, and each wave
// (48 co x 48 ci x 3 kx) issues 54 MFMAs per 48 transposing reads instead of 36 per 36.
//
// Structure: 8 waves, wave-group ping-pong (the main loop of conv3x3_halo_kernel): waves w and w + 4 share a SIMD; group 0
// reads the fragments of K-step t and issues its LDS-DMA pieces while group 1 multiplies step t - 1, then they swap: one
// block barrier per phase.  LDS: a ring of RING stages (gy slots first, x slots after), K-step t + RING - 1 is issued during
// step t.  Every fragment read is `base register + immediate`: the ring slot, the 32-pixel half of the step and the
// 16-pixel half of the fragment are compile-time offsets (the step loop is unrolled RING times), so the load phase has no
// address arithmetic.
//
// x image of a K-step (64 output pixels = 64 / SEG pieces of image rows, SEG = min(image width, 64) in {16, 32, 64}): each
// segment takes SS = SEG + 2 rounded up to a multiple of 8 tile rows -- row 0 = left neighbour of the segment (zeros at the
// image's left edge: out-of-range DMA offset = the convolution's zero padding), rows 1..SEG = its pixels, row SEG + 1 =
// right neighbour; padding in y = the segment's x rows out of range.  Tiles are pixel-major ([row][channels], 16-byte
// chunks XOR-swizzled on the DMA source address and on the read address), fragments come out through ds_read_b64_tr_b16.
//
// Which pixel sits in which k slot of the MFMA is free as long as both operands agree: lane group g (= lane >> 4), half h,
// element q of a 32-pixel half-step hold pixel 16 h + 4 g + q.  A 32-lane half of a transposing read then covers 8
// CONSECUTIVE tile rows (at any first row: the x image is read at row offsets 0 / 1 / 2), the swizzle is keyed on row & 7,
// and because every segment stride is a multiple of 8 rows the offsets of h (16 pixels) and of the second half-step
// (32 pixels) are the same constant for every lane, whatever the segment length.
#include "common.h"

#include <type_traits>

namespace {

struct Kx3Args {
    const bf16* x;
    const bf16* gy;
    float* dw;
    float* dbias;
    int M, h, w, c_in, ldx, c_out, ldo;
    int tiles_ci, chunk_px, hw_shift, w_shift, plain;
    int xcd_order, base, ny, accum;
    unsigned x_bytes;
};

// physical 16-byte slot of logical chunk c in row r of a [rows][TW channels] tile: 8 consecutive rows (any first row) put
// their 32-byte pieces of one channel group on 8 different bank groups -> conflict-free transposing reads
template <int TW>
__device__ __forceinline__ int kx_swz(int c, int r) {
    if constexpr (TW % 128 == 0) {          // rows are whole bank rows: XOR the 32-byte granule index with r & 7
        return (c & ~15) | ((c & 15) ^ ((r & 7) << 1));
    } else if constexpr (TW % 64 == 0) {    // 64 / 192: consecutive rows are 4 granules apart: XOR the low two bits with (r >> 1) & 3
        return (c & ~7) | ((c & 7) ^ (((r >> 1) & 3) << 1));
    } else {                                // 96: rows are 6 granules apart (0, 6, 4, 2, 0, ...): rows 4 apart collide -> XOR bit 0
        static_assert(TW % 32 == 0, "tile width");
        return (c & ~3) | ((c & 3) ^ (((r >> 2) & 1) << 1));
    }
}

template <int OFF>
__device__ __forceinline__ bf16x4 lds_tr16(unsigned addr) {
    static_assert(OFF >= 0 && OFF < 65536, "ds offset field is 16 bits");
    bf16x4 r;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF) : "memory");
    return r;
}
template <int N>
__device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

constexpr int kx_seg_stride(int seg) { return (seg + 2 + 7) / 8 * 8; }                 // tile rows per segment: 24 / 40 / 72
constexpr int kx_x_pieces(int seg, int cx) { return ((64 / seg) * kx_seg_stride(seg) * cx + 63) / 64; }   // 1 KiB pieces of the x tile
constexpr int kx_stage_bytes(int tg, int tx, int seg) { return 64 * tg * 2 + kx_x_pieces(seg, tx / 8) * 1024; }

template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// Schedule variants (VAR), all bit-identical in their results:
//   0  every fragment read and every LDS-DMA piece in the load phase, K-step t + RING - 1 issued during step t
//   1  the reads of the second 32-pixel half threaded between the MFMAs of the first, all DMA pieces in the MFMA phase
//   2  ... reads threaded, lookahead RING - 2, gy pieces in the load phase, x pieces in the MFMA phase
//   3  ... reads threaded, lookahead RING - 2, all pieces in the load phase
//   4  all reads in the load phase, gy pieces in the load phase, x pieces in the MFMA phase
// Hazard rule behind the lookahead: a wave's MFMA phase t runs beside the other group's load phase t (group 0) or t + 1
// (group 1).  Reads threaded into the MFMA phase make a group read K-step t while the other group is already in load phase
// t + 1, so a piece issued in a LOAD phase may then only overwrite the slot of step t - 2 (lookahead RING - 2); pieces issued
// in an MFMA phase may always overwrite the slot of step t - 1.
// TAPS = 3: the kx-triple of a 3x3 layer.  TAPS = 1: the same loop for a 1x1 / linear layer (one tap, no neighbour rows, no ky):
// what remains of the design there is the ping-pong schedule and the immediate-offset reads.
template <int TG, int TX, int NWM, int NWN, int RING, int VAR, int SEG, int TAPS = 3>
__global__ __launch_bounds__(512, 1) void wgrad_kx3_kernel(const Kx3Args p) {
    constexpr int NW = 8, BKP = 64;
    static_assert(NWM * NWN == NW, "8 waves");
    static_assert(SEG == 16 || SEG == 32 || SEG == 64, "segment = min(image width, 64)");
    static_assert(TAPS == 3 || (TAPS == 1 && SEG == 64), "one tap: 64 consecutive rows per K-step");
    constexpr int CG = TG / 8, CX = TX / 8;
    constexpr int NSEG = BKP / SEG, SS = TAPS == 1 ? 64 : kx_seg_stride(SEG);   // segments per K-step, tile rows per segment
    constexpr int G_PIECES = BKP * CG / 64;                    // 1 KiB pieces of the gy tile
    constexpr int X_PIECES = TAPS == 1 ? CX : kx_x_pieces(SEG, CX);             // ... of the x tile
    static_assert(G_PIECES % NW == 0, "gy pieces divide over the waves");
    constexpr int G_IT = G_PIECES / NW, X_IT = (X_PIECES + NW - 1) / NW, P_IT = G_IT + X_IT;
    constexpr int XR = X_PIECES % NW;                          // waves below XR issue X_IT x pieces, the others X_IT - 1 (XR != 0)
    constexpr int G_BYTES = BKP * TG * 2, X_BYTES = X_PIECES * 1024;
    constexpr int KK_ROWS = SEG == 64 ? 32 : (32 / SEG) * SS, H_ROWS = SEG == 16 ? SS : 16;   // x-tile rows of 32 / 16 pixels
    constexpr int X_REGION = RING * G_BYTES;
    static_assert(RING * (G_BYTES + X_BYTES) <= 160 * 1024, "LDS");
    constexpr int WTG = TG / NWM, WTX = TX / NWN, MF = WTG / 16, NF = WTX / 16;
    static_assert(WTG % 16 == 0 && WTX % 16 == 0, "wave tile");
    constexpr bool RD_SPLIT = VAR == 1 || VAR == 2 || VAR == 3;
    constexpr int LOOK = (VAR == 2 || VAR == 3) ? RING - 2 : RING - 1;
    constexpr int IT_SPLIT = VAR == 1 ? 0 : ((VAR == 2 || VAR == 4) ? G_IT : P_IT);   // piece iterations [0, IT_SPLIT) go in the load phase
    static_assert(LOOK >= 2, "the wait at the end of load phase t covers K-step t + 1");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / NWN, wn = wave % NWN;

    int bx, chunk_id;
    if (p.xcd_order) {   // all (co, ky, ci) tiles of a pixel chunk on one XCD: they share gy / x through its L2
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
    const int ky = TAPS == 3 ? bx % 3 : 0;
    const int co_tile = TAPS == 3 ? bx / 3 : bx;
    const int co0 = co_tile * TG, ci0 = ci_tile * TX;
    const int p_begin = chunk_id * p.chunk_px;
    const int p_end = min(p.M, p_begin + p.chunk_px);
    const int nsteps = (p_end - p_begin) / BKP;                // (M and the chunk size are multiples of 64: host)
    if (nsteps <= 0) return;
    const int w = p.w, hw = p.h * p.w;
    const int dty = TAPS == 3 ? ky - 1 : 0;

    // ---- staging bookkeeping (fixed per thread) ---------------------------------------------------------------------------
    int g_voff[G_IT];
#pragma unroll
    for (int it = 0; it < G_IT; ++it) {
        const int id = (it * NW + wave) * 64 + lane;
        const int r = id / CG, sl = id - r * CG;
        g_voff[it] = (r * p.ldo + co0 + kx_swz<TG>(sl, r) * 8) * 2;
    }
    // need: three bits per segment -- bit 0 = pixel of the segment, bit 1 = its left neighbour, bit 2 = its right neighbour; 0 = unused row
    int x_voff[X_IT], x_need[X_IT];
#pragma unroll
    for (int it = 0; it < X_IT; ++it) {
        const int id = (it * NW + wave) * 64 + lane;
        const int R = id / CX, sl = id - R * CX;
        const int sg = R / SS, o = R - sg * SS;                                  // segment, row inside it (0 = left neighbour)
        const int need = TAPS == 1 ? 1 : (o == 0 ? 2 : (o == SEG + 1 ? 4 : (o > SEG + 1 ? 0 : 1)));
        x_need[it] = sg < NSEG ? need << (3 * sg) : 0;
        x_voff[it] = ((sg * SEG + o) * p.ldx + ci0 + kx_swz<TX>(sl, R) * 8) * 2;   // relative to pixel (pz + dty w - 1); one tap: to pixel pz
    }
    const long long dshift = TAPS == 1 ? 0 : ((long long)dty * w - 1) * p.ldx;                 // elements
    const bf16* xb = p.x + dshift;
    const unsigned xb_bytes = (unsigned)((long long)p.x_bytes - dshift * 2);
    const unsigned g_bytes = (unsigned)((long long)p_end * p.ldo * 2);

    // piece iteration IT of K-step `step` into ring slot `slot`: IT < G_IT a gy piece, else an x piece
    auto x_have = [&](int step) {   // one scalar mask per step against one per-lane bit: no divergent control flow around the DMA issue
        int have = 0;
        if constexpr (TAPS == 1) return 1;     // (rows beyond the tensor read as zeros through the descriptor's extent)
#pragma unroll
        for (int sg = 0; sg < NSEG; ++sg) {
            const int pz = p_begin + step * BKP + sg * SEG;
            const int x0 = pz & (w - 1);
            const int uy = ((pz & (hw - 1)) >> p.w_shift) + dty;
            if ((unsigned)uy < (unsigned)p.h) have |= (1 | (x0 != 0 ? 2 : 0) | (x0 + SEG < w ? 4 : 0)) << (3 * sg);
        }
        return have;
    };
    auto issue_piece = [&](auto it_c, int step, int slot, int have) {
        constexpr int IT = decltype(it_c)::value;
        const int pz = p_begin + step * BKP;
        if constexpr (IT < G_IT) {
            buffer_load_lds16(p.gy, g_bytes, smem + slot * G_BYTES + (IT * NW + wave) * 1024, g_voff[IT], pz * p.ldo * 2);
        } else {
            constexpr int XI = IT - G_IT;
            if (XR != 0 && XI == X_IT - 1 && wave >= XR) return;
            buffer_load_lds16(xb, xb_bytes, smem + X_REGION + slot * X_BYTES + (XI * NW + wave) * 1024,
                              (x_need[XI] & have) ? x_voff[XI] : OOB_OFFSET, pz * p.ldx * 2);
        }
    };
    auto stage_issue = [&](int step, int slot) {
        const int have = x_have(step);
        static_for<P_IT>([&](auto it_c) { issue_piece(it_c, step, slot, have); });
    };

    // ---- fragment addressing: lane (g = lane>>4, q = (lane>>2)&3, pp = lane&3) supplies the row of pixel 4g + q (+ 16 h + 32 kk),
    // columns base + 4pp .. +3.  Everything but the lane's own part is an immediate.
    const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    const unsigned smem_addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    unsigned a_base[MF];
#pragma unroll
    for (int i = 0; i < MF; ++i) {
        const int r = 4 * g + q;
        const int col = wm * WTG + i * 16 + 4 * pp;
        a_base[i] = smem_addr + r * (TG * 2) + kx_swz<TG>(col >> 3, r) * 16 + (pp & 1) * 8;
    }
    unsigned b_base[TAPS][NF];
#pragma unroll
    for (int tp = 0; tp < TAPS; ++tp) {
        const int R = 4 * g + q + tp;                 // tile row of tap kx = tp for pixel 4g + q (row 0 of a segment = left neighbour)
#pragma unroll
        for (int j = 0; j < NF; ++j) {
            const int col = wn * WTX + j * 16 + 4 * pp;
            b_base[tp][j] = smem_addr + X_REGION + R * (TX * 2) + kx_swz<TX>(col >> 3, R) * 16 + (pp & 1) * 8;
        }
    }
    // (a slot offset beyond the 16-bit field: second base registers for the upper half of the ring)
    constexpr int A_IN = 32 * TG * 2 + 16 * TG * 2, B_IN = (KK_ROWS + H_ROWS) * TX * 2;     // largest offsets inside a slot
    constexpr bool A_TWO = (RING - 1) * G_BYTES + A_IN >= 65536, B_TWO = (RING - 1) * X_BYTES + B_IN >= 65536;
    static_assert(!A_TWO || ((RING - 3) * G_BYTES + A_IN < 65536 && G_BYTES + A_IN < 65536), "gy slot offsets");
    static_assert(!B_TWO || ((RING - 3) * X_BYTES + B_IN < 65536 && X_BYTES + B_IN < 65536), "x slot offsets");
    unsigned a_base2[MF], b_base2[TAPS][NF];
#pragma unroll
    for (int i = 0; i < MF; ++i) a_base2[i] = a_base[i] + (A_TWO ? 2 * G_BYTES : 0);
#pragma unroll
    for (int tp = 0; tp < TAPS; ++tp)
#pragma unroll
        for (int j = 0; j < NF; ++j) b_base2[tp][j] = b_base[tp][j] + (B_TWO ? 2 * X_BYTES : 0);

    f32x4 acc[TAPS][MF][NF];
#pragma unroll
    for (int tp = 0; tp < TAPS; ++tp)
#pragma unroll
        for (int i = 0; i < MF; ++i)
#pragma unroll
            for (int j = 0; j < NF; ++j) acc[tp][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // bias gradient = column sums of gy: the gy fragments are in registers anyway (lane L holds 8 pixels of output channel
    // L & 15), the owning waves add them up with vector ALU work in their LOAD phase; fragment i of a wave row belongs to the
    // wave with wn == i % NWN.  The 3 x tiles_ci blocks of a pixel chunk that share a co tile all hold the same gy fragments:
    // they take the K-steps in turn (block `mine` sums steps mine, mine + nshare, ...), so no block carries the extra ~80
    // vector instructions in every step -- with ONE owner block per co tile (ky = 1, ci tile 0) that block, and with it
    // the launch (one block per CU), ran 25-30 % longer than the bias-free kernel.  Every block ADDS its share into dbias
    // with atomics (single-chunk launches too): the caller passes a zeroed dbias or one that holds a running sum.
    const bool do_bias = p.dbias != nullptr;
    const int nshare = TAPS * p.tiles_ci;
    int bias_ctr = ky * p.tiles_ci + ci_tile;     // steps until this block's next turn
    bool own = false, own_prev = false;
    constexpr int NB = (MF + NWN - 1) / NWN;
    float bsum[NB];
#pragma unroll
    for (int k = 0; k < NB; ++k) bsum[k] = 0.f;

    auto join = [](bf16x4 lo, bf16x4 hi) { return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]}; };
    bf16x4 alo[2][MF], ahi[2][MF], blo[2][TAPS][NF], bhi[2][TAPS][NF];
    const int grp = wave >> 2;

    // read number R of the NR = 2 MF + 6 NF transposing reads of 32-pixel half KK of ring slot SLOT
    constexpr int NR = 2 * MF + 2 * TAPS * NF, NM = TAPS * MF * NF;
    auto frag_read = [&](auto kk_c, auto r_c, auto slot_c) {
        constexpr int KK = decltype(kk_c)::value, R = decltype(r_c)::value, SLOT = decltype(slot_c)::value;
        if constexpr (R < 2 * MF) {
            constexpr int i = R / 2, h = R % 2;
            constexpr int OFF = ((A_TWO && SLOT >= 2) ? (SLOT - 2) * G_BYTES : SLOT * G_BYTES) + KK * 32 * TG * 2 + h * 16 * TG * 2;
            const unsigned ab = (A_TWO && SLOT >= 2) ? a_base2[i] : a_base[i];
            if constexpr (h == 0) alo[KK][i] = lds_tr16<OFF>(ab);
            else ahi[KK][i] = lds_tr16<OFF>(ab);
        } else {
            constexpr int rr = R - 2 * MF, tp = rr / (2 * NF), j = (rr / 2) % NF, h = rr % 2;
            constexpr int OFF = ((B_TWO && SLOT >= 2) ? (SLOT - 2) * X_BYTES : SLOT * X_BYTES) + KK * KK_ROWS * TX * 2 + h * H_ROWS * TX * 2;
            const unsigned bb = (B_TWO && SLOT >= 2) ? b_base2[tp][j] : b_base[tp][j];
            if constexpr (h == 0) blo[KK][tp][j] = lds_tr16<OFF>(bb);
            else bhi[KK][tp][j] = lds_tr16<OFF>(bb);
        }
    };
    auto mfma_one = [&](auto kk_c, auto m_c) {
        constexpr int KK = decltype(kk_c)::value, M = decltype(m_c)::value;
        constexpr int i = M / (TAPS * NF), tp = (M / NF) % TAPS, j = M % NF;
        acc[tp][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(join(alo[KK][i], ahi[KK][i]), join(blo[KK][tp][j], bhi[KK][tp][j]), acc[tp][i][j], 0, 0, 0);
    };
    auto bias_add = [&](auto kk_c) {
        constexpr int KK = decltype(kk_c)::value;
#pragma unroll
        for (int i = 0; i < MF; ++i)
            if (i % NWN == wn) {
                float a = 0.f;
#pragma unroll
                for (int e = 0; e < 4; ++e) a += (float)alo[KK][i][e] + (float)ahi[KK][i][e];
                bsum[i / NWN] += a;
            }
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;

#pragma unroll
    for (int st = 0; st < LOOK; ++st)
        if (st < nsteps) stage_issue(st, st);
    wait_vm<0>();
    if (grp == 1) __builtin_amdgcn_s_barrier();

    // one K-step with the ring slot as a compile-time constant
    auto step = [&](int t, auto slot_c) {
        constexpr int SLOT = decltype(slot_c)::value;
        constexpr int FILL = (SLOT + LOOK) % RING;
        const bool more = t + LOOK < nsteps;
        own_prev = own;
        own = do_bias && bias_ctr == 0;
        bias_ctr = bias_ctr == 0 ? nshare - 1 : bias_ctr - 1;
        // ---- load phase ----------------------------------------------------------------------------------------------------
        __builtin_amdgcn_s_barrier();
        if constexpr (RD_SPLIT) {   // (the second half's gy fragments of the previous step are still in registers: their bias sums first)
            if (own_prev) bias_add(I1{});
            __builtin_amdgcn_sched_barrier(0);
        }
        static_for<NR>([&](auto r_c) { frag_read(I0{}, r_c, slot_c); });
        if constexpr (!RD_SPLIT) static_for<NR>([&](auto r_c) { frag_read(I1{}, r_c, slot_c); });
        __builtin_amdgcn_sched_barrier(0);
        const int have = more ? x_have(t + LOOK) : 0;
        if (more) static_for<IT_SPLIT>([&](auto it_c) { issue_piece(it_c, t + LOOK, FILL, have); });
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if (own) {
            bias_add(I0{});
            if constexpr (!RD_SPLIT) bias_add(I1{});
        }
        // every piece of K-step t + 1 has landed; younger ones stay in flight: (LOOK - 2) whole steps + what this phase issued
        {
            constexpr int NLG = IT_SPLIT < G_IT ? IT_SPLIT : G_IT;                      // gy pieces issued in a load phase
            constexpr int NLX = IT_SPLIT > G_IT ? IT_SPLIT - G_IT : 0;                  // x piece iterations issued in a load phase
            constexpr int P_HI = G_IT + X_IT, P_LO = G_IT + X_IT - (XR != 0 ? 1 : 0);   // pieces per step: waves below XR / the others
            constexpr int L_HI = NLG + NLX, L_LO = NLG + NLX - ((XR != 0 && NLX == X_IT) ? 1 : 0);
            if (!more) wait_vm<0>();
            else if (XR != 0 && wave < XR) wait_vm<(LOOK - 2) * P_HI + L_HI>();
            else wait_vm<(LOOK - 2) * P_LO + L_LO>();
        }
        // ---- MFMA phase ----------------------------------------------------------------------------------------------------
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_s_setprio(1);
        if constexpr (!RD_SPLIT && IT_SPLIT == P_IT) {
            static_for<NM>([&](auto m_c) { mfma_one(I0{}, m_c); });
            static_for<NM>([&](auto m_c) { mfma_one(I1{}, m_c); });
        } else {
            // hand schedule: the NR reads of the second half after the first NR MFMAs, one each; the remaining DMA pieces
            // spread over the phase, each after a whole MFMA
            constexpr int NP = P_IT - IT_SPLIT;                   // piece iterations issued in this phase
            static_for<NM>([&](auto m_c) {
                constexpr int M = decltype(m_c)::value;
                mfma_one(I0{}, m_c);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (RD_SPLIT) {
                    constexpr int r0 = M * NR / NM, r1 = (M + 1) * NR / NM;
                    static_for<r1 - r0>([&](auto d_c) { frag_read(I1{}, std::integral_constant<int, r0 + decltype(d_c)::value>{}, slot_c); });
                }
                if constexpr (NP > 0) {   // piece k after MFMA (2 k + 1) NM / (2 NP) of the 2 NM in this phase, first half only when it fits
                    static_for<NP>([&](auto k_c) {
                        constexpr int K = decltype(k_c)::value;
                        if constexpr ((2 * K + 1) * 2 * NM / (2 * NP) == M) {
                            if (more) issue_piece(std::integral_constant<int, IT_SPLIT + K>{}, t + LOOK, FILL, have);
                        }
                    });
                }
                __builtin_amdgcn_sched_barrier(0);
            });
            if constexpr (RD_SPLIT) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
            }
            static_for<NM>([&](auto m_c) {
                constexpr int M = decltype(m_c)::value + NM;
                mfma_one(I1{}, m_c);
                if constexpr (NP > 0) {
                    __builtin_amdgcn_sched_barrier(0);
                    static_for<NP>([&](auto k_c) {
                        constexpr int K = decltype(k_c)::value;
                        if constexpr ((2 * K + 1) * 2 * NM / (2 * NP) == M) {
                            if (more) issue_piece(std::integral_constant<int, IT_SPLIT + K>{}, t + LOOK, FILL, have);
                        }
                    });
                    __builtin_amdgcn_sched_barrier(0);
                }
            });
        }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
    };

    for (int t = 0; t < nsteps; t += RING) {
        step(t, std::integral_constant<int, 0>{});
        if (t + 1 < nsteps) step(t + 1, std::integral_constant<int, 1>{});
        if (t + 2 < nsteps) step(t + 2, std::integral_constant<int, 2 % RING>{});
        if constexpr (RING == 4) {
            if (t + 3 < nsteps) step(t + 3, std::integral_constant<int, 3 % RING>{});
        }
    }
    if constexpr (RD_SPLIT) {
        if (own) bias_add(I1{});
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();

    // ---- fp32 results into dw[co][ky*3 + kx][ci]; D layout: row = (lane>>4)*4 + reg (co), col = lane&15 (ci)
    const size_t ldw = (size_t)(TAPS * TAPS) * p.c_in;
#pragma unroll
    for (int i = 0; i < MF; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int co = co0 + wm * WTG + i * 16 + g * 4 + r;
#pragma unroll
            for (int tp = 0; tp < TAPS; ++tp) {
                float* rowp = p.dw + (size_t)co * ldw + (size_t)(ky * TAPS + tp) * p.c_in;
#pragma unroll
                for (int j = 0; j < NF; ++j) {
                    const int ci = ci0 + wn * WTX + j * 16 + (lane & 15);
                    if (p.plain && !p.accum) rowp[ci] = acc[tp][i][j][r];
                    else if (p.plain) rowp[ci] += acc[tp][i][j][r];
                    else atomicAdd(rowp + ci, acc[tp][i][j][r]);
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
                if (lane < 16) atomicAdd(p.dbias + co, v);
            }
    }
}

int g_kx3_ring = 0;     // 0 = default (4 where it fits), 3 / 4: A/B
int g_kx3_var = -1;     // schedule variant (see the kernel); -1 = default (4 with a ring of 4)
int g_kx3_blocks = 0;   // 0 = cost model, else target number of blocks

template <int TG, int TX, int NWM, int NWN, int RING, int VAR, int SEG, int TAPS = 3>
int kx3_launch_v(const Kx3Args& a, dim3 grid, hipStream_t s) {
    constexpr int BYTES = RING * (TAPS == 1 ? 64 * (TG + TX) * 2 : kx_stage_bytes(TG, TX, SEG));
    static_assert(BYTES <= 160 * 1024, "LDS");
    static TvPerDeviceOnce attr_once;
    if (attr_once.first()) {
        (void)hipFuncSetAttribute((const void*)wgrad_kx3_kernel<TG, TX, NWM, NWN, RING, VAR, SEG, TAPS>, hipFuncAttributeMaxDynamicSharedMemorySize, BYTES);
    }
    hipLaunchKernelGGL((wgrad_kx3_kernel<TG, TX, NWM, NWN, RING, VAR, SEG, TAPS>), grid, dim3(512), BYTES, s, a);
    return 0;
}

template <int TG, int TX, int NWM, int NWN, int RING, int SEG, int TAPS = 3>
int kx3_launch_r(Kx3Args a, hipStream_t s, bool plan_only) {
    const int tiles_co = a.c_out / TG;
    a.tiles_ci = a.c_in / TX;
    const long long base = (long long)tiles_co * TAPS * a.tiles_ci;
    // split-K over pixel chunks: one block per CU; with the XCD-grouped order a chunk's `base` tiles share an XCD (32 CUs).
    // minimise  rounds x (pixels per block) x t_pixel  +  atomic bytes / 1.3 TB/s  (MI355X_MICROARCH "Global float atomics")
    const double t_px = 2.0 * TG * TX * TAPS * 256.0 / 1200e12;
    const double tile_bytes = 4.0 * TG * TX * TAPS;
    long long split = 1;
    bool xcd = false;
    double best = 1e30;
    const long long smax = a.M / 512 > 0 ? a.M / 512 : 1;
    if (g_kx3_blocks) {
        split = (g_kx3_blocks + base - 1) / base;
        xcd = base <= 32 && split >= 8;
    } else {
        for (long long sp = 1; sp <= smax && sp <= 4096; ++sp) {
            const double rounds = (double)((long long)((base * sp + 255) / 256));
            const double t = rounds * ((double)a.M / sp) * t_px + (sp > 1 ? base * sp * tile_bytes / 1.3e12 : 0.0);
            if (t < best) { best = t; split = sp; xcd = false; }
        }
        if (base <= 32) {
            for (long long sp = 8; sp <= smax && sp <= 4096; sp += 8) {
                const double rounds = (double)((long long)((base * (sp / 8) + 31) / 32));
                const double t = rounds * ((double)a.M / sp) * (t_px * 0.85) + base * sp * tile_bytes / 1.3e12;
                if (t < best) { best = t; split = sp; xcd = true; }
            }
        }
    }
    long long chunk = (a.M + split - 1) / split;
    if (chunk < 512) chunk = 512;
    chunk = (chunk + 63) / 64 * 64;
    a.chunk_px = (int)chunk;
    const int ny = (int)((a.M + chunk - 1) / chunk);
    a.plain = (ny == 1) ? 1 : 0;
    if (plan_only) return 100 + a.plain;
    a.base = (int)base;
    a.ny = ny;
    a.xcd_order = (ny > 1 && xcd) ? 1 : 0;
    dim3 grid((unsigned)base, (unsigned)ny);
    if (a.xcd_order) grid = dim3((unsigned)(8 * base * ((ny + 7) / 8)), 1);
    if constexpr (TAPS == 1) {
        if (g_kx3_var == 0) return kx3_launch_v<TG, TX, NWM, NWN, RING, 0, SEG, 1>(a, grid, s);      // (A/B)
        return kx3_launch_v<TG, TX, NWM, NWN, RING, 4, SEG, 1>(a, grid, s);
    } else if constexpr (RING == 4 && SEG == 64) {   // (the lookahead RING - 2 variants need a ring of 4; A/B variants on the 64-pixel form only)
        switch (g_kx3_var < 0 ? 4 : g_kx3_var) {
            case 1: return kx3_launch_v<TG, TX, NWM, NWN, RING, 1, SEG>(a, grid, s);
            case 2: return kx3_launch_v<TG, TX, NWM, NWN, RING, 2, SEG>(a, grid, s);
            case 3: return kx3_launch_v<TG, TX, NWM, NWN, RING, 3, SEG>(a, grid, s);
            case 4: return kx3_launch_v<TG, TX, NWM, NWN, RING, 4, SEG>(a, grid, s);
            default: break;
        }
    }
    if (g_kx3_var != 0) return kx3_launch_v<TG, TX, NWM, NWN, RING, 4, SEG>(a, grid, s);    // schedule 4 works at any ring depth >= 3
    return kx3_launch_v<TG, TX, NWM, NWN, RING, 0, SEG>(a, grid, s);
}

template <int TG, int TX, int NWM, int NWN, int SEG>
int kx3_launch_s(const Kx3Args& a, hipStream_t s, bool plan_only) {
    if constexpr (4 * kx_stage_bytes(TG, TX, SEG) <= 160 * 1024) {
        if (g_kx3_ring != 3) return kx3_launch_r<TG, TX, NWM, NWN, 4, SEG>(a, s, plan_only);
    }
    return kx3_launch_r<TG, TX, NWM, NWN, 3, SEG>(a, s, plan_only);
}

template <int TG, int TX, int NWM, int NWN>
int kx3_launch(const Kx3Args& a, hipStream_t s, bool plan_only) {
    if (a.w >= 64) return kx3_launch_s<TG, TX, NWM, NWN, 64>(a, s, plan_only);
    if (a.w == 32) return kx3_launch_s<TG, TX, NWM, NWN, 32>(a, s, plan_only);
    return kx3_launch_s<TG, TX, NWM, NWN, 16>(a, s, plan_only);
}

}  // namespace

// A/B hooks (tools/, tests): ring depth (0 = default, 3, 4), block target for the split-K choice (0 = cost model)
extern "C" int tv_set_wgrad_kx3(int enable, int ring, int blocks);
int g_kx3_enable = 2;     // 0 = off, 1 = 3x3 / stride-1 layers, 2 = linear / 1x1 layers too (one-tap instantiation); + 10 t: tile A/B
int g_kx3_prefer128 = 0;
extern "C" int tv_set_wgrad_kx3(int enable, int ring, int blocks) {   // ring = depth + 10 x schedule variant
    g_kx3_enable = enable;
    g_kx3_ring = ring % 10;
    g_kx3_var = ring >= 10 ? ring / 10 : (ring == 0 ? -1 : 0);     // a bare ring depth selects schedule 0 (A/B); 0 = all defaults
    g_kx3_blocks = blocks % 100000;
    g_kx3_prefer128 = blocks / 100000;       // A/B: 2 = 192 x 96 x 3 tiles even where 128 x 128 x 3 divides the layer too
    return 0;
}

// Called by wgrad_impl (wgrad_tn.hip).  Returns -1 when the layer is not this kernel's (the caller goes on to the single-tap
// kernel), 100 / 101 for plan_only (accumulates / overwrites), 0 after a launch.
int tv_wgrad_kx3_try(const tv_conv_desc* d, const void* x, const void* gy, float* dw, float* dbias, hipStream_t s, bool plan_only, int accum) {
    if (!g_kx3_enable) return -1;
    auto log2_exact = [](int v) { int sh = 0; while ((1 << sh) < v) ++sh; return ((1 << sh) == v) ? sh : -1; };
    // linear / 1x1 layers through the one-tap instantiation: nothing to share between taps, but the schedule and the
    // immediate-offset reads carry over -- tools/probes/ab_wgrad_bias.py 64, with bias, single-tap kernel -> this: 1536->6144 @16
    // 0.368 -> 0.305 ms, 6144->1536 0.290 -> 0.262, 1536->4608 0.265 -> 0.218, 768->3072 @32 0.345 -> 0.313, 3072->768 0.328 ->
    // 0.294, 384->1536 @64 0.394 -> 0.320, 1536->384 0.354 -> 0.314 (809-1181 TFLOP/s; 45.5 -> 39.7 ms per micro-batch).
    // 256 x 128 / 128 x 256 tiles measured 0-60 % slower than 192 x 192, schedule 0 2 % slower than schedule 4.
    if (d->kh == 1 && d->kw == 1) {
        const long long Ml = (long long)d->batch * d->h_out * d->w_out;
        const bool ok = g_kx3_enable % 10 >= 2 && d->stride == 1 && d->pad == 0 && d->up_shift == 0 && d->dil_mask == 0 && d->h_in == d->h_out &&
                        d->w_in == d->w_out && Ml % 64 == 0 && Ml * d->ldx * 2 < (1ll << 31) && Ml * d->ldo * 2 < (1ll << 31) &&
                        d->c_out % 192 == 0 && d->c_in % 192 == 0;
        if (!ok) return -1;
        Kx3Args a;
        a.x = (const bf16*)x; a.gy = (const bf16*)gy; a.dw = dw; a.dbias = dbias;
        a.M = (int)Ml; a.h = 1; a.w = 64; a.c_in = d->c_in; a.ldx = d->ldx; a.c_out = d->c_out; a.ldo = d->ldo;
        a.tiles_ci = 1; a.chunk_px = 0; a.hw_shift = 6; a.w_shift = 6; a.plain = 0;
        a.xcd_order = 0; a.base = 1; a.ny = 1; a.accum = accum;
        a.x_bytes = (unsigned)(Ml * d->ldx * 2);
        // A/B: tv_set_wgrad_kx3(2 + 10 t, ...): t = 1: 256 x 128 tiles, t = 2: 128 x 256 tiles where they divide the layer
        const int tsel = g_kx3_enable / 10;
        if (tsel == 1 && d->c_out % 256 == 0 && d->c_in % 128 == 0) return kx3_launch_r<256, 128, 4, 2, 3, 64, 1>(a, s, plan_only);
        if (tsel == 2 && d->c_out % 128 == 0 && d->c_in % 256 == 0) return kx3_launch_r<128, 256, 2, 4, 3, 64, 1>(a, s, plan_only);
        return kx3_launch_r<192, 192, 4, 2, 3, 64, 1>(a, s, plan_only);
    }
    const int wsh = log2_exact(d->w_out), hwsh = log2_exact(d->h_out * d->w_out);
    const long long M = (long long)d->batch * d->h_out * d->w_out;
    const long long xbytes = M * d->ldx * 2, gbytes = M * d->ldo * 2;
    const bool geo = d->kh == 3 && d->kw == 3 && d->stride == 1 && d->pad == 1 && d->up_shift == 0 && d->dil_mask == 0 &&
                     d->h_in == d->h_out && d->w_in == d->w_out && wsh >= 4 && hwsh >= 6 &&
                     xbytes + (long long)(d->w_in + 1) * d->ldx * 2 < (1ll << 31) && gbytes < (1ll << 31);
    if (!geo) return -1;
    const bool t192 = d->c_out % 192 == 0 && d->c_in % 96 == 0;
    const bool t128 = d->c_out % 128 == 0 && d->c_in % 128 == 0;
    if (!t192 && !t128) return -1;
    Kx3Args a;
    a.x = (const bf16*)x; a.gy = (const bf16*)gy; a.dw = dw; a.dbias = dbias;
    a.M = (int)M; a.h = d->h_out; a.w = d->w_out; a.c_in = d->c_in; a.ldx = d->ldx; a.c_out = d->c_out; a.ldo = d->ldo;
    a.tiles_ci = 1; a.chunk_px = 0; a.hw_shift = hwsh; a.w_shift = wsh; a.plain = 0;
    a.xcd_order = 0; a.base = 1; a.ny = 1; a.accum = accum;
    a.x_bytes = (unsigned)xbytes;
    // both shapes divide the 384 / 768 / 1536-channel layers: the 128 x 128 x 3 tile measured 3-8 % faster there
    // (tools/probes/ab_wgrad3.py: 384 @64 0.561 -> 0.544 ms, 768 @32 0.555 -> 0.541, 1536 @16 0.578 -> 0.534)
    if (t192 && !(t128 && g_kx3_prefer128 != 2)) return kx3_launch<192, 96, 4, 2>(a, s, plan_only);
    return kx3_launch<128, 128, 2, 4>(a, s, plan_only);
}
