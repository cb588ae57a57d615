// conv3x3_halo_kernel: 3x3 / stride-1 / pad-1 convolutions and their data gradients on spatial tiles with a staged halo
// (see the block comment below); arguments, helpers and epilogues in igemm_common.h, host dispatch in igemm_nt.hip.
#include "igemm_common.h"

namespace {

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
#ifdef TV_ABL_K1   // (ablation: one channel chunk = 9 of the 27 tap steps at 192 channels: fixed costs by difference)
    const int cch = 1;
#else
    const int cch = p.c_in / BK;
#endif

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
#if TV_HALO_A_NT
        if (j < A_PIECES) buffer_load_lds16_nt(p.x, p.x_bytes, dst + j * 1024, VOFF_REGS ? a_voff_r[VOFF_REGS ? it : 0] : a_voff_of(j), ch * (BK * 2));
#else
        if (j < A_PIECES) buffer_load_lds16(p.x, p.x_bytes, dst + j * 1024, VOFF_REGS ? a_voff_r[VOFF_REGS ? it : 0] : a_voff_of(j), ch * (BK * 2));
#endif
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
#if TV_HALO_P8
        // Round 3: a tap step in FOUR phases of 12 MFMAs (quadrants: 2 fragment rows x 3 fragment columns x both halves of
        // BK), each with its own small load section -- 4-10 fragment reads and ONE DMA piece -- instead of one load phase of 20
        // reads + a burst of 3-4 pieces against one MFMA phase of 48: the pieces of a burst queue at the address pipe
        // (~150 wave-cycles each against ~60 alone; the ablation of the two-phase form: 2.19 ms, DMA issue off 1.54), which made
        // the load phase longer than the partner's MFMA phase.  Same ping-pong (group 1 one barrier behind), same ring, same
        // counted wait (at the end of the step's last load section), same results bit for bit; the structure that took the
        // generic 256-row tiles from 0.345 to 0.307 ms (igemm_nt.hip, eight-phase loop).
        //   L1 reads A0 + B_lo, issues weight piece 0 | L2 reads B_hi, piece 1 | L3 reads A1, piece 2 | L4 reads B_lo, halo piece, wait
        //   hazards: the slot of slab t+2 was last read in L4 of step t-1 by either group, complete (lgkmcnt) before the barrier
        //   that ends that section; the same group's L1 of step t follows two barriers later.
        static_assert(MF % 2 == 0 && NF % 2 == 0 && B_IT <= 3 && TV_PP_NM == 0, "halo eight-phase loop: one weight piece per load section");
        constexpr int MH = MF / 2, NH = NF / 2;
        __builtin_amdgcn_s_barrier();                 // the prologue's pieces have landed for every wave
        if (grp == 1) __builtin_amdgcn_s_barrier();   // the stagger: group 1 runs one barrier behind
        int bcur = 0;
        bf16x8 qa[2][MH], qb[2][NH];
        for (int ch = 0; ch < cch; ++ch) {
            const char* acur = a_buf + (ch & 1) * A_BYTES;
            char* anxt = a_buf + ((ch + 1) & 1) * A_BYTES;
            const bool more = ch + 1 < cch;
            static_for<0, 9>([&](auto tap_c) {
                constexpr int tap = decltype(tap_c)::value;
                constexpr int toff = (tap / 3) * HWD + (tap % 3);
                const char* const bslot = b_buf + bcur * B_BYTES;
                char* const bfill = b_buf + ((bcur + 2 >= BST) ? bcur + 2 - BST : bcur + 2) * B_BYTES;
                const bool b_go = (tap + 2 < 9) || more;
                const int b_koff = ((tap + 2 < 9) ? (tap + 2) * p.c_in + ch * BK : (tap + 2 - 9) * p.c_in + (ch + 1) * BK) * 2;
                int hpb = hp_base;
                asm volatile("" : "+v"(hpb));   // pin the address arithmetic to this tap
                auto rd_a = [&](auto a_c) {
                    constexpr int a = decltype(a_c)::value;
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                        for (int i = 0; i < MH; ++i) {
                            const int hp = hpb + (MH * a + i) * HWD + toff;
                            qa[kk][i] = *(const bf16x8*)(acur + hp * (BK * 2) + (((kk * 4 + fq) ^ (hp & 7)) << 4));
                        }
                };
                auto rd_b = [&](auto b_c) {
                    constexpr int b = decltype(b_c)::value;
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk) {
                        const int coff = ((kk * 4 + fq) ^ sw) * 16;
#pragma unroll
                        for (int j = 0; j < NH; ++j) qb[kk][j] = *(const bf16x8*)(bslot + b_row_off + bfrag_off(NH * b + j) * (BK * 2) + coff);
                    }
                };
                auto close_load = [&]() {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_sched_barrier(0);
                    __builtin_amdgcn_s_barrier();
                    __builtin_amdgcn_sched_barrier(0);
                };
                auto mm = [&](auto a_c, auto b_c) {
                    constexpr int a = decltype(a_c)::value, b = decltype(b_c)::value;
                    __builtin_amdgcn_s_setprio(1);
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                        for (int i = 0; i < MH; ++i)
#pragma unroll
                            for (int j = 0; j < NH; ++j)
                                acc[MH * a + i][NH * b + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qb[kk][j], qa[kk][i], acc[MH * a + i][NH * b + j], 0, 0, 0);
                    __builtin_amdgcn_s_setprio(0);
                    __builtin_amdgcn_sched_barrier(0);
                    __builtin_amdgcn_s_barrier();
                    __builtin_amdgcn_sched_barrier(0);
                };
                using I0 = std::integral_constant<int, 0>;
                using I1 = std::integral_constant<int, 1>;
                rd_b(I0{});
                __builtin_amdgcn_sched_barrier(0);
                rd_a(I0{});
                __builtin_amdgcn_sched_barrier(0);
                if (0 < B_IT && b_go) issue_b_piece(bfill, 0 < B_IT ? 0 : 0, b_koff);
                close_load();
                mm(I0{}, I0{});
                rd_b(I1{});
                __builtin_amdgcn_sched_barrier(0);
                if (1 < B_IT && b_go) issue_b_piece(bfill, 1 < B_IT ? 1 : 0, b_koff);
                close_load();
                mm(I0{}, I1{});
                rd_a(I1{});
                __builtin_amdgcn_sched_barrier(0);
                if (2 < B_IT && b_go) issue_b_piece(bfill, 2 < B_IT ? 2 : 0, b_koff);
                close_load();
                mm(I1{}, I1{});
                rd_b(I0{});
                __builtin_amdgcn_sched_barrier(0);
                if (tap < ATAPS && tap < A_IT && more) issue_a(anxt, tap, ch + 1);
                // in flight may stay, in issue order: [the previous step's halo piece] [this step's weight pieces] [this step's
                // halo piece]: the slab of the NEXT step (issued one step ago) has landed
                if (more) wait_vmcnt<nsure(tap - 1) + B_IT + nsure(tap)>();
                else if (tap + 2 < 9) wait_vmcnt<B_IT>();
                else wait_vmcnt<0>();
                close_load();
                mm(I1{}, I0{});
                bcur = (bcur + 1 == BST) ? 0 : bcur + 1;
            });
        }
        if (grp == 0) __builtin_amdgcn_s_barrier();   // (both groups have passed the same number of barriers)
#else
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
#endif
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
#ifdef TV_ABL_NO_EPI   // (ablation: main loop only)
    {
        float chk = 0.f;
#pragma unroll
        for (int i = 0; i < MF; ++i)
#pragma unroll
            for (int j = 0; j < NF; ++j) chk += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
        if (chk != 123.456f) return;
    }
#endif
    f32x4 bvals[NF];
    load_bias<WTN>(p, lane, n0 + wn * WTN, bvals);
    // wave-tile row r -> output pixel: fragment row i = r / 16 is tile row wm*MF + i, r % 16 the column
    const int pix0 = (b * p.h_out + y0 + wm * MF) * p.w_out + x0;
    epilogue<WTM, WTN, EPI, true>(p, acc, bvals, smem, wave, lane, n0 + wn * WTN, [&](int r) { return pix0 + (r >> 4) * p.w_out + (r & 15); });
    TV_T(6);
    TV_PROBE_DUMP(wave, lane);
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
int launch_halo_impl(IgemmArgs& a, hipStream_t s) {
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

}  // namespace

#ifdef TV_EXP_GN_EPI
extern "C" int tv_set_gn_epi_probe(void* dev_buf) {
    return hipMemcpyToSymbol(HIP_SYMBOL(g_gn_epi_buf), &dev_buf, sizeof(void*)) == hipSuccess ? 0 : 1;
}
#endif
namespace tvi {
int launch_halo(IgemmArgs& a, hipStream_t s) { return launch_halo_impl(a, s); }
#ifdef TV_PROBE
int set_halo_probe(void* dev_buf) { return hipMemcpyToSymbol(HIP_SYMBOL(g_probe_dev), &dev_buf, sizeof(void*)) == hipSuccess ? 0 : 1; }
#endif
}  // namespace tvi
