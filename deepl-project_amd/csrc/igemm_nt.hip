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
#include "igemm_common.h"

namespace tvi {
int g_cfg_bm = 0;      // 0 = heuristic, else 128 / 256
int g_cfg_bn = 0;      // 0 = heuristic, 256 = 256-wide N tiles whenever c_out % 256 == 0, 128 = never
int g_halo_ring = 3;    // weight ring depth of the halo kernel (2 / 3; 3 falls back to 2 where the LDS is too small)
bool g_halo_w4 = false;   // experiment: 256x192 halo tile with 4 waves (one per SIMD, 128x96 wave tiles)
bool g_xcd_order = true;  // column tiles of a row tile on one XCD (block order in the kernels)
int g_supertile = 0;      // 8-wave GEMM tiles: 0 = super-tile block order chosen per shape, 1 = row-tile-major (round 3), r * 100 + c = forced
bool g_epi_modes = true;   // compile-time epilogue forms (tv_set_igemm_epilogue(0): the generic one everywhere, for A/B timing)

// which epilogue form a call takes (see epilogue<>): 1 = residual add only, 2 = saved-derivative multiply, 0 = the rest
int epilogue_mode(const IgemmArgs& a) {
    if (a.rope) return 0;
    if (!g_epi_modes || a.shuffle || a.pre || a.out_bytes == 0) return 0;
    if (a.aux) return (a.aux_act == TV_ACT_DERIV && !a.res) ? 2 : 0;
    return (a.res && a.act == TV_ACT_NONE) ? 1 : 0;
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

}  // namespace tvi

namespace {

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

// Persistent tile walk (round 3).  A block of the plain two-stage DMA loop (the 256 x 256 tile) on a single-tap layer (1x1 /
// linear) walks a LIST of output tiles as ONE K-loop: the staging state wraps from the last K-step of a tile to step 0 of the
// block's next tile (both operands move through the scalar offset of their DMA pieces), so the first stage of the next tile
// is issued before the last K-step of the current one and lands under that step and under the register epilogue, which uses
// no LDS.  What this removes per tile: block dispatch, the offset set-up and -- the large part, 3.4 us of a 20-50 us tile --
// the first DMA round trip, which every CU of a launch used to pay at the same moment (the chip-wide prologue burst runs at
// ~11 B/clk/CU).  The wait that opens the next tile is COUNTED: vmcnt counts loads, stores and DMA together in issue order,
// so behind the prefetched pieces sit the epilogue's stores -- exactly NST per wave on a full tile -- and `vmcnt(NST)` retires
// the pieces without draining the stores (a persistent trial on the halo kernel in round 2 waited vmcnt(0) there: 1.00x).
// Which tiles: the blocks with id = x mod 8 share an XCD (round-robin placement) and own the row tiles = x mod 8, listed row
// tile by row tile (the column tiles of a row tile adjacent); block b of the P on that XCD takes list entries b, b + P, ...:
// at any moment the XCD's blocks sit on neighbouring entries, i.e. the column tiles of ONE or two row tiles, and share those
// activation rows through its L2 as the one-tile-per-block order did.  (A first form -- each block walking the column tiles
// of its own row tile -- re-read that row tile from HBM once per column tile wherever the row tiles of an XCD's blocks exceed
// its L2: K = 3072 -> N = 768 ran 0.269 -> 0.334 ms.)
template <int BM, int BN, int WGM, int WGN, int BK, int STAGES, int MODE>
constexpr bool igemm_persist_ok() {
    constexpr int MF = BM / WGM / 16, NF = BN / WGN / 16, NW = WGM * WGN;
    constexpr bool pipe_all = (MF + NF) * 4 * (BK / 32) <= TV_PIPE_ALL_MAX;
    constexpr bool pingpong = MODE != 0 && STAGES == 2 && NW == 8 && pipe_all && !TV_NO_PINGPONG;
    constexpr bool pipe2 = MODE != 0 && pipe_all && BK == 64 && !pingpong && !TV_NO_PIPE2;
    return MODE == 2 && STAGES == 2 && !pipe2 && !pingpong && BM * BN >= 128 * 128 && (NF % 2 == 0) && !TV_EPI_LDS && TV_GENERIC_BURST;
}
// store instructions per wave of a register epilogue form on a full tile (epilogue_direct: per fragment row two per pair of
// 32-channel blocks + one for a block without a partner; twice that where the derivative is saved as well)
template <int WTM, int WTN>
__device__ __forceinline__ constexpr int epi_store_count(bool saves) {
    constexpr int MF = WTM / 16, NC = WTN / 32;
    return MF * (2 * (NC / 2) + (NC & 1)) * (saves ? 2 : 1);
}

template <int BM, int BN, int WGM, int WGN, int BK, int STAGES, int MODE, int EPI, bool PERS = false, int PFORM = 0, bool P8 = false>
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
    // PERS instantiations serve single-tap layers only (1x1 / linear, stride 1: K = c_in): the gather state of the taps
    // (pixel coordinates per staged row) is dead after the first offset set-up instead of living across every epilogue
    constexpr bool PERSIST = PERS && igemm_persist_ok<BM, BN, WGM, WGN, BK, STAGES, MODE>();
    int tile_n, tile_m;
    int ntile = 1;   // tiles this block walks
    [[maybe_unused]] int walk_i = 0, walk_x = 0;   // persistent walk: entry of the XCD's tile list being COMPUTED, XCD label
    if constexpr (PERSIST) {
        walk_x = blockIdx.x & 7;
        walk_i = blockIdx.x >> 3;
        const int nlist = ((p.tiles_m - walk_x + 7) >> 3) * p.tiles_n;   // row tiles = walk_x mod 8, times the column tiles
        if (walk_i >= nlist) return;
        ntile = (nlist - 1 - walk_i) / p.nchunks + 1;                    // (p.nchunks = blocks per XCD label)
        const int r = walk_i / p.tiles_n;
        tile_n = walk_i - r * p.tiles_n;
        tile_m = r * 8 + walk_x;
    } else if (p.xcd_order == 2) {
        // Super-tile order (round 4).  An XCD runs ~32 blocks side by side; in the row-tile-major order above those are the
        // tiles_n column tiles of ONE row tile, which share the row panel through the XCD's L2 and NOTHING of the weights:
        // every row tile re-streams the whole weight matrix from beyond L2 (1536 -> 6144, 64 images: 64 x 18.9 MB = 1.2 GB
        // per launch for 69 MB of operands, profiles/r03_p8_gemm_pmc.json).  Here consecutive ids of an XCD form sup_r x sup_c
        // super-tiles: a row panel is shared by sup_c blocks and a weight panel by sup_r, and since the blocks of a super-tile
        // start together and advance along K in step, the XCD's L2 only has to hold the K-window they are in.
        const int lin = blockIdx.x, j = lin >> 3, per = p.sup_r * p.sup_c;
        const int sidx = (j / per) * 8 + (lin & 7), w = j % per;
        const int nsn = (p.tiles_n + p.sup_c - 1) / p.sup_c;
        const int sm = sidx / nsn, sn = sidx - sm * nsn;
        const int dr = w / p.sup_c;
        tile_m = sm * p.sup_r + dr;
        tile_n = sn * p.sup_c + (w - dr * p.sup_c);
        if (tile_m >= p.tiles_m || tile_n >= p.tiles_n) return;
    } else if (p.xcd_order) {
        const int lin = blockIdx.x, j = lin >> 3;
        tile_n = j % p.tiles_n;
        tile_m = (j / p.tiles_n) * 8 + (lin & 7);
        if (tile_m >= p.tiles_m) return;
    } else {
        tile_n = blockIdx.x % p.tiles_n;
        tile_m = blockIdx.x / p.tiles_n;
    }
    int m0 = tile_m * BM;
    int n0 = tile_n * BN;

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
    // persistent walk: list entry being STAGED, byte offsets of its activation rows / weight rows (the per-lane offsets are
    // tile-local there, see persist_reinit)
    [[maybe_unused]] int st_i = walk_i, st_am = m0 * p.ldx * 2, st_nb = n0 * p.K * 2;
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
                buffer_load_lds16(p.x, p.x_bytes, sbase + j * 1024, a_voff[it], koff * 2 + (PERSIST ? st_am : 0));
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
                buffer_load_lds16(p.w, p.w_bytes, sbase + A_BYTES + j * 1024, b_voff[it], kb * 2 + (PERSIST ? st_nb : 0));
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
        if constexpr (PERSIST) {   // single tap: the K-step is the channel chunk; behind the last one, step 0 of the next column tile
            ++st_t;
            ++st_ch;
            if (st_t == nk) {   // on to the block's next list entry (behind its last tile nothing is issued any more)
                st_i += p.nchunks;
                const int r = st_i / p.tiles_n, c = st_i - r * p.tiles_n;
                st_am = (r * 8 + walk_x) * (BM * p.ldx * 2);
                st_nb = c * (BN * p.K * 2);
                st_t = st_ch = 0;
            }
            return;
        }
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
    int fi = lane & 15, fq = lane >> 4;
    int sw = swz_of<BK>(fi);
    int a_row_off = (wm * WTM + fi) * (BK * 2);
    int b_row_off = A_BYTES + (wn * WTN + bfrag_lane_row(fi)) * (BK * 2);
    // Persistent loop: the main loop's per-lane state (DMA offsets, fragment addresses: ~20 registers) is REBUILT behind every
    // epilogue from a lane id the optimiser cannot see through, so that it does not live across the epilogue (nor the
    // epilogue's own lane arithmetic across the main loop): with both alive at once the EPI 0 instantiation spilled, and a
    // reload inside the K-loop waits vmcnt(0), i.e. for the DMA burst just issued.  Single tap: a staged row's pixel is m.
    [[maybe_unused]] auto persist_reinit = [&]() {
        int tl = threadIdx.x;
        asm volatile("" : "+v"(tl));
        const int ln = tl & 63, sr = ln / CPR, ss = ln % CPR;
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {   // tile-local: the tile's first row / weight row rides in the scalar offset
            const int row = (it * NW + wave) * RPI + sr;
            a_voff[it] = (row * p.ldx + (ss ^ swz_of<BK>(row & 15)) * 8) * 2;
        }
#pragma unroll
        for (int it = 0; it < B_IT; ++it) {
            const int row = (it * NW + wave) * RPI + sr;
            b_voff[it] = (row * p.K + (ss ^ swz_of<BK>(bfrag_reader(row % WTN))) * 8) * 2;
        }
        fi = ln & 15;
        fq = ln >> 4;
        sw = swz_of<BK>(fi);
        a_row_off = (wm * WTM + fi) * (BK * 2);
        b_row_off = A_BYTES + (wn * WTN + bfrag_lane_row(fi)) * (BK * 2);
    };

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
#ifdef TV_ABL_NO_LDSREAD
#pragma unroll
                for (int i = 0; i < MF; ++i) { af[i] = bf16x8{1, 1, 1, 1, 1, 1, 1, 1}; asm volatile("" : "+v"(af[i])); }
#pragma unroll
                for (int j = 0; j < NF; ++j) { bfr[j] = bf16x8{1, 1, 1, 1, 1, 1, 1, 1}; asm volatile("" : "+v"(bfr[j])); }
#else
#pragma unroll
                for (int i = 0; i < MF; ++i) af[i] = *(const bf16x8*)(sbase + a_row_off + i * 16 * (BK * 2) + coff);
#pragma unroll
                for (int j = 0; j < NF; ++j) bfr[j] = *(const bf16x8*)(sbase + b_row_off + bfrag_off(j) * (BK * 2) + coff);
#endif
#pragma unroll
                for (int i = 0; i < MF; ++i)
#pragma unroll
                    for (int j = 0; j < NF; ++j) {
#ifdef TV_ABL_NO_MFMA
                        asm volatile("" ::"v"(bfr[j]), "v"(af[i]));
#else
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
#endif
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
    if constexpr (P8) {
        // ---- eight-phase wave-group ping-pong (round 3; the 256 x 256 tile on single-tap layers) ------------------------------
        // Ablation of the plain loop on K = 1536 -> N = 6144 (profiles/r03_kernel_experiments.txt item 13): 0.351 ms, without the
        // DMA issue 0.274, without the fragment reads 0.275, without both 0.204 (= MFMAs + epilogue): the two halves of the data
        // path cost their full time ON TOP of the matrix work -- all eight waves issue the DMA burst together, read together
        // and multiply together.  Here a K-step is four phases of 16 MFMAs (one 64 x 32 quadrant of the wave's 128 x 64 tile
        // over the whole BK), each with its own small load section (4-12 fragment reads, 2 DMA pieces: no burst), and the two
        // wave groups (waves 0-3 / 4-7 = row halves, partners on every SIMD) run ONE BARRIER APART, so that a SIMD always has
        // one wave multiplying while the other loads (the structure of the guide's 256-square 8-phase template):
        //   wave:  [L_k: reads of quadrant k, 2 DMA pieces, waits]  barrier  [M_k: 16 MFMAs at priority 1]  barrier
        // A stage buffer is four REGIONS, each read in one load section and re-staged (K-step t+2) right behind its last read:
        //   L1 reads A0 + B0   issues B0 of step t+1   (B0 is read again in L4)
        //   L2 reads B1        issues A0 of step t+2
        //   L3 reads A1        issues B1 of step t+2
        //   L4 reads B0        issues A1 of step t+2,  then vmcnt(6): everything up to B0 of step t+1 has landed
        // A_a = the 64-row halves a of both groups' rows, B_b = the 32-channel halves b of the four waves' slabs; every wave
        // issues 2 pieces per region.  Hazards (g0 = waves 0-3, one barrier AHEAD of g1):
        //   WAR  a region's reads complete (lgkmcnt(0)) BEFORE the barrier that ends their load section; its re-staging DMA
        //        is issued in the same group's NEXT load section, two barriers later: g1's reads of it are complete too.
        //   RAW  every wave waits vmcnt in L4(t) before that section's barrier; step t+1 is first read in L1(t+1), two barriers
        //        later for either group: all eight waves' pieces have landed.
        // Two tile shapes: 256 x 256 (2 x 4 waves of 128 x 64: quadrants of 4 x 2 fragments, 16 MFMAs each) and 256 x 192
        // (4 x 2 waves of 64 x 96: row halves of 2 fragments, channel parts of 2 and 4 fragments -- a 32-channel block is the
        // unit the weight rows of a slab can be cut at -- i.e. phases of 8 / 16 / 16 / 8 MFMAs).  Pieces per wave and K-step:
        // A_a 2 each, B_0 2 (1), B_1 2: 8 (7); behind B_0 of step t+1 come A_0, B_1, A_1 of step t+2 = 6 pieces either way.
        static_assert(BM == 256 && (BN == 256 || BN == 192) && NW == 8 && BK == 64 && STAGES == 2 && MODE == 2 && RPI == 8, "8-phase loop: 256-row tiles, 8 waves, BK 64");
        static_assert(WGM * 2 <= 8 && (WGM == 2 || WGM == 4), "the two wave groups are row halves");
        constexpr int MH = MF / 2;                 // fragment rows per row half
        constexpr int NB0 = 2, NB1 = NF - 2;       // fragment columns of the two channel parts (32 channels | the rest)
        constexpr int PH = WTM / 2 / 8;            // 8-row pieces per (wave row, row half)
        constexpr int SLAB = WTN / 8, P1 = SLAB - 4;   // pieces per weight slab, of them in part 1
        constexpr int QB0 = (WGN * 4) / 8;         // pieces per wave of region B_0 (2 or 1)
        const int grp = wave >> 2;
        // piece ownership by region: the r-th piece of a region (in row order) belongs to wave r % 8
        int aq_voff[2][2], aq_voff2[2][2], aq_lds[2][2], bq_voff[2][2], bq_lds[2][2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int r = wave + 8 * q;
                const int j = (r / PH) * (2 * PH) + a * PH + (r % PH);
                const int row = j * RPI + srow, m = m0 + row;
                aq_voff[a][q] = (m < p.M) ? (m * p.ldx + (sslot ^ swz_of<BK>(row & 15)) * 8) * 2 : OOB_OFFSET;   // single tap: pixel = m
                aq_voff2[a][q] = (m < p.M) ? (m * p.ldx2 + (sslot ^ swz_of<BK>(row & 15)) * 8) * 2 : OOB_OFFSET; // rows of the second source
                aq_lds[a][q] = j * 1024;
            }
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int r = wave + 8 * q;
                const int j = b == 0 ? (r / 4) * SLAB + (r % 4) : (r / P1) * SLAB + 4 + (r % P1);
                const int row = j * RPI + srow;
                const int c = (sslot ^ swz_of<BK>(bfrag_reader(row % WTN))) * 8;
                const int n = n0 + row;
                bq_voff[b][q] = (n < p.N) ? (n * p.K + c) * 2 : OOB_OFFSET;
                bq_lds[b][q] = A_BYTES + j * 1024;
            }
        auto issue_A = [&](int t, int a) {
            char* sb = smem + (t & 1) * STAGE;
            if (p.k1_steps > 0 && t >= p.k1_steps) {   // (wave-uniform) K-concatenated rows: this K-step lies in the second source
#pragma unroll
                for (int q = 0; q < 2; ++q) buffer_load_lds16(p.x2, p.x2_bytes, sb + aq_lds[a][q], aq_voff2[a][q], (t - p.k1_steps) * (BK * 2));
            } else {
#pragma unroll
                for (int q = 0; q < 2; ++q) buffer_load_lds16(p.x, p.x_bytes, sb + aq_lds[a][q], aq_voff[a][q], t * (BK * 2));
            }
        };
        auto issue_B = [&](int t, int b) {
            char* sb = smem + (t & 1) * STAGE;
#pragma unroll
            for (int q = 0; q < 2; ++q)
                if (b == 1 || q < QB0) buffer_load_lds16(p.w, p.w_bytes, sb + bq_lds[b][q], bq_voff[b][q], t * (BK * 2));
        };
        // Fragment registers: one A half at a time; B0 stays resident for its two quadrants (phases 1 and 4) and B0 of the NEXT
        // K-step is read in phase 4 into a second set (TV_P8_B0PF; balances the load sections: 8 / 4 / 8 / 4 reads on the
        // 256 x 256 tile instead of 12 / 4 / 8 / 4): for that the wait that retires step t+1 sits in L3 -- behind B0 of step
        // t+1 come only A0 and B1 of step t+2 then: vmcnt(4) -- and L4 reads two barriers later.
        bf16x8 fa[2][MH], fb0[2][2][NB0], fb1[2][NB1];
        auto read_A = [&](const char* sb, auto a_c) {
            constexpr int a = decltype(a_c)::value;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int coff = ((kk * 4 + fq) ^ sw) * 16;
#pragma unroll
                for (int i = 0; i < MH; ++i) fa[kk][i] = *(const bf16x8*)(sb + a_row_off + (MH * a + i) * 16 * (BK * 2) + coff);
            }
        };
        auto read_B0 = [&](const char* sb, auto set_c) {
            constexpr int S = decltype(set_c)::value;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int coff = ((kk * 4 + fq) ^ sw) * 16;
#pragma unroll
                for (int j = 0; j < NB0; ++j) fb0[S][kk][j] = *(const bf16x8*)(sb + b_row_off + bfrag_off(j) * (BK * 2) + coff);
            }
        };
        auto read_B1 = [&](const char* sb) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int coff = ((kk * 4 + fq) ^ sw) * 16;
#pragma unroll
                for (int j = 0; j < NB1; ++j) fb1[kk][j] = *(const bf16x8*)(sb + b_row_off + bfrag_off(NB0 + j) * (BK * 2) + coff);
            }
        };
        auto close_load = [&]() {   // my reads are complete (data ready, region free for its next DMA), then the phase barrier
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        };
        auto mfma_q = [&](auto a_c, auto b_c, auto set_c) {
            constexpr int a = decltype(a_c)::value, b = decltype(b_c)::value, S = decltype(set_c)::value;
            constexpr int J0 = b ? NB0 : 0, NJ = b ? NB1 : NB0;
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int i = 0; i < MH; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        bf16x8 bv;
                        if constexpr (b == 0) bv = fb0[S][kk][j < NB0 ? j : 0];
                        else bv = fb1[kk][j < NB1 ? j : 0];
                        acc[MH * a + i][J0 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bv, fa[kk][i], acc[MH * a + i][J0 + j], 0, 0, 0);
                    }
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        };
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        // prologue: step 0 whole, of step 1 everything but B0 (the steady-state order)
        issue_A(0, 0); issue_B(0, 0); issue_B(0, 1); issue_A(0, 1);
        if (nk > 1) { issue_A(1, 0); issue_B(1, 1); issue_A(1, 1); wait_vmcnt<6>(); }
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();               // step 0 has landed for every wave
#if TV_P8_B0PF
        read_B0(smem, I0{});                        // B0 of step 0 (its region is re-staged in L1 of step 1 at the earliest)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
        if (grp == 1) __builtin_amdgcn_s_barrier(); // group 1 runs one barrier behind
        // one K-step with B0 in register set S (the next step's goes to set 1 - S)
        auto kstep = [&](int t, auto set_c) {
            constexpr int S = decltype(set_c)::value;
            using SC = std::integral_constant<int, TV_P8_B0PF ? S : 0>;
            using SN = std::integral_constant<int, TV_P8_B0PF ? 1 - S : 0>;
            const char* sb = smem + (t & 1) * STAGE;
            // phase 1: quadrant (0, 0)
#if !TV_P8_B0PF
            read_B0(sb, SC{});
            __builtin_amdgcn_sched_barrier(0);
#endif
            read_A(sb, I0{});
            __builtin_amdgcn_sched_barrier(0);
            if (t + 1 < nk) issue_B(t + 1, 0);
            close_load();
            mfma_q(I0{}, I0{}, SC{});
            // phase 2: quadrant (0, 1)
            read_B1(sb);
            __builtin_amdgcn_sched_barrier(0);
            if (t + 2 < nk) issue_A(t + 2, 0);
            close_load();
            mfma_q(I0{}, I1{}, SC{});
            // phase 3: quadrant (1, 1)
            read_A(sb, I1{});
            __builtin_amdgcn_sched_barrier(0);
            if (t + 2 < nk) issue_B(t + 2, 1);
#if TV_P8_B0PF
            if (t + 2 < nk) wait_vmcnt<4>();         // behind B0 of step t+1: A0, B1 of step t+2 -> step t+1 has landed
            else if (t + 1 < nk) wait_vmcnt<0>();
#endif
            close_load();
            mfma_q(I1{}, I1{}, SC{});
            // phase 4: quadrant (1, 0)
#if TV_P8_B0PF
            if (t + 1 < nk) read_B0(smem + ((t + 1) & 1) * STAGE, SN{});   // next step's B0: two barriers behind every wave's wait
            __builtin_amdgcn_sched_barrier(0);
            if (t + 2 < nk) issue_A(t + 2, 1);
#else
            read_B0(sb, SC{});
            __builtin_amdgcn_sched_barrier(0);
            if (t + 2 < nk) { issue_A(t + 2, 1); wait_vmcnt<6>(); }   // B0 of step t+1 and everything older has landed
            else if (t + 1 < nk) wait_vmcnt<0>();
#endif
            close_load();
            mfma_q(I1{}, I0{}, SC{});
        };
        for (int t = 0; t < nk; t += 2) {
            kstep(t, I0{});
            if (t + 1 < nk) kstep(t + 1, I1{});
        }
        if (grp == 0) __builtin_amdgcn_s_barrier();
    } else if constexpr (PIPE2) {
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
    } else if constexpr (DMA && PERSIST) {
        // one K-loop over the block's ntile column tiles (STAGES == 2, burst issue): step g stages step g + 1, whichever tile
        // that belongs to; a tile's epilogue follows its last step
        const int total = ntile * nk;
        constexpr bool full_rows = true;   // (the host sends only M % BM == 0 here: every staged row and every store is in range)
        persist_reinit();
        stage_issue(smem);
        int cur = 0, t = 0;
        bool after_epi = false;
        for (int g = 0; g < total; ++g) {
            if (after_epi) {
                // behind the prefetched pieces of this step: the previous tile's stores (NST per wave on a full tile; a ragged
                // one may have skipped some: drain)
                constexpr int NST = epi_store_count<WTM, WTN>(PFORM == EF_GELU_D || PFORM == EF_SILU_D);
                static_assert(NST <= 63, "vmcnt field");
                if (!full_rows || TV_PERSIST_DRAIN) wait_vmcnt<0>();
                else wait_vmcnt<NST>();
                after_epi = false;
            } else {
                wait_vmcnt<0>();
            }
            TV_T(0);
            __builtin_amdgcn_s_barrier();
            TV_T(1);
            if (g + 1 < total) stage_issue(smem + (cur ^ 1) * STAGE);
            compute(smem + cur * STAGE, nullptr, std::false_type{});
            TV_T(4);
            cur ^= 1;
            if (++t == nk) {   // the tile is complete
                t = 0;
                __builtin_amdgcn_sched_barrier(0);
                f32x4 bvals[NF];
                int mrow0 = m0 + wm * WTM, elane = threadIdx.x;
                asm volatile("" : "+s"(mrow0));   // per tile: the epilogue's row / offset / lane arithmetic must not be hoisted
                asm volatile("" : "+v"(elane));   // out of the tile loop (it would live in ~60 registers across the main loop)
                elane &= 63;
                load_bias<WTN>(p, elane, n0 + wn * WTN, bvals);
                // ONE register form per instantiation (PFORM): the run-time switch over all of them inside the tile loop cost the
                // EPI 0 kernel 251 registers and 104 SGPRs (lane-spilled in the K-loop): 3-9 % SLOWER than one tile per block
                epilogue_direct<WTM, WTN, PFORM>(p, acc, bvals, elane, n0 + wn * WTN, [&](int r) { return mrow0 + r; });
                __builtin_amdgcn_sched_barrier(0);
                persist_reinit();   // (unconditional: the old values are dead behind every epilogue, nothing to keep or spill)
#pragma unroll
                for (int i = 0; i < MF; ++i)
#pragma unroll
                    for (int j = 0; j < NF; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
                walk_i += p.nchunks;
                {
                    const int r = walk_i / p.tiles_n;
                    n0 = (walk_i - r * p.tiles_n) * BN;
                    m0 = (r * 8 + walk_x) * BM;
                }
                after_epi = true;
            }
        }
        TV_PROBE_DUMP(wave, lane);
        return;
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
#ifdef TV_ABL_NO_DMA
            stage_advance();
#else
            stage_issue(smem + nxt * STAGE);
#endif
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
    epilogue<WTM, WTN, EPI, MODE == 2 && (BM * BN >= 128 * 128)>(p, acc, bvals, smem, wave, lane, n0 + wn * WTN, [&](int r) { return mrow0 + r; });
    TV_T(6);
    TV_PROBE_DUMP(wave, lane);
}

bool g_use_dma = true;
int g_nt_threshold_mb = 64;  // epilogue stores are non-temporal for outputs of at least this many MiB (-1: never; tv_set_igemm_nt_threshold): bench 148.7 (never) / 149.4 (always) / 149.6 (64) / 149.2 (192) / 149.0 (512) images/s on one box
int g_loop8 = 1;       // eight-phase ping-pong loop for single-tap layers on the 256 x 256 tile (tv_set_igemm_persist(-1 / -2): off / on)
int g_persist = 0;     // persistent tile walk: 0 off (default: measured -0.3 % over the linear layers, profiles/r03_kernel_experiments.txt item 12), 1 heuristic walk length, n > 1 forced walk of n tiles
int g_cfg_stages = 0;  // 0 = heuristic, else 2 / 3 / 4
int g_cfg_bk = 0;      // 0 = largest that divides c_in, else 32 / 64
int g_addr_mode = 0;   // 0 = buffer DMA when the tensors are < 2 GiB, 1 = force 64-bit global DMA
bool g_use_halo = true;  // 3x3 stride-1 convolutions through conv3x3_halo_kernel when the shape qualifies





// register form of an EPI 0 launch (see epilogue<>): the common elementwise combinations; everything else -> LDS loop
int epilogue_form(const IgemmArgs& a) {
    if (!g_epi_modes) return EF_GENERIC;
    if (a.shuffle) {   // shuffled stores: the forms the model uses (a 64-channel pair of blocks must stay inside one phase)
        if (a.rope || ((a.N >> 2) % 64) != 0 || (a.N & 3)) return EF_GENERIC;
        if (a.shuffle == 1) {
            if (a.aux) return (a.aux_act == TV_ACT_DERIV && !a.res && !a.pre && a.act == TV_ACT_NONE) ? EF_DERIV_S1 : EF_GENERIC;
            if (a.pre || a.act != TV_ACT_NONE) return EF_GENERIC;
            return a.res ? EF_RES_S1 : EF_PLAIN_S1;
        }
        if (a.aux || a.res) return EF_GENERIC;
        if (a.pre) return (a.pre_deriv && a.act == TV_ACT_SILU) ? EF_SILU_D_S2 : EF_GENERIC;
        return a.act == TV_ACT_NONE ? EF_PLAIN_S2 : EF_GENERIC;
    }
    if (a.rope) return (!a.aux && !a.res && !a.pre && a.act == TV_ACT_NONE && a.rope_cols % 32 == 0) ? EF_ROPE : EF_GENERIC;
    if (a.aux) {
        if (!a.res || a.pre || a.act != TV_ACT_NONE) return EF_GENERIC;
        return a.aux_act == TV_ACT_DERIV ? EF_RES_DERIV : (a.aux_act == TV_ACT_ADD ? EF_RES2 : EF_GENERIC);
    }
    if (a.res) return EF_GENERIC;
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
        if (a_in.x2) {   // two-source rows exist in the eight-phase loop only
            constexpr bool P8OK = BM == 256 && ((BN == 256 && WGM == 2 && WGN == 4) || (BN == 192 && WGM == 4 && WGN == 2)) && BK == 64 && STAGES == 2 && MODE == 2;
            if (!P8OK || !g_loop8) return -2;
        }
        const int tiles_m = (a_in.M + BM - 1) / BM;
        IgemmArgs a = a_in;
        a.tiles_m = tiles_m;
        a.xcd_order = (g_xcd_order && a.tiles_n > 1) ? 1 : 0;
        a.tpb = 1;
        a.nchunks = a.tiles_n;
        a.sup_r = 1;
        a.sup_c = a.tiles_n;
        constexpr bool POK = igemm_persist_ok<BM, BN, WGM, WGN, BK, STAGES, MODE>() && BM == 256 && BN == 256 && BK == 64;
        dim3 grid((unsigned)(a.xcd_order ? 8 * a.tiles_n * ((tiles_m + 7) / 8) : tiles_m * a.tiles_n)), block(WGM * WGN * 64);
        if (a.xcd_order && g_supertile != 1 && (WGM * WGN == 8)) {
            // one block per CU (8-wave tiles): 32 block slots per XCD.  Measured (tools/probes/ab_supertile.py, same process,
            // interleaved; profiles/r04_kernel_experiments.txt): with 16 or more column tiles (N >= 4096: the 1536 -> 6144 and
            // 1536 -> 4608 layers and the data gradients of their transposes) the 4 x 8 super-tile is 3-5 % faster than the
            // row-tile-major order; with 12 or fewer column tiles the row-tile-major order already puts 3-5 row tiles of an XCD
            // side by side and is as fast or faster (ragged super-tiles cost up to 20 %), so it stays
            int br = 1, bc = a.tiles_n;
            if (g_supertile > 1) {          // (tuning hook: forced shape r * 100 + c)
                br = g_supertile / 100;
                bc = g_supertile % 100;
            } else if (a.tiles_n >= 16 && tiles_m >= 4) {
                br = 4;
                bc = 8;
            }
            if (br < 1) br = 1;
            if (bc < 1) bc = 1;
            if (bc > a.tiles_n) bc = a.tiles_n;
            if (!(br == 1 && bc == a.tiles_n)) {
                a.xcd_order = 2;
                a.sup_r = br;
                a.sup_c = bc;
                const int nsup = ((tiles_m + br - 1) / br) * ((a.tiles_n + bc - 1) / bc);
                grid = dim3((unsigned)(8 * br * bc * ((nsup + 7) / 8)));
            }
        }
        if constexpr (POK) {
            // tile walk: W tiles per block, P blocks per XCD label (tpb = W, nchunks = P).  The LDS epilogue parks the tile in the
            // stage buffers a prefetch would be landing in, ragged row tiles would stage rows past the tensor: one tile per block
            const int em = epilogue_mode(a);
            const bool reg_epi = em != 0 || a.form == EF_PLAIN || a.form == EF_GELU_D || a.form == EF_RES_DERIV || a.form == EF_ROPE;
            const bool single_tap = a.kh == 1 && a.kw == 1 && a.stride == 1 && a.pad == 0 && a.up_shift == 0 && a.dil_mask == 0;
            if (g_persist != 0 && !a.x2 && reg_epi && single_tap && a.M % BM == 0 && a.N % BN == 0) {   // (two-source rows: eight-phase loop only)
                const int nlist = ((tiles_m + 7) / 8) * a.tiles_n;      // longest per-XCD list
                int W = (nlist + 31) / 32;                              // 32 block slots per XCD
                if (g_persist > 1) W = g_persist < nlist ? g_persist : nlist;   // (tuning hook: forced walk length)
                if (W >= 2) {
                    a.tpb = W;
                    a.nchunks = (nlist + W - 1) / W;
                    grid = dim3((unsigned)(8 * a.nchunks));
                }
            }
        }
        auto go = [&](auto epi) {
            constexpr int EPI_MODE = decltype(epi)::value;
            static TvPerDeviceOnce attr_once;
            if (attr_once.first()) {  // > 64 KiB of dynamic LDS needs the opt-in
                (void)hipFuncSetAttribute((const void*)igemm_nt_kernel<BM, BN, WGM, WGN, BK, STAGES, MODE, EPI_MODE>,
                                          hipFuncAttributeMaxDynamicSharedMemorySize, BYTES);
            }
            if constexpr (POK) {
                if (a.tpb > 1) {
                    auto gop = [&](auto form_c) {
                        constexpr int F = decltype(form_c)::value;
                        static TvPerDeviceOnce attr_once_p;
                        if (attr_once_p.first())
                            (void)hipFuncSetAttribute((const void*)igemm_nt_kernel<BM, BN, WGM, WGN, BK, STAGES, MODE, EPI_MODE, true, F>,
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, BYTES);
                        hipLaunchKernelGGL((igemm_nt_kernel<BM, BN, WGM, WGN, BK, STAGES, MODE, EPI_MODE, true, F>), grid, block, BYTES, s, a);
                    };
                    if constexpr (EPI_MODE == 1) gop(std::integral_constant<int, EF_RES>{});
                    else if constexpr (EPI_MODE == 2) gop(std::integral_constant<int, EF_DERIV>{});
                    else if (a.form == EF_GELU_D) gop(std::integral_constant<int, EF_GELU_D>{});
                    else if (a.form == EF_RES_DERIV) gop(std::integral_constant<int, EF_RES_DERIV>{});
                    else if (a.form == EF_ROPE) gop(std::integral_constant<int, EF_ROPE>{});
                    else gop(std::integral_constant<int, EF_PLAIN>{});
                    return;
                }
            }
            if constexpr (BM == 256 && ((BN == 256 && WGM == 2 && WGN == 4) || (BN == 192 && WGM == 4 && WGN == 2)) && BK == 64 && STAGES == 2 && MODE == 2) {
                const bool single_tap8 = a.kh == 1 && a.kw == 1 && a.stride == 1 && a.pad == 0 && a.up_shift == 0 && a.dil_mask == 0;
                if (g_loop8 && single_tap8) {   // 1x1 / linear layers: the eight-phase ping-pong loop
                    static TvPerDeviceOnce attr_once_8;
                    if (attr_once_8.first())
                        (void)hipFuncSetAttribute((const void*)igemm_nt_kernel<BM, BN, WGM, WGN, BK, STAGES, MODE, EPI_MODE, false, 0, true>,
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, BYTES);
                    hipLaunchKernelGGL((igemm_nt_kernel<BM, BN, WGM, WGN, BK, STAGES, MODE, EPI_MODE, false, 0, true>), grid, block, BYTES, s, a);
                    return;
                }
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

extern "C" int tv_set_igemm_nt_threshold(int mib) {   // tuning hook (A/B): see g_nt_threshold_mb
    g_nt_threshold_mb = mib;
    return 0;
}

extern "C" int tv_set_igemm_persist(int on) {   // tuning hook (A/B timing, tests): see g_persist; -1 / -2: eight-phase loop off / on
    if (on == -1) g_loop8 = 0;
    else if (on == -2) g_loop8 = 1;
    else g_persist = on;
    return 0;
}

extern "C" int tv_set_igemm_supertile(int v) {   // tuning hook (A/B timing, tests): see g_supertile
    g_supertile = v;
    return 0;
}

extern "C" int tv_set_igemm_epilogue(int on) {   // 0: generic (run-time) epilogue everywhere, for A/B timing and tests
    g_epi_modes = on != 0;
    return 0;
}

#ifdef TV_PROBE
extern "C" int tv_set_igemm_probe(void* dev_buf) {   // 16 blocks x 8 waves x 8 counters (u64), or null
    const bool ok = hipMemcpyToSymbol(HIP_SYMBOL(g_probe_dev), &dev_buf, sizeof(void*)) == hipSuccess;
    return (tvi::set_halo_probe(dev_buf) == 0 && ok) ? 0 : 1;
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
struct Cat2Spec {      // second source of K-concatenated rows: columns k1 .. c_in - 1 come from x2 (row pitch ldx2)
    const void* x2;
    int k1, ldx2;
};
static int igemm_nt_impl(const tv_conv_desc* d, const void* x, const void* w, const float* bias, const void* residual,
                         void* pre_act, void* out, const void* aux, int aux_act, void* stream, RopeSpec rope = RopeSpec{nullptr, 0, 0},
                         Cat2Spec cat = Cat2Spec{nullptr, 0, 0});

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
    TV_CHECK_ARG(aux_pre_act && aux_act >= 0 && aux_act <= TV_ACT_ADD && d && d->act == TV_ACT_NONE,
                 "tv_igemm_nt_actgrad: needs the saved pre-activation, a valid activation id and desc.act == NONE");
    return igemm_nt_impl(d, x, w, nullptr, residual, nullptr, out, aux_pre_act, aux_act, stream);
}

extern "C" int tv_igemm_nt_cat2(const tv_conv_desc* d, const void* x, const void* x2, int k1, int ldx2, const void* w, const float* bias,
                               const void* residual, const void* aux, int aux_act, void* out, void* stream) {
    TV_CHECK_ARG(x2 && d && d->act == TV_ACT_NONE && (!aux || (aux_act >= 0 && aux_act <= TV_ACT_ADD && !bias)),
                 "tv_igemm_nt_cat2: needs the second source, desc.act == NONE, and no bias together with aux");
    return igemm_nt_impl(d, x, w, bias, residual, nullptr, out, aux, aux ? aux_act : TV_ACT_NONE, stream, RopeSpec{nullptr, 0, 0}, Cat2Spec{x2, k1, ldx2});
}

static int igemm_nt_impl(const tv_conv_desc* d, const void* x, const void* w, const float* bias, const void* residual,
                         void* pre_act, void* out, const void* aux, int aux_act, void* stream, RopeSpec rope, Cat2Spec cat) {
    TV_CHECK_ARG(d && x && w && out, "tv_igemm_nt: null pointer");
    if (cat.x2) {
        TV_CHECK_ARG(d->kh == 1 && d->kw == 1 && d->stride == 1 && d->pad == 0 && d->up_shift == 0 && d->dil_mask == 0 && d->store_shuffle == 0 && !rope.tab,
                     "tv_igemm_nt_cat2: single-tap layers only");
        TV_CHECK_ARG(cat.k1 > 0 && cat.k1 % 64 == 0 && d->c_in > cat.k1 && (d->c_in - cat.k1) % 64 == 0 && d->ldx >= cat.k1 && d->ldx % 8 == 0 &&
                     cat.ldx2 >= d->c_in - cat.k1 && cat.ldx2 % 8 == 0, "tv_igemm_nt_cat2: k1=%d c_in=%d ldx=%d ldx2=%d", cat.k1, d->c_in, d->ldx, cat.ldx2);
    }
    TV_CHECK_ARG(d->c_in > 0 && d->c_in % 32 == 0, "tv_igemm_nt: c_in=%d must be a multiple of 32", d->c_in);
    TV_CHECK_ARG(d->c_out > 0 && d->c_out % 8 == 0, "tv_igemm_nt: c_out=%d must be a multiple of 8", d->c_out);
    TV_CHECK_ARG(cat.x2 || (d->ldx >= d->c_in && d->ldx % 8 == 0), "tv_igemm_nt: ldx=%d (c_in=%d) must be >= c_in and a multiple of 8", d->ldx, d->c_in);
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
    {   // x / d as umulhi(x, m) >> s for 0 <= x < 2^31 (round-up method with 31 + ceil(log2 d) bits: exact on that range)
        auto fd = [](int d, unsigned& m, int& sh) {
            if (d <= 1) { m = 0; sh = 0; return; }
            int L = 0;
            while ((1ll << L) < d) ++L;
            m = (unsigned)(((1ull << (31 + L)) + (unsigned long long)d - 1) / (unsigned long long)d);
            sh = L - 1;
        };
        fd(d->h_out * d->w_out, a.dv_hw_m, a.dv_hw_s);
        fd(d->w_out, a.dv_w_m, a.dv_w_s);
        fd(a.N >> 2, a.dv_cq_m, a.dv_cq_s);
    }
    {   // buffer-descriptor extents (0 = too large for 32-bit offsets -> global-address DMA)
        const long long rows = (long long)d->batch * d->h_in * d->w_in;
        const long long xb = (rows - 1) * d->ldx * 2 + (long long)(cat.x2 ? cat.k1 : d->c_in) * 2;
        const long long wb = (long long)a.N * a.K * 2;
        a.x_bytes = xb < (1ll << 31) ? (unsigned)xb : 0u;
        a.w_bytes = wb < (1ll << 31) ? (unsigned)wb : 0u;
        a.x2 = (const bf16*)cat.x2;
        a.ldx2 = cat.x2 ? cat.ldx2 : d->ldx;
        a.k1_steps = cat.x2 ? cat.k1 / 64 : 0;
        const long long x2b = cat.x2 ? (rows - 1) * cat.ldx2 * 2 + (long long)(d->c_in - cat.k1) * 2 : 0;
        a.x2_bytes = (unsigned)x2b;
        if (cat.x2 && (a.x_bytes == 0 || a.w_bytes == 0 || x2b >= (1ll << 31))) return TV_ERR_UNSUPPORTED;
    }
    {   // extent of the output tensor (shuffled stores: 4 M pixels of N / 4 channels; the polyphase grid is smaller still)
        const long long ob = d->store_shuffle ? ((4ll * M - 1) * d->ldo + (a.N >> 2)) * 2 : ((M - 1) * (long long)d->ldo + a.N) * 2;
        a.out_bytes = ob < (1ll << 31) ? (unsigned)ob : 0u;
        a.store_nt = (g_nt_threshold_mb >= 0 && ob >= ((long long)g_nt_threshold_mb << 20)) ? 1 : 0;
        if (a.out_bytes == 0) a.form = EF_GENERIC;   // (the register forms carry 32-bit offsets; epilogue_mode() checks the same)
    }
    hipStream_t s = (hipStream_t)stream;
    const bool bk64 = (d->c_in % 64 == 0) && g_cfg_bk != 32;
    int rc = (g_use_dma && g_use_halo && g_addr_mode != 1 && bk64) ? launch_halo(a, s) : -1;
    if (rc != 0) rc = bk64 ? launch_bk<64>(a, s) : launch_bk<32>(a, s);
    if (rc != 0 && cat.x2) return TV_ERR_UNSUPPORTED;     // (the tile this shape takes has no two-source loop: the caller concatenates)
    if (rc != 0) {
        tv_set_error("tv_igemm_nt: no kernel for this configuration");
        return TV_ERR_ARG;
    }
    TV_CHECK_LAUNCH("tv_igemm_nt");
    return TV_OK;
}

