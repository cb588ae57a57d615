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
// global->LDS with 16-byte LDS-DMA (global_load_lds_dwordx4) into a ring of STAGES buffers.
// The loads of K-steps t+1 .. t+STAGES-1 are in flight while K-step t is multiplied: one raw
// s_barrier per K-step and a COUNTED s_waitcnt vmcnt(N) (never 0 in steady state when STAGES > 2),
// so the DMA spans barriers (cdna_hip_programming.md section 5, "Pipelining across barriers",
// T3/T4).  The LDS image is lane-linear (DMA constraint); bank conflicts are removed by
// XOR-swizzling the 16-byte chunk index on the SOURCE address and on the ds_read_b128 address
// (rule 21).
//
// The MFMA is issued as D' = W_frag x A_frag^T so that every lane ends up with 4*NF
// consecutive output channels of one pixel: bias / activation / residual / store work on
// contiguous channel runs.
#include "common.h"

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
    int tiles_n;
    int shuffle;
    int act;
    int aux_act;
    int hw_shift, w_shift;  // log2 of h_out*w_out / w_out when both are powers of two, else -1
    unsigned x_bytes, w_bytes;  // MODE 2: extents of the two buffers (< 2 GiB)
};

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int BK>
__device__ __forceinline__ int swz_of(int i) {  // i: row index inside a 16-row fragment
    if constexpr (BK == 64)
        return (i >> 1) & 7;
    else
        return (0x78 >> (2 * ((i >> 2) & 3))) & 3;
}

// MODE 0: register-staged loads + ds_write (bring-up / debugging)
// MODE 1: global_load_lds with 64-bit per-lane addresses (tensors >= 2 GiB)
// MODE 2: buffer_load ... lds: SGPR descriptor + 32-bit per-lane offset fixed per tap + scalar K offset; padding
//         = out-of-range offset (the hardware writes zeros) -- no vector ALU work per DMA in the steady state
template <int BM, int BN, int WGM, int WGN, int BK, int STAGES, int MODE>
__global__ __launch_bounds__(WGM* WGN * 64) void igemm_nt_kernel(const IgemmArgs p) {
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

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave % WGN;
    const int tile_n = blockIdx.x % p.tiles_n;
    const int tile_m = blockIdx.x / p.tiles_n;
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
        const int fi = ((rl / (4 * NF)) << 2) | (row & 3);
        const int c = (sslot ^ swz_of<BK>(fi)) * 8;
        const int n = n0 + row;
        b_ok[it] = (j < B_INSTR) && (n < p.N);
        b_src[it] = p.w + (size_t)(b_ok[it] ? n : 0) * p.K + c;
        b_voff[it] = b_ok[it] ? (n * p.K + c) * 2 : OOB_OFFSET;
    }

    // running state of the "next K-step to stage"
    const int cch = p.c_in / BK;  // channel chunks per tap
    const int nk = p.kh * p.kw * cch;
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

    // issue the global loads of K-step st_t (DMA: straight into LDS stage `sbase`)
    auto stage_issue = [&](char* sbase) {
        const int koff = st_ch * BK;
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            const int j = it * NW + wave;
            if (A_INSTR % NW != 0 && j >= A_INSTR) {
                if constexpr (PADDED && BUF) buffer_load_lds16(p.x, p.x_bytes, smem + STAGES * STAGE, OOB_OFFSET, 0);
                else if constexpr (PADDED) __builtin_amdgcn_global_load_lds(TV_GLB(p.zeros + lane * 16), TV_LDS(smem + STAGES * STAGE), 16, 0, 0);
                break;
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
        }
        const int kb = st_t * BK;
#pragma unroll
        for (int it = 0; it < B_IT; ++it) {
            const int j = it * NW + wave;
            if (B_INSTR % NW != 0 && j >= B_INSTR) {
                if constexpr (PADDED && BUF) buffer_load_lds16(p.w, p.w_bytes, smem + STAGES * STAGE, OOB_OFFSET, 0);
                else if constexpr (PADDED) __builtin_amdgcn_global_load_lds(TV_GLB(p.zeros + lane * 16), TV_LDS(smem + STAGES * STAGE), 16, 0, 0);
                break;
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
        // advance to the following K-step
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
    const int b_row_off = A_BYTES + (wn * WTN + (fi >> 2) * (4 * NF) + (fi & 3)) * (BK * 2);

    f32x4 acc[MF][NF];
#pragma unroll
    for (int i = 0; i < MF; ++i)
#pragma unroll
        for (int j = 0; j < NF; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // Fragment reads are software-pipelined against the MFMAs: where the registers allow (<= 80 fragment VGPRs) ALL reads
    // of the K-step (both 32-deep halves) are issued up front, so the second half's ds_read_b128s fly under the first
    // half's MFMAs instead of exposing their latency a second time; the 256x256 tile keeps one half in flight at a time.
    constexpr bool PIPE_ALL = (MF + NF) * 4 * (BK / 32) <= 80;
    auto compute = [&](const char* sbase) {
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
                for (int j = 0; j < NF; ++j) bfr[kk][j] = *(const bf16x8*)(sbase + b_row_off + j * 4 * (BK * 2) + coff);
#endif
            }
            __builtin_amdgcn_sched_barrier(0);   // keep every read ahead of the MFMAs (the scheduler would sink them again)
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
                    }
        } else {
#pragma unroll
            for (int kk = 0; kk < BK / 32; ++kk) {
                const int coff = ((kk * 4 + fq) ^ sw) * 16;
                bf16x8 af[MF], bfr[NF];
#pragma unroll
                for (int i = 0; i < MF; ++i) af[i] = *(const bf16x8*)(sbase + a_row_off + i * 16 * (BK * 2) + coff);
#pragma unroll
                for (int j = 0; j < NF; ++j) bfr[j] = *(const bf16x8*)(sbase + b_row_off + j * 4 * (BK * 2) + coff);
#pragma unroll
                for (int i = 0; i < MF; ++i)
#pragma unroll
                    for (int j = 0; j < NF; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
            }
        }
    };

    // ---- main loop -------------------------------------------------------------------------
    if constexpr (DMA) {
        constexpr int LA = STAGES - 1;  // K-steps of lookahead
#pragma unroll
        for (int s = 0; s < LA; ++s)
            if (s < nk) stage_issue(smem + s * STAGE);
        int cur = 0, nxt = LA % STAGES;
        for (int t = 0; t < nk; ++t) {
            // K-step t must have landed; up to min(LA-1, nk-1-t) younger K-steps may stay in flight
            const int younger = min(LA - 1, nk - 1 - t);
            if (LA >= 3 && younger == 2) wait_vmcnt<2 * NI>();
            else if (LA >= 2 && younger == 1) wait_vmcnt<NI>();
            else wait_vmcnt<0>();
#ifndef TV_ABL_NO_BARRIER
            __builtin_amdgcn_s_barrier();
#endif
#ifndef TV_ABL_NO_DMA
            if (t + LA < nk) stage_issue(smem + nxt * STAGE);
#endif
            compute(smem + cur * STAGE);
            cur = (cur + 1 == STAGES) ? 0 : cur + 1;
            nxt = (nxt + 1 == STAGES) ? 0 : nxt + 1;
        }
    } else {
        stage_issue(smem);
        stage_write(smem);
        for (int t = 0; t < nk; ++t) {
            __syncthreads();
            if (t + 1 < nk) stage_issue(nullptr);
            compute(smem + (t & 1) * STAGE);
            if (t + 1 < nk) stage_write(smem + ((t + 1) & 1) * STAGE);
        }
    }

    // ---- epilogue ---------------------------------------------------------------------------------
    // The accumulator layout scatters a row over lanes (8-byte pieces); stored directly the tile costs ~25 % of a
    // K=1728 convolution (measured: K=64 launch 0.44 ms of 1.56 ms).  Instead every wave parks its tile (+bias,
    // bf16) in its own slice of the now idle stage buffers and streams it out row by row, 16 bytes per lane:
    // pre-activation store, activation, residual add and the output store are all full-line accesses.
    constexpr int ERS = WTN * 2 + 16;          // LDS row stride of the parked tile (16 B pad: bank spread)
    constexpr int EB = WTM * ERS;              // bytes per wave
    __syncthreads();                           // every wave is done reading the stage buffers
    char* ebuf = smem + wave * EB;
#pragma unroll
    for (int i = 0; i < MF; ++i) {
#pragma unroll
        for (int j = 0; j < NF; ++j) {
            const int nl = fq * (4 * NF) + j * 4;
            const int n = n0 + wn * WTN + nl;
            f32x4 v = acc[i][j];
            if (p.bias && n < p.N) {
                const f32x4 bv = *(const f32x4*)(p.bias + n);
                v += bv;
            }
            bf16x4 pv = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
            *(bf16x4*)(ebuf + (i * 16 + fi) * ERS + nl * 2) = pv;
        }
    }
    // (same wave writes and reads: LDS executes a wave's accesses in order, no barrier needed)
    constexpr int CPW = WTN / 8;               // 16-byte chunks per tile row
    const int hw = p.h_out * p.w_out;
    const int cq = p.N >> 2;
#pragma unroll 2
    for (int idx = lane; idx < WTM * CPW; idx += 64) {
        const int r = idx / CPW, c8 = idx - r * CPW;
        const int m = m0 + wm * WTM + r;
        const int n = n0 + wn * WTN + c8 * 8;
        if (m >= p.M || n >= p.N) continue;
        size_t off;
        if (p.shuffle) {
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
        bf16x8 z = *(const bf16x8*)(ebuf + r * ERS + c8 * 16);
        if (p.pre) *(bf16x8*)(p.pre + off) = z;
        if (p.aux) {  // gradient w.r.t. a pre-activation: (acc + residual gradient) * act'(saved pre-activation)
            const bf16x8 av = *(const bf16x8*)(p.aux + off);
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (float)z[e];
            if (p.res) {
                const bf16x8 rv = *(const bf16x8*)(p.res + off);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += (float)rv[e];
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) z[e] = (bf16)(v[e] * tv_act_grad_rt(p.aux_act, (float)av[e]));
        } else if (p.act != TV_ACT_NONE || p.res) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = tv_act_rt(p.act, (float)z[e]);
            if (p.res) {
                const bf16x8 rv = *(const bf16x8*)(p.res + off);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += (float)rv[e];
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) z[e] = (bf16)v[e];
        }
        *(bf16x8*)(p.out + off) = z;
    }
}

bool g_use_dma = true;
int g_cfg_bm = 0;      // 0 = heuristic, else 128 / 256
int g_cfg_stages = 0;  // 0 = heuristic, else 2 / 3 / 4
int g_cfg_bk = 0;      // 0 = largest that divides c_in, else 32 / 64
int g_cfg_bn = 0;      // 0 = heuristic, 256 = 256-wide N tiles whenever c_out % 256 == 0, 128 = never
int g_addr_mode = 0;   // 0 = buffer DMA when the tensors are < 2 GiB, 1 = force 64-bit global DMA

constexpr int LDS_MAX = 160 * 1024;

template <int BM, int BN, int WGM, int WGN, int BK, int STAGES, int MODE>
int launch_one(const IgemmArgs& a, hipStream_t s) {
    constexpr int RING = STAGES * (BM + BN) * BK * 2 + (STAGES > 2 ? 1024 : 0);  // + dummy-DMA scratch slot
    constexpr int EPI = (WGM * WGN) * (BM / WGM) * ((BN / WGN) * 2 + 16);         // parked output tile (epilogue)
    constexpr int BYTES = RING > EPI ? RING : EPI;
    if constexpr (BYTES > LDS_MAX) {
        return -1;
    } else {
        const int tiles_m = (a.M + BM - 1) / BM;
        dim3 grid((unsigned)(tiles_m * a.tiles_n)), block(WGM * WGN * 64);
        static bool attr_done = false;
        if (!attr_done) {  // > 64 KiB of dynamic LDS needs the opt-in
            (void)hipFuncSetAttribute((const void*)igemm_nt_kernel<BM, BN, WGM, WGN, BK, STAGES, MODE>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, BYTES);
            attr_done = true;
        }
        hipLaunchKernelGGL((igemm_nt_kernel<BM, BN, WGM, WGN, BK, STAGES, MODE>), grid, block, BYTES, s, a);
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
    const long long m256 = (a.M + 255) / 256;
    // Tile heuristic (tools/gemm_sweep.py, mb 64): 256x256 tiles (8 waves, 128x64 per wave: half the LDS bytes per
    // MFMA) win 5-15 % where c_out % 256 == 0 and the grid still has >= 512 blocks; 256-row tiles help the N=192
    // convolutions a little when M is huge; everything else runs 128-row tiles at 2-3 blocks per CU.
    if (N % 192 == 0 && N % 128 != 0) {
        a.tiles_n = N / 192;
        const int bm = g_cfg_bm ? g_cfg_bm : (m256 * a.tiles_n >= 1024 ? 256 : 128);
        return launch_big<192, BK, MODE>(a, bm, stages, s);
    }
    const bool want256 = g_cfg_bn ? (g_cfg_bn == 256) : (g_cfg_bm == 0 && m256 * (N / 256) >= 512);
    if (want256 && N % 256 == 0) {
        a.tiles_n = N / 256;
        return launch_big<256, BK, MODE>(a, g_cfg_bm ? g_cfg_bm : 256, stages, s);
    }
    const int bm = g_cfg_bm ? g_cfg_bm : 128;
    if (N > 64) {
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

// tuning hook (tools/gemm_sweep.py): 0 restores the built-in heuristic
extern "C" int tv_set_igemm_config(int bm, int bn, int stages, int bk) {
    g_cfg_bm = bm;
    g_cfg_bn = bn;
    g_cfg_stages = stages;
    g_cfg_bk = bk;
    return 0;
}

static int igemm_nt_impl(const tv_conv_desc* d, const void* x, const void* w, const float* bias, const void* residual,
                         void* pre_act, void* out, const void* aux, int aux_act, void* stream);

extern "C" int tv_igemm_nt(const tv_conv_desc* d, const void* x, const void* w, const float* bias,
                           const void* residual, void* pre_act, void* out, void* stream) {
    return igemm_nt_impl(d, x, w, bias, residual, pre_act, out, nullptr, TV_ACT_NONE, stream);
}

extern "C" int tv_igemm_nt_actgrad(const tv_conv_desc* d, const void* x, const void* w, const void* residual,
                                   const void* aux_pre_act, int aux_act, void* out, void* stream) {
    TV_CHECK_ARG(aux_pre_act && aux_act >= 0 && aux_act <= 2 && d && d->act == TV_ACT_NONE,
                 "tv_igemm_nt_actgrad: needs the saved pre-activation, a valid activation id and desc.act == NONE");
    return igemm_nt_impl(d, x, w, nullptr, residual, nullptr, out, aux_pre_act, aux_act, stream);
}

static int igemm_nt_impl(const tv_conv_desc* d, const void* x, const void* w, const float* bias, const void* residual,
                         void* pre_act, void* out, const void* aux, int aux_act, void* stream) {
    TV_CHECK_ARG(d && x && w && out, "tv_igemm_nt: null pointer");
    TV_CHECK_ARG(d->c_in > 0 && d->c_in % 32 == 0, "tv_igemm_nt: c_in=%d must be a multiple of 32", d->c_in);
    TV_CHECK_ARG(d->c_out > 0 && d->c_out % 8 == 0, "tv_igemm_nt: c_out=%d must be a multiple of 8", d->c_out);
    TV_CHECK_ARG(d->ldx >= d->c_in && d->ldx % 8 == 0, "tv_igemm_nt: ldx=%d (c_in=%d) must be >= c_in and a multiple of 8", d->ldx, d->c_in);
    TV_CHECK_ARG(d->ldo % 8 == 0, "tv_igemm_nt: ldo=%d must be a multiple of 8", d->ldo);
    TV_CHECK_ARG(d->batch > 0 && d->h_in > 0 && d->w_in > 0 && d->h_out > 0 && d->w_out > 0, "tv_igemm_nt: empty geometry");
    TV_CHECK_ARG(d->kh > 0 && d->kw > 0 && d->stride > 0 && d->pad >= 0, "tv_igemm_nt: bad taps");
    TV_CHECK_ARG((d->up_shift | 1) == 1 && (d->dil_mask | 1) == 1, "tv_igemm_nt: up_shift/dil_mask must be 0 or 1");
    TV_CHECK_ARG(d->act >= 0 && d->act <= 2, "tv_igemm_nt: unknown activation %d", d->act);
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
    a.act = d->act;
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
    int rc = bk64 ? launch_bk<64>(a, s) : launch_bk<32>(a, s);
    if (rc != 0) {
        tv_set_error("tv_igemm_nt: no kernel for this configuration");
        return TV_ERR_ARG;
    }
    TV_CHECK_LAUNCH("tv_igemm_nt");
    return TV_OK;
}
