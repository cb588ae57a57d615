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
    int tiles_ci, chunk_px;
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

__device__ __forceinline__ bf16x4 lds_tr16(const char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(p));
}

template <int TG, int TX>
__global__ __launch_bounds__(256) void wgrad_tn_kernel(const WgradArgs p) {
    constexpr int NW = 4;
    constexpr int CG = TG / 8, CX = TX / 8;               // chunks per tile row
    constexpr int G_INSTR = TG / 16, X_INSTR = TX / 16;   // 1 KiB pieces per 32-pixel tile
    constexpr int G_IT = G_INSTR / NW, X_IT = X_INSTR / NW;
    constexpr int G_BYTES = 32 * TG * 2, X_BYTES = 32 * TX * 2, STAGE = G_BYTES + X_BYTES;
    constexpr int WTG = TG / 2, WTX = TX / 2, MF = WTG / 16, NF = WTX / 16;
    static_assert(G_INSTR % NW == 0 && X_INSTR % NW == 0, "tile widths must be multiples of 64");

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    const int taps = p.kh * p.kw;
    int bx = blockIdx.x;
    const int ci_tile = bx % p.tiles_ci;
    bx /= p.tiles_ci;
    const int tap = bx % taps;
    const int co_tile = bx / taps;
    const int ky = tap / p.kw, kx = tap - ky * p.kw;
    const int co0 = co_tile * TG, ci0 = ci_tile * TX;

    const int p_begin = blockIdx.y * p.chunk_px;
    const int p_end = min(p.M, p_begin + p.chunk_px);
    const int nsteps = (p_end - p_begin + 31) >> 5;
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

    auto stage_issue = [&](int step, char* sbase) {
        const int pz = p_begin + step * 32;
#pragma unroll
        for (int it = 0; it < G_IT; ++it) {
            const int px = pz + g_row[it];
            const bool ok = g_cok[it] && px < p_end;
            const void* src = ok ? (const void*)(p.gy + (size_t)px * p.ldo + g_col[it])
                                 : (const void*)(p.zeros + lane * 16);
            __builtin_amdgcn_global_load_lds(TV_GLB(src), TV_LDS(sbase + (it * NW + wave) * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int it = 0; it < X_IT; ++it) {
            const int px = pz + x_row[it];
            bool ok = x_cok[it] && px < p_end;
            const int b = px / hw;
            const int r = px - b * hw;
            const int oy = r / p.w_out;
            const int ox = r - oy * p.w_out;
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
    f32x4 accb[MF];
#pragma unroll
    for (int i = 0; i < MF; ++i) {
        accb[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < NF; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const bool do_bias = (p.dbias != nullptr) && tap == 0 && ci_tile == 0 && wn == 0;
    const bf16 one = (bf16)1.0f;
    const bf16x8 ones = {one, one, one, one, one, one, one, one};

    auto compute = [&](const char* sbase) {
        bf16x8 af[MF], bfr[NF];
#pragma unroll
        for (int i = 0; i < MF; ++i) {
            const bf16x4 lo = lds_tr16(sbase + a_off[0][i]);
            const bf16x4 hi = lds_tr16(sbase + a_off[1][i]);
            af[i] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        }
#pragma unroll
        for (int j = 0; j < NF; ++j) {
            const bf16x4 lo = lds_tr16(sbase + b_off[0][j]);
            const bf16x4 hi = lds_tr16(sbase + b_off[1][j]);
            bfr[j] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        }
#pragma unroll
        for (int i = 0; i < MF; ++i)
#pragma unroll
            for (int j = 0; j < NF; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        if (do_bias) {
#pragma unroll
            for (int i = 0; i < MF; ++i)
                accb[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], ones, accb[i], 0, 0, 0);
        }
    };

    stage_issue(0, smem);
    for (int t = 0; t < nsteps; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t + 1 < nsteps) stage_issue(t + 1, smem + ((t + 1) & 1) * STAGE);
        compute(smem + (t & 1) * STAGE);
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
                if (ci < p.c_in) atomicAdd(rowp + ci, acc[i][j][r]);
            }
            if (do_bias && (lane & 15) == 0) atomicAdd(p.dbias + co, accb[i][r]);
        }
    }
}

template <int TG, int TX>
int launch(const WgradArgs& a0, hipStream_t s) {
    WgradArgs a = a0;
    constexpr int STAGE = 32 * (TG + TX) * 2;
    const int tiles_co = (a.c_out + TG - 1) / TG;
    a.tiles_ci = (a.c_in + TX - 1) / TX;
    const long long base = (long long)tiles_co * a.tiles_ci * a.kh * a.kw;
    long long split = (2048 + base - 1) / base;
    long long chunk = (a.M + split - 1) / split;
    if (chunk < 512) chunk = 512;
    chunk = (chunk + 31) / 32 * 32;
    a.chunk_px = (int)chunk;
    const int ny = (int)((a.M + chunk - 1) / chunk);
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)wgrad_tn_kernel<TG, TX>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE);
        attr_done = true;
    }
    dim3 grid((unsigned)base, (unsigned)ny), block(256);
    hipLaunchKernelGGL((wgrad_tn_kernel<TG, TX>), grid, block, 2 * STAGE, s, a);
    return 0;
}

}  // namespace

extern "C" int tv_wgrad_tn(const tv_conv_desc* d, const void* x, const void* gy, float* dw, float* dbias,
                           void* stream) {
    TV_CHECK_ARG(d && x && gy && dw, "tv_wgrad_tn: null pointer");
    TV_CHECK_ARG(d->c_in > 0 && d->c_in % 8 == 0, "tv_wgrad_tn: c_in=%d must be a multiple of 8", d->c_in);
    TV_CHECK_ARG(d->c_out > 0 && d->c_out % 8 == 0, "tv_wgrad_tn: c_out=%d must be a multiple of 8", d->c_out);
    TV_CHECK_ARG(d->ldx >= d->c_in && d->ldx % 8 == 0 && d->ldo >= d->c_out && d->ldo % 8 == 0,
                 "tv_wgrad_tn: ldx/ldo must cover the channels and be multiples of 8");
    TV_CHECK_ARG(d->store_shuffle == 0, "tv_wgrad_tn: call with the transposed problem for shuffle-store layers");
    TV_CHECK_ARG(d->batch > 0 && d->h_in > 0 && d->w_in > 0 && d->h_out > 0 && d->w_out > 0, "tv_wgrad_tn: empty geometry");
    TV_CHECK_ARG((d->up_shift | 1) == 1 && (d->dil_mask | 1) == 1, "tv_wgrad_tn: up_shift/dil_mask must be 0 or 1");
    const long long M = (long long)d->batch * d->h_out * d->w_out;
    TV_CHECK_ARG(M < (1ll << 31) && (long long)d->batch * d->h_in * d->w_in < (1ll << 31), "tv_wgrad_tn: too many pixels");
    if (tv_init() != TV_OK) return TV_ERR_INIT;

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
    a.tiles_ci = 1; a.chunk_px = 0;
    hipStream_t s = (hipStream_t)stream;
    const bool g192 = (d->c_out % 192 == 0) && (d->c_out % 128 != 0);
    const bool x192 = (d->c_in % 192 == 0) && (d->c_in % 128 != 0);
    const bool g64 = d->c_out <= 64, x64 = d->c_in <= 64;
    if (g192 && x192) launch<192, 192>(a, s);
    else if (g192) { if (x64) launch<192, 64>(a, s); else launch<192, 128>(a, s); }
    else if (x192) { if (g64) launch<64, 192>(a, s); else launch<128, 192>(a, s); }
    else if (g64 && x64) launch<64, 64>(a, s);
    else if (g64) launch<64, 128>(a, s);
    else if (x64) launch<128, 64>(a, s);
    else launch<128, 128>(a, s);
    TV_CHECK_LAUNCH("tv_wgrad_tn");
    return TV_OK;
}
