/*
 * transvae_hip.h -- C ABI of libtransvae_hip.so, the gfx950 (MI355X) kernels
 * behind the TransVAE forward/backward path.
 *
 * The reference (benabbouosama/DEEPL-Project, R/ = transvae-implementation/)
 * has no FFI of its own: its hot path is a Python torch.nn.Module that
 * dispatches to stock ATen ops.  Each entry point
 * below therefore names the ATen call site(s) it replaces.  The Python shim in
 * deepl-project_amd/transvae/hip/ binds these with ctypes and raises
 * RuntimeError(tv_last_error()) on a non-zero status.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller; nothing is
 *     allocated or freed inside, nothing synchronises the device
 *   - activations are bf16, NHWC (= token-major [B, H*W, C]), fp32 accumulate
 *   - gradients of parameters and all statistics are fp32
 *   - `stream` is a hipStream_t passed as void* (0 = default stream)
 *   - return value: 0 on success, TV_ERR_* otherwise
 */
#ifndef TRANSVAE_HIP_H
#define TRANSVAE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TV_OK 0
#define TV_ERR_ARG 1     /* bad shape / unsupported configuration */
#define TV_ERR_LAUNCH 2  /* HIP reported a launch error */
#define TV_ERR_INIT 3    /* no device / allocation of the zero page failed */
#define TV_ERR_UNSUPPORTED 4 /* tv_igemm_nt_cat2 only: this shape's tile has no two-source loop -- concatenate and call tv_igemm_nt (no error text) */

#define TV_ACT_NONE 0
#define TV_ACT_GELU 1 /* exact (erf) GELU, R/transvae/modules/conv.py:56,86 */
#define TV_ACT_SILU 2 /* R/transvae/modules/upsample.py:35,96 */
/* Training-only variants of the saved tensor (no reference counterpart: autograd saves the pre-activation there).
 * desc.act | TV_ACT_SAVE_DERIV: `pre_act` receives act'(pre-activation) instead of the pre-activation, computed from the
 * same erf / exponential as the activation itself; tv_igemm_nt_actgrad(..., aux_act = TV_ACT_DERIV) then multiplies by
 * the saved tensor directly -- the backward epilogue carries no transcendental arithmetic. */
#define TV_ACT_DERIV 3
/* aux_act of tv_igemm_nt_actgrad only: the second tensor is ADDED, out = conv(x, w) + residual + aux -- two branch values
 * joining the residual stream in one fp32 sum with ONE rounding (the collapsed Conv-FFN tail: t + W_out u + (W_out W3) c,
 * R/transvae/modules/conv.py:85-104) */
#define TV_ACT_ADD 4
#define TV_ACT_SAVE_DERIV 16

/* library ------------------------------------------------------------------ */
int tv_init(void);                 /* allocates the device zero page; idempotent */
const char* tv_last_error(void);   /* message of the last failing call (thread local) */
int tv_abi_version(void);

/*
 * Geometry of one implicit-GEMM convolution / linear layer.
 *   x   : [batch, h_in, w_in, *] bf16, `ldx` elements between pixels (>= c_in)
 *   out : [batch, h_out, w_out, c_out] bf16, `ldo` elements between pixels
 * Virtual input coordinate of output (oy,ox), tap (ky,kx):
 *     uy = oy*stride + ky - pad ,  valid iff 0 <= uy < (h_in << up_shift)
 *                                        and (uy & dil_mask) == 0
 *     iy = uy >> up_shift                      (same for x)
 *   up_shift=1, dil_mask=0 : conv over a nearest-x2 upsampled input that is never
 *                            materialised (R/transvae/modules/upsample.py:94-95)
 *   up_shift=1, dil_mask=1 : zero-dilated input = data-gradient of a stride-2 conv
 * A linear layer is kh=kw=1, stride=1, pad=0, h=w=1, batch=#tokens.
 */
typedef struct tv_conv_desc {
    int batch, h_in, w_in, c_in, ldx;
    int h_out, w_out, c_out, ldo;
    int kh, kw, stride, pad;
    int up_shift, dil_mask;
    int act;           /* TV_ACT_* applied after bias, before residual */
    int store_shuffle; /* 1: pixel_shuffle(2) on store: out is [batch, 2*h_out, 2*w_out, c_out/4],
                          output column n = (dy*2+dx)*(c_out/4) + c  (upsample.py:121-123)
                          2: polyphase form of nearest-x2 upsample + 3x3 conv (upsample.py:94-95): the GEMM runs on the
                          (H+1) x (W+1) grid of 2x2 input neighbourhoods (kh = kw = 2, pad = 1, h_out = H+1, w_out = W+1),
                          column quadrant (py*2+px) of cell (y, x) is output pixel (2y - py, 2x - px) of the
                          [batch, 2H, 2W, c_out/4] result; phases that fall outside are dropped */
} tv_conv_desc;

/*
 * out = act(conv(x, w) + bias) + residual        (bf16 MFMA, fp32 accumulate)
 *   w        : [c_out, kh, kw, c_in] bf16 ("KRSC"; a Linear weight [out,in] is kh=kw=1)
 *   bias     : [c_out] fp32 or NULL
 *   residual : like out, or NULL
 *   pre_act  : like out, receives conv+bias before the activation (for backward), or NULL
 * Replaces F.conv2d / nn.Linear at R/transvae/modules/blocks.py:34,37, conv.py:39,54-60,65,
 * attention.py:43-48, upsample.py:33-37,42,93-98,103, the conv_in / conv_out / conv_mu / conv_logvar of models/{encoder,decoder,transvae}.py,
 * and -- called with rotated/transposed weights -- their data gradients.
 * Requires c_in % 32 == 0, c_out % 8 == 0, ldx % 8 == 0, ldo % 8 == 0 (bias, residual, pre_act, out 16-byte aligned).
 */
int tv_igemm_nt(const tv_conv_desc* d, const void* x, const void* w, const float* bias,
                const void* residual, void* pre_act, void* out, void* stream);

/* The QKV projection with RoPE in its epilogue (north star: "fused RMSNorm+RoPE+QKV-proj"; R/transvae/modules/attention.py:
 * 43-48 Linear q/k/v, :76-78 head split, :132-199 RoPE2D): out = x w^T + bias, and output columns < rope_cols (the q and k
 * thirds, whole heads of 64) are rotated with the table of tv_rope_qk before the single rounding to bf16; row m is token
 * m % tokens_per_image of its image. */
int tv_igemm_nt_rope(const tv_conv_desc* d, const void* x, const void* w, const float* bias, void* out,
                     const float* rope_tab, int tokens_per_image, int rope_cols, void* stream);

/*
 * Data gradient fused with the activation backward of the PREVIOUS layer:
 *     out = (conv(x, w) + residual) * act'(aux_pre_act)
 * i.e. the gradient w.r.t. the pre-activation tensor saved by the producing layer (aux_pre_act has the shape of
 * out).  Replaces autograd's GELU / SiLU backward (conv.py:56,86; upsample.py:35,96) without a separate pass.
 * aux_act = TV_ACT_ADD:  out = conv(x, w) + residual + aux_pre_act  (a second residual; see TV_ACT_ADD).
 */
int tv_igemm_nt_actgrad(const tv_conv_desc* d, const void* x, const void* w, const void* residual,
                        const void* aux_pre_act, int aux_act, void* out, void* stream);

/*
 * The same GEMM over K-CONCATENATED rows that are never materialised (1x1 / Linear geometry only, desc.c_in = k1 + k2):
 *     out = [x | x2] w^T (+ bias) (+ residual) (aux as in tv_igemm_nt_actgrad)
 * x supplies columns 0 .. k1-1 (row pitch desc.ldx), x2 columns k1 .. c_in-1 (row pitch ldx2); k1 and c_in - k1 multiples of 64;
 * w is [c_out][c_in] over the concatenated columns.  Two uses in the collapsed Conv-FFN tail (R/transvae/modules/conv.py:85-104):
 * t + [u | c] [W_out | W_out W3]^T and the data gradient onto u, [g | gz_c] [W_out ; W1] * gelu'.  Only the eight-phase loop of
 * the 256-row tiles has the second source: other shapes return TV_ERR_UNSUPPORTED (no error text) and the caller concatenates.
 */
int tv_igemm_nt_cat2(const tv_conv_desc* d, const void* x, const void* x2, int k1, int ldx2, const void* w, const float* bias,
                     const void* residual, const void* aux, int aux_act, void* out, void* stream);

/*
 * Weight gradient of the same layer (fp32):
 *   dw[co][ky][kx][ci] (+)= sum_p gy[p][co] * x_gathered[p][ky][kx][ci]
 *   dbias[co]          (+)= sum_p gy[p][co]                       (dbias may be NULL)
 *   gy : [batch, h_out, w_out, c_out] bf16 with `ldo` between pixels
 * The reduction over pixels is split over workgroups when the layer has few tiles; the partial tiles are then ADDED
 * into dw with fp32 atomics and the caller must pass a zeroed dw.  With a single pixel chunk every element of dw
 * is WRITTEN exactly once and dw needs no initialisation: tv_wgrad_tn_overwrites(d) says which of the two
 * tv_wgrad_tn will do for this geometry (1 = overwrites, 0 = accumulates, < 0 = bad descriptor).
 * dbias is ALWAYS added to (the workgroups that share a range of output channels take the pixel steps in turn and each
 * adds its share): pass it zeroed, or holding a running sum.
 * store_shuffle layers are handled by the caller as the transposed problem (x := the hi-res
 * gradient gathered with 2x2/stride-2 taps, gy := the layer input), so store_shuffle must be 0.
 * Requires c_in % 8 == 0, c_out % 8 == 0.
 * Replaces autograd's conv/linear weight-gradient for the call sites above.
 */
int tv_wgrad_tn(const tv_conv_desc* d, const void* x, const void* gy, float* dw, float* dbias,
                void* stream);
int tv_wgrad_tn_overwrites(const tv_conv_desc* d);
/* tv_wgrad_tn that always ADDS into dw / dbias (they hold the sum of earlier micro-batches): gradient accumulation without
 * autograd's separate add pass over 4.2 GB of gradients per micro-batch (R/train.py:557-646 accumulates through
 * loss.backward() into .grad). */
int tv_wgrad_tn_acc(const tv_conv_desc* d, const void* x, const void* gy, float* dw, float* dbias, void* stream);

/*
 * fp32 -> bf16 weight repack.  src: [O, T, I] fp32 (T = kh*kw taps).
 *   dst   : [O, T, I] bf16 (forward operand)                      or NULL
 *   dst_t : [I, T, O] bf16 with taps reversed if flip_taps        or NULL
 *           (the operand of the data-gradient convolution)
 */
int tv_pack_weight(const float* src, void* dst, void* dst_t, int O, int T, int I, int flip_taps,
                   void* stream);

/* Operands DERIVED from a 3x3 weight by sums of taps, one launch each (replaces ~20-60 slice adds / copies of the host code):
 *   TV_DERIVE_UP_FWD         w fp32 [Cout,3,3,Cin] -> bf16 [4*Cout,2,2,Cin]: nn.Upsample(nearest, 2) + 3x3 conv
 *                            (R/transvae/modules/upsample.py:94-95) in polyphase form, row (2*py+px)*Cout + co, tap (ty,tx)
 *   TV_DERIVE_UP_DGRAD       w -> bf16 [Cin,4,4,Cout]: its adjoint, a 4x4 / stride-2 / pad-1 convolution of the high-resolution gradient
 *   TV_DERIVE_UP_WGRAD_FOLD  fp32 [Cin,4,4,Cout] (weight gradient of that adjoint) -> fp32 [Cout,3,3,Cin] (gradient of w);
 *                            accumulate != 0 adds into dst (gradient accumulation over micro-batches)
 *   TV_DERIVE_S2_PARITY      w -> bf16 [4*Cin,2,2,Cout]: data-gradient operand of the 3x3 / stride-2 convolution by output parity
 *                            (R/transvae/modules/upsample.py:33-37), taps outside a class's footprint zero
 * Sums are taken in fp32 in a fixed order (ky outer, kx inner) and rounded once. */
#define TV_DERIVE_UP_FWD 1
#define TV_DERIVE_UP_DGRAD 2
#define TV_DERIVE_UP_WGRAD_FOLD 3
#define TV_DERIVE_S2_PARITY 4
int tv_conv3x3_derived(const float* src, void* dst, int form, int c_out, int c_in, int accumulate, void* stream);

/* GroupNorm(32)+SiLU (R/transvae/modules/blocks.py:33,36,60-65; decoder.py:93,128-129) --------- */
/* Reductions over all pixels of an image span workgroups: each block writes a partial sum and a
 * finalize kernel adds the partials in block order (no atomics => bit-reproducible).  `partials`
 * is scratch of tv_gn_partial_count(batch, hw, C) floats. */
long long tv_gn_partial_count(int batch, int hw, int C);
/* per-(b,channel) sums about the pivot piv = x[b, pixel 0, c]: stats[b][c][0] = sum (x - piv), [1] = sum (x - piv)^2
 * (fp32; tv_gn_silu_fwd merges the channels of a group with Chan's (mean, M2) update -- no E[x^2] - mean^2 cancellation) */
int tv_gn_stats(const void* x, float* stats, float* partials, int batch, int hw, int C, void* stream);
/* y = silu(groupnorm(x)); stats from tv_gn_stats; writes mean/rstd per (b,group) to mr[b][G][2] */
int tv_gn_silu_fwd(const void* x, const float* stats, const float* gamma, const float* beta,
                   float* mr, void* y, int batch, int hw, int C, int G, float eps, void* stream);
/* y = silu(groupnorm(x)) again from the mean / rstd a tv_gn_silu_fwd call wrote to mr: the fused-op recompute of a
 * checkpointed ResBlock (R/transvae/models/encoder.py:97-99,117-118: activation checkpointing) -- bit-identical to the forward */
int tv_gn_silu_apply(const void* x, const float* mr, const float* gamma, const float* beta, void* y,
                     int batch, int hw, int C, int G, void* stream);
/* backward pass 1: red[b][c][0] = sum dh, red[b][c][1] = sum dh*xhat   (dh = dy * silu'(h)) */
int tv_gn_silu_bwd_reduce(const void* x, const void* dy, const float* mr, const float* gamma,
                          const float* beta, float* red, float* partials, int batch, int hw, int C,
                          int G, void* stream);
/* backward pass 2: dx (+= dres if given); dgamma/dbeta (fp32 [C]) accumulated from red */
int tv_gn_silu_bwd_apply(const void* x, const void* dy, const void* dres, const float* mr,
                         const float* red, const float* gamma, const float* beta, void* dx,
                         float* dgamma, float* dbeta, int batch, int hw, int C, int G, void* stream);

/* Token-wise norms (R/transvae/modules/blocks.py:179-194, attention.py:39-41,71-73) ------------
 * mode 0: y = x * rsqrt(mean(x^2)+eps_rms)                     (RMSNorm, weight folded downstream)
 * mode 1: u = x*w*rsqrt(mean(x^2)+eps_rms); y = (u-mean u)*rsqrt(var u + eps_ln)
 *         (RMSNorm followed by the affine-free LayerNorm shared by norm_q/k/v)
 * rows: x,y [T, C] bf16; w fp32 [C] (mode 1 only). */
int tv_rownorm_fwd(const void* x, const float* w, void* y, int T, int C, int mode, float eps_rms,
                   float eps_ln, void* stream);
/* dx (+= dres if given) and, mode 1, dw[C] += ...  (fp32 atomics) */
int tv_rownorm_bwd(const void* x, const float* w, const void* dy, const void* dres, void* dx,
                   float* dw, int T, int C, int mode, float eps_rms, float eps_ln, void* stream);

/* RoPE (R/transvae/modules/attention.py:132-199), in place on the q and k thirds of
 * qkv [B, N, 3, heads, 64] bf16.  tab: [N, 4, 32] fp32 = cos1,sin1,cos2,sin2 per pair.
 * transpose=1 applies the adjoint (backward). */
int tv_rope_qk(void* qkv, const float* tab, int B, int N, int heads, int transpose, void* stream);

/* softmax(q k^T * scale) v, non-causal, head_dim 64 (attention.py:88-92).
 * qkv [B,N,3,heads,64] bf16; o [B,N,heads,64] bf16; lse [B,heads,N] fp32 (natural log). */
int tv_attn_fwd(const void* qkv, void* o, float* lse, int B, int N, int heads, float scale,
                void* stream);
/* dqkv [B,N,3,heads,64] bf16 from do; delta [2,B,heads,N] fp32 is scratch (the kernels keep -delta and -lse log2(e) there).  rope_tab (may be NULL): the table of tv_rope_qk;
 * when given, q and k in `qkv` are the ROTATED projections (tv_igemm_nt_rope) and dq / dk are stored as gradients w.r.t.
 * the un-rotated ones (the adjoint of attention.py:156-197 applied to the fp32 accumulators in the store). */
int tv_attn_bwd(const void* qkv, const void* o, const void* d_o, const float* lse, float* delta,
                const float* rope_tab, void* dqkv, int B, int N, int heads, float scale, void* stream);

/* elementwise ------------------------------------------------------------------------------- */
/* dz = dy * act'(z)   (n bf16 elements, n % 8 == 0) */
int tv_act_bwd(const void* z, const void* dy, void* dz, long long n, int act, void* stream);
/* a += b (bf16, n % 8 == 0) */
int tv_add_(void* a, const void* b, long long n, void* stream);
/* NCHW fp32 [B,C,H,W] -> NHWC bf16 [B,H,W,Cpad] (channels >= C zero-filled) and back */
int tv_nchw_to_nhwc(const float* src, void* dst, int B, int C, int H, int W, int Cpad, void* stream);
int tv_nhwc_to_nchw(const void* src, float* dst, int B, int C, int H, int W, int Cpad, void* stream);
/* 3x3/pad-1 patches of an NCHW fp32 image -> [B*H*W, Kpad] bf16 rows ordered (ky,kx,c), for the
 * stem conv_in (R/transvae/models/encoder.py:52) */
int tv_im2col3x3(const float* src, void* dst, int B, int C, int H, int W, int Kpad, void* stream);
/* dst[b,y,x,c] = sum of the 2x2 block of src[b,2y+dy,2x+dx,c]  (adjoint of nearest x2) */
int tv_pool2x2_sum(const void* src, void* dst, int B, int H, int W, int C, void* stream);

/* Train-step glue around the path ------------------------------------------------------------------
 * Multi-tensor AdamW with the global-norm clip and the non-finite guard on the device, replacing the caller pattern
 *   clip_grad_norm_(model.parameters(), 1.0); optimizer.step()        (R/train.py:610-618, AdamW at :681-687)
 *   "skip the step when the loss is not finite"                       (R/train_2.py:328-338)
 * in two launches, plus one launch that refreshes every bf16 operand of the next forward / backward. */
typedef struct tv_opt_tensor {
    float* param;        /* fp32 master weights, updated in place */
    const float* grad;   /* fp32 gradient (same element order as param) */
    float* exp_avg;      /* Adam first moment  (fp32, updated in place) */
    float* exp_avg_sq;   /* Adam second moment (fp32, updated in place) */
    void* shadow_bf16;   /* optional: bf16 copy of param written by the update (NULL = none) */
    long long numel;
} tv_opt_tensor;
/* elements handled by one workgroup; chunk table = int pairs (tensor index, chunk index inside the tensor) */
int tv_opt_chunk_elems(void);
/* ctrl: 8 device floats owned by the caller, persistent across steps:
 *   [0] step count t   [1] gradient L2 norm   [2] clip coefficient   [3] 1 if this step is skipped (non-finite norm)
 *   [4] skipped steps so far   [5] 1 - beta1^t   [6] sqrt(1 - beta2^t)
 * Computes the global norm over all gradients (partials: n_chunks floats of scratch; fixed-order, bit-reproducible),
 * coef = min(1, max_norm / (norm + 1e-6)) (max_norm <= 0: no clipping), advances t unless the norm is non-finite.
 * compute_norm = 0 skips the reduction (norm reported as 0, never skipped). */
int tv_opt_grad_norm(const tv_opt_tensor* table_dev, const int* chunks_dev, int n_chunks, float* partials,
                     float* ctrl, float max_norm, float beta1, float beta2, int compute_norm, void* stream);
/* AdamW update of every tensor in the table with g * ctrl[2] as the gradient; no-op when ctrl[3] != 0 */
int tv_opt_adamw(const tv_opt_tensor* table_dev, const int* chunks_dev, int n_chunks, const float* ctrl,
                 float lr, float beta1, float beta2, float eps, float weight_decay, void* stream);
/* shadow_bf16 = (bf16) param for every tensor that has one (first fill) */
int tv_opt_cast_shadows(const tv_opt_tensor* table_dev, const int* chunks_dev, int n_chunks, void* stream);
/* every transposed operand in one launch: src bf16 [O,T,I] -> dst_t bf16 [I,T',O] (T' reversed if flip);
 * tile_start = exclusive prefix sum of ceil(O/64)*ceil(I/64)*T over the forms */
typedef struct tv_pack_form {
    const void* src;
    void* dst_t;
    int O, T, I, flip;
    long long tile_start;
} tv_pack_form;
int tv_pack_weight_multi(const tv_pack_form* forms_dev, int n_forms, long long total_tiles, void* stream);

/* Parameter folds ------------------------------------------------------------------------------------
 * A LayerNorm / RMSNorm affine in front of a Linear folds into the projection (attention.py:39-48,71-78; blocks.py:146-149):
 *   Wf[r][c] = W[r][c] * gamma[c],   bf[r] = sum_c W[r][c] * beta[c]     (beta / bf NULL together: no bias term)
 * and the gradients back onto W, gamma, beta in one pass (fp32, [R, C] row-major; column sums in a fixed order). */
int tv_fold_cols(const float* W, const float* gamma, const float* beta, float* Wf, float* bf, int R, int C, void* stream);
long long tv_fold_partial_count(int R, int C);     /* floats of scratch for tv_fold_cols_bwd */
int tv_fold_cols_bwd(const float* dWf, const float* dbf, const float* W, const float* gamma, const float* beta,
                     float* dW, float* dgamma, float* dbeta, float* partials, int R, int C, void* stream);

/* Closed-form loss terms on the path's outputs, value and gradient in one pass ---------------------
 *   out[0] = l1_weight * mean |f(recon) - target|     f = identity (R/transvae/losses/vae_loss.py:83-84) or sigmoid (P/...:80-84)
 *   out[1] = kl_weight * -0.5 * sum(1 + lv - mu^2 - exp(lv)) / kl_denom     (R/...:94-96: kl_denom = B*H_lat*W_lat;
 *            P/...:96-102: kl_denom = numel), lv = clamp(logvar, lo, hi) when lo < hi (R/train_2.py:316-318)
 *   out[2] = out[0] + out[1]
 * d_recon / d_mu / d_logvar (each may be NULL) receive the gradient of out[2]; all tensors fp32, any layout shared by the
 * tensors of a pair (they are walked flat).  partials: tv_vae_loss_partial_count(n_img, n_lat) floats of scratch. */
long long tv_vae_loss_partial_count(long long n_img, long long n_lat);
int tv_vae_loss_l1_kl(const float* recon, const float* target, const float* mu, const float* logvar,
                      float* d_recon, float* d_mu, float* d_logvar, float* partials, float* out,
                      long long n_img, long long n_lat, float l1_weight, float kl_weight, float kl_denom,
                      int sigmoid, float logvar_lo, float logvar_hi, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TRANSVAE_HIP_H */
