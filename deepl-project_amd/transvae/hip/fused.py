"""Block-level autograd Functions: one forward / one hand-orchestrated backward per TransVAE block.

Per-op autograd (ops.ConvFn etc.) is correct but leaves three kinds of avoidable HBM passes in the
backward, all visible in the rocprof summary (profiles/): a separate GELU/SiLU-backward pass per
activation, autograd's add kernels wherever a tensor has two consumers (every residual), and zero
fills.  Here each block's backward is written out explicitly, so that

  * activation backward rides in the epilogue of the data-gradient GEMM that produces the incoming
    gradient (`tv_igemm_nt_actgrad`: out = (acc + residual) * act'(pre)); the forward epilogue saves act'(pre)
    itself (`conv_forward(..., want_pre="deriv")`, from the erf / exponential it computes anyway), so the tensors
    called `pre_*` below hold the DERIVATIVE and the backward epilogue is one multiply per element,
  * the second gradient of a residual stream is added inside GroupNorm-backward / row-norm-backward
    (`dres`) or a GEMM epilogue (`residual`),
  * nothing is materialised twice.

Math and parameter layouts are those of ops.py; reference lines are cited there and in modules/*.py.
"""
from __future__ import annotations

import os

import torch

from . import _lib as L
from . import ops
from .ops import BF16, _p, _stream, conv_dgrad, conv_forward, conv_wgrad

GELU, SILU, NONE = L.ACT_GELU, L.ACT_SILU, L.ACT_NONE
# what the forward saves for an activation: its derivative (default) or the pre-activation (TV_SAVE_DERIV=0, A/B timing)
_SAVE_DERIV = os.environ.get("TV_SAVE_DERIV", "1") != "0"
_WANT = "deriv" if _SAVE_DERIV else True


def _aux_act(act_id: int) -> int:
    return L.ACT_DERIV if _SAVE_DERIV else act_id


# ------------------------------------------------------------------------------------------------
# raw GroupNorm / row-norm / attention calls
# ------------------------------------------------------------------------------------------------
# Ablation hooks for UPPER BOUNDS on two fusions that are not built (VERDICT r03 row g; results are wrong by construction, only the
# step time is read; bench.py refuses TV_* variables unless --allow-tuning-env and prints them into config.tuning_env):
#   TV_ABL_SKIP_GN_STATS=1    no tv_gn_stats pass: statistics from a cached constant (mean = the pivot pixel, variance 1) -- what
#                             GroupNorm statistics in the producing convolution's epilogue could save AT MOST (the epilogue work is free here)
#   TV_ABL_SKIP_ROWNORM_FWD=1 no x-hat pass in front of the QKV / Conv-FFN projections (the first x-hat of a shape is reused: finite, wrong)
#                             -- what x-hat inside the GEMM could save at most
_ABL_SKIP_GN_STATS = os.environ.get("TV_ABL_SKIP_GN_STATS") == "1"
_ABL_SKIP_ROWNORM_FWD = os.environ.get("TV_ABL_SKIP_ROWNORM_FWD") == "1"
_abl_stats = {}


def gn_silu_fwd(x, gamma, beta, groups, eps):
    B, H, W, Cc = x.shape
    lib = L.load()
    if _ABL_SKIP_GN_STATS:
        key = (B, H * W, Cc, x.device)
        stats = _abl_stats.get(key)
        if stats is None:
            stats = torch.zeros((B, Cc, 2), dtype=torch.float32, device=x.device)
            stats[..., 1] = float(H * W)
            _abl_stats[key] = stats
    else:
        stats = torch.empty((B, Cc, 2), dtype=torch.float32, device=x.device)
        part = torch.empty((lib.tv_gn_partial_count(B, H * W, Cc),), dtype=torch.float32, device=x.device)
        L.check(lib.tv_gn_stats(_p(x), _p(stats), _p(part), B, H * W, Cc, _stream()), "tv_gn_stats")
    mr = torch.empty((B, groups, 2), dtype=torch.float32, device=x.device)
    y = torch.empty_like(x)
    L.check(lib.tv_gn_silu_fwd(_p(x), _p(stats), _p(gamma), _p(beta), _p(mr), _p(y), B, H * W, Cc, groups, eps, _stream()),
            "tv_gn_silu_fwd")
    return y, mr


def gn_silu_apply(x, mr, gamma, beta, groups):
    """silu(groupnorm(x)) recomputed from saved mean / rstd (bit-identical to gn_silu_fwd's output)."""
    B, H, W, Cc = x.shape
    y = torch.empty_like(x)
    L.check(L.load().tv_gn_silu_apply(_p(x), _p(mr), _p(gamma), _p(beta), _p(y), B, H * W, Cc, groups, _stream()), "tv_gn_silu_apply")
    return y


def gn_silu_bwd(x, dy, dres, mr, gamma, beta, groups):
    """(dx [+ dres], dgamma, dbeta)"""
    B, H, W, Cc = x.shape
    lib = L.load()
    red = torch.empty((B, Cc, 2), dtype=torch.float32, device=x.device)
    part = torch.empty((lib.tv_gn_partial_count(B, H * W, Cc),), dtype=torch.float32, device=x.device)
    L.check(lib.tv_gn_silu_bwd_reduce(_p(x), _p(dy), _p(mr), _p(gamma), _p(beta), _p(red), _p(part), B, H * W, Cc, groups,
                                      _stream()), "tv_gn_silu_bwd_reduce")
    dx = torch.empty_like(x)
    dg = ops.zeros_f32((Cc,), x.device)
    db = ops.zeros_f32((Cc,), x.device)
    L.check(lib.tv_gn_silu_bwd_apply(_p(x), _p(dy), _p(dres), _p(mr), _p(red), _p(gamma), _p(beta), _p(dx), _p(dg), _p(db),
                                     B, H * W, Cc, groups, _stream()), "tv_gn_silu_bwd_apply")
    return dx, dg, db


def rownorm_fwd(x, w, mode, eps_rms, eps_ln):
    T, Cc = x.shape
    if _ABL_SKIP_ROWNORM_FWD:       # (timing ablation: the first x-hat of this shape stands in for every later one -- finite, wrong)
        hit = _abl_stats.get(("rownorm", T, Cc, mode, x.device))
        if hit is not None:
            return hit
    y = torch.empty_like(x)
    L.check(L.load().tv_rownorm_fwd(_p(x), _p(w), _p(y), T, Cc, mode, eps_rms, eps_ln, _stream()), "tv_rownorm_fwd")
    if _ABL_SKIP_ROWNORM_FWD:
        _abl_stats[("rownorm", T, Cc, mode, x.device)] = y
    return y


def rownorm_bwd(x, w, dy, dres, mode, eps_rms, eps_ln):
    T, Cc = x.shape
    dx = torch.empty_like(x)
    dw = ops.zeros_f32((Cc,), x.device) if mode == 1 else None
    L.check(L.load().tv_rownorm_bwd(_p(x), _p(w), _p(dy), _p(dres), _p(dx), _p(dw), T, Cc, mode, eps_rms, eps_ln, _stream()),
            "tv_rownorm_bwd")
    return dx, dw


def _c(t):
    return t.contiguous() if t is not None else None


# ------------------------------------------------------------------------------------------------
# parameter folds: norm affine in front of a Linear -> one projection (two launches instead of ~40 small PyTorch kernels)
# ------------------------------------------------------------------------------------------------
class FoldFn(torch.autograd.Function):
    """(Wf, bf) for a list of (W [R, C], gamma [C], beta [C] | None) triples stacked along the rows:
    Wf = cat_s(W_s * gamma_s),  bf = cat_s(W_s @ beta_s)  (bf is None when no triple has a beta)."""

    @staticmethod
    def forward(ctx, n, *args):
        Ws, gs, bs = args[:n], args[n:2 * n], args[2 * n:3 * n]
        ops._need_gpu(*Ws)
        lib = L.load()
        R, Cc = Ws[0].shape
        has_b = bs[0] is not None
        Wf = torch.empty((n * R, Cc), dtype=torch.float32, device=Ws[0].device)
        bf = torch.empty((n * R,), dtype=torch.float32, device=Ws[0].device) if has_b else None
        Ws = [w.contiguous() for w in Ws]
        gs = [g.contiguous() for g in gs]
        bs = [b.contiguous() if b is not None else None for b in bs]
        for s in range(n):
            ops._require(Ws[s].dtype == torch.float32 and tuple(Ws[s].shape) == (R, Cc) and gs[s].numel() == Cc, "fold: fp32 [R, C] weights")
            L.check(lib.tv_fold_cols(_p(Ws[s]), _p(gs[s]), _p(bs[s]), _p(Wf[s * R:]), _p(bf[s * R:]) if has_b else None, R, Cc, _stream()),
                    "tv_fold_cols")
        ctx.n, ctx.has_b = n, has_b
        ctx.save_for_backward(*Ws, *gs, *[b for b in bs if b is not None])
        return (Wf, bf) if has_b else (Wf, None)

    @staticmethod
    def backward(ctx, dWf, dbf):
        n, has_b = ctx.n, ctx.has_b
        saved = ctx.saved_tensors
        Ws, gs = saved[:n], saved[n:2 * n]
        bs = saved[2 * n:] if has_b else [None] * n
        lib = L.load()
        R, Cc = Ws[0].shape
        dWf = dWf.contiguous()
        if has_b:
            dbf = dbf.contiguous() if dbf is not None else torch.zeros((n * R,), dtype=torch.float32, device=dWf.device)
        outW, outg, outb = [], [], []
        part = torch.empty((lib.tv_fold_partial_count(R, Cc),), dtype=torch.float32, device=dWf.device)
        for s in range(n):
            dW = torch.empty_like(Ws[s])
            dg = torch.empty_like(gs[s])
            db = torch.empty_like(bs[s]) if has_b else None
            L.check(lib.tv_fold_cols_bwd(_p(dWf[s * R:]), _p(dbf[s * R:]) if has_b else None, _p(Ws[s]), _p(gs[s]), _p(bs[s]), _p(dW), _p(dg),
                                         _p(db), _p(part), R, Cc, _stream()), "tv_fold_cols_bwd")
            outW.append(dW)
            outg.append(dg)
            outb.append(db)
        return (None, *outW, *outg, *outb)


def fold(weights, gammas, betas=None):
    n = len(weights)
    betas = list(betas) if betas is not None else [None] * n
    return FoldFn.apply(n, *weights, *gammas, *betas)


import os

# Weight gradients do not feed the data-gradient chain, so they CAN run on a side stream (TV_WGRAD_SIDE_STREAM=1) to
# fill the partial last wave of the dgrad launches.  Measured on MI355X (Large, bs256): 104.5 img/s with, 105.5 without
# -- the two GEMM streams just compete for the same CUs -- so it is OFF by default.  When on: outputs are allocated on
# the main stream (the caching allocator must see their consumer stream), the side stream waits for the producer of
# `gz`, and the main stream joins the side stream before the Function returns.
_SIDE = {"on": os.environ.get("TV_WGRAD_SIDE_STREAM", "0") == "1", "streams": {}}


def _side_stream(device):
    st = _SIDE["streams"].get(device)
    if st is None:
        st = torch.cuda.Stream(device=device)
        _SIDE["streams"][device] = st
    return st


def _stash(ctx, *pairs):
    """pairs of (input index, tensor): remember which nn.Parameter each weight / bias input is, for in-place accumulation"""
    ctx.gparams = {i: ops.param_of(t) for i, t in pairs}


def _wg(ctx, iw, ib, geo, w, x, gz):
    """weight / bias gradient of one layer, skipped when neither input of the Function needs it (frozen parameters)."""
    need_w, need_b = ctx.needs_input_grad[iw], ctx.needs_input_grad[ib]
    if not (need_w or need_b):
        return None, None
    if need_w and geo.mode != "shuf":
        gp = getattr(ctx, "gparams", None)
        views = ops.grad_views(gp.get(iw), w, gp.get(ib), need_b) if gp else None
        if views is not None:      # micro-batch >= 2 of a step: add straight into param.grad, nothing for autograd to do
            if geo.mode == "c3up":
                conv_wgrad(geo, w, x, gz, need_b, out=views, accumulate=True)
            else:
                ops.wgrad_acc(geo.fwd_desc(0), x, gz, views[0], views[1])
            ops.acc_stats["in_place"] += 1
            return None, None
        ops.acc_stats["autograd"] += 1
    if _SIDE["on"] and geo.mode not in ("shuf", "c3up"):
        out = ops.conv_wgrad_alloc(geo, w, need_b, x, gz)
        main = torch.cuda.current_stream()
        side = _side_stream(x.device)
        side.wait_stream(main)                      # gz (and the zero fills) are ready
        with torch.cuda.stream(side):
            dw, db = conv_wgrad(geo, w, x, gz, need_b, out=out)
        ctx._side_used = True
    else:
        dw, db = conv_wgrad(geo, w, x, gz, need_b)
    return (dw if need_w else None), db


def _join(ctx, device):
    if getattr(ctx, "_side_used", False):
        torch.cuda.current_stream().wait_stream(_side_stream(device))


# ------------------------------------------------------------------------------------------------
# ResBlock (identity shortcut)   R/transvae/modules/blocks.py:48-68
# ------------------------------------------------------------------------------------------------
class ResBlockFn(torch.autograd.Function):
    """recompute = the activation-checkpointing toggle (R/transvae/models/encoder.py:97-99,117-118) as a FUSED-OP recompute:
    the two GroupNorm+SiLU outputs (the operands of the two weight gradients) are not saved; the backward rebuilds each from
    the tensor it normalised and the saved mean / rstd (one HBM pass each, bit-identical), so a block keeps 2 full-resolution
    tensors instead of 4 and no convolution runs twice."""

    @staticmethod
    def forward(ctx, x, g1, b1, w1, c1b, g2, b2, w2, c2b, eps1, eps2, recompute=False):
        ops._need_gpu(x)
        g1, b1, g2, b2 = _c(g1), _c(b1), _c(g2), _c(b2)
        a1, mr1 = gn_silu_fwd(x, g1, b1, 32, eps1)
        h1, _, geo1, w1c = conv_forward(a1, w1, c1b, None, "c3s1", NONE, False)
        a2, mr2 = gn_silu_fwd(h1, g2, b2, 32, eps2)
        out, _, geo2, w2c = conv_forward(a2, w2, c2b, x, "c3s1", NONE, False)
        ctx.geo = (geo1, geo2)
        _stash(ctx, (3, w1), (4, c1b), (7, w2), (8, c2b))
        ctx.recompute = bool(recompute)
        if recompute:
            ctx.save_for_backward(x, h1, mr1, mr2, g1, b1, g2, b2, w1c, w2c)
        else:
            ctx.save_for_backward(x, a1, h1, a2, mr1, mr2, g1, b1, g2, b2, w1c, w2c)
        return out

    @staticmethod
    def backward(ctx, g):
        if ctx.recompute:
            x, h1, mr1, mr2, g1, b1, g2, b2, w1, w2 = ctx.saved_tensors
            a1 = a2 = None
        else:
            x, a1, h1, a2, mr1, mr2, g1, b1, g2, b2, w1, w2 = ctx.saved_tensors
        geo1, geo2 = ctx.geo
        g = g.contiguous()
        da2 = conv_dgrad(geo2, w2, g, h1.shape)
        if a2 is None and (ctx.needs_input_grad[7] or ctx.needs_input_grad[8]):
            a2 = gn_silu_apply(h1, mr2, g2, b2, 32)
        dw2, dc2b = _wg(ctx, 7, 8, geo2, w2, a2, g)
        del a2
        dh1, dg2, db2 = gn_silu_bwd(h1, da2, None, mr2, g2, b2, 32)
        del da2
        da1 = conv_dgrad(geo1, w1, dh1, x.shape)
        if a1 is None and (ctx.needs_input_grad[3] or ctx.needs_input_grad[4]):
            a1 = gn_silu_apply(x, mr1, g1, b1, 32)
        dw1, dc1b = _wg(ctx, 3, 4, geo1, w1, a1, dh1)
        del a1
        dx, dg1, db1 = gn_silu_bwd(x, da1, g, mr1, g1, b1, 32)    # + skip-connection gradient, fused
        _join(ctx, x.device)
        return dx, dg1, db1, dw1, dc1b, dg2, db2, dw2, dc2b, None, None, None


# ------------------------------------------------------------------------------------------------
# attention branch of a TransVAE block:  t + proj(attn(rope(qkv(LN-hat(RMSNorm(t))))))
# ------------------------------------------------------------------------------------------------
class AttnBranchFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, t, w_rms, wqkv, bqkv, wp, bp, tab, B, N, heads, scale, eps_rms, eps_ln):
        ops._need_gpu(t)
        lib = L.load()
        w_rms = _c(w_rms)
        Cc = t.shape[1]
        xh = rownorm_fwd(t, w_rms, 1, eps_rms, eps_ln)
        # QKV projection with RoPE in the GEMM epilogue (q and k thirds rotated before the one rounding to bf16)
        qkv, _, geo_q, wq = conv_forward(xh, wqkv, _c(bqkv), None, "linear", NONE, False,
                                         rope=(tab, N, 2 * Cc) if tab is not None else None)
        o = torch.empty((B * N, Cc), dtype=BF16, device=t.device)
        lse = torch.empty((B, heads, N), dtype=torch.float32, device=t.device)
        L.check(lib.tv_attn_fwd(_p(qkv), _p(o), _p(lse), B, N, heads, scale, _stream()), "tv_attn_fwd")
        out, _, geo_p, wpc = conv_forward(o, wp, _c(bp), t, "linear", NONE, False)
        ctx.geo = (geo_q, geo_p)
        _stash(ctx, (2, wqkv), (3, bqkv), (4, wp), (5, bp))
        ctx.meta = (B, N, heads, scale, eps_rms, eps_ln)
        ctx.save_for_backward(t, w_rms, xh, qkv, o, lse, wq, wpc, tab)
        return out

    @staticmethod
    def backward(ctx, g):
        t, w_rms, xh, qkv, o, lse, wq, wp, tab = ctx.saved_tensors
        geo_q, geo_p = ctx.geo
        B, N, heads, scale, eps_rms, eps_ln = ctx.meta
        lib = L.load()
        g = g.contiguous()
        do = conv_dgrad(geo_p, wp, g, o.shape)
        dwp, dbp = _wg(ctx, 4, 5, geo_p, wp, o, g)
        delta = torch.empty((2, B, heads, N), dtype=torch.float32, device=t.device)
        dqkv = torch.empty_like(qkv)
        # (the adjoint of RoPE is applied to dq / dk inside the attention-backward stores)
        L.check(lib.tv_attn_bwd(_p(qkv), _p(o), _p(do), _p(lse), _p(delta), _p(tab), _p(dqkv), B, N, heads, scale, _stream()),
                "tv_attn_bwd")
        dxh = conv_dgrad(geo_q, wq, dqkv, xh.shape)
        dwq, dbq = _wg(ctx, 2, 3, geo_q, wq, xh, dqkv)
        dt, dw_rms = rownorm_bwd(t, w_rms, dxh, g, 1, eps_rms, eps_ln)     # + residual gradient, fused
        _join(ctx, t.device)
        return dt, dw_rms, dwq, dbq, dwp, dbp, None, None, None, None, None, None, None


# ------------------------------------------------------------------------------------------------
# Conv-FFN branch:  t + W_out (u + W3 gelu(K3x3 gelu(W1 u))) ,  u = gelu(W_in RMS-hat(t))
# The tail runs in collapsed form (ops.ffn_collapsed_operands):  t + W_out u + (W_out W3) c2 + (W_out b3 + b_out).
# ------------------------------------------------------------------------------------------------
class ConvFFNBranchFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, t, w_in, b_in, w1, b1, w2, b2, w3, b3, w_out, b_out, B, H, W, eps_rms):
        ops._need_gpu(t)
        T, d = t.shape
        train = any(ctx.needs_input_grad)
        b3, b_out = _c(b3), _c(b_out)
        r = rownorm_fwd(t, None, 0, eps_rms, 1e-5)
        u, pre_u, geo_in, w_in_c = conv_forward(r, w_in, _c(b_in), None, "linear", GELU, train and _WANT)
        c1, pre_c1, geo1, w1c = conv_forward(u, w1, _c(b1), None, "linear", GELU, train and _WANT)
        mid = c1.shape[1]
        c2, pre_c2, geo2, w2c = conv_forward(c1.view(B, H, W, mid), w2, _c(b2), None, "c3s1", GELU, train and _WANT)
        w3c, w_out_c = w3.contiguous(), w_out.contiguous()
        wc_f, wc_t, bc, wcat_f = ops.ffn_collapsed_operands(w_out_c, w3c, b3, b_out)
        # t + [u | c2] [W_out | Wc]^T + bc over the K-concatenation (one GEMM, nothing of [T, 4d + d] exists); shapes whose tile
        # has no two-source loop: two GEMMs, the branch values joining the residual stream in ONE fp32 sum (TV_ACT_ADD) -- an
        # intermediate t + ... in bf16 would round the stream a second time
        out = ops.gemm_rows2(u, c2.view(T, mid), wcat_f, d, bias=bc, residual=t)
        if out is None:
            o1 = ops.gemm_rows(c2.view(T, mid), wc_f, d, bias=bc)                       # (W_out W3) c2 + W_out b3 + b_out
            wo_f, _ = ops.pack_weight(w_out_c.view(d, 1, w_out_c.shape[1]), True, False, False)
            out = ops.gemm_rows(u, wo_f.view(d, -1), d, residual=t, aux=o1, aux_act=L.ACT_ADD)   # t + W_out u + o1
        geo_out = ops._Geo("linear", u, w_out_c)
        ctx.geo = (geo_in, geo1, geo2, geo_out)
        _stash(ctx, (1, w_in), (2, b_in), (3, w1), (4, b1), (5, w2), (6, b2), (7, w3), (8, b3), (9, w_out), (10, b_out))
        ctx.meta = (B, H, W, eps_rms)
        ctx.save_for_backward(t, r, u, pre_u, c1, pre_c1, c2, pre_c2, w_in_c, w1c, w2c, w3c, w_out_c, wc_t, b3)
        return out

    @staticmethod
    def backward(ctx, g):
        t, r, u, pre_u, c1, pre_c1, c2, pre_c2, w_in, w1, w2, w3, w_out, wc_t, b3 = ctx.saved_tensors
        geo_in, geo1, geo2, geo_out = ctx.geo
        B, H, W, eps_rms = ctx.meta
        T, d = t.shape
        mid = c1.shape[1]
        hid = u.shape[1]
        g = g.contiguous()
        need = ctx.needs_input_grad
        # ---- the composite: G = g^T c2 (+ sum_t g), then the chain rule onto W_out, W3, b3 (deferred while in-place)
        gp = ctx.gparams
        p_out, p3, pb3 = gp.get(9), gp.get(7), gp.get(8)
        need_chain = need[7] or need[8] or need[9]
        dw3 = db3 = dwo_chain = None
        deferred = False
        if need_chain:
            can_defer = need[9] and need[7] and p3 is not None and p_out is not None
            views = ops.grad_views(p_out, w_out, gp.get(10), need[10]) if can_defer else None
            dG = ops._rows_desc(T, mid, d)
            if views is not None or (can_defer and ops._defer_chain):
                # in-place micro-batch (or any backward pass train_step marks): accumulate G and the column sums only; the
                # chain rule runs once per optimizer step (ops.flush_deferred_grads)
                ent = ops.defer_ffn_grad(p_out, p3, pb3, w_out, w3, b3, need[8])
                ops.wgrad_acc(dG, c2.view(T, mid), g, ent["G"], ent["gs"])
                deferred = True
            else:
                pend = ops.take_deferred_ffn(p_out)           # (a syncing micro-batch after in-place ones: fold their share in)
                G = pend["G"] if pend is not None else ops.zeros_f32((d, mid), t.device)
                gs = pend["gs"] if pend is not None else ops.zeros_f32((d,), t.device)
                ops.wgrad_acc(dG, c2.view(T, mid), g, G, gs)
                dwo_chain, dw3, db3 = ops.ffn_chain_grads(G, gs if b3 is not None else None, w_out, w3, b3)
        dw_out, db_out = _wg(ctx, 9, 10, geo_out, w_out, u, g)                          # direct part: g^T u, sum_t g
        if dwo_chain is not None and need[9]:
            dw_out = dwo_chain if dw_out is None else dw_out.add_(dwo_chain)
        gz_c2 = ops.gemm_rows(g, wc_t, mid, aux=pre_c2.view(T, mid), aux_act=_aux_act(GELU))    # (g Wc) * gelu'(pre_c2)
        gz_c2 = gz_c2.view(B, H, W, mid)
        gz_c1 = conv_dgrad(geo2, w2, gz_c2, (B, H, W, mid), aux=pre_c1.view(B, H, W, mid), aux_act=_aux_act(GELU))
        dw2, db2 = _wg(ctx, 5, 6, geo2, w2, c1.view(B, H, W, mid), gz_c2)
        gz_c1 = gz_c1.view(T, mid)
        # d u = (g W_out + gz_c1 W1) * gelu'(pre_u) as ONE GEMM over the K-concatenation [g | gz_c1] (ops.ffn_du_operand)
        w_du = ops.ffn_du_operand(w_out, w1)
        gz_u = ops.gemm_rows2(g, gz_c1, w_du, hid, aux=pre_u, aux_act=_aux_act(GELU)) if _SAVE_DERIV else None
        if gz_u is None:
            cat = torch.cat([g, gz_c1], dim=1)
            gz_u = ops.gemm_rows(cat, w_du, hid, aux=pre_u, aux_act=_aux_act(GELU))
            del cat
        dw1, db1 = _wg(ctx, 3, 4, geo1, w1, u, gz_c1)
        dr = conv_dgrad(geo_in, w_in, gz_u, r.shape)
        dw_in, db_in = _wg(ctx, 1, 2, geo_in, w_in, r, gz_u)
        dt, _ = rownorm_bwd(t, None, dr, g, 0, eps_rms, 1e-5)                           # + residual gradient, fused
        _join(ctx, t.device)
        if deferred:
            dw3 = db3 = None
        return (dt, dw_in, db_in, dw1, db1, dw2, db2, dw3 if need[7] else None, db3 if need[8] else None,
                dw_out if need[9] else None, db_out, None, None, None, None)


# ------------------------------------------------------------------------------------------------
# Downsample / Upsample with DC path   R/transvae/modules/upsample.py:44-66, 110-128
# ------------------------------------------------------------------------------------------------
class DownsampleFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w0, b0, w2, b2, wdc, bdc):
        ops._need_gpu(x)
        dc, gdc, wdcc = None, None, None
        if wdc is not None:
            dc, _, gdc, wdcc = conv_forward(x, wdc, _c(bdc), None, "unshuf", NONE, False)
        h, pre_h, g0, w0c = conv_forward(x, w0, _c(b0), None, "c3s1", SILU, any(ctx.needs_input_grad) and _WANT)
        out, _, g2, w2c = conv_forward(h, w2, _c(b2), dc, "c3s2", NONE, False)
        ctx.geo = (g0, g2, gdc)
        _stash(ctx, (1, w0), (2, b0), (3, w2), (4, b2), (5, wdc), (6, bdc))
        ctx.save_for_backward(x, h, pre_h, w0c, w2c, wdcc)
        return out

    @staticmethod
    def backward(ctx, g):
        x, h, pre_h, w0, w2, wdc = ctx.saved_tensors
        g0, g2, gdc = ctx.geo
        g = g.contiguous()
        gz_h = conv_dgrad(g2, w2, g, h.shape, aux=pre_h, aux_act=_aux_act(SILU))
        dw2, db2 = _wg(ctx, 3, 4, g2, w2, h, g)
        dx = conv_dgrad(g0, w0, gz_h, x.shape)
        dw0, db0 = _wg(ctx, 1, 2, g0, w0, x, gz_h)
        dwdc = dbdc = None
        if wdc is not None:
            dx = conv_dgrad(gdc, wdc, g, x.shape, residual=dx)                          # DC-path gradient + main path, fused
            dwdc, dbdc = _wg(ctx, 5, 6, gdc, wdc, x, g)
        _join(ctx, x.device)
        return dx, dw0, db0, dw2, db2, dwdc, dbdc


class UpsampleFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w1, b1, w3, b3, wdc, bdc):
        ops._need_gpu(x)
        dc, gdc, wdcc = None, None, None
        if wdc is not None:
            dc, _, gdc, wdcc = conv_forward(x, wdc, _c(bdc), None, "shuf", NONE, False)
        h, pre_h, g1, w1c = conv_forward(x, w1, _c(b1), None, "c3up", SILU, any(ctx.needs_input_grad) and _WANT)
        out, _, g3, w3c = conv_forward(h, w3, _c(b3), dc, "c3s1", NONE, False)
        ctx.geo = (g1, g3, gdc)
        _stash(ctx, (1, w1), (2, b1), (3, w3), (4, b3), (5, wdc), (6, bdc))
        ctx.save_for_backward(x, h, pre_h, w1c, w3c, wdcc)
        return out

    @staticmethod
    def backward(ctx, g):
        x, h, pre_h, w1, w3, wdc = ctx.saved_tensors
        g1, g3, gdc = ctx.geo
        g = g.contiguous()
        gz_h = conv_dgrad(g3, w3, g, h.shape, aux=pre_h, aux_act=_aux_act(SILU))
        dw3, db3 = _wg(ctx, 3, 4, g3, w3, h, g)
        dx = conv_dgrad(g1, w1, gz_h, x.shape)
        dw1, db1 = _wg(ctx, 1, 2, g1, w1, x, gz_h)
        dwdc = dbdc = None
        if wdc is not None:
            dx = conv_dgrad(gdc, wdc, g, x.shape, residual=dx)
            dwdc, dbdc = _wg(ctx, 5, 6, gdc, wdc, x, g)
        _join(ctx, x.device)
        return dx, dw1, db1, dw3, db3, dwdc, dbdc
