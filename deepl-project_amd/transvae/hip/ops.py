"""torch.autograd wrappers over the C ABI of libtransvae_hip.so.

Every function here launches hand-written gfx950 kernels on the *current* HIP stream through
raw device pointers; there is no CPU or PyTorch-op fallback -- CPU tensors raise.  Activations
are bf16 NHWC ([B, H, W, C] contiguous; token matrices are [T, C]); parameters stay fp32 master
copies owned by the nn.Module and are repacked to bf16 per call (the repack of all 1.05 B
parameters costs ~2 ms, three orders of magnitude below a train step).

Layer "modes" of :class:`ConvFn` (geometry table in include/transvae_hip.h):
    linear  nn.Linear / 1x1 conv on a token matrix
    c3s1    Conv2d 3x3 stride 1 pad 1            (R/transvae/modules/blocks.py:34,37 ...)
    c3s2    Conv2d 3x3 stride 2 pad 1            (upsample.py:36)
    c3up    nearest-x2 upsample + Conv2d 3x3     (upsample.py:94-95), upsample never materialised
    unshuf  pixel_unshuffle(2) + 1x1             (upsample.py:60-61)  == 2x2 / stride-2 conv
    shuf    1x1 + pixel_shuffle(2)               (upsample.py:121-123) == GEMM with shuffled store
"""
from __future__ import annotations

import contextlib
import ctypes as C
import functools
from typing import Optional

import torch

from . import _lib as L

BF16 = torch.bfloat16


def _p(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _require(cond: bool, msg: str):
    """Operand checks that guard memory safety are real errors (they must survive `python -O`)."""
    if not cond:
        raise RuntimeError("transvae.hip: " + msg)


def hip_entry(fn):
    """Decorator for the public, reference-style entry points (Module.forward / encode / decode ...).

    * The kernels launch on the CURRENT device's current stream and the library keeps per-device state (zero page,
      LDS opt-ins): the call therefore runs under `torch.cuda.device(x.device)`, so a model moved with `.to('cuda:1')`
      works without the caller having called `torch.cuda.set_device(1)`.
    * The path has its own precision policy (bf16 storage, fp32 accumulation, fp32 parameter algebra).  Under the
      reference trainers' `torch.autocast` (R/train.py:589-593, R/train_2.py:303-312) the small parameter folds
      (W @ b, W * g) would silently turn bf16; autocast is switched off inside the call.
    """
    @functools.wraps(fn)
    def wrapper(self, x, *args, **kwargs):
        if isinstance(x, torch.Tensor) and x.is_cuda:
            with torch.cuda.device(x.device), torch.autocast("cuda", enabled=False):
                return fn(self, x, *args, **kwargs)
        return fn(self, x, *args, **kwargs)   # CPU tensors: the first op raises (no CPU path)
    return wrapper


def _need_gpu(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("transvae.hip: this op only runs on a HIP device (MI355X); there is no CPU fallback")


def _act_id(act: Optional[str]) -> int:
    return {None: L.ACT_NONE, "none": L.ACT_NONE, "gelu": L.ACT_GELU, "silu": L.ACT_SILU}[act]


# ---------------------------------------------------------------------------------------------
# raw calls
# ---------------------------------------------------------------------------------------------
_pack_cache = None   # dict while a packed_weight_cache() block is active


@contextlib.contextmanager
def packed_weight_cache():
    """Reuse the bf16 repack of a PARAMETER across the micro-batches of one optimizer step.

    Only views of (or) nn.Parameters are cached, keyed on the parameter object and its version counter, so tensors
    computed per forward (LayerNorm-folded projections) are always repacked and an in-place update invalidates the
    entry.  The cache lives for the duration of the block (transvae/parallel.py wraps the micro-batch loop).
    Gradients deferred by the collapsed Conv-FFN tail (see ffn_collapsed_operands) are flushed when the outermost block ends."""
    global _pack_cache
    prev, _pack_cache = _pack_cache, {}
    try:
        yield
    finally:
        _pack_cache = prev
        if prev is None:
            flush_deferred_grads()


# ---- bf16 operands owned by transvae.optim.FusedAdamW -------------------------------------------------------------
# The optimizer writes a bf16 copy of every weight in its update pass (same element order as the fp32 master: for conv
# weights in channels_last memory that IS the forward operand [Cout, KH*KW, Cin]) and refreshes every transposed
# (data-gradient) operand that has been asked for in one multi-tensor launch after each step.  An entry is valid while
# the parameter's autograd version and storage are the ones recorded at the last refresh: load_state_dict, init code or
# any other in-place update bumps the version and the operands fall back to per-call packing until the optimizer's next
# step.  (Updates through `param.data` bypass the version counter -- the usual caveat of `.data`.)
class _ParamOperands:
    __slots__ = ("ref", "shadow", "version", "ptr", "forms")

    def __init__(self, p, shadow):
        import weakref
        self.ref = weakref.ref(p)
        self.shadow = shadow
        self.version = p._version
        self.ptr = p.data_ptr()
        self.forms = {}          # (element offset, O, T, I, flip) -> bf16 [I, T, O]


_param_operands = {}   # id(param) -> _ParamOperands


def register_param_shadows(params, shadows: dict):
    for p in params:
        sh = shadows.get(id(p))
        if sh is not None:
            _param_operands[id(p)] = _ParamOperands(p, sh)


def _operands_of(base):
    ent = _param_operands.get(id(base))
    if ent is None:
        return None
    if ent.ref() is not base:
        del _param_operands[id(base)]
        return None
    if ent.version != base._version or ent.ptr != base.data_ptr():
        return None
    return ent


def refresh_param_operands(params):
    """After FusedAdamW's update: the bf16 copies are fresh; re-derive every registered transposed operand from them in
    one launch and record the parameters' versions."""
    forms = []
    keep = []
    total = 0
    for p in params:
        ent = _param_operands.get(id(p))
        if ent is None or ent.ref() is not p or ent.ptr != p.data_ptr():
            continue
        ent.version = p._version
        flat = ent.shadow.as_strided((p.numel(),), (1,))
        for (off, O, T, I, flip), dst in ent.forms.items():
            forms.append((flat.data_ptr() + 2 * off, dst.data_ptr(), O | (T << 32), I | (int(flip) << 32), total))
            total += ((O + 63) // 64) * ((I + 63) // 64) * T
    if not forms:
        return
    dev = params[0].device
    host = torch.tensor(forms, dtype=torch.int64).pin_memory()
    tab = host.to(dev, non_blocking=True)
    L.check(L.load().tv_pack_weight_multi(_p(tab), len(forms), total, _stream()), "tv_pack_weight_multi")
    _refresh_keep[0] = (host, tab)


_refresh_keep = [None]


def pack_weight(w: torch.Tensor, want_fwd: bool, want_t: bool, flip: bool):
    """w: fp32 [O, T, I] contiguous -> (bf16 [O,T,I] | None, bf16 [I,T,O] | None)."""
    O, T, I = w.shape
    _require(w.dtype == torch.float32 and w.is_contiguous(), "pack_weight needs a contiguous fp32 [O, T, I] tensor")
    base = w._base if w._base is not None else w
    is_param = isinstance(base, torch.nn.Parameter)
    if is_param:
        ent = _operands_of(base)
        if ent is not None:          # operands kept current by the optimizer: no kernel here
            off = (w.data_ptr() - base.data_ptr()) // 4
            if 0 <= off and off + O * T * I <= base.numel():
                d = ent.shadow.as_strided((O, T, I), (T * I, I, 1), off) if want_fwd else None
                dt = None
                if want_t:
                    fk = (off, O, T, I, bool(flip))
                    dt = ent.forms.get(fk)
                    if dt is None:   # first request: pack once here, the optimizer refreshes it from now on
                        dt = torch.empty((I, T, O), dtype=BF16, device=w.device)
                        L.check(L.load().tv_pack_weight(_p(w), None, _p(dt), O, T, I, int(flip), _stream()), "tv_pack_weight")
                        ent.forms[fk] = dt
                return d, dt
    key = None
    if _pack_cache is not None and is_param:
        key = (id(base), base._version, w.data_ptr(), O, T, I, want_fwd, want_t, flip)
        hit = _pack_cache.get(key)
        if hit is not None:
            return hit
    d = torch.empty((O, T, I), dtype=BF16, device=w.device) if want_fwd else None
    dt = torch.empty((I, T, O), dtype=BF16, device=w.device) if want_t else None
    lib = L.load()
    L.check(lib.tv_pack_weight(_p(w), _p(d), _p(dt), O, T, I, int(flip), _stream()), "tv_pack_weight")
    if key is not None:
        _pack_cache[key] = (d, dt)
    return d, dt


def _derived_weight(w: torch.Tensor, kind: str, build):
    """Operands DERIVED from a parameter by a chain of small torch ops (polyphase / parity forms): same cache, same key
    rule as pack_weight -- they cost 50-100 launch-bound kernels each, once per optimizer step instead of per micro-batch."""
    if _pack_cache is None:
        return build()
    base = w._base if w._base is not None else w
    if not isinstance(base, torch.nn.Parameter):
        return build()
    key = (id(base), base._version, w.data_ptr(), tuple(w.shape), kind)
    hit = _pack_cache.get(key)
    if hit is None:
        hit = _pack_cache[key] = build()
    return hit


def _desc(**kw) -> L.ConvDesc:
    d = L.ConvDesc()
    for k in ("up_shift", "dil_mask", "act", "store_shuffle", "pad"):
        setattr(d, k, 0)
    d.stride = 1
    for k, v in kw.items():
        setattr(d, k, int(v))
    return d


# ---- launches over more than 2 GiB of activations --------------------------------------------------------------------------
# The fast staging paths of the kernels address an operand through a buffer descriptor with 32-bit byte offsets, i.e. one
# launch sees < 2 GiB of each activation tensor (a micro-batch of 64 Large images at 256 x 256 is 1.6 GB per 192-channel
# tensor; 128 images are 3.2 GB).  Images are independent through every layer, so a larger tensor is simply launched in
# batch chunks -- forward and data gradient chunk by chunk, the weight gradient with the later chunks ADDING
# (tv_wgrad_tn_acc) -- instead of dropping to the generic 64-bit addressing path of the kernels (2-3x slower).
_LAUNCH_BYTES = (1 << 31) - (1 << 26)


def _batch_chunks(batch: int, tensors, align: int = 1):
    """None when one launch covers it, else [(first row, rows)] with every listed tensor's slice below the launch limit.
    `tensors` hold `batch` leading rows each (images, or tokens for 'linear'); chunks are multiples of `align` rows."""
    per_row = 0
    for t in tensors:
        if t is not None:
            per_row = max(per_row, (t.numel() // batch) * t.element_size())
    if per_row * batch < _LAUNCH_BYTES:
        return None
    rows = max(align, (_LAUNCH_BYTES // per_row) // align * align)
    n = -(-batch // rows)
    rows = -(-(-(-batch // n)) // align) * align          # even chunks
    return [(s, min(rows, batch - s)) for s in range(0, batch, rows)]


def _row_ptr(t, batch: int, row: int):
    """device pointer of row `row` of a tensor with `batch` leading rows (None stays None)"""
    if t is None:
        return None
    return C.c_void_p(t.data_ptr() + row * (t.numel() // batch) * t.element_size())


def _desc_rows(desc: L.ConvDesc, rows: int) -> L.ConvDesc:
    d = L.ConvDesc()
    C.memmove(C.byref(d), C.byref(desc), C.sizeof(L.ConvDesc))
    d.batch = rows
    return d


def igemm(desc: L.ConvDesc, x, w, bias, residual, pre, out):
    lib = L.load()
    ch = _batch_chunks(desc.batch, (x, residual, pre, out))
    if ch is None:
        L.check(lib.tv_igemm_nt(C.byref(desc), _p(x), _p(w), _p(bias), _p(residual), _p(pre), _p(out), _stream()), "tv_igemm_nt")
        return
    B = desc.batch
    for s, c in ch:
        L.check(lib.tv_igemm_nt(C.byref(_desc_rows(desc, c)), _row_ptr(x, B, s), _p(w), _p(bias), _row_ptr(residual, B, s),
                                _row_ptr(pre, B, s), _row_ptr(out, B, s), _stream()), "tv_igemm_nt")


def wgrad(desc: L.ConvDesc, x, gy, dw, dbias, _acc: bool = False):
    """dw (and dbias) of one layer.  Chunked launches: the first chunk follows the kernel's own overwrite / accumulate
    plan for ITS geometry (callers size-check with wgrad_plan_desc), later chunks add."""
    lib = L.load()
    ch = _batch_chunks(desc.batch, (x, gy))
    fn_first = lib.tv_wgrad_tn_acc if _acc else lib.tv_wgrad_tn
    if ch is None:
        L.check(fn_first(C.byref(desc), _p(x), _p(gy), _p(dw), _p(dbias), _stream()), "tv_wgrad_tn")
        return
    B = desc.batch
    for i, (s, c) in enumerate(ch):
        fn = fn_first if i == 0 else lib.tv_wgrad_tn_acc
        L.check(fn(C.byref(_desc_rows(desc, c)), _row_ptr(x, B, s), _row_ptr(gy, B, s), _p(dw), _p(dbias), _stream()), "tv_wgrad_tn")


def wgrad_acc(desc: L.ConvDesc, x, gy, dw, dbias):
    """dw += ..., dbias += ...  (the buffers hold the gradients of earlier micro-batches)"""
    wgrad(desc, x, gy, dw, dbias, _acc=True)


def wgrad_plan_desc(desc: L.ConvDesc, x, gy) -> L.ConvDesc:
    """The descriptor of the FIRST launch of wgrad(desc, x, gy, ...): what tv_wgrad_tn_overwrites must be asked about."""
    ch = _batch_chunks(desc.batch, (x, gy))
    return desc if ch is None else _desc_rows(desc, ch[0][1])


# In-place gradient accumulation across the micro-batches of one optimizer step (transvae.parallel.train_step switches it on
# for every micro-batch after the first, and never for one whose backward must fire DDP's reduction hooks): the block
# Functions then add their weight / bias gradients straight into `param.grad` and hand autograd nothing to accumulate.
_accum_grads = False
acc_stats = {"in_place": 0, "autograd": 0}     # weight gradients added in place / handed to autograd (diagnostics, tests)
in_place_params = set()    # id(parameter) of every weight / bias a kernel has added into in place (transvae.parallel reads it)


_defer_chain = False     # the Conv-FFN composite's chain rule waits for flush_deferred_grads() (set per micro-batch by train_step)


@contextlib.contextmanager
def defer_chain_grads(on: bool = True):
    """Inside: ConvFFNBranchFn only accumulates the composite's gradient G; the chain rule onto W_out / W3 / b3 runs once, in
    flush_deferred_grads() (end of the surrounding packed_weight_cache() block, or of this block when there is none).  For
    backward passes after which nobody reads those gradients before the flush -- i.e. not the pass DDP reduces in."""
    global _defer_chain
    prev, _defer_chain = _defer_chain, bool(on)
    try:
        yield
    finally:
        _defer_chain = prev
        if _pack_cache is None:
            flush_deferred_grads()


@contextlib.contextmanager
def accumulate_grads_in_place(on: bool = True):
    global _accum_grads
    prev, _accum_grads = _accum_grads, bool(on)
    try:
        yield
    finally:
        _accum_grads = prev
        if _pack_cache is None:      # (inside a packed_weight_cache() block -- one optimizer step -- the flush waits for its end)
            flush_deferred_grads()


def param_of(t: Optional[torch.Tensor]):
    """The nn.Parameter a weight / bias input is (a view of), or None (derived tensors: folds, pads, concatenations)."""
    if t is None:
        return None
    base = t._base if t._base is not None else t
    return base if isinstance(base, torch.nn.Parameter) else None


def grad_views(pw, w: torch.Tensor, pb, need_b: bool, force: bool = False):
    """(dw buffer laid out like `w` inside pw.grad, bias gradient buffer | None) when both can be accumulated in place.
    force: also outside the in-place mode (the flush of deferred gradients adds into gradients that exist)."""
    if not (_accum_grads or force):
        return None
    why = None
    if pw is None or pw.grad is None:
        why = "no parameter / no gradient yet"
    else:
        gw = pw.grad
        off = w.storage_offset() - pw.storage_offset()
        if gw.dtype != torch.float32 or gw.stride() != pw.stride():
            why = f"grad layout {gw.dtype} {gw.stride()} vs {pw.stride()}"
        elif w.numel() != pw.numel() or not w.is_contiguous() or off < 0:
            why = "weight view does not cover the parameter"
        elif w.untyped_storage().data_ptr() != pw.untyped_storage().data_ptr():
            why = "weight is a repacked copy, not a view"
        elif need_b and (pb is None or pb.grad is None or pb.grad.dtype != torch.float32 or not pb.grad.is_contiguous()):
            why = "bias gradient not available in place"
    if why is not None:
        if _ACC_DEBUG:
            print("[transvae.hip] in-place accumulation skipped:", why, tuple(w.shape), flush=True)
        return None
    dw = gw.as_strided(tuple(w.shape), tuple(w.stride()), gw.storage_offset() + off)
    in_place_params.add(id(pw))
    if need_b:
        in_place_params.add(id(pb))
    return dw, (pb.grad if need_b else None)


_ACC_DEBUG = __import__("os").environ.get("TV_DEBUG_ACC") == "1"


_ones_cache = {}


def _ones(n: int, device) -> torch.Tensor:
    key = (device, )
    t = _ones_cache.get(key)
    if t is None or t.shape[0] < n:
        t = torch.ones((n, 8), dtype=BF16, device=device)
        _ones_cache[key] = t
    return t


# ---------------------------------------------------------------------------------------------
# convolution / linear
# ---------------------------------------------------------------------------------------------
class _Geo:
    """Shapes of one layer instance, derived from mode + tensor shapes."""

    def __init__(self, mode: str, x: torch.Tensor, w: torch.Tensor):
        self.mode = mode
        if mode == "linear":
            _require(x.dim() == 2 and w.dim() == 2, "operand check failed: x.dim() == 2 and w.dim() == 2")
            self.B, self.H, self.W, self.Cin = x.shape[0], 1, 1, x.shape[1]
            self.Cout, self.KH, self.KW = w.shape[0], 1, 1
            self.Ho, self.Wo = 1, 1
            self.out_shape = (self.B, self.Cout)
        else:
            _require(x.dim() == 4 and w.dim() == 4, "operand check failed: " + repr((mode, x.shape, w.shape)))
            self.B, self.H, self.W, self.Cin = x.shape
            self.Cout, self.KH, self.KW = w.shape[0], w.shape[1], w.shape[2]
            if mode == "c3s1":
                self.Ho, self.Wo = self.H, self.W
            elif mode in ("c3s2", "unshuf"):
                _require(self.H % 2 == 0 and self.W % 2 == 0, "operand check failed: self.H % 2 == 0 and self.W % 2 == 0")
                self.Ho, self.Wo = self.H // 2, self.W // 2
            elif mode == "c3up":
                self.Ho, self.Wo = 2 * self.H, 2 * self.W
            elif mode == "shuf":
                _require(self.KH == 1 and self.KW == 1 and self.Cout % 4 == 0, "operand check failed: self.KH == 1 and self.KW == 1 and self.Cout % 4 == 0")
                self.Ho, self.Wo = self.H, self.W          # GEMM grid; stored to [B,2H,2W,Cout/4]
            else:
                raise ValueError(f"unknown conv mode {mode}")
            if mode == "shuf":
                self.out_shape = (self.B, 2 * self.H, 2 * self.W, self.Cout // 4)
            else:
                self.out_shape = (self.B, self.Ho, self.Wo, self.Cout)
        _require(w.shape[-1] == self.Cin, "operand check failed: " + repr((mode, tuple(x.shape), tuple(w.shape))))
        exp_k = {"linear": (1, 1), "c3s1": (3, 3), "c3s2": (3, 3), "c3up": (3, 3), "unshuf": (2, 2), "shuf": (1, 1)}[mode]
        _require((self.KH, self.KW) == exp_k, "operand check failed: " + repr((mode, tuple(w.shape))))

    def fwd_desc(self, act: int) -> L.ConvDesc:
        m = self.mode
        common = dict(batch=self.B, h_in=self.H, w_in=self.W, c_in=self.Cin, ldx=self.Cin,
                      h_out=self.Ho, w_out=self.Wo, c_out=self.Cout, ldo=self.Cout,
                      kh=self.KH, kw=self.KW, act=act)
        if m == "linear":
            return _desc(**common)
        if m == "c3s1":
            return _desc(**common, stride=1, pad=1)
        if m == "c3s2":
            return _desc(**common, stride=2, pad=1)
        if m == "c3up":
            return _desc(**common, stride=1, pad=1, up_shift=1)
        if m == "unshuf":
            return _desc(**common, stride=2, pad=0)
        if m == "shuf":
            common["ldo"] = self.Cout // 4
            return _desc(**common, store_shuffle=1)
        raise ValueError(m)


# ---- the three passes of one layer as plain functions (shared by ConvFn and the fused block Functions) ----
def conv_forward(x, w, bias, residual, mode: str, act_id: int, want_pre, rope=None):
    """Returns (out, pre_activation | None, geometry, contiguous fp32 weight).
    rope = (table [N, 4, 32] fp32, tokens per image N, columns to rotate): 'linear' only -- the QKV projection with RoPE in
    its epilogue (tv_igemm_nt_rope).
    want_pre = "deriv": the second tensor is act'(pre-activation) instead (pass it to conv_dgrad with
    aux_act = L.ACT_DERIV); falsy: nothing is saved."""
    _need_gpu(x, w)
    _require(x.dtype == BF16 and x.is_contiguous(), "activations must be contiguous bf16 NHWC")
    _require(w.dtype == torch.float32, f"weights must be fp32 master copies, got {w.dtype} (parameter algebra under autocast?)")
    _require(x.device == w.device and x.device.index == torch.cuda.current_device(),
             f"tensors on {x.device} / {w.device} but the current device is cuda:{torch.cuda.current_device()}")
    w = w.contiguous()
    g = _Geo(mode, x, w)
    out = torch.empty(g.out_shape, dtype=BF16, device=x.device)
    pre = torch.empty_like(out) if (want_pre and act_id != L.ACT_NONE) else None
    if pre is not None and want_pre == "deriv":
        act_id |= L.ACT_SAVE_DERIV
    if residual is not None:
        _require(residual.shape == out.shape and residual.dtype == BF16 and residual.is_contiguous(),
                 "residual must be contiguous bf16 of the output's shape")
    if bias is not None:
        _require(bias.dtype == torch.float32 and bias.is_contiguous() and bias.numel() == g.Cout,
                 f"bias must be contiguous fp32 [{g.Cout}], got {bias.dtype} {tuple(bias.shape)}")
    if mode == "c3up":   # polyphase: one 2x2-footprint GEMM on the (H+1) x (W+1) cell grid, four phases as column quadrants
        wb = _derived_weight(w, "up_fwd", lambda: _up_fwd_weight(w, g.Cout, g.Cin))
        d = _desc(batch=g.B, h_in=g.H, w_in=g.W, c_in=g.Cin, ldx=g.Cin, h_out=g.H + 1, w_out=g.W + 1, c_out=4 * g.Cout,
                  ldo=g.Cout, kh=2, kw=2, stride=1, pad=1, act=act_id, store_shuffle=2)
        igemm(d, x, wb, bias.repeat(4) if bias is not None else None, residual, pre, out)
        return out, pre, g, w
    wb, _ = pack_weight(w.view(g.Cout, g.KH * g.KW, g.Cin), True, False, False)
    if rope is not None:
        tab, tokens, cols = rope
        _require(mode == "linear" and residual is None and pre is None and act_id == L.ACT_NONE, "rope epilogue: plain projection only")
        _require(tab.dtype == torch.float32 and tab.is_contiguous() and tuple(tab.shape) == (tokens, 4, 32) and g.B % tokens == 0,
                 "rope table must be contiguous fp32 [tokens, 4, 32] and the rows whole images")
        d = g.fwd_desc(act_id)
        ch = _batch_chunks(d.batch, (x, out), align=int(tokens)) or [(0, d.batch)]
        for s0, c0 in ch:
            L.check(L.load().tv_igemm_nt_rope(C.byref(_desc_rows(d, c0)), _row_ptr(x, d.batch, s0), _p(wb), _p(bias), _row_ptr(out, d.batch, s0),
                                              _p(tab), int(tokens), int(cols), _stream()), "tv_igemm_nt_rope")
        return out, pre, g, w
    igemm(g.fwd_desc(act_id), x, wb, bias, residual, pre, out)
    return out, pre, g, w


def _igemm_bwd(desc, gz, wt, residual, aux, aux_act, dx):
    """dx = conv(gz, wt) [+ residual] [* act'(aux)]"""
    if aux is None:
        igemm(desc, gz, wt, None, residual, None, dx)
    else:
        lib = L.load()
        B = desc.batch
        for s, c in (_batch_chunks(B, (gz, residual, aux, dx)) or [(0, B)]):
            L.check(lib.tv_igemm_nt_actgrad(C.byref(_desc_rows(desc, c)), _row_ptr(gz, B, s), _p(wt), _row_ptr(residual, B, s),
                                            _row_ptr(aux, B, s), aux_act, _row_ptr(dx, B, s), _stream()), "tv_igemm_nt_actgrad")


# nearest-x2 upsample followed by a 3x3 / pad-1 convolution, in polyphase form.  Output row Y = 2y' - py of the [2H]
# grid reads the input rows y'-1, y' of the [H] grid (cell y' of the (H+1)-cell grid of neighbouring row pairs), because
# upsampled rows 2r, 2r+1 are both input row r:
#     py = 0:  row y'-1 gets w[ky=0],        row y' gets w[1] + w[2]
#     py = 1:  row y'-1 gets w[0] + w[1],    row y' gets w[2]                       (and the same along x)
# so each output phase is a 2x2 convolution of the INPUT: 4 taps per output pixel instead of 9, and nothing of the
# upsampled tensor is ever read.  _UP_SETS[p][t] = the ky folded into tap t of phase p.
_UP_SETS = (((0,), (1, 2)), ((0, 1), (2,)))


def _derive(w: torch.Tensor, form: int, Cout: int, Cin: int, out: torch.Tensor, accumulate: bool = False) -> torch.Tensor:
    """One tv_conv3x3_derived launch (csrc/elementwise.hip: tapsum_kernel); w fp32 contiguous in its [*, taps, *] layout."""
    _require(w.dtype == torch.float32 and w.is_contiguous() and out.is_contiguous(), "derived operand needs contiguous fp32 input")
    L.check(L.load().tv_conv3x3_derived(_p(w), _p(out), form, Cout, Cin, int(accumulate), _stream()), "tv_conv3x3_derived")
    return out


def _up_fwd_weight(w: torch.Tensor, Cout: int, Cin: int) -> torch.Tensor:
    """Forward operand: bf16 [4*Cout, 2, 2, Cin], row (2*py+px)*Cout + co, tap (ty, tx)."""
    return _derive(w.view(Cout, 3, 3, Cin), L.DERIVE_UP_FWD, Cout, Cin, torch.empty((4 * Cout, 2, 2, Cin), dtype=BF16, device=w.device))


# The adjoint is a 4x4 / stride-2 / pad-1 convolution of the high-resolution gradient: input row r receives output rows
# Y = 2r + d, d = -1 .. 2 (tap t = d + 1), through  d = -1: w[2],  d = 0: w[1] + w[2],  d = 1: w[0] + w[1],  d = 2: w[0].
_UP_ADJ = ((2,), (1, 2), (0, 1), (0,))


def _up_dgrad_weight(w: torch.Tensor, Cout: int, Cin: int) -> torch.Tensor:
    """Data-gradient operand: bf16 [Cin, 4, 4, Cout]."""
    return _derive(w.view(Cout, 3, 3, Cin), L.DERIVE_UP_DGRAD, Cout, Cin, torch.empty((Cin, 4, 4, Cout), dtype=BF16, device=w.device))


def _up_fold_wgrad(d16: torch.Tensor, Cout: int, Cin: int, out: Optional[torch.Tensor] = None, accumulate: bool = False) -> torch.Tensor:
    """[Cin, 4, 4, Cout] gradient of the 4x4 adjoint operand -> [Cout, 3, 3, Cin] gradient of the 3x3 weight
    (tap t of the adjoint contains w[ky] for ky in _UP_ADJ[t], so dw[ky] sums the taps that contain it: ((2,3), (1,2), (0,1))).
    out + accumulate: add into an existing gradient buffer (in-place accumulation over micro-batches)."""
    if out is None:
        out = torch.empty((Cout, 3, 3, Cin), dtype=torch.float32, device=d16.device)
    return _derive(d16, L.DERIVE_UP_WGRAD_FOLD, Cout, Cin, out, accumulate)


def _s2_parity_weight(w: torch.Tensor, Cout: int, Cin: int) -> torch.Tensor:
    """Data-gradient operand of a 3x3 / stride-2 / pad-1 convolution, by output parity (see conv_dgrad):
    bf16 [4*Cin, 2, 2, Cout];  row (2*py+px)*Cin + ci, tap (ty, tx) holds w[:, ky, kx, ci] with ky = 1 for py = 0 and
    ky = 2, 0 for ty = 0, 1 when py = 1 (kx likewise); taps outside the class's footprint stay zero."""
    return _derive(w.view(Cout, 3, 3, Cin), L.DERIVE_S2_PARITY, Cout, Cin, torch.empty((4 * Cin, 2, 2, Cout), dtype=BF16, device=w.device))


# ---- Conv-FFN tail in collapsed form ---------------------------------------------------------------------------------------
# R/transvae/modules/conv.py:85-104:  u <- u + W3 c + b3 ;  out = W_out u + b_out.  Nothing non-linear sits between the two
# projections, so   out = W_out u + (W_out W3) c + (W_out b3 + b_out) :  the [T, 4d] tensor u + W3 c is never formed, the
# d -> 4d GEMM (and, in the backward pass, its data gradient from the [T, 4d] gradient and its weight gradient) becomes a
# d -> d GEMM on the composite  Wc = W_out W3  [d, mid]  (a quarter of the FLOPs, none of the [T, 4d] passes), and the chain
# rule hands the composite's gradient  G = g^T c  [d, mid]  (g = d out) back to the parameters with two small products:
#     dW_out += G W3^T      dW3 = W_out^T G      db3 = W_out^T sum_t g      db_out = sum_t g
# Wc / bc are operands derived from parameters (built once per optimizer step inside a packed_weight_cache() block, like
# the polyphase forms); G accumulates over the in-place micro-batches of a step and the two products run once, at its end
# (flush_deferred_grads, called by transvae.parallel.train_step through the packed_weight_cache() block, and by
# accumulate_grads_in_place() when no such block is open).  Outside the in-place mode the Function applies the chain rule
# at once, so a plain loss.backward() / optimizer.step() loop (R/train.py:577-620) gets complete gradients.
def _rows_desc(rows: int, c_in: int, c_out: int) -> L.ConvDesc:
    return _desc(batch=rows, h_in=1, w_in=1, c_in=c_in, ldx=c_in, h_out=1, w_out=1, c_out=c_out, ldo=c_out, kh=1, kw=1)


def gemm_rows(x: torch.Tensor, wb: torch.Tensor, c_out: int, bias=None, residual=None, aux=None, aux_act: int = 0) -> torch.Tensor:
    """out[T, c_out] = x[T, K] wb[c_out, K]^T (+ bias) (+ residual) (x act'(aux)); wb is a packed bf16 operand."""
    T, K = x.shape
    out = torch.empty((T, c_out), dtype=BF16, device=x.device)
    d = _rows_desc(T, K, c_out)
    if aux is None:
        igemm(d, x, wb, bias, residual, None, out)
    else:
        _require(bias is None, "gemm_rows: bias and a derivative multiply do not combine")
        _igemm_bwd(d, x, wb, residual, aux, aux_act, out)
    return out


def gemm_rows2(x1: torch.Tensor, x2: torch.Tensor, wb: torch.Tensor, c_out: int, bias=None, residual=None, aux=None, aux_act: int = 0):
    """out[T, c_out] = [x1 | x2] wb[c_out, K1 + K2]^T (+ bias) (+ residual) (aux as in gemm_rows) WITHOUT forming the
    concatenation (tv_igemm_nt_cat2: the eight-phase loop reads K-steps beyond K1 from x2).  None when the shape's tile has
    no two-source loop (small problems): the caller takes its own fallback."""
    T, K1 = x1.shape
    K2 = x2.shape[1]
    _require(x1.dtype == BF16 and x2.dtype == BF16 and x1.is_contiguous() and x2.is_contiguous() and x2.shape[0] == T and
             tuple(wb.shape[-2:]) == (c_out, K1 + K2) or wb.numel() == c_out * (K1 + K2), "gemm_rows2: operand check failed")
    if K1 % 64 or K2 % 64 or _batch_chunks(T, (x1, x2, residual, aux)) is not None:
        return None
    d = _rows_desc(T, K1 + K2, c_out)
    d.ldx = K1
    out = torch.empty((T, c_out), dtype=BF16, device=x1.device)
    rc = L.load().tv_igemm_nt_cat2(C.byref(d), _p(x1), _p(x2), K1, K2, _p(wb), _p(bias), _p(residual), _p(aux), int(aux_act), _p(out), _stream())
    if rc == 4:          # TV_ERR_UNSUPPORTED
        return None
    L.check(rc, "tv_igemm_nt_cat2")
    return out


def ffn_collapsed_operands(w_out: torch.Tensor, w3: torch.Tensor, b3, b_out):
    """(Wc bf16 [d, mid], Wc^T bf16 [mid, d], bc fp32 [d], [W_out | Wc] bf16 [d, hid + mid]) for w_out [d, hid], w3 [hid, mid]
    fp32 parameter views."""
    d, hid = w_out.shape
    mid = w3.shape[1]

    def build():
        wo_f, _ = pack_weight(w_out.view(d, 1, hid), True, False, False)        # bf16 [d, hid]
        _, w3_t = pack_weight(w3.view(hid, 1, mid), False, True, False)         # bf16 [mid, hid]
        wc_f = gemm_rows(wo_f.view(d, hid), w3_t.view(mid, hid), mid)           # Wc[o, j] = sum_h W_out[o, h] W3[h, j]
        wc_t = gemm_rows(w3_t.view(mid, hid), wo_f.view(d, hid), d)             # its transpose (data-gradient operand)
        if b3 is not None:
            bc = torch.mv(w_out, b3) if b_out is None else torch.addmv(b_out, w_out, b3)
        else:
            bc = b_out
        return wc_f, wc_t, (bc.contiguous() if bc is not None else None), torch.cat([wo_f.view(d, hid), wc_f], dim=1)
    if _pack_cache is None:
        return build()
    bo, b3b = param_of(w_out), param_of(w3)
    if bo is None or b3b is None:
        return build()
    key = ("ffn_collapsed", id(bo), bo._version, id(b3b), b3b._version, w_out.data_ptr(), w3.data_ptr(),
           None if b3 is None else (b3.data_ptr(), b3._version), None if b_out is None else (b_out.data_ptr(), b_out._version))
    hit = _pack_cache.get(key)
    if hit is None:
        hit = _pack_cache[key] = build()
    return hit


def ffn_du_operand(w_out: torch.Tensor, w1: torch.Tensor) -> torch.Tensor:
    """bf16 [hid, d + mid] = [W_out^T | W1^T]: the operand of the ONE data-gradient GEMM onto u,
    d u = [g | gz_c1] [W_out ; W1]  (K = d + mid), instead of g W_out written as a [T, 4d] tensor and read back as the
    residual of gz_c1 W1.  Built from the optimizer-kept transposed forms; cached per step inside packed_weight_cache()."""
    d, hid = w_out.shape
    mid = w1.shape[0]

    def build():
        _, wo_t = pack_weight(w_out.view(d, 1, hid), False, True, False)        # bf16 [hid, 1, d]
        _, w1_t = pack_weight(w1.view(mid, 1, hid), False, True, False)         # bf16 [hid, 1, mid]
        return torch.cat([wo_t.view(hid, d), w1_t.view(hid, mid)], dim=1)
    bo, b1 = param_of(w_out), param_of(w1)
    if _pack_cache is None or bo is None or b1 is None:
        return build()
    key = ("ffn_du", id(bo), bo._version, id(b1), b1._version, w_out.data_ptr(), w1.data_ptr())
    hit = _pack_cache.get(key)
    if hit is None:
        hit = _pack_cache[key] = build()
    return hit


def ffn_chain_grads(G: torch.Tensor, gs, w_out: torch.Tensor, w3: torch.Tensor, b3=None, out_w=None, out_w3=None):
    """The composite's gradient back to the parameters: (dW_out chain part [d, hid], dW3 [hid, mid], db3 [hid] | None), fp32.
    G fp32 [d, mid], gs = sum_t g fp32 [d] | None (with b3: the rank-one term  sum_t g (x) b3  of dW_out, and db3).
    out_w / out_w3: gradient buffers to ADD into (in place) instead.  G enters the two products as a bf16 hi + lo pair
    (G = bf16(G) + bf16(G - bf16(G)) to 2^-17): the products keep the precision of an fp32 gradient against bf16 weights."""
    d, hid = w_out.shape
    mid = w3.shape[1]
    wo_f, _ = pack_weight(w_out.view(d, 1, hid), True, False, False)            # bf16 [d, hid]
    _, w3_t = pack_weight(w3.view(hid, 1, mid), False, True, False)             # bf16 [mid, hid]
    d1 = _rows_desc(mid, hid, d)     # dW_out[o, h] += sum_j G[o, j] W3[h, j]: over the mid "pixels" j, gy = G^T [mid, d], x = W3^T [mid, hid]
    d2 = _rows_desc(d, mid, hid)     # dW3[h, j]     = sum_o W_out[o, h] G[o, j]: over the d "pixels" o, gy = W_out [d, hid], x = G [d, mid]
    dwo = out_w if out_w is not None else zeros_f32((d, hid), G.device)
    dw3 = out_w3 if out_w3 is not None else zeros_f32((hid, mid), G.device)
    part = G
    for _ in range(2):
        Gb, GbT = pack_weight(part.view(d, 1, mid), True, True, False)           # bf16 [d, mid], [mid, d]
        wgrad_acc(d1, w3_t.view(mid, hid), GbT.view(mid, d), dwo, None)
        wgrad_acc(d2, Gb.view(d, mid), wo_f.view(d, hid), dw3, None)
        part = part - Gb.view(d, mid).float()
    db3 = None
    if gs is not None and b3 is not None:
        dwo.addr_(gs, b3)                                                        # u + W3 c + b3: the bias rides along with u
        db3 = torch.mv(w_out.t(), gs)
    return dwo, dw3, db3


_deferred_ffn = {}    # id(W_out parameter) -> dict(G, gs, w_out, w3, refs of the parameters)


def defer_ffn_grad(pw_out, pw3, pb3, w_out, w3, b3, need_b3: bool):
    """The accumulators (G fp32 [d, mid], gs fp32 [d]) of one Conv-FFN for the in-place micro-batches of a step."""
    import weakref
    ent = _deferred_ffn.get(id(pw_out))
    if ent is None or ent["pw_out"]() is not pw_out:
        d, hid = w_out.shape
        ent = dict(G=zeros_f32((d, w3.shape[1]), w_out.device), gs=zeros_f32((d,), w_out.device), w_out=w_out, w3=w3, b3=b3, need_b3=need_b3,
                   pw_out=weakref.ref(pw_out), pw3=weakref.ref(pw3), pb3=weakref.ref(pb3) if pb3 is not None else None)
        _deferred_ffn[id(pw_out)] = ent
        in_place_params.add(id(pw3))
        if pb3 is not None:
            in_place_params.add(id(pb3))
    return ent


def take_deferred_ffn(pw_out):
    """Pending accumulators of this Conv-FFN (None if there are none): the caller folds them into the gradients it returns."""
    ent = _deferred_ffn.get(id(pw_out)) if pw_out is not None else None
    if ent is None or ent["pw_out"]() is not pw_out:
        return None
    del _deferred_ffn[id(pw_out)]
    return ent


def flush_deferred_grads():
    """Apply the chain rule of every pending composite gradient and ADD the results into the parameters' gradients."""
    if not _deferred_ffn:
        return
    ents = list(_deferred_ffn.values())
    _deferred_ffn.clear()
    with torch.no_grad():
        for ent in ents:
            pw_out, pw3 = ent["pw_out"](), ent["pw3"]()
            pb3 = ent["pb3"]() if ent["pb3"] is not None else None
            if pw_out is None or pw3 is None:
                continue
            for pp in (pw_out, pw3):      # (a pass that deferred on the step's first micro-batch: no gradient of W3 exists yet)
                if pp.grad is None:
                    pp.grad = torch.zeros_like(pp, memory_format=torch.preserve_format)
            vo = grad_views(pw_out, ent["w_out"], None, False, force=True)
            v3 = grad_views(pw3, ent["w3"], None, False, force=True)
            dwo, dw3, db3 = ffn_chain_grads(ent["G"], ent["gs"] if ent["b3"] is not None else None, ent["w_out"], ent["w3"], ent["b3"],
                                            out_w=vo[0] if vo is not None else None, out_w3=v3[0] if v3 is not None else None)
            if not (pb3 is not None and ent["need_b3"]):
                db3 = None
            if vo is None:       # (no gradient buffer of the expected layout: fall back to torch's accumulation)
                g_ = dwo.view_as(ent["w_out"])
                pw_out.grad = g_.clone().view_as(pw_out) if pw_out.grad is None else pw_out.grad.add_(g_.reshape(pw_out.shape))
            if v3 is None:
                g_ = dw3.reshape(pw3.shape)
                pw3.grad = g_.clone() if pw3.grad is None else pw3.grad.add_(g_)
            if db3 is not None:
                pb3.grad = db3 if pb3.grad is None else pb3.grad.add_(db3)


def conv_dgrad(g: _Geo, w, gz, x_shape, residual=None, aux=None, aux_act: int = 0):
    """Gradient w.r.t. the layer input.  Optional fusions (one kernel, no extra pass):
    residual: a second gradient of the same tensor to add;  aux/aux_act: multiply by act'(aux), i.e. return the
    gradient w.r.t. the pre-activation `aux` of the layer that produced this layer's input."""
    lib = L.load()
    m = g.mode
    T = g.KH * g.KW
    dev = gz.device
    if m == "unshuf":  # GEMM rows n = (dy,dx,c): transpose the flattened [Cout, 4*Cin] matrix
        _, wt = pack_weight(w.view(g.Cout, 1, T * g.Cin), False, True, False)
    elif m in ("c3s2", "c3up"):
        wt = None      # (parity / polyphase formulations below build their own operands)
    else:              # [Cin][taps (reversed for 3x3)][Cout]
        _, wt = pack_weight(w.view(g.Cout, T, g.Cin), False, True, m == "c3s1")
    dx = torch.empty(x_shape, dtype=BF16, device=dev)
    if m == "linear":
        d = _desc(batch=g.B, h_in=1, w_in=1, c_in=g.Cout, ldx=g.Cout, h_out=1, w_out=1, c_out=g.Cin, ldo=g.Cin, kh=1, kw=1)
        _igemm_bwd(d, gz, wt, residual, aux, aux_act, dx)
    elif m == "c3s1":
        d = _desc(batch=g.B, h_in=g.H, w_in=g.W, c_in=g.Cout, ldx=g.Cout, h_out=g.H, w_out=g.W, c_out=g.Cin, ldo=g.Cin,
                  kh=3, kw=3, stride=1, pad=1)
        _igemm_bwd(d, gz, wt, residual, aux, aux_act, dx)
    elif m == "c3s2":   # (sizes are even: _Geo asserts it)
        # By output parity: dx[2y+py, 2x+px] only sees the taps with ky = 1 (py = 0) or ky in {2, 0} at gz rows y, y+1
        # (py = 1), likewise in x.  One GEMM over the low-resolution grid with a 2x2 footprint and 4*Cin columns -- class
        # (py, px) in column quadrant 2*py+px, its unused taps zero -- stored pixel-shuffled: 16 tap-GEMMs instead of the
        # 36 of a zero-dilated 3x3 on the full-resolution grid (9 are the algorithmic minimum).
        wd = _derived_weight(w, "s2_parity", lambda: _s2_parity_weight(w, g.Cout, g.Cin))
        d = _desc(batch=g.B, h_in=g.Ho, w_in=g.Wo, c_in=g.Cout, ldx=g.Cout, h_out=g.Ho, w_out=g.Wo, c_out=4 * g.Cin, ldo=g.Cin,
                  kh=2, kw=2, stride=1, pad=0, store_shuffle=1)
        _igemm_bwd(d, gz, wd, residual, aux, aux_act, dx)
    elif m == "c3up":   # adjoint of the polyphase form: 4x4 / stride-2 / pad-1 convolution of the high-resolution gradient
        wd = _derived_weight(w, "up_dgrad", lambda: _up_dgrad_weight(w, g.Cout, g.Cin))
        d = _desc(batch=g.B, h_in=g.Ho, w_in=g.Wo, c_in=g.Cout, ldx=g.Cout, h_out=g.H, w_out=g.W, c_out=g.Cin, ldo=g.Cin,
                  kh=4, kw=4, stride=2, pad=1)
        _igemm_bwd(d, gz, wd, residual, aux, aux_act, dx)
    elif m == "unshuf":
        # dx[b,2oy+dy,2ox+dx,c] = sum_co gz[b,oy,ox,co] W[co,dy,dx,c]  -> GEMM with shuffled store
        d = _desc(batch=g.B, h_in=g.Ho, w_in=g.Wo, c_in=g.Cout, ldx=g.Cout, h_out=g.Ho, w_out=g.Wo, c_out=4 * g.Cin, ldo=g.Cin,
                  kh=1, kw=1, store_shuffle=1)
        _igemm_bwd(d, gz, wt, residual, aux, aux_act, dx)
    elif m == "shuf":
        # dx[p,ci] = sum_{q,c} gz_hi[pix(p,q),c] W[(q,c),ci]  -> 2x2/stride-2 gather conv over gz_hi
        cq = g.Cout // 4
        d = _desc(batch=g.B, h_in=2 * g.H, w_in=2 * g.W, c_in=cq, ldx=cq, h_out=g.H, w_out=g.W, c_out=g.Cin, ldo=g.Cin,
                  kh=2, kw=2, stride=2, pad=0)
        _igemm_bwd(d, gz, wt, residual, aux, aux_act, dx)
    else:
        raise ValueError(m)
    return dx


def _wgrad_overwrites(desc: L.ConvDesc) -> bool:
    """True when tv_wgrad_tn writes every element of dw exactly once for this geometry (dw needs no pre-zeroing).  dbias is
    NOT covered by the answer: it is always ADDED to with fp32 atomics (include/transvae_hip.h) and must be passed zeroed or
    holding a running sum.  The answer follows the kernel's split-K choice (and its tuning hooks), so the library is asked
    every time."""
    r = L.load().tv_wgrad_tn_overwrites(C.byref(desc))
    if r < 0:
        raise RuntimeError("tv_wgrad_tn_overwrites: bad descriptor")
    return r == 1


# Small zero-initialised fp32 buffers (bias / norm-weight gradients, split-K weight gradients of small layers) are carved
# out of a pre-zeroed slab: a backward pass asked for ~770 of them per micro-batch, each a separate ~5 us fill kernel on a
# GPU that is otherwise saturated (profiles/: FillFunctor 0.7 % of the step).  A region is handed out once and never reused;
# the slab lives as long as any tensor carved from it.
_ZERO_SLAB_BYTES = 256 << 20
_ZERO_SMALL_BYTES = 40 << 20       # (round 4: up to the largest weight gradient, 6144 x 1536 fp32 = 36 MiB -- the ~150 split-K
                                   # gradient buffers of a step's first micro-batch were a fill kernel each)
_zero_slabs = {}     # (device, stream) -> [slab tensor (uint8), offset]


def zeros_f32(shape, device) -> torch.Tensor:
    n = 1
    for v in shape:
        n *= int(v)
    nbytes = n * 4
    if nbytes == 0 or nbytes > _ZERO_SMALL_BYTES or device.type != "cuda":
        return torch.zeros(shape, dtype=torch.float32, device=device)
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    ent = _zero_slabs.get(key)
    need = (nbytes + 255) & ~255
    if ent is None or ent[1] + need > _ZERO_SLAB_BYTES:
        ent = [torch.zeros((_ZERO_SLAB_BYTES,), dtype=torch.uint8, device=device), 0]
        _zero_slabs[key] = ent
    off = ent[1]
    ent[1] = off + need
    return ent[0][off:off + nbytes].view(torch.float32).view(shape)


def _grad_buffer(shape, device, zero: bool):
    return zeros_f32(shape, device) if zero else torch.empty(shape, dtype=torch.float32, device=device)


def conv_wgrad_alloc(g: _Geo, w, need_db: bool, x=None, gz=None):
    """Outputs of conv_wgrad (allocated on the stream that will consume them), zeroed only where the kernel accumulates.
    x / gz: the operands, when the launch may be chunked (> 2 GiB): the plan is that of the first chunk."""
    if g.mode == "shuf":
        db = zeros_f32((g.Cout,), w.device) if need_db else None
        return zeros_f32((g.Cin, 2, 2, g.Cout // 4), w.device), db
    d0 = g.fwd_desc(0)
    zero = not _wgrad_overwrites(wgrad_plan_desc(d0, x, gz) if x is not None else d0)
    db = zeros_f32((g.Cout,), w.device) if need_db else None       # (the bias gradient is always ADDED to: include/transvae_hip.h)
    return _grad_buffer(tuple(w.shape), w.device, zero), db


def conv_wgrad(g: _Geo, w, x, gz, need_db: bool, out=None, accumulate: bool = False):
    """(dw in w's layout, dbias | None), fp32.  `out` = buffers from conv_wgrad_alloc (optional).
    accumulate ('c3up' only, with out = (dw, dbias | None) holding the gradients of earlier micro-batches): add in place."""
    dev = x.device
    if g.mode == "c3up":
        # weight gradient of the polyphase form = that of its adjoint conv (gathered operand: the high-resolution
        # gradient under 4x4 / stride-2 taps; the other operand: the layer input), folded back onto the 3x3 taps
        d = _desc(batch=g.B, h_in=g.Ho, w_in=g.Wo, c_in=g.Cout, ldx=g.Cout, h_out=g.H, w_out=g.W, c_out=g.Cin, ldo=g.Cin,
                  kh=4, kw=4, stride=2, pad=1)
        d16 = _grad_buffer((g.Cin, 4, 4, g.Cout), dev, not _wgrad_overwrites(wgrad_plan_desc(d, gz, x)))
        wgrad(d, gz, x, d16, None)
        db = gz.view(-1, g.Cout).sum(0, dtype=torch.float32) if need_db else None
        if accumulate:
            _up_fold_wgrad(d16, g.Cout, g.Cin, out=out[0], accumulate=True)
            if need_db:
                out[1].add_(db)
            return out
        dw = _up_fold_wgrad(d16, g.Cout, g.Cin)
        return dw, db
    if out is not None and g.mode != "shuf":
        dw, db = out
        wgrad(g.fwd_desc(0), x, gz, dw, db)
        return dw, db
    if g.mode != "shuf":
        dw, db = conv_wgrad_alloc(g, w, need_db, x, gz)
        wgrad(g.fwd_desc(0), x, gz, dw, db)
        return dw, db
    db = zeros_f32((g.Cout,), dev) if need_db else None
    cq = g.Cout // 4
    d = _desc(batch=g.B, h_in=2 * g.H, w_in=2 * g.W, c_in=cq, ldx=cq, h_out=g.H, w_out=g.W, c_out=g.Cin, ldo=g.Cin,
              kh=2, kw=2, stride=2, pad=0)
    dwt = zeros_f32((g.Cin, 2, 2, cq), dev)
    wgrad(d, gz, x, dwt, None)     # the transposed problem: gathered operand = hi-res gradient
    dw = dwt.permute(1, 2, 3, 0).reshape(g.Cout, 1, 1, g.Cin).contiguous()
    if need_db:
        d1 = _desc(batch=g.B, h_in=2 * g.H, w_in=2 * g.W, c_in=cq, ldx=cq, h_out=g.H, w_out=g.W, c_out=8, ldo=8,
                   kh=2, kw=2, stride=2, pad=0)
        tmp = zeros_f32((8, 2, 2, cq), dev)
        wgrad(d1, gz, _ones(g.B * g.H * g.W, dev), tmp, None)
        db = tmp[0].reshape(g.Cout).contiguous()
    return dw, db


def act_backward(pre, gy, act_id: int):
    gz = torch.empty_like(gy)
    L.check(L.load().tv_act_bwd(_p(pre), _p(gy), _p(gz), gy.numel(), act_id, _stream()), "tv_act_bwd")
    return gz


class ConvFn(torch.autograd.Function):
    """out = act(conv(x, w) + bias) + residual   on bf16 NHWC activations.

    w is the fp32 master weight already viewed as [Cout, KH, KW, Cin] (or [Cout, Cin] for
    'linear'); its gradient is returned in the same layout, so the caller's permute / cat /
    scaling of the nn.Parameter stays ordinary autograd.
    """

    @staticmethod
    def forward(ctx, x, w, bias, residual, mode: str, act: Optional[str]):
        act_id = _act_id(act)
        out, pre, g, wc = conv_forward(x, w, bias, residual, mode, act_id, x.requires_grad or w.requires_grad)
        ctx.geo, ctx.act_id = g, act_id
        ctx.has_bias, ctx.has_res = bias is not None, residual is not None
        ctx.save_for_backward(x, wc, pre)
        return out

    @staticmethod
    def backward(ctx, gy):
        x, w, pre = ctx.saved_tensors
        g: _Geo = ctx.geo
        gy = gy.contiguous()
        _require(gy.dtype == BF16, "operand check failed: gy.dtype == BF16")
        gres = gy if ctx.has_res else None
        gz = act_backward(pre, gy, ctx.act_id) if ctx.act_id != L.ACT_NONE else gy
        need_dx, need_dw, need_db = ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.has_bias and ctx.needs_input_grad[2]
        dx = conv_dgrad(g, w, gz, x.shape) if need_dx else None
        dw = db = None
        if need_dw or need_db:
            dw, db = conv_wgrad(g, w, x, gz, need_db)
            if not need_dw:
                dw = None
        return dx, dw, db, gres, None, None


def conv(x, w, bias=None, residual=None, mode: str = "c3s1", act: Optional[str] = None):
    return ConvFn.apply(x, w, bias, residual, mode, act)


def linear(x, w, bias=None, residual=None, act: Optional[str] = None):
    return ConvFn.apply(x, w, bias, residual, "linear", act)


# ---------------------------------------------------------------------------------------------
# GroupNorm + SiLU
# ---------------------------------------------------------------------------------------------
class GroupNormSiluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, groups: int, eps: float):
        _need_gpu(x, gamma, beta)
        _require(x.dtype == BF16 and x.is_contiguous() and x.dim() == 4, "operand check failed: x.dtype == BF16 and x.is_contiguous() and x.dim() == 4")
        B, H, W, Cc = x.shape
        lib = L.load()
        gamma = gamma.contiguous()
        beta = beta.contiguous()
        stats = torch.empty((B, Cc, 2), dtype=torch.float32, device=x.device)
        part = torch.empty((lib.tv_gn_partial_count(B, H * W, Cc),), dtype=torch.float32, device=x.device)
        L.check(lib.tv_gn_stats(_p(x), _p(stats), _p(part), B, H * W, Cc, _stream()), "tv_gn_stats")
        mr = torch.empty((B, groups, 2), dtype=torch.float32, device=x.device)
        y = torch.empty_like(x)
        L.check(lib.tv_gn_silu_fwd(_p(x), _p(stats), _p(gamma), _p(beta), _p(mr), _p(y), B, H * W, Cc, groups, eps, _stream()),
                "tv_gn_silu_fwd")
        ctx.groups = groups
        ctx.save_for_backward(x, gamma, beta, mr)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, gamma, beta, mr = ctx.saved_tensors
        B, H, W, Cc = x.shape
        lib = L.load()
        gy = gy.contiguous()
        red = torch.empty((B, Cc, 2), dtype=torch.float32, device=x.device)
        part = torch.empty((lib.tv_gn_partial_count(B, H * W, Cc),), dtype=torch.float32, device=x.device)
        L.check(lib.tv_gn_silu_bwd_reduce(_p(x), _p(gy), _p(mr), _p(gamma), _p(beta), _p(red), _p(part), B, H * W, Cc, ctx.groups,
                                          _stream()), "tv_gn_silu_bwd_reduce")
        dx = torch.empty_like(x)
        dg = zeros_f32((Cc,), x.device)
        db = zeros_f32((Cc,), x.device)
        L.check(lib.tv_gn_silu_bwd_apply(_p(x), _p(gy), None, _p(mr), _p(red), _p(gamma), _p(beta), _p(dx), _p(dg), _p(db),
                                         B, H * W, Cc, ctx.groups, _stream()), "tv_gn_silu_bwd_apply")
        return dx, dg, db, None, None


def group_norm_silu(x, gamma, beta, groups: int = 32, eps: float = 1e-5):
    return GroupNormSiluFn.apply(x, gamma, beta, groups, eps)


# ---------------------------------------------------------------------------------------------
# token-row norms
# ---------------------------------------------------------------------------------------------
class RowNormFn(torch.autograd.Function):
    """mode 0: x*rsqrt(mean x^2 + eps)           (RMSNorm; its weight is folded into the next GEMM)
    mode 1: LayerNorm-hat(RMSNorm(x)*w)        (the x-hat shared by norm_q / norm_k / norm_v)"""

    @staticmethod
    def forward(ctx, x, w, mode: int, eps_rms: float, eps_ln: float):
        _need_gpu(x, w)
        _require(x.dtype == BF16 and x.is_contiguous() and x.dim() == 2, "operand check failed: x.dtype == BF16 and x.is_contiguous() and x.dim() == 2")
        T, Cc = x.shape
        lib = L.load()
        if w is not None:
            w = w.contiguous()
        y = torch.empty_like(x)
        L.check(lib.tv_rownorm_fwd(_p(x), _p(w), _p(y), T, Cc, mode, eps_rms, eps_ln, _stream()), "tv_rownorm_fwd")
        ctx.mode, ctx.eps = mode, (eps_rms, eps_ln)
        ctx.save_for_backward(x, w)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        T, Cc = x.shape
        lib = L.load()
        gy = gy.contiguous()
        dx = torch.empty_like(x)
        dw = zeros_f32((Cc,), x.device) if ctx.mode == 1 else None
        L.check(lib.tv_rownorm_bwd(_p(x), _p(w), _p(gy), None, _p(dx), _p(dw), T, Cc, ctx.mode, ctx.eps[0], ctx.eps[1], _stream()),
                "tv_rownorm_bwd")
        return dx, dw, None, None, None


def rms_hat(x, eps: float = 1e-6):
    return RowNormFn.apply(x, None, 0, eps, 1e-5)


def rms_ln_hat(x, w, eps_rms: float = 1e-6, eps_ln: float = 1e-5):
    return RowNormFn.apply(x, w, 1, eps_rms, eps_ln)


# ---------------------------------------------------------------------------------------------
# attention (RoPE + flash attention, head_dim 64)
# ---------------------------------------------------------------------------------------------
class AttentionFn(torch.autograd.Function):
    """qkv: [B, N, 3*C] bf16 (q | k | v, each [heads, 64]); returns o [B, N, C].

    RoPE is applied IN PLACE on the q and k thirds of `qkv` (the caller must not reuse the
    un-rotated projections); the backward applies the adjoint to dq, dk.
    """

    @staticmethod
    def forward(ctx, qkv, rope_tab, heads: int, scale: float):
        _need_gpu(qkv)
        _require(qkv.dtype == BF16 and qkv.is_contiguous() and qkv.dim() == 3, "operand check failed: qkv.dtype == BF16 and qkv.is_contiguous() and qkv.dim() == 3")
        B, N, C3 = qkv.shape
        _require(C3 == 3 * heads * 64, "operand check failed: C3 == 3 * heads * 64")
        lib = L.load()
        if rope_tab is not None:
            _require(rope_tab.dtype == torch.float32 and rope_tab.shape == (N, 4, 32) and rope_tab.is_contiguous(), "operand check failed: rope_tab.dtype == torch.float32 and rope_tab.shape == (N, 4, 32) and rope_tab.is_contiguous()")
            L.check(lib.tv_rope_qk(_p(qkv), _p(rope_tab), B, N, heads, 0, _stream()), "tv_rope_qk")
        o = torch.empty((B, N, heads * 64), dtype=BF16, device=qkv.device)
        lse = torch.empty((B, heads, N), dtype=torch.float32, device=qkv.device)
        L.check(lib.tv_attn_fwd(_p(qkv), _p(o), _p(lse), B, N, heads, scale, _stream()), "tv_attn_fwd")
        ctx.heads, ctx.scale = heads, scale
        ctx.save_for_backward(qkv, o, lse, rope_tab)
        return o

    @staticmethod
    def backward(ctx, go):
        qkv, o, lse, rope_tab = ctx.saved_tensors
        B, N, _ = qkv.shape
        heads = ctx.heads
        lib = L.load()
        go = go.contiguous()
        delta = torch.empty((2, B, heads, N), dtype=torch.float32, device=qkv.device)   # scratch: -delta, -lse log2(e)
        dqkv = torch.empty_like(qkv)
        L.check(lib.tv_attn_bwd(_p(qkv), _p(o), _p(go), _p(lse), _p(delta), None, _p(dqkv), B, N, heads, ctx.scale, _stream()),
                "tv_attn_bwd")
        if rope_tab is not None:
            L.check(lib.tv_rope_qk(_p(dqkv), _p(rope_tab), B, N, heads, 1, _stream()), "tv_rope_qk")
        return dqkv, None, None, None


def attention(qkv, rope_tab, heads: int, scale: float):
    return AttentionFn.apply(qkv, rope_tab, heads, scale)


# ---------------------------------------------------------------------------------------------
# API-boundary layout conversion (NCHW fp32 <-> NHWC bf16) and the stem's im2col
# ---------------------------------------------------------------------------------------------
class ToNhwcFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, cpad: int):
        _need_gpu(x)
        x = x.float().contiguous()
        B, Cc, H, W = x.shape
        y = torch.empty((B, H, W, cpad), dtype=BF16, device=x.device)
        L.check(L.load().tv_nchw_to_nhwc(_p(x), _p(y), B, Cc, H, W, cpad, _stream()), "tv_nchw_to_nhwc")
        ctx.C = Cc
        return y

    @staticmethod
    def backward(ctx, gy):
        gy = gy.contiguous()
        B, H, W, cpad = gy.shape
        gx = torch.empty((B, ctx.C, H, W), dtype=torch.float32, device=gy.device)
        L.check(L.load().tv_nhwc_to_nchw(_p(gy), _p(gx), B, ctx.C, H, W, cpad, _stream()), "tv_nhwc_to_nchw")
        return gx, None


class ToNchwFn(torch.autograd.Function):
    """x: [B,H,W,Cpad] bf16 -> fp32 [B, C, H, W] taking channels [c0, c0+C)."""

    @staticmethod
    def forward(ctx, x, c0: int, Cc: int):
        _need_gpu(x)
        _require(x.dtype == BF16 and x.is_contiguous(), "operand check failed: x.dtype == BF16 and x.is_contiguous()")
        B, H, W, cpad = x.shape
        y = torch.empty((B, Cc, H, W), dtype=torch.float32, device=x.device)
        src = C.c_void_p(x.data_ptr() + 2 * c0)
        L.check(L.load().tv_nhwc_to_nchw(src, _p(y), B, Cc, H, W, cpad, _stream()), "tv_nhwc_to_nchw")
        ctx.c0, ctx.cpad = c0, cpad
        return y

    @staticmethod
    def backward(ctx, gy):
        gy = gy.float().contiguous()
        B, Cc, H, W = gy.shape
        if ctx.c0 == 0:
            gx = torch.empty((B, H, W, ctx.cpad), dtype=BF16, device=gy.device)
            L.check(L.load().tv_nchw_to_nhwc(_p(gy), _p(gx), B, Cc, H, W, ctx.cpad, _stream()), "tv_nchw_to_nhwc")
        else:  # channel slice in the middle: convert compactly, then place (tiny tensors: mu / logvar)
            tmp = torch.empty((B, H, W, Cc), dtype=BF16, device=gy.device)
            L.check(L.load().tv_nchw_to_nhwc(_p(gy), _p(tmp), B, Cc, H, W, Cc, _stream()), "tv_nchw_to_nhwc")
            gx = torch.zeros((B, H, W, ctx.cpad), dtype=BF16, device=gy.device)
            gx[..., ctx.c0:ctx.c0 + Cc] = tmp
        return gx, None, None


def to_nhwc(x, cpad: int):
    return ToNhwcFn.apply(x, cpad)


def to_nchw(x, c0: int, Cc: int):
    return ToNchwFn.apply(x, c0, Cc)


def im2col3x3(x: torch.Tensor, kpad: int) -> torch.Tensor:
    """NCHW fp32 image -> [B*H*W, kpad] bf16 rows of 3x3 patches ordered (ky,kx,c).  No gradient
    flows to the image (the stem's input is data)."""
    _need_gpu(x)
    x = x.detach().float().contiguous()
    B, Cc, H, W = x.shape
    y = torch.empty((B * H * W, kpad), dtype=BF16, device=x.device)
    L.check(L.load().tv_im2col3x3(_p(x), _p(y), B, Cc, H, W, kpad, _stream()), "tv_im2col3x3")
    return y
