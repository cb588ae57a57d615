"""AdamW for the TransVAE path as three multi-tensor HIP launches per optimizer step.

Caller pattern replaced (R/train.py:610-618, 681-687; R/train_2.py:328-338):

    torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)     # ~750 small launches + a pass over 4.2 GB of gradients
    optimizer.step()                                            # torch.optim.AdamW(..., fused=True)
    (non-finite loss -> skip the step)

Here (`fused_clip_step`, called by transvae.parallel.clip_and_step):

    tv_opt_grad_norm      global L2 norm of every gradient, clip coefficient, non-finite flag, step counter  (device)
    tv_opt_adamw          the update, with the clip folded in as the un-scale of the gradient, skipped on the device when
                          the norm is not finite; writes the bf16 copy of every weight in the same pass
    tv_pack_weight_multi  every transposed (data-gradient) bf16 operand refreshed from those copies in one launch

so the ~840 `tv_pack_weight` launches of the next step's first micro-batch disappear (transvae.hip.ops serves the operands
from this optimizer's copies as long as the parameter has not been modified by anyone else).

State layout is torch.optim.AdamW's (`state[p] = {step, exp_avg, exp_avg_sq}`), so `state_dict()` / `load_state_dict()`
interchange with the reference's checkpoints (`optimizer_state_dict`, R/train.py:753-769).  The arithmetic is ATen's fused
AdamW in fp32 (tests/test_train_gpu.py compares the two).  No CPU path: parameters must live on a HIP device.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from .hip import _lib as L
from .hip import ops


def _dense(t: torch.Tensor) -> bool:
    return t.is_contiguous() or (t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last))


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-2,
                 bf16_operands: bool = True):
        if lr < 0 or eps < 0 or weight_decay < 0 or not (0 <= betas[0] < 1 and 0 <= betas[1] < 1):
            raise ValueError("FusedAdamW: bad hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay))
        ps = [p for g in self.param_groups for p in g["params"]]
        if not ps:
            raise ValueError("FusedAdamW: no parameters")
        dev = ps[0].device
        if dev.type != "cuda" or any(p.device != dev for p in ps):
            raise RuntimeError("FusedAdamW: all parameters must live on one HIP device (there is no CPU path)")
        if any(p.dtype != torch.float32 or not _dense(p) for p in ps):
            raise RuntimeError("FusedAdamW: parameters must be dense fp32 tensors")
        self._dev = dev
        self._chunk = L.load().tv_opt_chunk_elems()
        self._ctrl = torch.zeros(8, dtype=torch.float32, device=dev)
        self._tables = {}       # participation key -> (chunk table on device, n_chunks, partials)
        self._shadow = {}       # id(param) -> bf16 copy (same strides)
        self._bf16_operands = bf16_operands
        if bf16_operands:
            with torch.cuda.device(dev):
                for p in ps:
                    if p.dim() >= 2:
                        self._shadow[id(p)] = torch.empty_strided(p.shape, p.stride(), dtype=torch.bfloat16, device=dev)
                self._cast_shadows(ps)
                ops.register_param_shadows(ps, self._shadow)

    # ------------------------------------------------------------------ tables
    def _state_of(self, p):
        st = self.state[p]
        if "exp_avg" not in st:
            st["step"] = self._ctrl[0]           # a view: every parameter shares the device step counter
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            if st["exp_avg"].stride() != p.stride():
                raise RuntimeError("FusedAdamW: optimizer state does not share the parameter's memory layout")
        return st

    def _chunk_table(self, ps):
        key = tuple(id(p) for p in ps)
        ent = self._tables.get(key)
        if ent is None:
            rows = []
            for i, p in enumerate(ps):
                n = (p.numel() + self._chunk - 1) // self._chunk
                rows.append(torch.stack([torch.full((n,), i, dtype=torch.int32), torch.arange(n, dtype=torch.int32)], 1))
            tab = torch.cat(rows, 0).contiguous()
            ent = (tab.to(self._dev), tab.shape[0], torch.empty(tab.shape[0], dtype=torch.float32, device=self._dev))
            self._tables[key] = ent
        return ent

    def _pointer_table(self, ps, with_grad: bool):
        rows = []
        self._relaid = []
        for p in ps:
            st = self._state_of(p) if with_grad else None
            g = p.grad if with_grad else None
            if with_grad:
                if g.dtype != torch.float32 or g.device != p.device:
                    raise RuntimeError("FusedAdamW: gradients must be fp32 on the parameter's device")
                if g.stride() != p.stride():      # (autograd and DDP's bucket views follow the parameter's layout; anything else: re-lay once)
                    g = torch.empty_like(p).copy_(g)
                    self._relaid.append(g)
            sh = self._shadow.get(id(p))
            rows.append((p.data_ptr(), g.data_ptr() if with_grad else 0, st["exp_avg"].data_ptr() if with_grad else 0,
                         st["exp_avg_sq"].data_ptr() if with_grad else 0, sh.data_ptr() if sh is not None else 0, p.numel()))
        host = torch.tensor(rows, dtype=torch.int64).pin_memory()
        return host.to(self._dev, non_blocking=True), host     # (keep the pinned source alive until the copy has run)

    def _cast_shadows(self, ps):
        ps = [p for p in ps if id(p) in self._shadow]
        if not ps:
            return
        chunks, n, _ = self._chunk_table(ps)
        tab, keep = self._pointer_table(ps, False)
        L.check(L.load().tv_opt_cast_shadows(C.c_void_p(tab.data_ptr()), C.c_void_p(chunks.data_ptr()), n, ops._stream()),
                "tv_opt_cast_shadows")
        self._keep = keep

    # ------------------------------------------------------------------ step
    @torch.no_grad()
    def fused_clip_step(self, max_norm: Optional[float] = None):
        """Clip (global L2 norm, R/train.py:610-612; None / <= 0: no clipping), non-finite guard, AdamW, operand refresh.
        Returns (gradient norm, skipped flag) as device scalars -- no host sync."""
        lib = L.load()
        with torch.cuda.device(self._dev):
            stream = ops._stream()
            all_ps = [p for g in self.param_groups for p in g["params"] if p.grad is not None]
            if not all_ps:
                return torch.zeros((), device=self._dev), torch.zeros((), device=self._dev)
            beta1, beta2 = self.param_groups[0]["betas"]
            if any(g["betas"] != (beta1, beta2) for g in self.param_groups):
                raise RuntimeError("FusedAdamW: all parameter groups must share betas (one device step counter)")
            chunks, n, partials = self._chunk_table(all_ps)
            tab, keep = self._pointer_table(all_ps, True)
            L.check(lib.tv_opt_grad_norm(C.c_void_p(tab.data_ptr()), C.c_void_p(chunks.data_ptr()), n, C.c_void_p(partials.data_ptr()),
                                         C.c_void_p(self._ctrl.data_ptr()), float(max_norm or 0.0), beta1, beta2, 1, stream),
                    "tv_opt_grad_norm")
            single = len(self.param_groups) == 1
            for g in self.param_groups:
                ps = all_ps if single else [p for p in g["params"] if p.grad is not None]
                if not ps:
                    continue
                if single:
                    c2, n2, t2 = chunks, n, tab
                else:
                    c2, n2, _ = self._chunk_table(ps)
                    t2, k2 = self._pointer_table(ps, True)
                    keep = (keep, k2)
                L.check(lib.tv_opt_adamw(C.c_void_p(t2.data_ptr()), C.c_void_p(c2.data_ptr()), n2, C.c_void_p(self._ctrl.data_ptr()),
                                         float(g["lr"]), beta1, beta2, float(g["eps"]), float(g["weight_decay"]), stream),
                        "tv_opt_adamw")
            self._keep = keep
            # the update went through raw pointers: tell autograd (saved tensors of a live graph would be stale) ...
            torch._C._autograd._unsafe_set_version_counter(tuple(all_ps), tuple(p._version + 1 for p in all_ps))
            # ... and re-validate / refresh the bf16 operands derived from the fresh copies
            if self._bf16_operands:
                ops.refresh_param_operands(all_ps)
            norm, skipped = self._ctrl[1].clone(), self._ctrl[3].clone()
        return norm, skipped

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        self.fused_clip_step(None)
        return loss

    # ------------------------------------------------------------------ (de)serialisation: torch.optim.AdamW's layout
    def state_dict(self):
        sd = super().state_dict()
        for st in sd["state"].values():
            if "step" in st:
                st["step"] = st["step"].detach().clone().cpu()      # a plain scalar per parameter, like torch.optim.AdamW
        return sd

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        step = None
        for p in (p for g in self.param_groups for p in g["params"]):
            st = self.state.get(p)
            if st and "step" in st:
                v = float(st["step"])
                step = v if step is None else max(step, v)
        if step is not None:
            self._ctrl[0] = step
        for p in (p for g in self.param_groups for p in g["params"]):
            st = self.state.get(p)
            if st and "exp_avg" in st:
                st["step"] = self._ctrl[0]
                for k in ("exp_avg", "exp_avg_sq"):
                    if st[k].stride() != p.stride() or st[k].dtype != torch.float32:
                        t = torch.empty_strided(p.shape, p.stride(), dtype=torch.float32, device=p.device)
                        t.copy_(st[k])
                        st[k] = t
