"""Checkpoints in the reference's file format.

The reference trainer writes ``{'epoch', 'global_step', 'model_state_dict', 'optimizer_state_dict', 'args'}`` with
``torch.save`` (R/train.py:753-769) and restores it with ``load_state_dict`` on the model and the optimizer
(R/train.py:708-719).  The model's ``state_dict`` keys and shapes are identical here (tests/test_model_host.py), so a
run can move between the two implementations in either direction; the only difference is memory layout -- conv
weights live channels_last here -- which ``load_state_dict``'s ``copy_`` absorbs for the parameters and which
:func:`load_checkpoint` re-applies to the optimizer moments so that fused optimizers see one layout per parameter.
"""
from __future__ import annotations

from typing import Any, Dict, Optional

import torch
import torch.nn as nn

__all__ = ["save_checkpoint", "load_checkpoint"]


def _unwrap(model: nn.Module) -> nn.Module:
    return model.module if hasattr(model, "module") and isinstance(model.module, nn.Module) else model


def save_checkpoint(model: nn.Module, optimizer: Optional[torch.optim.Optimizer], epoch: int, global_step: int, path: str,
                    args: Optional[Dict[str, Any]] = None) -> None:
    """Write the reference's checkpoint dictionary (R/train.py:753-769).  Tensors are stored contiguous in their
    logical (OIHW) order, exactly what the reference's own ``state_dict`` would hold."""
    sd = {k: v.detach().contiguous() for k, v in _unwrap(model).state_dict().items()}
    ckpt = {"epoch": int(epoch), "global_step": int(global_step), "model_state_dict": sd,
            "optimizer_state_dict": optimizer.state_dict() if optimizer is not None else {},
            "args": dict(args) if args is not None else {}}
    torch.save(ckpt, path)


def load_checkpoint(path: str, model: nn.Module, optimizer: Optional[torch.optim.Optimizer] = None,
                    map_location: Any = None) -> Dict[str, Any]:
    """Restore a checkpoint written by either implementation (R/train.py:708-719).

    Returns ``{'epoch', 'global_step', 'args'}`` (``global_step`` defaults to 0 for old files, as in the reference)."""
    ckpt = torch.load(path, map_location=map_location, weights_only=True)
    _unwrap(model).load_state_dict(ckpt["model_state_dict"])
    if optimizer is not None and ckpt.get("optimizer_state_dict"):
        optimizer.load_state_dict(ckpt["optimizer_state_dict"])
        for group in optimizer.param_groups:          # moments follow their parameter's memory layout
            for p in group["params"]:
                st = optimizer.state.get(p)
                if not st:
                    continue
                for k, v in st.items():
                    if torch.is_tensor(v) and v.shape == p.shape and v.stride() != p.stride():
                        st[k] = torch.empty_like(p, dtype=v.dtype).copy_(v)
    return {"epoch": int(ckpt.get("epoch", 0)), "global_step": int(ckpt.get("global_step", 0)), "args": ckpt.get("args", {})}
