"""``torch.optim.Adam`` (what the reference trains with, ``/root/reference/src/train.py:469``) as ONE kernel launch per step
(``csrc/optim.hip``) plus one ``_foreach_add_`` on the step counters.

At dataset scale the parameter set is ~20 small tensors; torch's fused multi-tensor Adam in capturable mode spends ~40 us on
them -- a tenth of a hipGraph-replayed Cora step.  Same arithmetic (non-amsgrad, L2 ``weight_decay``, bias corrections from a
per-parameter step counter kept on the device, so the optimizer is capturable by construction); fp32 and bf16 device
parameters (bf16: moments in bf16 as torch keeps them, fp32 arithmetic); anything else falls back to ``torch.optim.Adam``'s functional form.
"""
from __future__ import annotations

import ctypes
from typing import Iterable

import torch

from . import _lib
from ._lib import check, stream_of, on_device


class FusedAdam(torch.optim.Optimizer):
    """``torch.optim.Adam(params, lr, betas, eps, weight_decay)`` -- the form the reference uses -- on fp32 / bf16 device
    parameters.  NOT covered, and refused rather than ignored: ``amsgrad``, ``maximize``, a tensor ``lr``."""

    def __init__(self, params: Iterable, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0,
                 amsgrad: bool = False, maximize: bool = False):
        if amsgrad or maximize:
            raise NotImplementedError("FusedAdam covers plain Adam only (no amsgrad / maximize): use torch.optim.Adam")
        if isinstance(lr, torch.Tensor):
            raise NotImplementedError("FusedAdam takes a float learning rate (the kernel receives it by value)")
        if lr < 0 or eps < 0 or not (0 <= betas[0] < 1) or not (0 <= betas[1] < 1) or weight_decay < 0:
            raise ValueError("FusedAdam: invalid hyper-parameter")
        # `capturable` is what allset_amd.graphs.GraphedTrainStep checks: this optimizer keeps its step counters on the device
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, capturable=True))

    def _state(self, p):
        st = self.state[p]
        if not st:
            st["step"] = torch.zeros((), dtype=torch.float32, device=p.device)
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        step = st["step"]
        # a state loaded from a non-capturable torch.optim.Adam checkpoint carries a Python number or a CPU tensor here; the kernel
        # takes the counter's DEVICE address
        if not isinstance(step, torch.Tensor) or step.device != p.device or step.dtype != torch.float32 or step.dim() != 0:
            st["step"] = torch.as_tensor(float(step), dtype=torch.float32).to(p.device).reshape(())
        for k in ("exp_avg", "exp_avg_sq"):
            if st[k].device != p.device:
                st[k] = st[k].to(p.device)
        return st

    def step_counters(self):
        """The device step counters ``step()`` will advance: one per parameter that has a gradient and takes the fused kernel.  A
        caller that advances them itself (``graphs.GraphedTrainStep``: in the launch that reduces the gradients) then calls
        ``step(counters_advanced=True)``."""
        out = []
        for group in self.param_groups:
            for p in group["params"]:
                if (p.grad is not None and p.is_cuda and p.dtype in (torch.float32, torch.bfloat16) and p.is_contiguous()
                        and p.grad.is_contiguous() and p.grad.dtype == p.dtype and not p.grad.is_sparse):
                    out.append(self._state(p)["step"])
        return out

    @torch.no_grad()
    def step(self, closure=None, counters_advanced: bool = False):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.load()
        cap = int(lib.allset_adam_max_tensors())
        from . import dense
        dense.weights_changed()      # (the kernel writes parameters through raw pointers: no version counter moves -- drop prebuilt plane images)
        for group in self.param_groups:
            b1, b2 = group["betas"]
            live = [p for p in group["params"] if p.grad is not None]
            fast = [p for p in live if p.is_cuda and p.dtype in (torch.float32, torch.bfloat16) and p.is_contiguous()
                    and p.grad.is_contiguous() and p.grad.dtype == p.dtype and not p.grad.is_sparse]
            slow = [p for p in live if all(p is not q for q in fast)]
            if slow:     # anything else: torch's own functional Adam on the same state layout
                sts = [self._state(p) for p in slow]
                import importlib
                importlib.import_module("torch.optim.adam").adam(slow, [p.grad for p in slow], [s["exp_avg"] for s in sts], [s["exp_avg_sq"] for s in sts], [],
                                      [s["step"] for s in sts], amsgrad=False, beta1=b1, beta2=b2, lr=group["lr"],
                                      weight_decay=group["weight_decay"], eps=group["eps"], maximize=False,
                                      capturable=all(p.is_cuda for p in slow),
                                      foreach=None, fused=None)
            if not fast:
                continue
            sts = [self._state(p) for p in fast]
            if not counters_advanced:
                torch._foreach_add_([s["step"] for s in sts], 1.0)
            by_dev = {}
            for p, s in zip(fast, sts):
                by_dev.setdefault((p.device, p.dtype), []).append((p, s))
            for (dev, dt), items in by_dev.items():
                for k0 in range(0, len(items), cap):
                    part = items[k0:k0 + cap]
                    n = len(part)
                    arr = lambda vals: (ctypes.c_void_p * n)(*vals)
                    with on_device(dev):
                        check(lib.allset_adam_step_dtype(1 if dt == torch.bfloat16 else 0, arr([p.data_ptr() for p, _ in part]), arr([p.grad.data_ptr() for p, _ in part]),
                                                   arr([s["exp_avg"].data_ptr() for _, s in part]),
                                                   arr([s["exp_avg_sq"].data_ptr() for _, s in part]),
                                                   arr([s["step"].data_ptr() for _, s in part]),
                                                   (ctypes.c_int64 * n)(*[p.numel() for p, _ in part]), n, float(group["lr"]),
                                                   float(b1), float(b2), float(group["eps"]), float(group["weight_decay"]),
                                                   stream_of(dev)), "allset_adam_step_dtype")
        return loss
