"""The loss of the reference's training loop -- ``NLLLoss()(F.log_softmax(out, dim=1)[train_idx], y[train_idx])``
(``/root/reference/src/train.py:479-480``) -- as one forward and one backward kernel (``csrc/loss.hip``).

At dataset scale the torch composition (log_softmax, two index kernels, nll forward / backward, a reduction, softmax
backward) is ~8 launches of 5-16 us, a tenth of a hipGraph-replayed Cora step.  The split is passed as a 0/1 weight per row
(:func:`split_mask`), so the gradient kernel writes every row of ``d loss / d logits`` and no scatter is needed.
Device fp32 logits only; anything else (CPU tensors in unit tests) runs the reference composition.
"""
from __future__ import annotations

from ctypes import byref, c_int64

import torch
import torch.nn.functional as F
from torch.autograd.function import once_differentiable

from . import _lib
from ._lib import check, ptr, stream_of, on_device

Tensor = torch.Tensor


def split_mask(idx: Tensor, n: int) -> Tensor:
    """float32 [n] with 1 at the rows of ``idx`` (unique row ids, as the reference's splits are)."""
    w = torch.zeros(n, dtype=torch.float32, device=idx.device)
    w[idx] = 1.0
    return w


_TICKETS = {}      # (device index, stream handle) -> zeroed uint32 the forward kernel counts its workgroups on (and re-arms)


def _ticket(dev: torch.device) -> Tensor:
    """One per (device, stream): launches on one stream are ordered, so they may share the counter the kernel re-arms; a loss forward
    on ANOTHER stream of the same device (an eager evaluation loss beside a captured training step on a side stream) gets its own."""
    key = (dev.index, int(torch.cuda.current_stream(dev).cuda_stream))
    t = _TICKETS.get(key)
    if t is None:
        t = _TICKETS[key] = torch.zeros(1, dtype=torch.int32, device=dev)      # (zeroed on the stream that launches next)
    return t


class _NllLogSoftmax(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, y, w, inv_count):
        dev = logits.device
        n, C = logits.shape
        lib = _lib.load()
        npart = c_int64(0)
        check(lib.allset_nll_partials(n, byref(npart)), "allset_nll_partials")
        # [workgroup sums | their total].  The total is written by the LAST workgroup to arrive on the ticket; outside a graph
        # capture it starts as NaN, so a ticket wedged by an earlier aborted launch shows up as a NaN loss instead of stale memory
        # (inside a capture the fill would be one more node of a launch-bound step; a replay that aborts takes the process with it)
        if torch.cuda.is_current_stream_capturing():
            partials = torch.empty(npart.value + 1, dtype=torch.float32, device=dev)
        else:
            partials = torch.full((npart.value + 1,), float("nan"), dtype=torch.float32, device=dev)
        with on_device(dev):
            check(lib.allset_nll_logsoftmax_fwd_total(ptr(logits), logits.stride(0), ptr(y), ptr(w), inv_count, ptr(partials), npart.value,
                                                      ptr(_ticket(dev)), ptr(partials[npart.value:]), n, C, stream_of(dev)),
                  "allset_nll_logsoftmax_fwd_total")
        ctx.save_for_backward(logits, y, w)
        ctx.inv_count = inv_count
        return partials[npart.value]

    @staticmethod
    @once_differentiable
    def backward(ctx, gout):
        logits, y, w = ctx.saved_tensors
        dev = logits.device
        n, C = logits.shape
        g = torch.empty((n, C), dtype=torch.float32, device=dev)
        gout = gout.reshape(1).float().contiguous()
        with on_device(dev):
            check(_lib.load().allset_nll_logsoftmax_bwd(ptr(logits), logits.stride(0), ptr(y), ptr(w), ctx.inv_count, ptr(gout), ptr(g), C,
                                                        n, C, stream_of(dev)), "allset_nll_logsoftmax_bwd")
        return g, None, None, None


def nll_log_softmax(logits: Tensor, y: Tensor, mask: Tensor, count: float) -> Tensor:
    """``mean over the rows with mask == 1 of -log_softmax(logits)[r, y[r]]`` -- ``count`` = number of such rows (a host
    number, known when the split is made: no device read-back per step).  ``y``: int64 class per row, for all rows."""
    if logits.is_cuda and logits.dtype == torch.float32 and logits.dim() == 2 and logits.stride(1) == 1:
        if y.dtype != torch.int64 or mask.dtype != torch.float32:
            raise _lib.AllSetHipError(f"nll_log_softmax needs int64 labels and a float32 mask (got {y.dtype}, {mask.dtype})")
        return _NllLogSoftmax.apply(logits, y.contiguous(), mask.contiguous(), 1.0 / float(count))
    out = F.log_softmax(logits.float(), dim=1)
    picked = out.gather(1, y.view(-1, 1)).squeeze(1)
    return -(picked * mask).sum() / float(count)


def split_ids(split_idx: dict, n: int, device) -> Tensor:
    """int8 [n]: 0 / 1 / 2 for the rows of ``split_idx['train' | 'valid' | 'test']``, -1 elsewhere."""
    sp = torch.full((n,), -1, dtype=torch.int8, device=device)
    for k, name in enumerate(("train", "valid", "test")):
        sp[split_idx[name].to(device)] = k
    return sp


def split_metrics(logits: Tensor, y: Tensor, split: Tensor, counts: Tensor) -> Tensor:
    """float32 [6] ON THE DEVICE: accuracy of the train / valid / test rows and their mean NLL of ``log_softmax(logits)`` -- what the
    reference's ``evaluate`` returns (train.py:169-199), without its host round trips.  ``split``: :func:`split_ids`;
    ``counts``: float32 [3] set sizes (device)."""
    if logits.is_cuda and logits.dtype == torch.float32 and logits.dim() == 2 and logits.stride(1) == 1:
        if y.dtype != torch.int64 or split.dtype != torch.int8:
            raise _lib.AllSetHipError(f"split_metrics needs int64 labels and int8 split ids (got {y.dtype}, {split.dtype})")
        dev = logits.device
        n, C = logits.shape
        lib = _lib.load()
        npart = c_int64(0)
        check(lib.allset_nll_partials(n, byref(npart)), "allset_nll_partials")
        partials = torch.empty((npart.value, 6), dtype=torch.float32, device=dev)
        with on_device(dev):
            check(lib.allset_split_metrics(ptr(logits), logits.stride(0), ptr(y.contiguous()), ptr(split), ptr(partials), npart.value, n, C,
                                           stream_of(dev)), "allset_split_metrics")
        sums = partials.sum(0) if npart.value > 1 else partials[0]
    else:
        out = F.log_softmax(logits.float(), dim=1)
        ok = (out.argmax(dim=1) == y).float()
        nll = -out.gather(1, y.view(-1, 1)).squeeze(1)
        sel = torch.stack([(split == k).float() for k in range(3)])
        sums = torch.cat([sel @ ok, sel @ nll])
    return sums / torch.cat([counts, counts]).clamp(min=1.0)
