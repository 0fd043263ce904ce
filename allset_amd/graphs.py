"""hipGraph capture of the launch-bound regime: at dataset scale (Cora: 2.7k vertices, 13k incidences) one
``SetGNN`` training step is ~150 kernel launches of a few microseconds each, so the step time is the host's launch
rate, not the GPU's.  Capturing the whole step once and replaying it removes the host from the loop.

The reference has no counterpart (``src/train.py:480-560`` runs an eager PyTorch loop); this is the MI355X side of the
same loop -- same arithmetic, one graph launch per epoch.

What makes capture possible here:
  * every kernel of the path is stream-ordered with no host read-back once the incidence is built
    (``models.SetGNN._incidences`` caches the CSR pair on the first call, which must happen BEFORE capture);
  * dropout masks come from a device-resident counter (``dense.device_seed_counter``): the kernels read it when they
    start and the graph advances it once per step (in the launch that reduces the parameter gradients), so every replay
    draws fresh masks although the host-side seeds were frozen at capture;
  * the optimizer must be capturable (``torch.optim.Adam(..., capturable=True)``; add ``fused=True``: the unfused
    capturable Adam launches ~2 tiny kernels per parameter for its bias corrections, 20 % of a Cora-sized step).

Torch's ``CUDAGraph`` on ROCm records hipGraph nodes from whatever is launched on the capturing stream; the C-ABI
takes the stream explicitly, so its launches are captured like torch's own.
"""
from __future__ import annotations

from typing import Callable, Optional

import torch

from . import dense
from .optim import FusedAdam
from ._lib import AllSetHipError

Tensor = torch.Tensor


def _side_stream_warmup(fn: Callable[[], None], iters: int) -> "torch.cuda.Stream":
    """Runs ``fn`` on a side stream and returns that stream: capturing on the SAME stream (``torch.cuda.graph(g, stream=s)``) keeps
    per-stream state created during warm-up valid for the capture -- the loss kernel's ticket counter (losses._ticket) would
    otherwise be allocated and zero-filled inside the capture, one more node in every replay."""
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(iters):
            fn()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    return s


class GraphedForward:
    """Eval-mode forward of ``model`` on a fixed ``data`` captured as one graph.

    ``out = GraphedForward(model, data)()`` replays it; the result tensor is static (overwritten by the next
    replay).  New features for the same hypergraph: ``gf(x_new)`` copies them into the captured input first.
    Parameters are read at replay time, so the graph stays valid while training updates them in place.
    """

    def __init__(self, model: torch.nn.Module, data, warmup: int = 2, constant_features: bool = False):
        if not data.x.is_cuda:
            raise AllSetHipError("GraphedForward needs device tensors")
        self.model, self.data = model, data
        self.constant_features = bool(constant_features)
        was_training = model.training
        model.eval()
        import contextlib
        # constant_features: the captured kernels may read data.x through its cached non-zero structure (bag-of-words rows:
        # dense.sparse_rows) -- new features can then NOT be copied in between replays
        with torch.no_grad(), torch.cuda.device(data.x.device), (dense.constant_features() if constant_features else contextlib.nullcontext()):
            st = _side_stream_warmup(lambda: model(data), max(1, warmup))
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph, stream=st):
                self.out = model(data)
        model.train(was_training)

    def __call__(self, x: Optional[Tensor] = None) -> Tensor:
        if x is not None:
            if self.constant_features:
                raise AllSetHipError("GraphedForward(constant_features=True): the captured kernels read the features' non-zero structure; "
                                     "capture with constant_features=False to replace them between replays")
            self.data.x.copy_(x)
        self.graph.replay()
        return self.out


class GraphedCallable:
    """A no-grad callable over STATIC device tensors (closed over by ``fn``) captured as one graph: ``out = gc()`` replays it and
    returns ``fn``'s result tensor (static, overwritten by the next replay).  For the small per-epoch tails of a training loop
    (metrics of a replayed forward, copies into a history buffer) whose 5-10 eager launches cost more host time than the
    replayed step takes on the device."""

    def __init__(self, fn: Callable[[], Tensor], device, warmup: int = 2):
        with torch.no_grad(), torch.cuda.device(device):
            st = _side_stream_warmup(lambda: fn(), max(1, warmup))
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph, stream=st):
                self.out = fn()

    def __call__(self) -> Tensor:
        self.graph.replay()
        return self.out


class GraphedTrainStep:
    """One full training step -- zero_grad, forward (train mode, dropout live), loss, backward, optimizer step --
    captured as one graph.  ``loss = step()`` replays it and returns the static loss tensor (a device scalar; read
    it with ``.item()`` only when you need it).

    ``loss_fn(logits) -> scalar`` closes over static label / index tensors, e.g.
    ``lambda out: F.nll_loss(F.log_softmax(out[train_idx], 1), y[train_idx])`` (``src/train.py:519-524``).

    ``data`` (features, incidence, norm) is a CONSTANT of the captured step: kernels may read structures derived from it at
    capture time (the CSR pair of the incidence; the non-zero structure of sparse raw features, ``dense.sparse_rows``).

    Capture needs a few real warm-up steps (allocator, autograd and optimizer state must exist before recording).
    With ``restore=True`` (default) parameters and optimizer state are put back afterwards -- in place, the graph holds
    their addresses -- so the first replay is step 1 of training, not step ``warmup + 1``: state the optimizer created
    during warm-up is zeroed, which is Adam's initial state.
    """

    def __init__(self, model: torch.nn.Module, data, loss_fn: Callable[[Tensor], Tensor],
                 optimizer: torch.optim.Optimizer, warmup: int = 3, train_mode: bool = True, restore: bool = True):
        dev = data.x.device
        if not data.x.is_cuda:
            raise AllSetHipError("GraphedTrainStep needs device tensors")
        for grp in optimizer.param_groups:
            if not grp.get("capturable", False):
                raise AllSetHipError("GraphedTrainStep needs a capturable optimizer (torch.optim.Adam(..., capturable=True))")
        self.model, self.data, self.optimizer = model, data, optimizer
        self.counter = torch.zeros(1, dtype=torch.int64, device=dev)
        self.counter.fill_(int(torch.empty((), dtype=torch.int64).random_().item()) & 0x3FFFFFFFFFFFFFFF)
        model.train(train_mode)           # False: a dropout-free (deterministic) step, e.g. for tests / fine-tuning

        one = self._one = torch.ones((), dtype=torch.float32, device=dev)     # (an attribute: the graph reads this address at every replay)

        fused = isinstance(optimizer, FusedAdam)

        def one_step():
            optimizer.zero_grad(set_to_none=True)
            loss = loss_fn(model(data))
            # ONE launch reduces every parameter-gradient partial of the backward pass and advances the step's counters (the dropout
            # seed counter; FusedAdam's step counters): dense.deferred_param_grads
            with dense.deferred_param_grads(bump_i64=self.counter, bump_f32=optimizer.step_counters if fused else None):
                loss.backward(one if (loss.dim() == 0 and loss.dtype == one.dtype) else None)      # (no ones_like launch per step)
            if fused:
                optimizer.step(counters_advanced=True)
            else:
                optimizer.step()
            return loss

        params = [p for grp in optimizer.param_groups for p in grp["params"]]
        if restore:
            with torch.no_grad():
                p_snap = [p.detach().clone() for p in params]
                s_snap = {p: {k: v.clone() for k, v in optimizer.state.get(p, {}).items() if torch.is_tensor(v)}
                          for p in params}
                # module buffers move during warm-up too (BatchNorm running_mean / running_var / num_batches_tracked)
                b_snap = [(b, b.detach().clone()) for b in model.buffers()]
        with torch.cuda.device(dev), dense.device_seed_counter(self.counter):
            st = _side_stream_warmup(one_step, max(1, warmup))
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph, stream=st):
                self.loss = one_step()
        if restore:
            with torch.no_grad():
                for p, snap in zip(params, p_snap):
                    p.copy_(snap)
                    for k, v in optimizer.state.get(p, {}).items():
                        if torch.is_tensor(v):
                            v.copy_(s_snap[p][k]) if k in s_snap[p] else v.zero_()
                for b, snap in b_snap:
                    b.copy_(snap)

    def __call__(self) -> Tensor:
        self.graph.replay()
        return self.loss
