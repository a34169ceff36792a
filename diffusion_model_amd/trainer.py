"""The training hot loop of the reference (main_stl10.py:449-516, main.py:159-188) on the HIP kernels.

* parameters, gradients, Adam moments and the EMA copy each live in ONE flat fp32 buffer
  (``FlatParams``); module parameters are views, so ``state_dict()`` / checkpoints are unchanged;
* a micro-step is: fused q-sample + target -> hand-scheduled UNet forward -> fused weighted-MSE
  loss + dL/dout -> hand-scheduled backward accumulating straight into the flat gradient;
* an optimizer step is 3 launches: squared-norm partials, norm + clip coefficient, fused
  clip-scale + AdamW + EMA (the 1/world_size gradient average rides in the same multiply);
* data parallel: one process per GPU, the flat gradient is all-reduced (RCCL) in a few large
  contiguous buckets, each enqueued the moment its layers' backward has been launched, so the
  collective overlaps the rest of the backward; only on accumulation boundaries.
"""
from __future__ import annotations

import math
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import os

import torch
import torch.nn as nn

from . import kernels as K
from .unet import Grads, invalidate_packed_weights, repack_all, reseed_dropout


def lr_lambda(step: int, total_steps: int) -> float:
    """Reference main_stl10.py:405-412: 5 % linear warm-up, then cosine."""
    warm = max(1, int(0.05 * total_steps))
    if step < warm:
        return float(step + 1) / float(max(1, warm))
    prog = (step - warm) / float(max(1, total_steps - warm))
    return 0.5 * (1.0 + math.cos(math.pi * prog))


class FlatParams:
    """Re-homes a module's parameters into one flat fp32 buffer (16-byte aligned slices)."""

    def __init__(self, model: nn.Module):
        self.model = model
        self.params = list(model.parameters())
        dev = self.params[0].device
        self.offsets: List[Tuple[int, int]] = []
        off = 0
        for p in self.params:
            self.offsets.append((off, p.numel()))
            off += (p.numel() + 3) // 4 * 4
        self.total = off
        self.flat = torch.zeros(off, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(off, dtype=torch.float32, device=dev)
        self.grad_views: Dict[int, torch.Tensor] = {}
        with torch.no_grad():
            for p, (o, n) in zip(self.params, self.offsets):
                self.flat[o:o + n].copy_(p.detach().reshape(-1))
                p.data = self.flat[o:o + n].view(p.shape)
                self.grad_views[id(p)] = self.grad[o:o + n].view(p.shape)
        invalidate_packed_weights()

    def view_like(self, flat: torch.Tensor, module: nn.Module) -> None:
        """Point ``module``'s parameters (same structure as the model) at slices of ``flat``."""
        for p, (o, n) in zip(module.parameters(), self.offsets):
            p.data = flat[o:o + n].view(p.shape)

    def ranges_for(self, names: Sequence[str]) -> List[Tuple[int, int]]:
        """Merged [start, end) ranges of the parameters whose name starts with one of ``names``."""
        out: List[List[int]] = []
        for (n, _), (o, cnt) in zip(self.model.named_parameters(), self.offsets):
            if any(n.startswith(pref) for pref in names):
                end = o + (cnt + 3) // 4 * 4
                if out and out[-1][1] == o:
                    out[-1][1] = end
                else:
                    out.append([o, end])
        return [(a, b) for a, b in out]


class GradBuckets:
    """Data-parallel gradient exchange over the flat gradient buffer: a handful of large contiguous
    buckets in reverse execution order (head, Layer4, ..., Layer1, tail), each all-reduced (SUM)
    asynchronously as soon as the backward of its layers has been enqueued."""

    def __init__(self, fp: FlatParams, stage_names: Sequence[str], process_group=None):
        import torch.distributed as dist
        self.dist = dist
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.fp = fp
        self.stage_ranges: Dict[str, List[Tuple[int, int]]] = {}
        covered = 0
        for st, prefixes in stage_names:
            r = fp.ranges_for(prefixes)
            self.stage_ranges[st] = r
            covered += sum(b - a for a, b in r)
        assert covered == fp.total, "every parameter must belong to exactly one bucket"
        self.pending = []
        self.force = False
        # DM_DP_BUCKETS=1: ONE all-reduce of the whole flat gradient (16.7 MB for the STL10 UNet) behind the last stage
        # instead of one per stage -- fewer collective launches, no overlap with the backward (measurement switch)
        self.single = os.environ.get("DM_DP_BUCKETS", "") == "1"
        self.last_stage = stage_names[-1][0]
        # Transport.  RCCL ("nccl") reduces device memory in place on its own stream.  A host-side backend (gloo) cannot
        # touch device memory: instead of torch's gloo-CUDA shim (pinned staging on pool streams that gloo WORKER
        # THREADS synchronise and copy back from, concurrently with this thread's launches) the staging is done here,
        # in stream order, by this thread: D2H copy of the bucket on the issuing stream + an event; wait() synchronises
        # the event, reduces the HOST tensor and copies it back on the current stream.
        self.host_staged = bool(fp.grad.is_cuda and dist.is_initialized() and dist.get_backend(process_group) != "nccl")
        self.host = torch.empty(fp.total, dtype=torch.float32, pin_memory=True) if self.host_staged else None
        self.issued = self.completed = 0      # collectives issued / finished (stall diagnostics: describe())
        self.stage_events = []                # (stage, event recorded on the issuing stream right behind the launch)

    def describe(self) -> str:
        """One line for a stall dump: how many collectives were issued / completed and which stage events have fired."""
        ev = ", ".join("%s:%s" % (s, "done" if e.query() else "PENDING") for s, e in self.stage_events)
        return "collectives issued %d completed %d; stage events [%s]" % (self.issued, self.completed, ev)

    def launch(self, stage: str) -> None:
        if self.world == 1 and not self.force:   # (force: rehearse the collective path on one rank, bench --rehearse-dp)
            return
        if self.host_staged:
            if self.single and stage != self.last_stage:
                return
            ranges = [(0, self.fp.total)] if self.single else self.stage_ranges[stage]
            for a, b in ranges:
                self.host[a:b].copy_(self.fp.grad[a:b], non_blocking=True)   # on the issuing (weight-gradient) stream
            ev = torch.cuda.Event()
            ev.record()
            self.pending.append((ranges, ev))
            self.stage_events.append((stage, ev))
            self.issued += len(ranges)
            return
        # DM_DP_SYNC=1 (measurement switch): blocking-style collectives.  They are issued with the weight-gradient
        # stream current (unet hook -> kernels.call_on_side), so "blocking" orders that stream behind them, not the host
        # and not the main stream.
        sync = os.environ.get("DM_DP_SYNC", "") == "1"
        if self.single:
            if stage == self.last_stage:
                w = self.dist.all_reduce(self.fp.grad, op=self.dist.ReduceOp.SUM, group=self.pg, async_op=not sync)
                self.issued += 1
                if not sync:
                    self.pending.append(w)
            return
        for a, b in self.stage_ranges[stage]:
            w = self.dist.all_reduce(self.fp.grad[a:b], op=self.dist.ReduceOp.SUM, group=self.pg, async_op=not sync)
            self.issued += 1
            if not sync:
                self.pending.append(w)
        if torch.cuda.is_available() and self.fp.grad.is_cuda:
            ev = torch.cuda.Event()
            ev.record()
            self.stage_events.append((stage, ev))

    def wait(self) -> None:
        if self.host_staged:
            for ranges, ev in self.pending:
                ev.synchronize()                      # the bucket's bytes are in pinned host memory
                for a, b in ranges:
                    self.dist.all_reduce(self.host[a:b], op=self.dist.ReduceOp.SUM, group=self.pg)   # host tensor, blocking
                    self.fp.grad[a:b].copy_(self.host[a:b], non_blocking=True)   # current stream; its producers are done (event)
                    self.completed += 1
            self.pending = []
            self.stage_events = []
            return
        for w in self.pending:
            w.wait()
            self.completed += 1
        self.pending = []
        self.stage_events = []


def unet_stage_names(model) -> List[Tuple[str, List[str]]]:
    L = model.num_layers
    stages = [("head", ["late_conv1", "late_conv2", "output_conv"])]
    stages += [(f"Layer{i + 1}", [f"Layer{i + 1}."]) for i in range(L - 1, -1, -1)]
    stages += [("tail", ["input_conv", "time_mlp"])]
    return stages


class Trainer:
    """One object per process (= per GPU)."""

    def __init__(self, model, scheduler, *, lr=5e-5, weight_decay=1e-4, betas=(0.9, 0.99), eps=1e-8,
                 clip=1.0, ema_decay=0.995, accum=2, total_steps=100000, start_step=0,
                 low_t_weight: Optional[float] = 0.01, target="v", adamw=True, ema=None,
                 distributed=False, process_group=None):
        self.model, self.scheduler = model, scheduler
        self.lr, self.wd, self.betas, self.eps = lr, weight_decay, betas, eps
        self.clip, self.ema_decay, self.accum = clip, ema_decay, max(1, accum)
        self.total_steps, self.low_t_weight, self.target, self.adamw = total_steps, low_t_weight, target, adamw
        self.fp = FlatParams(model)
        dev = self.fp.flat.device
        self.m = torch.zeros_like(self.fp.flat)
        self.v = torch.zeros_like(self.fp.flat)
        self.ema = ema
        self.ema_flat = None
        if ema is not None:
            self.ema_flat = torch.zeros_like(self.fp.flat)
            with torch.no_grad():
                for p, (o, n) in zip(ema.module.parameters(), self.fp.offsets):
                    self.ema_flat[o:o + n].copy_(p.detach().reshape(-1))
            self.fp.view_like(self.ema_flat, ema.module)
        self.grads = Grads(self.fp.grad_views)
        self.norm = torch.zeros(2, dtype=torch.float32, device=dev)
        self.opt_step = 0          # optimizer steps taken
        self.sched_step = start_step
        self.micro = 0
        self.defer_optimizer = False   # tests: inspect the accumulated gradient before stepping
        self.world, self.rank = 1, 0
        self.buckets = None
        self.pg = process_group
        self.last_skipped = False  # micro_step(check_finite=True) found a non-finite loss and skipped the batch
        self._graph = self._gin = self._gloss = self._gctr = None   # enable_graph() / enable_step_graph()
        self._gwhole, self._gring, self._gopt = False, None, None
        if distributed:
            import torch.distributed as dist
            if dev.type == "cuda":
                # before the first collective of a lazily initialised process group creates the RCCL communicator (and
                # its streams): our streams first, or they share hardware queues (kernels.reserve_streams).  A caller
                # that initialises the group eagerly (device_id=...) calls reserve_streams itself, before that.
                K.reserve_streams(dev)
            self.buckets = GradBuckets(self.fp, unet_stage_names(model), process_group)
            self.world = self.buckets.world
            self.rank = dist.get_rank(process_group) if dist.is_initialized() else 0
            self.broadcast_state()
        # dropout masks: a function of (process seed, rank, optimizer step) -- ranks differ, a resumed run does not
        # replay step 0's masks (unet.DropoutSeeds)
        reseed_dropout(None, self.rank, start_step)
        model._bwd_hook = None

    def broadcast_state(self, src: int = 0) -> None:
        """Data parallel: every rank continues from rank ``src``'s parameters, EMA copy and Adam moments (one
        broadcast per flat buffer).  Replicas built from process-local RNG state, or of which only one loaded a
        checkpoint, are made identical here; afterwards they stay identical because every rank applies the same
        all-reduced gradient."""
        if self.world <= 1:
            return
        import torch.distributed as dist
        staged = self.buckets is not None and self.buckets.host_staged   # host-side backend: stage here (GradBuckets)
        for buf in (self.fp.flat, self.ema_flat, self.m, self.v):
            if buf is None:
                continue
            if staged:
                h = buf.cpu()
                dist.broadcast(h, src=src, group=self.pg)
                buf.copy_(h)
            else:
                dist.broadcast(buf, src=src, group=self.pg)
        invalidate_packed_weights()

    # ------------------------------------------------------------------ one micro-batch
    def _loss_weights(self, B: int, chw: int, t: torch.Tensor) -> torch.Tensor:
        if self.low_t_weight is None:
            key = (B * chw, None, t.device)
            if getattr(self, "_wflat_key", None) != key:
                self._wflat = torch.full((B,), 1.0 / (B * chw), dtype=torch.float32, device=t.device)
                self._wflat_key = key
            return self._wflat
        # exp(-k t) / (B chw) for every timestep once (the same elementwise ops the per-batch statement ran, so the
        # rows are bit-identical): one gather per micro-batch instead of four small kernels
        key = (B * chw, self.low_t_weight, t.device)
        if getattr(self, "_wtab_key", None) != key:
            steps = torch.arange(self.scheduler.alpha_t.numel(), device=t.device)
            self._wtab, self._wtab_key = torch.exp(-self.low_t_weight * steps.float()) / (B * chw), key
        return self._wtab[t]

    def _micro_body(self, x, context, t, e):
        """q-sample -> forward -> loss + dL/dout -> backward (weight gradients joined), all on device: the part of a
        micro-step a HIP graph can hold.  Returns the loss tensor [1]."""
        B = x.shape[0]
        chw = x.numel() // B
        if self._gctr is not None:
            K.bump_dropout_counter(self._gctr)
        xt, v = K.qsample_vtarget(x, e, self.scheduler.alpha_t[t].contiguous())
        tgt = v if self.target == "v" else e
        y, saved = self.model._fwd(xt, t, context)
        loss, dout = K.mse_loss_fwd_bwd(y, tgt, self._loss_weights(B, chw, t).contiguous(), grad_scale=1.0 / self.accum)
        self.model._bwd(dout, saved, self.grads, join=True)
        return loss

    def enable_graph(self, x: torch.Tensor, context, t: torch.Tensor, e: torch.Tensor, warmup: int = 3) -> None:
        """Capture the micro-step (``_micro_body``) for inputs of this shape as ONE HIP graph; ``micro_step`` then
        copies its arguments into the graph's static input buffers and replays it -- one host call instead of a few
        hundred kernel launches (the launch-bound small-shape configurations: reference main.py, 32x32x1).  The
        optimizer step stays eager (its scalars change every step).  Dropout masks still differ per replay: every
        dropout kernel adds a device counter to its seed, bumped by the graph's first node (dm_set_dropout_counter).
        Call at an accumulation boundary (the warm-up runs accumulate into the gradient buffer, which is re-zeroed);
        single-process only (the data-parallel bucket hooks issue collectives from Python)."""
        if self.buckets is not None and self.world > 1:
            raise RuntimeError("enable_graph: not available under data parallelism")
        if self.micro != 0:
            raise RuntimeError("enable_graph: call at an accumulation boundary")
        dev = x.device
        if self._graph is not None:
            self.disable_graph()
        self._gctr = torch.zeros(1, dtype=torch.int32, device=dev)
        K.set_dropout_counter(self._gctr)   # raises if another live Trainer holds the registration
        import weakref
        self._gfin = weakref.finalize(self, K.release_dropout_counter, self._gctr.data_ptr())
        K.pin_workspaces()                  # replay bakes workspace addresses in: outgrown buffers stay alive (kernels.workspace)
        self._gin = (x.clone(), None if context is None else context.clone(), t.clone(), e.clone())
        self.model._bwd_hook = None
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):          # eager warm-up off the default stream (workspaces, packed weights, tables)
            for _ in range(max(1, warmup)):
                self._micro_body(*self._gin)
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.fp.grad.zero_()
        graph = torch.cuda.CUDAGraph()
        K.CAPTURE_KEEPALIVE = []
        try:
            with torch.cuda.graph(graph):
                self._gloss = self._micro_body(*self._gin)
        finally:
            K.CAPTURE_KEEPALIVE = None
        self.fp.grad.zero_()                   # capture enqueues nothing, but keep the contract explicit
        self._graph = graph

    def disable_graph(self) -> None:
        if self._gctr is not None:
            K.release_dropout_counter(self._gctr.data_ptr())
            self._gfin.detach()
            K.unpin_workspaces()
        if getattr(self, "_gwhole", False):
            invalidate_packed_weights()    # the replayed repacks did not touch the host-side staleness keys
        self._graph = None
        self._gin = self._gloss = None
        self._gctr = None
        self._gwhole = False
        self._gring = self._gopt = None

    def micro_step(self, x: torch.Tensor, context, t: torch.Tensor, e: torch.Tensor, *,
                   step_now: Optional[bool] = None, check_finite: bool = False) -> torch.Tensor:
        """x,e: [B,C,H,W] fp32 on the device; t: [B] long (device); context: [B,D] fp32 or None.
        Returns the (un-synchronised) loss tensor [1].
        step_now overrides the accumulation gate (reference main_stl10.py:506: the last batch of an epoch
        steps even if the accumulation window is not full); check_finite mirrors main_stl10.py:484-486
        (a non-finite loss skips the batch before backward) at the price of one host sync."""
        last = ((self.micro + 1) % self.accum == 0) if step_now is None else bool(step_now)
        if self._graph is not None and getattr(self, "_gwhole", False):
            raise RuntimeError("a whole-step graph is live: use graph_step() (or disable_graph() first)")
        if self._graph is not None and not check_finite and x.shape == self._gin[0].shape and \
                (context is None) == (self._gin[1] is None):
            gx, gc, gt, ge = self._gin
            gx.copy_(x, non_blocking=True)
            ge.copy_(e, non_blocking=True)
            gt.copy_(t, non_blocking=True)
            if gc is not None:
                gc.copy_(context, non_blocking=True)
            self._graph.replay()
            self.last_skipped = False
            self.micro = 0 if last else self.micro + 1
            if last and not self.defer_optimizer:
                self.optimizer_step()
            return self._gloss
        B = x.shape[0]
        chw = x.numel() // B
        if self._gctr is not None:
            K.bump_dropout_counter(self._gctr)
        abar = self.scheduler.alpha_t[t].contiguous()
        xt, v = K.qsample_vtarget(x, e, abar)
        tgt = v if self.target == "v" else e
        y, saved = self.model._fwd(xt, t, context)
        w = self._loss_weights(B, chw, t)
        loss, dout = K.mse_loss_fwd_bwd(y, tgt, w.contiguous(), grad_scale=1.0 / self.accum)
        self.last_skipped = False
        if check_finite:
            ok = torch.isfinite(loss).all()
            if self.buckets is not None and self.world > 1:
                # data parallel: skip on ALL ranks or on none -- a rank that returned alone would never launch its
                # bucket all-reduces or the optimizer step, and the others would wait for it forever
                flag = ok.to(torch.int32)
                if self.buckets.host_staged:
                    flag = flag.cpu()
                self.buckets.dist.all_reduce(flag, op=self.buckets.dist.ReduceOp.MIN, group=self.pg)
                ok = flag
            if not bool(ok.item()):
                self.last_skipped = True
                return loss  # reference: "continue" -- no backward, no accumulation-counter change
        if self.buckets is not None and last:
            self.model._bwd_hook = self.buckets.launch
        self.model._bwd(dout, saved, self.grads, join=last)
        self.model._bwd_hook = None
        self.micro = 0 if last else self.micro + 1
        if last and not self.defer_optimizer:
            self.optimizer_step()
        return loss

    def optimizer_state_dict(self) -> dict:
        """The fused optimizer's state in torch.optim.AdamW.state_dict() form (the 'optimizer' entry of
        the reference checkpoint, main_stl10.py:533-539): per-parameter step / exp_avg / exp_avg_sq."""
        state = {}
        for i, (p, (o, n)) in enumerate(zip(self.fp.params, self.fp.offsets)):
            state[i] = {"step": torch.tensor(float(self.opt_step)),
                        "exp_avg": self.m[o:o + n].view_as(p).clone(),
                        "exp_avg_sq": self.v[o:o + n].view_as(p).clone()}
        group = {"lr": self.current_lr(), "betas": tuple(self.betas), "eps": self.eps, "weight_decay": self.wd,
                 "amsgrad": False, "maximize": False, "foreach": None, "capturable": False,
                 "differentiable": False, "fused": None, "initial_lr": self.lr,
                 "params": list(range(len(self.fp.params)))}
        return {"state": state, "param_groups": [group]}

    def load_optimizer_state_dict(self, sd: dict) -> None:
        for i, (p, (o, n)) in enumerate(zip(self.fp.params, self.fp.offsets)):
            st = sd["state"].get(i)
            if st is None:
                continue
            self.m[o:o + n].copy_(st["exp_avg"].reshape(-1))
            self.v[o:o + n].copy_(st["exp_avg_sq"].reshape(-1))
            self.opt_step = int(float(st["step"]))

    # ------------------------------------------------------------------ clip + AdamW + EMA
    def current_lr(self) -> float:
        if not self.total_steps:  # no LambdaLR (reference main.py:150: plain Adam at a constant rate)
            return self.lr
        return self.lr * lr_lambda(self.sched_step, self.total_steps)

    def optimizer_step(self) -> None:
        K.join_side(self.fp.flat.device)  # weight gradients of un-joined micro-batches (side stream)
        if self.buckets is not None:
            self.buckets.wait()
        pre = 1.0 / self.world
        clipbuf = None
        if self.clip and self.clip > 0:
            K.grad_norm_clipcoef(self.fp.grad, self.clip, pre_scale=pre, out=self.norm)
            clipbuf = self.norm
        self.opt_step += 1
        K.adamw_ema_step(self.fp.flat, self.fp.grad, self.m, self.v, self.ema_flat, clipbuf, lr=self.current_lr(),
                         beta1=self.betas[0], beta2=self.betas[1], eps=self.eps, wd=self.wd, step=self.opt_step,
                         ema_decay=self.ema_decay, pre_scale=pre, adamw=self.adamw)
        self.sched_step += 1
        self.fp.grad.zero_()
        invalidate_packed_weights()
        repack_all(self.model)  # all bf16 shadows of the updated weights in one launch

    def _optimizer_body_dev(self) -> None:
        """The optimizer step with NO host-computed scalar: step count, learning rate and bias corrections come from the
        device struct ``self._gopt`` that a one-thread kernel advances (dm_opt_state_advance) -- the form a HIP graph can
        hold.  Same kernels and arithmetic as ``optimizer_step`` otherwise (single process: no bucket wait)."""
        K.join_side(self.fp.flat.device)
        clipbuf = None
        if self.clip and self.clip > 0:
            K.grad_norm_clipcoef(self.fp.grad, self.clip, pre_scale=1.0, out=self.norm)
            clipbuf = self.norm
        warm = max(1, int(0.05 * self.total_steps)) if self.total_steps else 0     # lr_lambda's warm-up length
        K.opt_state_advance(self._gopt, base_lr=self.lr, warmup_steps=warm, total_steps=self.total_steps or 0,
                            beta1=self.betas[0], beta2=self.betas[1])
        K.adamw_ema_step_dev(self.fp.flat, self.fp.grad, self.m, self.v, self.ema_flat, clipbuf, self._gopt,
                             beta1=self.betas[0], beta2=self.betas[1], eps=self.eps, wd=self.wd, ema_decay=self.ema_decay,
                             adamw=self.adamw)
        self.fp.grad.zero_()
        repack_all(self.model, force=True)

    def enable_step_graph(self, ring: Sequence[Tuple], warmup: int = 2) -> None:
        """Capture ONE WHOLE OPTIMIZER STEP -- ``accum`` micro-steps, clip, AdamW, EMA, gradient clear and the repack of
        the bf16 weight shadows -- as one HIP graph; ``graph_step()`` then costs the host one ``replay()``.  For the
        launch-bound small-shape configuration (reference main.py:159-188, 32x32x1: every kernel is tiny).

        ``ring``: a sequence of input batches ``(x, context-or-None, t, e)`` of one shape, resident on the device; the
        captured step fetches micro-batch k from ring slot (k mod len(ring)) with a device-side copy indexed by the
        same device counter that reseeds the dropout masks (dm_ring_fetch), so no host copy feeds a replay -- the data
        loader refills ring slots asynchronously, off the step's critical path.  The step count, LambdaLR rate and Adam
        bias corrections advance on the device (dm_opt_state_advance).  Single process only."""
        if self.buckets is not None and self.world > 1:
            raise RuntimeError("enable_step_graph: not available under data parallelism")
        if self.micro != 0:
            raise RuntimeError("enable_step_graph: call at an accumulation boundary")
        if self._graph is not None:
            self.disable_graph()
        x0, c0, t0, e0 = ring[0]
        dev = x0.device
        import weakref
        self._gctr = torch.zeros(1, dtype=torch.int32, device=dev)
        K.set_dropout_counter(self._gctr)
        self._gfin = weakref.finalize(self, K.release_dropout_counter, self._gctr.data_ptr())
        K.pin_workspaces()
        # the ring: one contiguous [nslots, ...] tensor per input; static per-micro-step landing buffers
        self._gring = (torch.stack([b[0] for b in ring]).contiguous(),
                       None if c0 is None else torch.stack([b[1] for b in ring]).contiguous(),
                       torch.stack([b[2] for b in ring]).contiguous(), torch.stack([b[3] for b in ring]).contiguous())
        self._gin = (torch.empty_like(x0), None if c0 is None else torch.empty_like(c0), torch.empty_like(t0),
                     torch.empty_like(e0))
        self._gopt = K.opt_state_new(dev, self.opt_step, self.sched_step)
        self.model._bwd_hook = None

        def step_body():
            loss = None
            for _ in range(self.accum):
                for ring_t, dst in zip(self._gring, self._gin):
                    if ring_t is not None:
                        K.ring_fetch(ring_t, dst, self._gctr)   # slot = counter % nslots, BEFORE this micro-step's bump
                loss = self._micro_body(*self._gin)             # (bumps the counter first thing)
            self._optimizer_body_dev()
            return loss

        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        snap = [b.clone() for b in (self.fp.flat, self.m, self.v) + ((self.ema_flat,) if self.ema_flat is not None else ())]
        with torch.cuda.stream(side):          # eager warm-up off the default stream (workspaces, packed weights, tables)
            for _ in range(max(1, warmup)):
                step_body()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        # the warm-up steps were real optimizer steps: put parameters, moments, EMA, counters and the schedule back
        for buf, sv in zip((self.fp.flat, self.m, self.v) + ((self.ema_flat,) if self.ema_flat is not None else ()), snap):
            buf.copy_(sv)
        self.fp.grad.zero_()
        self._gctr.zero_()
        self._reset_opt_state()
        repack_all(self.model, force=True)
        torch.cuda.synchronize(dev)
        graph = torch.cuda.CUDAGraph()
        K.CAPTURE_KEEPALIVE = []
        try:
            with torch.cuda.graph(graph):
                self._gloss = step_body()
        finally:
            K.CAPTURE_KEEPALIVE = None
        self._graph = graph
        self._gwhole = True

    def _reset_opt_state(self):
        """(Re)initialise the device schedule IN PLACE (the captured kernels hold its address)."""
        from . import _lib
        _lib.check(_lib.lib().dm_opt_state_init(self._gopt.data_ptr(), int(self.opt_step), int(self.sched_step), K._stream()),
                   "dm_opt_state_init")
        return self._gopt

    def graph_step(self) -> torch.Tensor:
        """Replay the captured whole optimizer step (enable_step_graph): ONE host call.  Returns the last micro-batch's
        (un-synchronised) loss tensor."""
        if self._graph is None or not getattr(self, "_gwhole", False):
            raise RuntimeError("graph_step: call enable_step_graph() first")
        self._graph.replay()
        self.opt_step += 1          # host mirrors of the device-side counters (checkpointing, current_lr())
        self.sched_step += 1
        return self._gloss

    def grad_norm(self) -> float:
        return float(self.norm[0].item())
