"""Replay of a fixed-shape training step as hipGraphs.

The flow decoder is a chain of several hundred short dependent kernels per step: launched eagerly from Python the step is bound by host
launch work (12 ms at B = 32 on an MI355X against 6 ms of GPU time).  `GraphedTrainStep` captures forward + loss + backward (+ clip, RAdam)
once per input shape and replays it; parameters, gradients and the returned loss live at fixed addresses, new batches are copied into
static input buffers.  What the capture needs from the model is already in place: dropout seeds are re-drawn on the device, weight-gradient
job tables come from pinned memory, the flat parameter storage is built during the eager warm-up passes.

    step = GraphedTrainStep(model, lambda m, tokens, tl, mels, ml: total_loss(m, tokens, tl, mels, ml), optimizer=opt, scheduler=sch)
    loss = step(tokens, token_lengths, mels, mel_lengths)

Data parallel (torch.distributed initialised with more than one rank, or `distributed.SINGLE_RANK_IS_DIST`): the step is captured as
THREE graphs around the gradient exchange, the structure bench.py measures - (1) forward, losses, backward up to and including the k-tap
weight gradients; their all-reduce (with the encoder's and the ActNorm / 1x1 gradients, ~100 of the 114 MB) is issued asynchronously and
runs under (2) the 1x1 weight-gradient groups and the weight-norm backward of their classes, whose 15 MB are reduced last; (3) clip +
RAdam on the reduced gradients.  The reference's step is single-process (Train.py:182-238).

Shapes: every new input shape costs `warmup` eager forward + backward passes WITHOUT a parameter update plus a capture, then exactly one
training step (the replay).  All graphs share one memory pool (activations of one shape are reused by the next capture) and at most
`max_graphs` shapes stay captured (least recently used first out).
Capture the model BEFORE any eager `backward()` of it ran on another stream in this process (or after every tensor of those earlier
steps - losses, outputs - has been released): autograd keeps a parameter's AccumulateGrad node on the stream of its first use while
an old graph is alive, and a captured backward that has to hop to that stream crashes hipStreamEndCapture on ROCm 7.2.
"""
from collections import OrderedDict

import torch

from . import _lib


class GraphedTrainStep:
    def __init__(self, model, loss_fn, warmup=2, optimizer=None, scheduler=None, max_grad_norm=None, max_graphs=16):
        """loss_fn(model, *inputs) -> scalar loss tensor (forward + losses).  warmup >= 1: eager forward + backward passes before the capture
        of a new shape (the very first one also runs the ActNorm data-dependent init and builds the flat parameter storage); they apply NO
        update.  optimizer (glow_tts_amd.optim.RAdam) / scheduler / max_grad_norm: the rest of `Train.py:218-233` - clip_grad_norm_,
        optimizer.step(), scheduler.step() - joins the captured step: the clip coefficient stays on the device, the step's hyper-parameters
        are sent from the host before every replay (`RAdam.advance_host`).  `steps_taken` counts the optimizer steps applied: one per call."""
        self.model, self.loss_fn, self.warmup = model, loss_fn, max(1, int(warmup))
        self.optimizer, self.scheduler, self.max_grad_norm = optimizer, scheduler, max_grad_norm
        self.params = [p for p in model.parameters() if p.requires_grad]
        self.stream = torch.cuda.Stream()
        self.graphs = OrderedDict()
        self.max_graphs = max(1, int(max_graphs))
        self.pool = None
        self.steps_taken = 0
        self._opt_ready = False

    # ------------------------------------------------------------------ pieces of a step
    def _dp(self):
        from . import distributed as gd
        return gd.is_dist()

    def _fwd_bwd(self, inputs):
        loss = self.loss_fn(self.model, *inputs)          # a scalar tensor or `alignment.LossTerms` (whose sum is computed by the first `detach()`)
        self.model.zero_grad(set_to_none=True)
        loss.backward()
        return loss

    def _update(self):
        coef = None
        if self.max_grad_norm is not None:
            from .optim import grad_norm_and_coef
            _, coef = grad_norm_and_coef(self.params, self.max_grad_norm)
        self.optimizer.step(grad_scale=coef)

    def _uncount_capture_pass(self):
        """A captured `optimizer.step()` advanced the step counters on the host without running a kernel."""
        for p in self.params:
            st = self.optimizer.state.get(p)
            if p.grad is not None and st is not None and "step" in st:
                st["step"] -= 1

    # ------------------------------------------------------------------ capture
    def _capture(self, inputs):
        """-> (entry, eager_loss): eager_loss is not None when the triggering batch's training step was already taken eagerly (the first
        shape ever: the optimizer's moments and hyper-parameter buffers must exist before a capture)."""
        static_in = [t.clone() if torch.is_tensor(t) else t for t in inputs]
        dp = self._dp()
        cur = torch.cuda.current_stream()
        applied = False
        # every eager pass that precedes the capture runs on the capture-side stream as well: a backward that first ran on another
        # stream makes the captured AccumulateGrad nodes hop streams, which hipStreamEndCapture does not survive on ROCm 7.2
        self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            for _ in range(self.warmup):
                eager_loss = self._fwd_bwd(static_in).detach()  # no update: a new shape must not cost extra optimizer / scheduler steps
            if self.optimizer is not None and not self._opt_ready:
                if dp:
                    self._reducer_all().reduce(average=False)
                self._update()                                 # allocates moments, job tables, hyper-parameter words: a real step on this batch
                if self.scheduler is not None:
                    self.scheduler.step()
                self._opt_ready, applied = True, True
                self.steps_taken += 1
        cur.wait_stream(self.stream)
        torch.cuda.synchronize()
        if dp:
            from .distributed import before_capture
            before_capture()                                   # the process group's watchdog holds no Work when the captures begin
        _lib.refill_spares()                                   # pinned job tables for this capture (earlier captures kept theirs)
        if self.pool is None:
            self.pool = torch.cuda.graph_pool_handle()         # one pool for every shape: a capture reuses what earlier ones have freed
        mode = "thread_local" if dp else "global"              # (RCCL's proxy / watchdog threads make their own runtime calls; see distributed.py)
        keep = []                                              # pinned job tables this entry's copy nodes read
        entry = {"in": static_in, "keep": keep, "tail": None, "opt": None, "early": None, "late": None}
        g = torch.cuda.CUDAGraph()
        with _lib.pinned_sink(keep):
            if dp:
                from . import decoder as D
                from .distributed import FlatGradReducer
                with D.defer_tail_wgrads():
                    with torch.cuda.graph(g, pool=self.pool, capture_error_mode=mode):
                        loss = self._fwd_bwd(static_in).detach()
                tail = torch.cuda.CUDAGraph()
                with torch.cuda.graph(tail, pool=self.pool, capture_error_mode=mode):
                    D.flush_tail_wgrads()
                stacks = getattr(self.model, "_dec_stacks", None)
                tail_ids = {id(p) for p in stacks.tail_leaves()} if stacks is not None else set()
                entry["tail"] = tail
                # (frozen: __call__ points every .grad at this entry's buffers before the replay - their addresses cannot change)
                entry["early"] = FlatGradReducer([p for p in self.params if id(p) not in tail_ids], frozen=True)
                entry["late"] = FlatGradReducer([p for p in self.params if id(p) in tail_ids], frozen=True)
                if self.optimizer is not None:
                    og = torch.cuda.CUDAGraph()                # the update reads the REDUCED gradients: its own graph behind the exchange
                    with torch.cuda.graph(og, pool=self.pool, capture_error_mode=mode):
                        self._update()
                    self._uncount_capture_pass()
                    entry["opt"] = og
            else:
                with torch.cuda.graph(g, pool=self.pool, capture_error_mode=mode):
                    loss = self._fwd_bwd(static_in)
                    if self.optimizer is not None:
                        self._update()
                    loss = loss.detach()                       # (the logged sum: behind the update on this stream)
                if self.optimizer is not None:
                    self._uncount_capture_pass()
        entry.update(graph=g, loss=loss, grads=[p.grad for p in self.params])
        return entry, (eager_loss.detach() if applied else None)

    def _reducer_all(self):
        from .distributed import FlatGradReducer
        return FlatGradReducer(self.params)

    # ------------------------------------------------------------------ one step
    def __call__(self, *inputs):
        key = tuple((tuple(t.shape), t.dtype) if torch.is_tensor(t) else t for t in inputs)
        e = self.graphs.get(key)
        if e is None:
            e, eager_loss = self._capture(inputs)
            self.graphs[key] = e
            while len(self.graphs) > self.max_graphs:          # least recently used shape: its graphs, static buffers and gradients go
                self.graphs.popitem(last=False)
            if eager_loss is not None:
                return eager_loss
        self.graphs.move_to_end(key)
        for s, t in zip(e["in"], inputs):
            if torch.is_tensor(t) and s.data_ptr() != t.data_ptr():
                s.copy_(t, non_blocking=True)
        for p, gr in zip(self.params, e["grads"]):             # several cached shapes: point .grad at this graph's buffers
            p.grad = gr
        opt = self.optimizer
        if e["tail"] is None:
            if opt is not None:
                opt.advance_host()                             # stream-ordered, before the replay on the same stream
            e["graph"].replay()
        else:
            e["graph"].replay()
            pending = e["early"].begin()                       # ~100 MB of gradients on the wire ...
            e["tail"].replay()                                 # ... under the 1x1 weight-gradient groups
            e["late"].reduce(average=False)
            e["early"].finish(pending)
            if e["opt"] is not None:
                opt.advance_host()
                e["opt"].replay()
        if opt is not None:
            self.steps_taken += 1
        if self.scheduler is not None:
            self.scheduler.step()
        return e["loss"]
