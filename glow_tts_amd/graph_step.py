"""Replay of a fixed-shape training step as ONE hipGraph.

The flow decoder is a chain of ~600 short dependent kernels per step: launched eagerly from Python the step is bound by host launch
work (14 ms at B = 32 on an MI355X against 7 ms of GPU time).  `GraphedTrainStep` captures forward + loss + backward once per input
shape and replays it; parameters, gradients and the returned loss live at fixed addresses, new batches are copied into static input
buffers.  What the capture needs from the model is already in place: dropout seeds are re-drawn on the device, weight-gradient job
tables come from pinned memory, the flat parameter storage is built during the eager warm-up steps.

    step = GraphedTrainStep(model, lambda m, tokens, tl, mels, ml: total_loss(m, tokens, tl, mels, ml))
    loss = step(tokens, token_lengths, mels, mel_lengths)      # gradients are in p.grad, as after loss.backward()
    optimizer.step()

Use one padded shape (or a few length buckets): every new shape costs `warmup` eager steps plus a capture.
Capture the model BEFORE any eager `backward()` of it ran on another stream in this process (or after every tensor of those earlier
steps - losses, outputs - has been released): autograd keeps a parameter's AccumulateGrad node on the stream of its first use while
an old graph is alive, and a captured backward that has to hop to that stream crashes hipStreamEndCapture on ROCm 7.2.
"""
import torch

from . import _lib


class GraphedTrainStep:
    def __init__(self, model, loss_fn, warmup=3, optimizer=None, scheduler=None, max_grad_norm=None):
        """loss_fn(model, *inputs) -> scalar loss tensor (forward + losses).  warmup >= 2: eager steps before the capture (the first
        one also runs the ActNorm data-dependent init and builds the flat parameter storage).
        optimizer (glow_tts_amd.optim.RAdam) / scheduler / max_grad_norm: the rest of `Train.py:218-233` - clip_grad_norm_, optimizer.step(),
        scheduler.step() - joins the graph: the clip coefficient stays on the device, the step's hyper-parameters are sent from the host
        before every replay (`RAdam.advance_host`).  NOTE: the warm-up steps of every new input shape are REAL optimizer / scheduler steps
        on the triggering batch; `steps_taken` counts every optimizer step actually applied (warm-up + replays), so a trainer that keeps
        its own step counter (Train.py:234 `self.steps += 1`) can stay in sync with the optimizer: `trainer.steps = step.steps_taken`."""
        self.model, self.loss_fn, self.warmup = model, loss_fn, max(2, int(warmup))
        self.optimizer, self.scheduler, self.max_grad_norm = optimizer, scheduler, max_grad_norm
        self.params = [p for p in model.parameters() if p.requires_grad]
        self.stream = torch.cuda.Stream()
        self.graphs = {}
        self.steps_taken = 0

    def _fwd_bwd(self, inputs):
        loss = self.loss_fn(self.model, *inputs)
        self.model.zero_grad(set_to_none=True)
        loss.backward()
        if self.optimizer is not None:
            coef = None
            if self.max_grad_norm is not None:
                from .optim import grad_norm_and_coef
                _, coef = grad_norm_and_coef(self.params, self.max_grad_norm)
            self.optimizer.step(grad_scale=coef)
        return loss

    def _capture(self, inputs):
        static_in = [t.clone() if torch.is_tensor(t) else t for t in inputs]
        cur = torch.cuda.current_stream()
        # every eager step that precedes the capture runs on the capture-side stream as well: a backward that first ran on another
        # stream makes the captured AccumulateGrad nodes hop streams, which hipStreamEndCapture does not survive on ROCm 7.2
        self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            for _ in range(self.warmup):
                self._fwd_bwd(static_in)
                if self.optimizer is not None:
                    self.steps_taken += 1
                if self.scheduler is not None:
                    self.scheduler.step()
        cur.wait_stream(self.stream)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        keep = []                                          # pinned job tables this graph's copy nodes read: owned by the graph's entry
        with _lib.pinned_sink(keep):
            with torch.cuda.graph(g):
                loss = self._fwd_bwd(static_in)            # (with an optimizer: this capture pass advanced the step counters once; the
        grads = [p.grad for p in self.params]              #  captured kernels only run at the replays)
        if self.optimizer is not None:
            for p in self.params:                          # exactly the parameters step() counted: those that have a gradient
                st = self.optimizer.state.get(p)
                if p.grad is not None and st is not None and "step" in st:
                    st["step"] -= 1
        return g, static_in, loss, grads, keep

    def __call__(self, *inputs):
        key = tuple((tuple(t.shape), t.dtype) if torch.is_tensor(t) else t for t in inputs)
        if key not in self.graphs:
            self.graphs[key] = self._capture(inputs)
        g, static_in, loss, grads, _ = self.graphs[key]
        for s, t in zip(static_in, inputs):
            if torch.is_tensor(t) and s.data_ptr() != t.data_ptr():
                s.copy_(t, non_blocking=True)
        for p, gr in zip(self.params, grads):              # several cached shapes: point .grad at this graph's buffers
            p.grad = gr
        if self.optimizer is not None:
            self.optimizer.advance_host()                  # stream-ordered, before the replay on the same stream
            self.steps_taken += 1
        g.replay()
        if self.scheduler is not None:
            self.scheduler.step()
        return loss
