"""Optimizer side of the training step, MI355X-native: drop-ins for the reference's `Radam.RAdam` (Radam.py:12-90),
`Noam_Scheduler.Modified_Noam_Scheduler` / `Noam_Scheduler` (Noam_Scheduler.py:5-29) and the `torch.nn.utils.clip_grad_norm_` call of
`Train.py:228-231`.

The reference updates ~500 parameter tensors in a Python loop with ~10 small torch kernels each; here the whole model is ONE
multi-tensor launch (`glowtts_radam_step`) over a device job table, and the global gradient norm is one two-stage reduction
(`glowtts_multi_grad_norm`).  Parameters whose data / gradients / moments are adjacent in memory - the decoder's stacked weight
classes (decoder.LeafStack) - collapse into one job each.  State keys and `state_dict()` layout are the reference's
(`step`, `exp_avg`, `exp_avg_sq`), so `Train.py:516,540` checkpoints load unchanged.  GPU only: no CPU fallback."""
import ctypes
import math

import torch

from . import _lib

c_i64, c_p = ctypes.c_int64, ctypes.c_void_p


class OptJob(ctypes.Structure):
    """Mirror of `glowtts_opt_job`."""
    _fields_ = [("p", c_p), ("g", c_p), ("m", c_p), ("v", c_p), ("n", c_i64), ("block0", c_i64)]


_decl = False


def _L():
    global _decl
    L = _lib.lib()
    if not _decl:
        L.glowtts_opt_chunk.restype = ctypes.c_int
        L.glowtts_multi_grad_norm.argtypes = [c_p, ctypes.c_int, ctypes.c_int, ctypes.c_float, c_p, c_p, c_p]
        L.glowtts_multi_grad_scale.argtypes = [c_p, ctypes.c_int, ctypes.c_int, c_p, c_p]
        L.glowtts_radam_step.argtypes = [c_p, ctypes.c_int, ctypes.c_int, c_p, c_p, c_p]
        _decl = True
    return L


class _JobTable:
    """Device table of coalesced jobs for a list of (p, g, m, v) tensor tuples (m / v may be None: gradient-only tables).
    Rebuilt only when an address changes - and always under hipGraph capture, where the upload becomes a copy node reading a pinned
    table owned by that graph (_lib.staged_upload): tables are never shared between captured graphs."""

    def __init__(self):
        self.sig, self.table, self.njobs, self.blocks = None, None, 0, 0

    def update(self, items, device, canonical=False):
        """canonical: one job per tensor in the order given, no coalescing - for REDUCTIONS (the gradient norm), whose summation order must
        not depend on where the allocator of this process happened to place the tensors: data-parallel replicas otherwise compute clip
        coefficients that differ in the last bit and drift apart (found by bench.py's replica check).  Elementwise launches may coalesce."""
        sig = (canonical,) + tuple((p.data_ptr(), g.data_ptr(), m.data_ptr() if m is not None else 0, p.numel()) for p, g, m, v in items)
        capturing = torch.cuda.is_current_stream_capturing()
        if sig == self.sig and not capturing:
            return
        chunk = _L().glowtts_opt_chunk()
        order = range(len(items)) if canonical else sorted(range(len(items)), key=lambda i: sig[1 + i][0])
        jobs, cur = [], None
        for i in order:
            p, g, m, v = items[i]
            ptrs = [p.data_ptr(), g.data_ptr(), m.data_ptr() if m is not None else 0, v.data_ptr() if v is not None else 0]
            n = p.numel()
            if not canonical and cur is not None and all(a + 4 * cur[4] == b or (a == 0 and b == 0) for a, b in zip(cur[:4], ptrs)):
                cur[4] += n                                    # adjacent in all four tensors: extend the run
            else:
                cur = ptrs + [n]
                jobs.append(cur)
        arr, b0 = (OptJob * len(items))(), 0                   # fixed size (one slot per tensor): coalescing may differ between steps
        for k, (pp, gp, mp, vp, n) in enumerate(jobs):
            arr[k].p, arr[k].g, arr[k].m, arr[k].v, arr[k].n, arr[k].block0 = pp, gp, mp or None, vp or None, n, b0
            b0 += (n + chunk - 1) // chunk
        self.table = _lib.staged_upload(bytes(arr), device)      # (a captured step gets a pinned table of its own)
        self.sig, self.njobs, self.blocks = (None if capturing else sig), len(jobs), b0


_CLIP_TABLES = {}


def clip_grad_norm_(parameters, max_norm, _table_key=None):
    """torch.nn.utils.clip_grad_norm_(parameters, max_norm) for fp32 device gradients (Train.py:228-231): returns the total L2 norm
    (0-d device tensor, no host sync) and scales every gradient in place by min(1, max_norm / (norm + 1e-6))."""
    params = [p for p in ([parameters] if torch.is_tensor(parameters) else list(parameters)) if p.grad is not None]
    if not params:
        return torch.zeros(())
    dev = params[0].grad.device
    key = _table_key if _table_key is not None else (id(params[0]), len(params))
    tab = _CLIP_TABLES.setdefault(key, _JobTable())
    tab.update([(p.grad, p.grad, None, None) for p in params], dev, canonical=True)
    partial = torch.empty(tab.blocks, device=dev)
    out = torch.empty(2, device=dev)
    L = _L()
    _lib.check(L.glowtts_multi_grad_norm(tab.table.data_ptr(), tab.njobs, tab.blocks, float(max_norm), partial.data_ptr(), out.data_ptr(),
                                         _lib.stream()), "glowtts_multi_grad_norm")
    _lib.check(L.glowtts_multi_grad_scale(tab.table.data_ptr(), tab.njobs, tab.blocks, out.data_ptr() + 4, _lib.stream()), "glowtts_multi_grad_scale")
    return out[0]


def radam_scalars(step, beta1, beta2):
    """(N_sma, step_size) of Radam.py:63-79 for the 1-based step count (host double arithmetic, like the reference)."""
    beta2_t = beta2 ** step
    n_max = 2.0 / (1.0 - beta2) - 1.0
    n_sma = n_max - 2.0 * step * beta2_t / (1.0 - beta2_t)
    if n_sma >= 5:
        size = math.sqrt((1 - beta2_t) * (n_sma - 4) / (n_max - 4) * (n_sma - 2) / n_sma * n_max / (n_max - 2)) / (1 - beta1 ** step)
    else:
        size = 1.0 / (1 - beta1 ** step)
    return n_sma, size


class RAdam(torch.optim.Optimizer):
    """Rectified Adam with the reference's constructor, update rule and state layout (Radam.py:12-90)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._tables, self._hyper = {}, {}

    def _init_state(self, plist):
        """exp_avg / exp_avg_sq for parameters that have none yet.  Parameters that are adjacent in memory (views of one flat tensor) get
        adjacent moments, so that the whole run is one job."""
        fresh = sorted([p for p in plist if len(self.state[p]) == 0], key=lambda p: p.data_ptr())
        i = 0
        while i < len(fresh):
            j, end = i + 1, fresh[i].data_ptr() + 4 * fresh[i].numel()
            while j < len(fresh) and fresh[j].data_ptr() == end:
                end += 4 * fresh[j].numel(); j += 1
            total = sum(p.numel() for p in fresh[i:j])
            m, v = torch.zeros(total, device=fresh[i].device), torch.zeros(total, device=fresh[i].device)
            off = 0
            for p in fresh[i:j]:
                st = self.state[p]
                st["step"] = 0
                st["exp_avg"], st["exp_avg_sq"] = m[off:off + p.numel()].view_as(p), v[off:off + p.numel()].view_as(p)
                off += p.numel()
            i = j

    def _write_hyper(self, key, group, step, dev):
        """The step's hyper-parameter words -> the device buffer the update kernel reads.  Stream-ordered copy through a ring of pinned
        slots (the host may run steps ahead of the stream); never part of a captured graph (see advance_host)."""
        beta1, beta2 = group["betas"]
        n_sma, size = radam_scalars(step, beta1, beta2)
        hp = self._hyper.get(key)
        if hp is None:
            if torch.cuda.is_current_stream_capturing():
                raise _lib.GlowTTSHipError("run one eager optimizer step before capturing a hipGraph (hyper-parameter buffer not allocated yet)")
            hp = self._hyper[key] = (_lib.PinnedRing(8), torch.empty(8, device=dev))
        words = torch.tensor([group["lr"], beta1, beta2, group["eps"], group["weight_decay"], size, 1.0 if n_sma >= 5 else 0.0, 0.0])
        hp[0].push(words, hp[1])
        return hp

    def advance_host(self):
        """Host half of a step, for replaying a captured hipGraph that contains `step()`: advance every step counter and send this step's
        hyper-parameter words (learning rate of the scheduler, step size, rectification flag) to the device buffer the captured update
        kernel reads - a stream-ordered copy issued BEFORE the replay on the replay's stream, not a node of the graph, so a host that
        runs ahead can neither skip nor repeat a step's values.  Call it on the stream the graph is replayed on."""
        for gi, group in enumerate(self.param_groups):
            by_step = {}
            for p in group["params"]:
                if p.grad is not None and len(self.state[p]):
                    self.state[p]["step"] += 1
                    by_step.setdefault(self.state[p]["step"], []).append(p)
            for step, ps in by_step.items():
                self._write_hyper((gi, len(ps), ps[0].data_ptr()), group, step, ps[0].device)

    @torch.no_grad()
    def step(self, closure=None, grad_scale=None):
        """One update.  grad_scale: optional device tensor [1] multiplied into every gradient on the fly (e.g. the clip coefficient of
        `grad_norm_and_coef`), instead of a separate in-place scaling pass."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        L = _L()
        for gi, group in enumerate(self.param_groups):
            plist = [p for p in group["params"] if p.grad is not None]
            if not plist:
                continue
            for p in plist:
                if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous() or not p.grad.is_contiguous():
                    raise _lib.GlowTTSHipError("glow_tts_amd.optim.RAdam updates contiguous fp32 device parameters (no CPU fallback)")
            self._init_state(plist)
            by_step = {}
            for p in plist:
                st = self.state[p]
                st["step"] += 1                                                      # Radam.py:62
                by_step.setdefault(st["step"], []).append(p)
            for step, ps in by_step.items():                                         # normally one entry: every parameter steps together
                key = (gi, len(ps), ps[0].data_ptr())
                tab = self._tables.setdefault(key, _JobTable())
                dev = ps[0].device
                tab.update([(p, p.grad, self.state[p]["exp_avg"], self.state[p]["exp_avg_sq"]) for p in ps], dev)
                # captured: the words arrive by advance_host() before every replay; eager: sent now
                hp = self._hyper[key] if (torch.cuda.is_current_stream_capturing() and key in self._hyper) else self._write_hyper(key, group, step, dev)
                _lib.check(L.glowtts_radam_step(tab.table.data_ptr(), tab.njobs, tab.blocks, hp[1].data_ptr(),
                                                grad_scale.data_ptr() if grad_scale is not None else None, _lib.stream()), "glowtts_radam_step")
        return loss


def grad_norm_and_coef(parameters, max_norm, _table_key="fused"):
    """(norm, coef) device tensors of the global gradient norm and the clip coefficient, WITHOUT touching the gradients: pass `coef` to
    `RAdam.step(grad_scale=coef)`."""
    params = [p for p in parameters if p.grad is not None]
    dev = params[0].grad.device
    from . import decoder
    decoder.stamp("clip_begin")
    tab = _CLIP_TABLES.setdefault((_table_key, id(params[0]), len(params)), _JobTable())
    tab.update([(p.grad, p.grad, None, None) for p in params], dev, canonical=True)
    partial, out = torch.empty(tab.blocks, device=dev), torch.empty(2, device=dev)
    _lib.check(_L().glowtts_multi_grad_norm(tab.table.data_ptr(), tab.njobs, tab.blocks, float(max_norm), partial.data_ptr(), out.data_ptr(),
                                            _lib.stream()), "glowtts_multi_grad_norm")
    return out[0], out[1:2]


class Modified_Noam_Scheduler(torch.optim.lr_scheduler._LRScheduler):
    """lr = base_lr * sqrt(base / (step + base)) - the reference's warm-up-free Noam variant (Noam_Scheduler.py:17-29); same constructor
    and the same attributes in `state_dict()` (`base`)."""

    def __init__(self, optimizer, base):
        self.base = base
        super().__init__(optimizer)

    def get_lr(self):
        e = max(1, self.last_epoch)
        return [b * self.base ** 0.5 * (e + self.base) ** (-0.5) for b in self.base_lrs]


class Noam_Scheduler(torch.optim.lr_scheduler._LRScheduler):
    """Noam_Scheduler.py:5-14."""

    def __init__(self, optimizer, warmup_steps):
        self.warmup_steps = warmup_steps
        super().__init__(optimizer)

    def get_lr(self):
        e = max(1, self.last_epoch)
        return [b * self.warmup_steps ** 0.5 * min(e ** (-0.5), e * self.warmup_steps ** -1.5) for b in self.base_lrs]
