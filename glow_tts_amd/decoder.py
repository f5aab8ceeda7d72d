"""Flow decoder (Modules.py:286-309 `Decoder`, :653-924) on the HIP path.

`decoder_forward` / `DecoderFunction` run Squeeze -> Stack x AIA -> Unsqueeze through the C ABI
(glowtts_squeeze_rows, glowtts_flow_forward/inverse/backward, glowtts_unsqueeze_rows,
glowtts_decoder_logdet).  This file only marshals pointers: weights arrive as *stacked* fp32 tensors
(one tensor per parameter kind, leading dim = flow), are packed into MFMA tile order once per step
(glowtts_pack_weight_batched) and every kept activation lives in buffers allocated here and handed to
the kernels.  There is no CPU fallback.
"""
import ctypes
import math

import torch

from . import _lib, ops
from ._lib import c_int, c_void_p

MAXL = 8
ROW_PAD = 2
c_i64 = ctypes.c_int64


class Packed(ctypes.Structure):
    _fields_ = [("w", c_void_p), ("npad", c_int), ("kchunks", c_int)]


class FlowDims(ctypes.Structure):
    _fields_ = [("B", c_int), ("T", c_int), ("C", c_int), ("H", c_int), ("L", c_int), ("ksize", c_int), ("precision", c_int),
                ("drop_p", ctypes.c_float), ("seed", ctypes.c_uint32), ("seed_ptr", c_void_p), ("act_bf16", c_int)]


class FlowParams(ctypes.Structure):
    _fields_ = [("an_logs", c_void_p), ("an_bias", c_void_p), ("winfo", c_void_p),
                ("start", Packed), ("in_", Packed * MAXL), ("rs", Packed * MAXL), ("end", Packed),
                ("start_t", Packed), ("in_t", Packed * MAXL), ("rs_t", Packed * MAXL), ("end_t", Packed),
                ("b_start", c_void_p), ("b_in", c_void_p * MAXL), ("b_rs", c_void_p * MAXL), ("b_end", c_void_p),
                ("cond", c_void_p), ("ldcond", c_i64), ("cond_rows", c_int), ("wn_img", c_void_p), ("wn_img_t", c_void_p)]


class FlowActs(ctypes.Structure):
    _fields_ = [("xin", c_void_p), ("xmid", c_void_p), ("xout", c_void_p),
                ("hs", c_void_p * MAXL), ("gates", c_void_p * MAXL), ("skip", c_void_p), ("outs", c_void_p),
                ("rowmask", c_void_p), ("acts", c_void_p * MAXL), ("skip_bf", c_void_p), ("xa_bf", c_void_p),
                ("next_an_logs", c_void_p), ("next_an_bias", c_void_p), ("next_winfo", c_void_p),
                ("next_xmid", c_void_p), ("next_xout", c_void_p), ("next_xa_bf", c_void_p), ("actnorm_done", c_int)]


class FlowGrads(ctypes.Structure):
    _fields_ = [("dx", c_void_p), ("dlogdet", c_void_p), ("douts", c_void_p), ("dskip", c_void_p),
                ("dh", c_void_p * MAXL), ("dins", c_void_p * MAXL), ("defer_wgrad", c_int),
                ("scratch", c_void_p), ("d_an", c_void_p),
                ("dw_start", c_void_p), ("db_start", c_void_p),
                ("dw_in", c_void_p * MAXL), ("db_in", c_void_p * MAXL),
                ("dw_rs", c_void_p * MAXL), ("db_rs", c_void_p * MAXL),
                ("dw_end", c_void_p), ("db_end", c_void_p), ("dcond", c_void_p), ("douts_bf", c_void_p),
                ("coupling_done", c_int), ("prev_xmid", c_void_p), ("prev_outs", c_void_p), ("prev_douts", c_void_p), ("prev_douts_bf", c_void_p),
                ("pitch_rows", c_void_p), ("pitch_ns", c_int), ("dh0_bf16", c_int)]


_declared = False


def _L():
    global _declared
    L = _lib.lib()
    if not _declared:
        L.glowtts_pack_weight_batched.argtypes = [c_void_p] + [c_int] * 8 + [c_void_p, ctypes.POINTER(c_int), ctypes.POINTER(c_int), c_void_p]
        L.glowtts_squeeze_rows.argtypes = [c_void_p] * 4 + [c_int] * 4 + [c_void_p]
        L.glowtts_unsqueeze_rows.argtypes = [c_void_p] * 3 + [c_int] * 5 + [ctypes.c_float, c_void_p]
        L.glowtts_inv1x1_prepare.argtypes = [c_void_p, c_void_p, c_int, c_void_p]
        L.glowtts_actnorm_stats.argtypes = [c_void_p] * 4 + [c_i64, c_int, c_void_p]
        L.glowtts_actnorm_stats_scratch_floats.argtypes = [c_i64, c_int]
        L.glowtts_actnorm_stats_scratch_floats.restype = c_i64
        L.glowtts_actnorm_bwd_blocks.argtypes = [c_i64]
        L.glowtts_actnorm_bwd_blocks.restype = c_i64
        L.glowtts_actnorm_from_stats.argtypes = [c_void_p] * 3 + [c_int, c_void_p]
        L.glowtts_actnorm_inv1x1.argtypes = [c_void_p] * 6 + [c_i64, c_int, c_int, c_void_p]
        L.glowtts_flow_forward.argtypes = [c_void_p] * 4
        L.glowtts_flow_inverse.argtypes = [c_void_p] * 4
        L.glowtts_flow_backward.argtypes = [c_void_p] * 5
        L.glowtts_wgrad_grouped.argtypes = [c_void_p] + [c_int] * 9 + [c_void_p]
        L.glowtts_wgrad_grouped_io.argtypes = [c_void_p] + [c_int] * 10 + [c_void_p]
        L.glowtts_wgrad_grouped_phased.argtypes = [c_void_p] + [c_int] * 5 + [c_void_p]
        L.glowtts_colsum_batched.argtypes = [c_void_p, c_void_p, c_int, c_int, c_int, c_i64, c_i64, c_void_p]
        L.glowtts_weightnorm_fwd.argtypes = [c_void_p] * 4 + [c_i64, c_int, c_void_p]
        L.glowtts_weightnorm_bwd.argtypes = [c_void_p] * 6 + [c_i64, c_int, c_void_p]
        L.glowtts_decoder_logdet.argtypes = [c_void_p, c_i64] + [c_void_p] * 5 + [c_int] * 5 + [c_void_p]
        L.glowtts_decoder_param_grads.argtypes = [c_void_p] * 7 + [c_int] * 4 + [c_void_p]
        L.glowtts_wavenet_image_bytes.argtypes = [c_int, c_int, ctypes.POINTER(c_i64)]
        L.glowtts_wavenet_pack_images.argtypes = [c_void_p] * 5 + [c_int] * 3 + [c_void_p] * 3
        L.glowtts_prep_job_init.argtypes = [c_void_p] * 4 + [c_int] * 8 + [c_void_p] + [c_i64] * 4 + [c_int, ctypes.POINTER(c_int)]
        L.glowtts_prep_launch.argtypes = [c_void_p, c_int, c_int, c_int, c_void_p]
        L.glowtts_wavenet_prep_jobs.argtypes = [c_void_p, c_int, ctypes.POINTER(c_int), ctypes.POINTER(c_int)] + [c_void_p] * 13 + [c_int] * 3 + [c_void_p, c_int, c_void_p]
        L.glowtts_cond_linear_fwd.argtypes = [c_void_p] * 6 + [c_int] * 4 + [c_void_p]
        L.glowtts_sum_slices_seg.argtypes = [c_void_p, c_int, c_i64, c_void_p, c_int, c_void_p]
        L.glowtts_cond_linear_bwd.argtypes = [c_void_p, c_i64] + [c_void_p] * 9 + [c_int] * 3 + [c_void_p]
        L.glowtts_cond_linear_bwd_scratch_floats.argtypes = [c_int] * 3
        L.glowtts_cond_linear_bwd_scratch_floats.restype = c_i64
        _declared = True
    return L


class WnBwdJob(ctypes.Structure):
    """Mirror of `glowtts_wn_bwd_job`."""
    _fields_ = [("dw", c_void_p), ("v", c_void_p), ("g", c_void_p), ("inv_norm", c_void_p), ("dv", c_void_p), ("dg", c_void_p),
                ("rows", c_i64), ("cols", c_int), ("reserved", c_int)]


class SumSeg(ctypes.Structure):
    """Mirror of `glowtts_sum_seg`."""
    _fields_ = [("dst", c_void_p), ("off", c_i64), ("n", c_i64)]


class WgradJob(ctypes.Structure):
    """Mirror of `glowtts_wgrad_job` (one weight-gradient problem of the grouped launch)."""
    _fields_ = [("dy", c_void_p), ("x", c_void_p), ("xmask", c_void_p), ("dw", c_void_p), ("dbias", c_void_p),
                ("lddy", c_i64), ("ldx", c_i64),
                ("m", c_int), ("ca", c_int), ("xpro", c_int), ("perm", c_int), ("perm_h", c_int),
                ("tile0", c_int), ("mt", c_int), ("nt", c_int), ("rows", c_int), ("reserved", c_int)]


class PrepJob(ctypes.Structure):
    """Mirror of `glowtts_prep_job` (weight norm + packing of one conv class into one image, csrc/prep_ops.hip)."""
    _fields_ = [("v", c_void_p), ("g", c_void_p), ("inv_out", c_void_p), ("packed", c_void_p),
                ("outer_stride", c_i64), ("inner_stride", c_i64), ("w_stride", c_i64), ("g_stride", c_i64),
                ("batch", c_int), ("inner", c_int), ("O", c_int), ("I", c_int), ("taps", c_int), ("transpose", c_int), ("perm", c_int), ("perm_h", c_int),
                ("o_ext", c_int), ("npad", c_int), ("kchunks", c_int), ("tiles", c_int), ("block0", c_int), ("reserved", c_int),
                ("twin", c_void_p), ("twin_outer", c_i64), ("twin_inner", c_i64), ("twin_half", c_i64), ("twin_batch", c_int), ("twin_npad", c_int)]


class PrepJobs:
    """Collects the weight-preparation jobs of a training step (every image of the decoder's convs, straight from the weight-norm pairs) and
    issues them as ONE launch (glowtts_prep_launch)."""
    CAP = 22                             # GLOWTTS_PREP_MAX_JOBS

    def __init__(self):
        self.jobs = (PrepJob * self.CAP)()
        self.n, self.blocks, self.max_cols = c_int(0), c_int(0), 1
        self.keep = []                       # tensors the jobs point into

    def add(self, v, g, inv, batch, inner, O, I, taps, transpose, perm, perm_h, packed_ptr, outer_stride, inner_stride=0, w_stride=0, g_stride=0):
        if self.n.value >= self.CAP:
            raise _lib.GlowTTSHipError("too many weight-preparation jobs")
        nb = c_int(0)
        _lib.check(_L().glowtts_prep_job_init(ctypes.byref(self.jobs[self.n.value]), v, g, inv, batch, inner, O, I, taps, int(transpose), perm, perm_h,
                                              packed_ptr, outer_stride, inner_stride, w_stride, g_stride, self.blocks.value, ctypes.byref(nb)), "prep_job_init")
        self.n.value += 1
        self.blocks.value += nb.value
        self.max_cols = max(self.max_cols, I * taps)

    def launch(self, device):
        if self.n.value == 0:
            return
        if TUNE["prep_twin"] and not getattr(self, "_twinned", False):
            # the In_l pairs are read ONCE: the forward image's tiles also write the two transposed half images (csrc/prep_ops.hip; no-op without that pattern)
            _L().glowtts_prep_jobs_twin_in.argtypes = [c_void_p, ctypes.POINTER(c_int), ctypes.POINTER(c_int)]
            _lib.check(_L().glowtts_prep_jobs_twin_in(ctypes.cast(self.jobs, c_void_p), ctypes.byref(self.n), ctypes.byref(self.blocks)), "glowtts_prep_jobs_twin_in")
            self._twinned = True
        # (the table is a HOST array: it travels in the launch's argument segment - no copy node in a captured step)
        _lib.check(_L().glowtts_prep_launch(ctypes.cast(self.jobs, c_void_p), self.n.value, self.blocks.value, self.max_cols, _lib.stream()), "glowtts_prep_launch")


_WSTREAM = {}


def _engine_callbacks_ok():
    """queue_callback is only legal while the autograd engine is running a backward pass (it is, whenever DecoderFunction.backward runs under
    loss.backward() / torch.autograd.grad)."""
    return hasattr(torch.autograd.Variable._execution_engine, "queue_callback")


# Data parallel overlap (bench.py, N > 1): the big k-tap weight gradients are launched first and their all-reduce starts while the
# "tail" - the 1x1 weight-gradient groups and the weight-norm backward of their classes - is still running.  With TAIL["defer"] set the
# backward queues the tail's launches here instead of issuing them; `flush_tail_wgrads()` issues them (bench.py captures that as a
# second hipGraph).  The gradient tensors of the tail classes exist (autograd has already handed them to .grad) but hold no data until then.
TAIL = {"defer": False, "pending": []}
# Measured design choices that tools / tests may flip programmatically (never read from the environment: DESIGN.md section 5 has the numbers).
#   wgrad_wide: 16-byte staging items in the weight-gradient kernel; fuse_coupling_bwd: the next flow's coupling backward rides in the
#   ActNorm / 1x1 backward pass; wgrad_split: weight gradients in n segments on a second stream (1 = one grouped launch per class);
#   act_bf16: WaveNet state / gates / gate gradients stored as bf16 in bf16 precision
#   fused_wn_bwd: its data gradients likewise (csrc/wavenet_fused_bwd.hip; every mode but GR's per-frame pitch), for flows 0 .. n-1 - the flows the backward
#       reaches LAST.  n: a count, True = all flows, -1 (default) = automatic: all flows when the batch leaves a quarter of the CUs free, else 7 of 12 (5 until the encoder's attention backward got shorter).  Alone the fused kernel is faster than the ten launches it
#       replaces (126 vs 155 us per flow, 151 vs 178 with cold caches: tools/bench_wn.py), but a workgroup that owns a whole CU (150 KB of LDS,
#       3 x 168 VGPRs per SIMD) for 126 us leaves the encoder stream's backward no CU to share, and while that stream is busy the step LOSES:
#       all 12 flows 6.33 vs 5.90 ms/step, 9 flows 5.95.  The encoder's backward has drained by the time the decoder's backward is half way:
#       the last 6 flows fused 5.70 vs 5.85 and 5.82 vs 5.92 ms/step on two boxes
#   fused_wn: the coupling network of a flow (Start .. End + coupling) as ONE launch (csrc/wavenet_fused.hip) where its shape contract holds
#   fused_wn_fwd_skip: training forward, the first n flows on the per-conv launches (they run beside the text encoder's forward, which the
#       CU-filling fused workgroups starve).  -1 (default) = automatic: 1 for a chip-filling batch, else 0.  The encoder's forward is what the
#       first half of the step waits for (enc_fwd_project 87 us after dec_fwd_end with every flow fused); one per-conv flow gives it the
#       CUs it needs: 5.55 / 5.56 vs 5.61 / 5.59 ms/step (2 flows: 5.57 / 5.56), same box, alternating runs.  (Round 3's first measurement,
#       before the alignment test compared path QUALITY instead of a count of moved frames, had left it off.)
#   bwd_packs_side: a scheduling experiment of round 3, off: the backward-only weight images
#       packed on a third stream joined when the backward starts (5.61 / 5.67 vs 5.68 / 5.66 ms/step: inside the spread)
#   enc_pack_split: round-4 experiment, off.  The encoder's weight images in two launches - the prenet's forward images on its stream, the other 95 % on a
#       third stream forked from the origin stream and joined in front of the transformer (forked from the encoder's stream - a fork of a fork - hipStreamEndCapture
#       crashed on ROCm 7.2): the encoder's projection then lands 7 us after the decoder's forward instead of 60-90, but the three packing kernels share the
#       memory system at the head of the step and the decoder's forward ends 45 us later: 5.24-5.30 against 5.20-5.23 ms/step (three alternating pairs)
#   prep_fused: training on the fused bf16 path forms w = g v / ||v|| and every weight image in ONE launch (csrc/prep_ops.hip) instead of 4 weight-norm +
#       ~22 packing launches (0.42 -> ~0.1 ms at the head of the step, round 4)
TUNE = {"chain_actnorm": True, "enc_ln_qkv": True, "enc_proj_ln": True, "enc_pack_split": False, "wgrad_dma": True, "wgrad_dma_k1": False, "prep_fused": True, "prep_early": True, "wgrad_wide": True, "fuse_coupling_bwd": True, "wgrad_split": 1, "act_bf16": True, "fused_wn": True, "fused_wn_bwd": -1, "fused_wn_bwd_from": 0, "fused_wn_fwd_skip": -1, "bwd_packs_side": 0, "fwd_packs_split": 0, "cond_hip": True, "drop_skip32": True, "wgrad_tail_splits": 2, "prep_bwd_late": False, "prep_bwd_gentle": True, "tail_aside": True, "wgrad_balance": False, "prep_twin": True}
STAMPS = {"buf": None, "names": []}      # tools/step_timeline.py: an int64 device buffer; stamp(name) appends a slot


_SEED_STATE = {}


def step_seed(device):
    """-> int32 tensor [1]: the dropout seed word of this training step.  Eager: one draw from torch's generator, as ever (torch.manual_seed reproduces the masks; every
    forward owns its word, so several forwards may precede a backward).  Under hipGraph capture: glowtts_step_seed - a counter in device memory advanced by a launch of
    the graph itself, so that every replay draws a new word without torch's philox machinery (an RNG launch on each stream's chain and two fills of the generator's
    offset words in front of every replay); the base is a hash of torch.initial_seed() at the first eager call (per-rank seeds give per-rank masks)."""
    key = str(device)
    st = _SEED_STATE.get(key)
    capturing = torch.cuda.is_current_stream_capturing()
    if st is None:
        if capturing:
            raise _lib.GlowTTSHipError("run one eager training step before capturing a hipGraph (the dropout seed state is created eagerly)")
        st = _SEED_STATE[key] = torch.zeros(2, dtype=torch.int32, device=device)
        st[1] = int((torch.initial_seed() * 0x9E3779B97F4A7C15 >> 33) & 0x7FFFFFFF)      # (torch.manual_seed decides it; no draw is consumed)
    if not capturing:
        return torch.randint(0, 2 ** 31 - 1, (1,), device=device, dtype=torch.int32)
    out = torch.empty(1, dtype=torch.int32, device=device)
    L = _L()
    L.glowtts_step_seed.argtypes = [c_void_p, c_void_p, c_void_p]
    _lib.check(L.glowtts_step_seed(st.data_ptr(), out.data_ptr(), _lib.stream()), "glowtts_step_seed")
    return out


def stamp(name):
    """Diagnostics: record when the current stream reaches this point (one tiny launch; only while tools/step_timeline.py armed it)."""
    if STAMPS["buf"] is None:
        return
    i = len(STAMPS["names"])
    if i >= STAMPS["buf"].numel():                       # (an armed eager run of many steps: the buffer is full, stop stamping)
        return
    STAMPS["names"].append(name)
    _L().glowtts_debug_stamp.argtypes = [c_void_p, c_void_p]
    _lib.check(_L().glowtts_debug_stamp(STAMPS["buf"].data_ptr() + 8 * i, _lib.stream()), "glowtts_debug_stamp")

WN_SLAB = 24576                          # GLOWTTS_WN_SLAB_BYTES
TAIL_STACKS = ("start_g", "start_v", "start_b", "rs_g", "rs_v", "rs_b", "rsl_g", "rsl_v", "rsl_b", "end_w", "end_b")


class defer_tail_wgrads:
    def __enter__(self):
        TAIL["defer"], TAIL["pending"] = True, []
        return self

    def __exit__(self, *exc):
        TAIL["defer"] = False
        return False


def flush_tail_wgrads():
    """Issues, on the current stream, the launches queued by the last backward under `defer_tail_wgrads()` (kept for re-capture)."""
    for fn in TAIL["pending"]:
        fn()


def _pack_stream(device):
    key = "pack:" + str(device)
    if key not in _WSTREAM:
        _WSTREAM[key] = torch.cuda.Stream(device=device)
    return _WSTREAM[key]


def _wgrad_stream(device):
    key = str(device)
    if key not in _WSTREAM:
        _WSTREAM[key] = torch.cuda.Stream(device=device)
    return _WSTREAM[key]


class WgradGroup:
    """Weight-gradient problems sharing (rows, taps, X prologue).  Problems are added in *segments* (one per flow); the whole
    job table is uploaded once and every segment is one glowtts_wgrad_grouped launch (tile indices restart per segment)."""

    def __init__(self, rows, taps, precision, xpro=ops.APRO_NONE, io_flags=0, tag="dec", dma_k1=None):
        self.rows, self.taps, self.precision, self.xpro, self.io_flags, self.tag = rows, taps, precision, xpro, io_flags, tag
        self.jobs, self.segments, self._tiles, self._start = [], [], 0, 0
        self.table = None
        # 16-byte staging items (glowtts_wgrad WIO_WIDE): both operands bf16, no prologue, and every job 8-channel / 16-byte aligned
        self._wide = io_flags in (0, ops.WIO_DY_BF16 | ops.WIO_X_BF16) and xpro == ops.APRO_NONE and precision == ops.BF16 and TUNE["wgrad_wide"]
        # the LDS-DMA / 16x16x32 kernel (csrc/wgrad_cl.hip wgrad_dma_kernel): both operands bf16 rows
        self._dma = self._wide and io_flags == (ops.WIO_DY_BF16 | ops.WIO_X_BF16) and bool(TUNE["wgrad_dma"]) and (taps > 1 or bool(TUNE["wgrad_dma_k1"] if dma_k1 is None else dma_k1))

    def add(self, dy, lddy, m, x, ldx, ca, dw, dbias, perm=ops.PERM_NONE, perm_h=0):
        j = WgradJob()
        j.dy, j.x, j.dw, j.dbias, j.lddy, j.ldx = dy, x, dw, dbias, lddy, ldx
        j.m, j.ca, j.xpro, j.perm, j.perm_h = m, ca, self.xpro, perm, perm_h
        if (m | ca | lddy | ldx) & 7 or (dy | x) & 15:
            self._wide = False
        if not self._wide or (self.rows + 512) * max(lddy, ldx) * 2 >= 2 ** 31:    # (the DMAs run three steps past the end; 32-bit buffer offsets)
            self._dma = False
        self.jobs.append(j)

    def end_segment(self):
        self.segments.append((self._start, len(self.jobs) - self._start, 0))
        self._start = len(self.jobs)

    def _tile(self):
        """Tile geometry of every job, once the group's kernel is known (the DMA kernel works on 192 x 64 tiles - 192 x 192 at one tap -, the staged one on 128 x 64)."""
        dma = self._dma and self._wide
        bm, bn = (192, 192 if self.taps == 1 else 64) if dma else (128, 64)
        for i, (start, n, _) in enumerate(self.segments):
            tiles = 0
            for j in self.jobs[start:start + n]:
                j.mt, j.nt, j.tile0 = (j.m + bm - 1) // bm, (j.ca + bn - 1) // bn, tiles
                tiles += j.mt * j.nt
            self.segments[i] = (start, n, tiles)

    def upload(self, device):
        if not self.jobs:
            return
        self._tile()
        # (a captured step gets a pinned table of its own: _lib.staged_upload)
        self.table = _lib.staged_upload(bytes((WgradJob * len(self.jobs))(*self.jobs)), device)

    @staticmethod
    def upload_all(groups, device):
        """One host-to-device copy for the job tables of several groups (one memcpy node in a captured step instead of one per group)."""
        groups = [g for g in groups if g.jobs]
        if not groups:
            return
        for g in groups:
            g._tile()
        raws = [bytes((WgradJob * len(g.jobs))(*g.jobs)) for g in groups]
        table = _lib.staged_upload(b"".join(raws), device)
        off = 0
        for g, r in zip(groups, raws):
            g.table = table[off:off + len(r)]
            off += len(r)

    def balance(self, B, rows_per_utt, device):
        """Two-phase form of a one-segment LDS-DMA group whose tile count exceeds the CU count by a fraction (the decoder's In_l group at B = 32: 288 one-per-CU
        tiles on 256 CUs = two rounds, the second with 32 tiles): the first jobs stay whole (at most one round), the rest are cut into S row splits over B / S
        utterances each - short tiles that fill the idle CUs beside the whole ones and ONE short round behind them (288 tiles: 1 + 1/8 rounds instead of 2).
        The splits write partial images (`self.part` [S, n]), summed in a fixed order by `sum_parts` (one glowtts_sum_slices_seg launch).  A cut between
        utterances is exact for any tap count: the pad rows around every utterance are zero.  -> True when the group was rebuilt."""
        if not (self._dma and self._wide) or len(self.segments) != 0 or self._start != 0 or not self.jobs or self.rows != B * rows_per_utt:
            return False
        cus = torch.cuda.get_device_properties(device).multi_processor_count
        bm, bn = 192, (192 if self.taps == 1 else 64)
        tiles = [((j.m + bm - 1) // bm) * ((j.ca + bn - 1) // bn) for j in self.jobs]
        if len(set(tiles)) != 1 or any(j.dbias is None for j in self.jobs):
            return False
        tpj, nj = tiles[0], len(self.jobs)
        unsplit = float(-(-nj * tpj // cus))
        best = None
        for S in (2, 4, 8, 16):
            if B % S:
                continue
            for a in range(nj, -1, -1):                        # (most whole jobs first: ties go to the fewest partial images)
                W = a * tpj
                if W > cus:
                    continue
                P = (nj - a) * S * tpj
                t, done = (1.0, min(P, (cus - W) * S)) if W else (0.0, 0)
                t += -(-(P - done) // cus) / S
                if best is None or t < best[0] - 1e-9:
                    best = (t, a, S)
        if best is None or best[0] > unsplit - 0.25 or best[1] == nj:
            return False
        _, a, S = best
        whole, cut = self.jobs[:a], self.jobs[a:]
        # destinations of the partial images: (dw, dbias) of every cut job, contiguous neighbours merged (glowtts_sum_slices_seg takes 12 segments)
        segs, tot = [], 0
        place = {}
        for j in cut:
            for ptr, n in ((j.dw, j.m * j.ca * self.taps), (j.dbias, j.m)):
                if n % 4 or ptr % 16:
                    return False
                place[ptr] = tot
                if segs and segs[-1][0] + 4 * segs[-1][1] == ptr:
                    segs[-1] = (segs[-1][0], segs[-1][1] + n, segs[-1][2])
                else:
                    segs.append((ptr, n, tot))
                tot += n
        if len(segs) > 12:
            return False
        self.part = torch.empty(S, tot, device=device)
        Rs = (B // S) * rows_per_utt
        jobs = list(whole)
        for j in cut:
            for s_ in range(S):
                c = WgradJob()
                ctypes.memmove(ctypes.byref(c), ctypes.byref(j), ctypes.sizeof(WgradJob))
                c.dy, c.x = j.dy + 2 * s_ * Rs * j.lddy, j.x + 2 * s_ * Rs * j.ldx          # (bf16 operands)
                c.dw = self.part.data_ptr() + 4 * (s_ * tot + place[j.dw])
                c.dbias = self.part.data_ptr() + 4 * (s_ * tot + place[j.dbias])
                c.rows = Rs
                jobs.append(c)
        self.jobs, self.part_segs, self.part_S, self.whole_tiles = jobs, segs, S, a * tpj
        return True

    def sum_parts(self):
        if getattr(self, "part", None) is None:
            return
        segs = (SumSeg * len(self.part_segs))(*[SumSeg(p0, off, n) for p0, n, off in self.part_segs])
        _lib.check(_L().glowtts_sum_slices_seg(self.part.data_ptr(), self.part_S, self.part.shape[1], segs, len(self.part_segs), _lib.stream()), "glowtts_sum_slices_seg")

    def launch_segment(self, i):
        start, n, tiles = self.segments[i]
        if n == 0:
            return
        if getattr(self, "part", None) is not None:
            _lib.check(_L().glowtts_wgrad_grouped_phased(self.table.data_ptr() + start * ctypes.sizeof(WgradJob), n, tiles, self.whole_tiles, self.rows, self.taps,
                                                         _lib.stream()), "glowtts_wgrad_grouped_phased")
            self.sum_parts()
            return
        _lib.check(_L().glowtts_wgrad_grouped_io(self.table.data_ptr() + start * ctypes.sizeof(WgradJob), n, tiles, self.rows, self.taps,
                                                 (self.taps - 1) // 2, self.xpro, self.precision, 1, 0,
                                                 self.io_flags | (ops.WIO_WIDE if self._wide else 0) | (ops.WIO_DMA if (self._dma and self._wide) else 0), _lib.stream()),
                   "glowtts_wgrad_grouped_io")


class PackedBatch:
    """`batch` same-shape conv weights packed by one launch; element i is the i-th weight."""

    def __init__(self, w, transpose, perm, perm_h, precision, g=None, jobs=None):
        """jobs (a PrepJobs) given: `w` is weight_v (g: weight_g [batch, O, 1, 1], or None for a plain weight) and the image is produced by the
        step's one weight-preparation launch instead of a launch of its own."""
        L = _L()
        w = w.contiguous()
        self.batch, O, I, taps = w.shape
        npad, kch = c_int(0), c_int(0)
        args = (self.batch, O, I, taps, int(transpose), perm, perm_h, precision)
        _lib.check(L.glowtts_pack_weight_batched(None, *args, None, ctypes.byref(npad), ctypes.byref(kch), None), "pack(size)")
        self.npad, self.kchunks = npad.value, kch.value
        self.stride = taps * self.kchunks * self.npad * 64
        self.data = torch.empty(self.batch * self.stride, dtype=torch.uint8, device=w.device)
        if jobs is not None:
            assert precision == ops.BF16
            jobs.add(w.data_ptr(), g.data_ptr() if g is not None else None, None, self.batch, 1, O, I, taps, transpose, perm, perm_h,
                     self.data.data_ptr(), self.stride)
            jobs.keep += [w, g, self.data]
            return
        _lib.check(L.glowtts_pack_weight_batched(_lib.ptr(w), *args, _lib.ptr(self.data), None, None, _lib.stream()), "pack")

    def at(self, i):
        return Packed(self.data.data_ptr() + i * self.stride, self.npad, self.kchunks)


class ImageSlices:
    """The convs of one weight class inside the per-flow weight images of the fused coupling-network kernel (same interface as PackedBatch):
    element i = (flow i // inner, layer i % inner) starts at flow * flow_stride + offset + layer * layer_stride."""

    def __init__(self, img, flow_stride, offset, inner, layer_stride, npad, kchunks):
        self.img, self.flow_stride, self.offset, self.inner, self.layer_stride, self.npad, self.kchunks = img, flow_stride, offset, inner, layer_stride, npad, kchunks

    def at(self, i):
        return Packed(self.img.data_ptr() + (i // self.inner) * self.flow_stride + self.offset + (i % self.inner) * self.layer_stride, self.npad, self.kchunks)


def fused_wn_supported(cfg):
    """Shape contract of glowtts_wavenet_fwd (include/glowtts_hip.h): the reference's default decoder."""
    return bool(TUNE["fused_wn"] and cfg.act_bf16 and cfg.H == 192 and cfg.k == 5 and 1 <= cfg.L <= 4 and 64 < cfg.C // 2 <= 96 and cfg.C % 8 == 0)


class DecoderConfig:
    """Shapes of the decoder, from Hyper_Parameters.yaml (Decoder.*, Sound.Mel_Dim)."""

    def __init__(self, mel_dim, n_flows, n_squeeze, n_split, hidden, n_layers, ksize, precision=ops.BF16):
        assert n_split == 4, "Invertible_1x1_Conv kernels are written for Decoder.Num_Split == 4 (the reference default)"
        self.Cm, self.F, self.ns, self.H, self.L, self.k = mel_dim, n_flows, n_squeeze, hidden, n_layers, ksize
        self.C = mel_dim * n_squeeze
        self.precision = precision
        assert n_layers <= MAXL and self.C % 4 == 0 and hidden % 4 == 0
        # bf16 precision: the GEMM-only activations (WaveNet state, gates, gate gradients) live in HBM as bf16
        # (glowtts_flow_dims.act_bf16); TUNE["act_bf16"] = False keeps them fp32 (same MFMA inputs, twice the traffic)
        self.act_bf16 = precision == ops.BF16 and hidden % 8 == 0 and TUNE["act_bf16"]
        self.act_dtype = torch.bfloat16 if self.act_bf16 else torch.float32


WEIGHT_KEYS = ("an_logs", "an_bias", "inv_w", "w_start", "b_start", "w_in", "b_in", "w_rs", "b_rs",
               "w_rs_last", "b_rs_last", "w_end", "b_end")
# DecoderFunction's other calling form (training on the fused bf16 path): the four weight-normalised convs arrive as their (weight_g, weight_v)
# stacks and the function forms w = g v / ||v|| inside its one weight-preparation launch (and applies the weight-norm backward itself)
WN_KEYS = ("w_start", "w_in", "w_rs", "w_rs_last")
WEIGHT_KEYS_GV = ("an_logs", "an_bias", "inv_w", "g_start", "v_start", "b_start", "g_in", "v_in", "b_in", "g_rs", "v_rs", "b_rs",
                  "g_rs_last", "v_rs_last", "b_rs_last", "w_end", "b_end")


# classes whose gradients come out of the deferrable tail of the backward (TAIL_STACKS, by weight key): everything but ActNorm / the 1x1 mixing / In_l
_TAIL_OF = {k: not (k.startswith(("an_", "inv_")) or k.endswith("_in")) for k in set(WEIGHT_KEYS + WEIGHT_KEYS_GV)}


class _Prepared:
    """Packed weight images + per-flow parameter structs for one set of stacked weights."""

    def __init__(self, cfg, W, need_bwd, cond=None, fused_bwd_ok=True, rows=None, GV=None, conditioned=None):
        """conditioned: whether a conditioning vector will be set (early_prepare runs before it exists); None = `cond is not None`.  fused_bwd_ok = False: the backward needs what only the per-conv kernels produce (GR mode: the per-row pitch conditioning and the
        Pitch_l weight gradient).  rows: B * (T + 4) of the batch this is prepared for (sizes the automatic choice of TUNE["fused_wn_bwd"]).
        GV (training, bf16, fused coupling network): {"w_start" | "w_in" | "w_rs" | "w_rs_last": (weight_g, weight_v)} stacked like W's entries, which
        are then absent from W - every image is produced from the weight-norm pairs by ONE launch (csrc/prep_ops.hip; `self.inv` keeps 1 / ||v||
        per class for the weight-norm backward) instead of 4 weight-norm + ~22 packing launches at the head of every step."""
        L = _L()
        F_, H, C, Lw = cfg.F, cfg.H, cfg.C, cfg.L
        P = cfg.precision
        dev = W["an_logs"].device
        self.keep = W
        stamp("dec_prepared_begin")
        self.winfo = torch.empty(F_, 36, device=dev)
        _lib.check(L.glowtts_inv1x1_prepare(_lib.ptr(W["inv_w"].contiguous()), _lib.ptr(self.winfo), F_, _lib.stream()), "inv1x1_prepare")
        jobs = PrepJobs() if GV is not None else None
        # (round 5) the images only the BACKWARD reads - the fused data-gradient image, the per-conv transposed images - are a second launch that the caller
        # may issue later, off the decoder's chain (`launch_bwd_images`: modules.GlowTTS.forward queues it on the encoder's stream behind the encoder's forward,
        # where it runs under the log-prior / MAS section of the step; DecoderFunction.backward issues it itself if nobody has).  Round 6: OFF by default
        # (TUNE["prep_bwd_late"]) - one launch for every image at the head of the step again: what the deferred launch saves at the head (~40 us) is less than the
        # two cross-stream edges it adds to the replayed graph cost (4.80 / 4.83 against 4.84 / 4.87 ms/step, two alternating rounds; an extra event edge
        # between the two branches was measured at + 0.5 ms in the same session)
        jobs_b = PrepJobs() if (jobs is not None and need_bwd and TUNE["prep_bwd_late"]) else None
        cur = [jobs]
        self.inv = None
        if GV is not None:
            assert fused_wn_supported(cfg) and P == ops.BF16, "weight preparation from (g, v) serves the fused bf16 path"
            self.inv = {"w_start": torch.empty(F_, H, device=dev), "w_in": torch.empty(F_, Lw, 2 * H, device=dev),
                        "w_rs": torch.empty(F_, max(Lw - 1, 1), 2 * H, device=dev), "w_rs_last": torch.empty(F_, H, device=dev)}

        def src(key, sl=slice(None)):
            """(weight or weight_v, weight_g or None) of class `key`, flows `sl`"""
            if GV is not None and key in GV:
                return GV[key][1][sl], GV[key][0][sl]
            return W[key][sl], None

        def batch(key, sl, shape, transpose, perm, perm_h):
            w, g = src(key, sl)
            return PackedBatch(w.reshape(shape), transpose, perm, perm_h, P, g=g, jobs=cur[0])

        def images(sl, nflows, img_fwd, nbwd, img_bwd, with_inv):
            """the fused kernels' weight images of flows `sl` (forward: nflows of them; transposed: the first nbwd)"""
            if GV is None:
                _lib.check(L.glowtts_wavenet_pack_images(_lib.ptr(W["w_start"][sl]), _lib.ptr(W["w_in"][sl]), _lib.ptr(W["w_rs"][sl]) if Lw > 1 else None,
                                                         _lib.ptr(W["w_rs_last"][sl]), _lib.ptr(W["w_end"][sl]), nflows if img_fwd is not None else nbwd, Lw, C // 2,
                                                         _lib.ptr(img_fwd), _lib.ptr(img_bwd), _lib.stream()), "wavenet_pack_images")
                return
            a = []
            for key in ("w_start", "w_in", "w_rs", "w_rs_last"):
                g, v = GV[key]
                inv = self.inv[key] if with_inv else None
                if key == "w_rs" and Lw == 1:
                    a += [None, None, None]
                else:
                    a += [v[sl].data_ptr(), g[sl].data_ptr(), inv[sl].data_ptr() if inv is not None else None]
            jb = cur[0]
            _lib.check(L.glowtts_wavenet_prep_jobs(ctypes.cast(jb.jobs, c_void_p), PrepJobs.CAP, ctypes.byref(jb.n), ctypes.byref(jb.blocks), *a,
                                                   W["w_end"][sl].data_ptr(), nflows, Lw, C // 2, _lib.ptr(img_fwd), nbwd, _lib.ptr(img_bwd)), "wavenet_prep_jobs")
            jb.max_cols = max(jb.max_cols, H * cfg.k)

        self.wn_img = None
        ksplit = 0
        if fused_wn_supported(cfg):
            # one image per flow holding every forward weight of its coupling network as 24-KiB slabs (glowtts_wavenet_pack_images)
            nb = c_i64(0)
            _lib.check(L.glowtts_wavenet_image_bytes(Lw, 0, ctypes.byref(nb)), "wavenet_image_bytes")
            nb = nb.value
            self.wn_img = torch.empty(F_, nb, dtype=torch.uint8, device=dev)
            # (experiment, TUNE["fwd_packs_split"] = k: only the first k flows' images here, the others on the side stream below - the
            # decoder's first flow then waits for k flows' worth of packing instead of twelve)
            ksplit = int(TUNE["fwd_packs_split"]) if (need_bwd and dev.type == "cuda" and GV is None) else 0
            ksplit = ksplit if 0 < ksplit < F_ else 0
            images(slice(None), ksplit or F_, self.wn_img, 0, None, True)
            S = WN_SLAB
            self.pk = {
                "start": ImageSlices(self.wn_img, nb, 0, 1, 0, 192, 3),
                "in": ImageSlices(self.wn_img, nb, 2 * S, Lw, 36 * S, 2 * H, H // 32),
                "rs_last": ImageSlices(self.wn_img, nb, (36 * (Lw - 1) + 32) * S, 1, 0, H, H // 32),
                "end": ImageSlices(self.wn_img, nb, (36 * (Lw - 1) + 35) * S, 1, 0, 192, H // 32),
            }
            if Lw > 1:
                self.pk["rs"] = ImageSlices(self.wn_img, nb, 32 * S, Lw - 1, 36 * S, 2 * H, H // 32)       # (PAIR-packed: read by the fused kernel only)
        # training: the first flows of the forward can stay on the per-conv launches (TUNE["fused_wn_fwd_skip"]) - they run beside the
        # text encoder's forward on the other stream, which a CU-filling fused workgroup starves
        nskip = int(TUNE["fused_wn_fwd_skip"])
        if nskip < 0:                                           # automatic: one flow when the batch's fused workgroups fill the chip (see fused_wn_bwd)
            nwg_ = -(-rows // (64 - 4 * (Lw - 1))) if rows else 0
            cus_ = torch.cuda.get_device_properties(dev).multi_processor_count if dev.type == "cuda" else 256
            nskip = 3 if 4 * nwg_ > 3 * cus_ else 0             # (round 4, with the early weight preparation: 1 / 2 / 3 / 4 flows = 5.36 / 5.31 / 5.28 / 5.29 ms/step on one box)
        nskip = min(nskip, F_) if (need_bwd and self.wn_img is not None) else 0
        pk_conv = None
        all_f = slice(None)
        if self.wn_img is None:
            self.pk = {
                "start": batch("w_start", all_f, (F_, H, C // 2, 1), False, ops.PERM_NONE, 0),
                "in": batch("w_in", all_f, (F_ * Lw, 2 * H, H, cfg.k), False, ops.PERM_PAIR, H),
                "rs_last": batch("w_rs_last", all_f, (F_, H, H, 1), False, ops.PERM_NONE, 0),
                "end": batch("w_end", all_f, (F_, C, H, 1), False, ops.PERM_PAIR, C // 2),
            }
            if Lw > 1:
                self.pk["rs"] = batch("w_rs", all_f, (F_ * (Lw - 1), 2 * H, H, 1), False, ops.PERM_NONE, 0)
        elif nskip > 0:
            # the per-conv kernels read the Start / In_l / last Res_Skip / End slabs of the fused image as they are; only Res_Skip_l (l < L - 1)
            # is PAIR-packed there (residual | skip per 32 channels) and is packed once more in the per-conv order for the skipped flows
            pk_conv = dict(self.pk)
            if Lw > 1:
                pk_conv["rs"] = batch("w_rs", slice(0, nskip), (nskip * (Lw - 1), 2 * H, H, 1), False, ops.PERM_NONE, 0)
        # (the per-flow structs below hold raw pointers into this image: it lives as long as they do - ADVICE r3: as a local it went back to
        # the caching allocator / the graph pool at the end of __init__ while flow 0's launches still read it)
        self.pk_conv = pk_conv
        # backward: the transposed image of the fused data-gradient kernel (glowtts_wavenet_bwd) where it applies - no conditioning gradient -
        # else the per-conv transposed images
        self.wn_img_t = None
        nfb = TUNE["fused_wn_bwd"]
        if nfb is True:
            nfb = F_
        elif int(nfb) < 0:
            # automatic: a fused workgroup owns its CU.  A batch whose 52-row windows leave a quarter of the CUs free (B = 16 x 800 frames: 125
            # workgroups) shares the chip with the encoder stream anyway - every flow takes the fused kernel (config 4: 4.28 vs 4.68 ms/step);
            # a chip-filling one (B = 32: 249) only the last 7 of 12 flows: there the in-graph timeline (bench.py --timeline) shows the two
            # streams finishing their weight gradients together.  (It was 5 of 12 until the encoder's attention kernels stopped waiting for
            # every load of their operand staging - backward 81 -> 46 us, forward 36 -> 25 us per layer: with the shorter encoder backward
            # 6 / 7 / 8 / 9 / 10 / 12 fused flows measure 5.61 / 5.54-5.59 / 5.54-5.61 / 5.59-5.73 / 5.88 / 6.08 ms/step against 5.58-5.66 for 5.)
            nwg = -(-rows // (64 - 4 * (Lw - 1))) if rows else None
            cus = torch.cuda.get_device_properties(dev).multi_processor_count if dev.type == "cuda" else 256
            # (round 5: 8 of 12 - the encoder's backward chain lost its gate passes, the relative-position sums and ~100 us of weight packing at its head:
            # 5.02 vs 5.07 ms/step for 7, three alternating pairs on one box; 9: 5.23)
            # (round 6: 9 of 12 - the head of the encoder's backward lost its dozen PyTorch launches (duration projection, prior split: enc_dgrads_done 4 006 ->
            # 3 580 us): 4.69 / 4.73 vs 4.77 / 4.78 ms/step for 8, two alternating rounds on one box; 10: 4.80, 12: 4.95)
            # (conditioned modes stay at 8 of 12: config 3 5.01 vs 5.11 ms/step for 9 - their encoder stream also carries the conditioning encoders' backward)
            is_cond = (cond is not None) if conditioned is None else bool(conditioned)
            nfb = F_ if (nwg is not None and 4 * nwg <= 3 * cus) else max(1, ((2 if is_cond else 3) * F_) // (3 if is_cond else 4))
        else:
            nfb = min(int(nfb), F_)                             # flows 0 .. nfb-1 take the fused kernel
        f0 = min(max(int(TUNE["fused_wn_bwd_from"]), 0), F_ - nfb)               # (experiments: the fused flows are f0 .. f0 + nfb - 1)
        # (experiment, TUNE["bwd_packs_side"]: the images only the backward reads are packed on a stream of their own, joined when the backward starts)
        self.bwd_side = None
        main_s = torch.cuda.current_stream(dev) if dev.type == "cuda" else None
        self.fwd_side_from = None
        if need_bwd and GV is None and (TUNE["bwd_packs_side"] or ksplit) and main_s is not None:
            self.bwd_side = _pack_stream(dev)
            self.bwd_side.wait_stream(main_s)
            torch.cuda.set_stream(self.bwd_side)
            if ksplit:
                self.fwd_side_from = ksplit
                images(slice(ksplit, None), F_ - ksplit, self.wn_img[ksplit:], 0, None, False)
        if jobs_b is not None:
            cur[0] = jobs_b                                                    # everything from here on is read by the backward only
        if need_bwd and self.wn_img is not None and fused_bwd_ok and nfb > 0:
            self.wn_img_t = torch.empty_like(self.wn_img[:nfb])                # (the fused flows only)
            images(slice(f0, None), nfb, None, nfb, self.wn_img_t, False)
        else:
            nfb = 0
        c0 = nfb if f0 == 0 else 0                                             # per-conv transposed images: flows c0 .. F-1 (element i = flow c0 + i)
        if need_bwd and c0 < F_:
            rest, nr = slice(c0, None), F_ - c0
            self.pk.update({
                "start_t": batch("w_start", rest, (nr, H, C // 2, 1), True, ops.PERM_NONE, 0),
                "in_t": batch("w_in", rest, (nr * Lw, 2 * H, H, cfg.k), True, ops.PERM_PAIR, H),
                "rs_last_t": batch("w_rs_last", rest, (nr, H, H, 1), True, ops.PERM_NONE, 0),
                "end_t": batch("w_end", rest, (nr, C, H, 1), True, ops.PERM_PAIR, C // 2),
            })
            if Lw > 1:
                self.pk["rs_t"] = batch("w_rs", rest, (nr * (Lw - 1), 2 * H, H, 1), True, ops.PERM_NONE, 0)
        if self.bwd_side is not None:
            torch.cuda.set_stream(main_s)
            for t in [self.wn_img_t] + [self.pk[k].data for k in ("start_t", "in_t", "rs_last_t", "end_t", "rs_t") if k in self.pk]:
                if t is not None:
                    t.record_stream(main_s)
        if jobs is not None:
            stamp("dec_prep_begin")
            jobs.launch(dev)
            stamp("dec_prep_end")
            self.prep_jobs = jobs                                              # (keeps the job table and the tensors it points into)
        self.bwd_pending = jobs_b if (jobs_b is not None and jobs_b.n.value > 0) else None
        self.ldo = self.pk["end"].npad
        self.ldin = self.pk["in"].npad
        self.cond, self._H, self._Lw = cond, H, Lw
        self.params = []
        for f in range(F_):
            p = FlowParams()
            p.an_logs = W["an_logs"][f].data_ptr()
            p.an_bias = W["an_bias"][f].data_ptr()
            p.winfo = self.winfo[f].data_ptr()
            fused_f = self.wn_img is not None and f >= nskip
            pkf = self.pk if fused_f or pk_conv is None else pk_conv
            p.start = pkf["start"].at(f)
            p.end = pkf["end"].at(f)
            p.b_start = W["b_start"][f].data_ptr()
            p.b_end = W["b_end"][f].data_ptr()
            p.wn_img = self.wn_img[f].data_ptr() if fused_f else None
            for l in range(Lw):
                p.in_[l] = pkf["in"].at(f * Lw + l)
                p.b_in[l] = W["b_in"][f, l].data_ptr()
                if l < Lw - 1:
                    p.rs[l] = pkf["rs"].at(f * (Lw - 1) + l)
                    p.b_rs[l] = W["b_rs"][f, l].data_ptr()
                else:
                    p.rs[l] = pkf["rs_last"].at(f)
                    p.b_rs[l] = W["b_rs_last"][f].data_ptr()
            p.wn_img_t = self.wn_img_t[f - f0].data_ptr() if f0 <= f < f0 + nfb else None
            if need_bwd and p.wn_img_t is None:
                fc = f - c0
                p.start_t = self.pk["start_t"].at(fc)
                p.end_t = self.pk["end_t"].at(fc)
                for l in range(Lw):
                    p.in_t[l] = self.pk["in_t"].at(fc * Lw + l)
                    p.rs_t[l] = self.pk["rs_t"].at(fc * (Lw - 1) + l) if l < Lw - 1 else self.pk["rs_last_t"].at(fc)
            self.params.append(p)
        self.set_cond(cond)

    def launch_bwd_images(self, gentle=False):
        """Issues, on the CURRENT stream, the weight-preparation launch of the images only the backward reads, if it is still pending."""
        jb = getattr(self, "bwd_pending", None)
        if jb is None:
            return
        self.bwd_pending = None
        dev = self.winfo.device
        if gentle:
            # beside another stream's latency-critical kernels: one workgroup per CU (an 83-KB LDS request instead of 31 KB x 5 workgroups, which leave no CU
            # with LDS for anybody else - the log-prior's 42-KB operand pass waited for them: 15 -> 45 us); this launch has ~150 us of slack
            jb.max_cols = max(jb.max_cols, 2600)
        jb.launch(dev)
        self.prep_jobs_b = jb
        cs = torch.cuda.current_stream(dev)
        for t in [self.wn_img_t] + [self.pk[k].data for k in ("start_t", "in_t", "rs_last_t", "end_t", "rs_t") if k in self.pk]:
            if t is not None:
                t.record_stream(cs)

    def set_cond(self, cond):
        """Attach the per-call conditioning [B, F, L, 2H] (None: unconditioned) to the per-flow parameter structs."""
        F_, H, Lw = len(self.params), self._H, self._Lw
        self.cond = cond
        for f, p in enumerate(self.params):
            p.cond = (cond.data_ptr() + 4 * f * Lw * 2 * H) if cond is not None else None
            p.ldcond = F_ * Lw * 2 * H if cond is not None else 0
            p.cond_rows = 0

    def set_cond_rows(self, f, rows):
        """Flow f reads per-ROW conditioning `rows` [R, L, 2H] (per-utterance terms already added in; GR-mode pitch, Modules.py:867-869)."""
        p = self.params[f]
        p.cond, p.ldcond, p.cond_rows = rows.data_ptr(), self._Lw * 2 * self._H, 1


def _dims(cfg, B, T, drop_p=0.0, seed=None, flow=0, chunk=0):
    """seed: None or a device int32/uint32 tensor with one element (re-drawn on device every step, hipGraph-safe).
    chunk: 0 (round 5's utterance-chunk experiment numbered its chunks' rows from 0 and gave each its own dropout key)."""
    return FlowDims(B, T, cfg.C, cfg.H, cfg.L, cfg.k, cfg.precision, float(drop_p), (1000003 * flow + 7368787 * chunk) & 0xFFFFFFFF,
                    seed.data_ptr() if seed is not None else None, int(cfg.act_bf16))


def squeeze_rows(cfg, mels, lengths, want_mask=True, out=None):
    """[B,Cm,Tm] -> rows [B*(T+4), C] (+ rowmask [B*(T+4)]).  out: write the rows there (a kept buffer) instead of a fresh tensor."""
    B, Cm, Tm = mels.shape
    T = Tm // cfg.ns
    R = B * (T + 2 * ROW_PAD)
    rows = out if out is not None else torch.empty(R, cfg.C, device=mels.device)
    mask = torch.empty(R, device=mels.device) if want_mask else None
    _lib.check(_L().glowtts_squeeze_rows(_lib.ptr(mels.contiguous()), _lib.ptr(rows), _lib.ptr(mask), _lib.ptr(lengths),
                                         B, Cm, Tm, cfg.ns, _lib.stream()), "squeeze_rows")
    return rows, mask, T


def unsqueeze_rows(cfg, rows, lengths, B, Tm, fill=None):
    mel = torch.empty(B, cfg.Cm, Tm, device=rows.device)
    if Tm % cfg.ns:
        mel.zero_()
    _lib.check(_L().glowtts_unsqueeze_rows(_lib.ptr(rows), _lib.ptr(mel), _lib.ptr(lengths), B, cfg.Cm, Tm, cfg.ns,
                                           0 if fill is None else 1, 0.0 if fill is None else float(fill), _lib.stream()), "unsqueeze_rows")
    return mel


class _Buffers:
    """Kept activations of a training forward (all flows)."""

    def __init__(self, cfg, prep, R, dev):
        F_, L, H, C = cfg.F, cfg.L, cfg.H, cfg.C
        self.x = torch.empty(F_ + 1, R, C, device=dev)
        self.xmid = torch.empty(F_, R, C, device=dev)
        self.hs = torch.empty(F_, L, R, H, device=dev, dtype=cfg.act_dtype)
        self.gates = torch.empty(F_, L, R, 2 * H, device=dev, dtype=cfg.act_dtype)
        self.actp = torch.empty(F_, L, R, H, device=dev, dtype=torch.bfloat16) if cfg.act_bf16 else None     # tanh * sigmoid
        self.skipb = torch.empty(F_, R, H, device=dev, dtype=torch.bfloat16) if cfg.act_bf16 else None      # bf16 copy of skip (End conv operand)
        self.skip = torch.empty(F_, R, H, device=dev)
        self.outs = torch.empty(F_, R, prep.ldo, device=dev)
        # bf16 copy of x_a = xmid[:, :C/2] (written by the flow's ActNorm + 1x1 pass): X of the Start conv's weight gradient
        self.xa_bf = torch.empty(F_, R, C // 2, device=dev, dtype=torch.bfloat16) if (cfg.act_bf16 and (C // 2) % 8 == 0) else None
        # the fp32 skip sum: scratch of the per-conv forward launches and X of the End conv's weight gradient where that reads fp32 operands.  A flow on the
        # fused forward launch whose backward takes the bf16 copy needs neither: its launch is told not to keep it (a->skip = NULL: 768 of 8 064 kept bytes per row)
        bf_tail = self.skipb is not None and self.xa_bf is not None
        self.keep_skip32 = [not (bf_tail and TUNE["drop_skip32"] and bool(prep.params[f].wn_img)) for f in range(F_)]

    def acts(self, f, L, rowmask, row0=0):
        """row0: first row of the utterance chunk the launch serves (every kept tensor is rows-major: a chunk is a row range of each)."""
        at = lambda t: t.data_ptr() + row0 * t.stride(-2) * t.element_size()
        a = FlowActs()
        a.xin, a.xmid, a.xout = at(self.x[f]), at(self.xmid[f]), at(self.x[f + 1])
        for l in range(L):
            a.hs[l] = at(self.hs[f, l])
            a.gates[l] = at(self.gates[f, l])
            if self.actp is not None:
                a.acts[l] = at(self.actp[f, l])
        if self.skipb is not None:
            a.skip_bf = at(self.skipb[f])
        if self.xa_bf is not None:
            a.xa_bf = at(self.xa_bf[f])
        a.skip, a.outs, a.rowmask = (at(self.skip[f]) if self.keep_skip32[f] else None), at(self.outs[f]), rowmask.data_ptr() + 4 * row0
        return a


def pitch_rows(cfg, pitches, rowmask, B, T):
    """Squeeze of the per-frame pitch (Modules.py:300-301: Squeeze(pitches.unsqueeze(1), mask) -> [B, ns, T]) in the rows layout:
    [R, ns], row of (utterance b, squeezed frame t) = pitches[b, ns*t : ns*t + ns] * mask'[t]; pad rows zero."""
    ns = cfg.ns
    p = pitches[:, :T * ns].reshape(B, T, ns).to(torch.float32)
    p = torch.nn.functional.pad(p, (0, 0, ROW_PAD, ROW_PAD)).reshape(B * (T + 2 * ROW_PAD), ns)
    return (p * rowmask.unsqueeze(1)).contiguous()


def _cond_rows(cfg, prep, f, prow, pitch_w, pitch_b, Tp):
    """Per-row conditioning of flow f: Pitch_l(pitch)[r] + per-utterance (speaker + prosody) terms -> [R, L, 2H] fp32."""
    cr = torch.einsum("rj,lnj->rln", prow, pitch_w[f]) + pitch_b[f]
    if prep.cond is not None:
        cr = cr + prep.cond[:, f].repeat_interleave(Tp, dim=0)
    return cr.contiguous()


def _params_at(prep, f, b0):
    """Flow f's parameter struct as utterance b0's chunk sees it: the per-utterance conditioning rows start at b0."""
    p = prep.params[f]
    if b0 == 0 or not p.cond:
        return p
    q = FlowParams.from_buffer_copy(p)
    q.cond = p.cond + 4 * b0 * p.ldcond
    return q


def _run_forward(cfg, prep, mels, lengths, drop_p=0.0, seed=None, pitch=None):
    """pitch: None or (pitches [B, Tm], pitch_w [F, L, 2H, ns], pitch_b [F, L, 2H]) - GR mode."""
    L = _L()
    B, _, Tm = mels.shape
    R = B * (Tm // cfg.ns + 2 * ROW_PAD)
    buf = _Buffers(cfg, prep, R, mels.device)
    _, rowmask, T = squeeze_rows(cfg, mels, lengths, out=buf.x[0])
    prow = pitch_rows(cfg, pitch[0], rowmask, B, T) if pitch is not None else None
    stamp("dec_fwd_begin")
    Tp = T + 2 * ROW_PAD

    def chain(ci, b0, nb):
        """All flows of utterances [b0, b0 + nb) on the current stream."""
        chained = False
        for f in range(cfg.F):
            if pitch is not None:
                cr = _cond_rows(cfg, prep, f, prow, pitch[1], pitch[2], Tp)
                prep.set_cond_rows(f, cr)
            if getattr(prep, "fwd_side_from", None) == f:
                torch.cuda.current_stream(mels.device).wait_stream(prep.bwd_side)
            acts = buf.acts(f, cfg.L, rowmask, b0 * Tp)
            acts.actnorm_done = int(chained)
            # a flow on the fused coupling launch also applies the NEXT flow's ActNorm + 1x1 conv in that launch's epilogue (one launch less per flow on the
            # decoder's forward chain; csrc/wavenet_fused.hip)
            chained = bool(TUNE["chain_actnorm"]) and f + 1 < cfg.F and bool(prep.params[f].wn_img) and pitch is None
            if chained:
                nxt, pn = buf.acts(f + 1, cfg.L, rowmask, b0 * Tp), prep.params[f + 1]
                acts.next_an_logs, acts.next_an_bias, acts.next_winfo = pn.an_logs, pn.an_bias, pn.winfo
                acts.next_xmid, acts.next_xout, acts.next_xa_bf = nxt.xmid, nxt.xout, nxt.xa_bf
            dims = _dims(cfg, nb, T, drop_p, seed, f, ci)
            _lib.check(L.glowtts_flow_forward(ctypes.byref(dims), ctypes.byref(_params_at(prep, f, b0)), ctypes.byref(acts), _lib.stream()),
                       "glowtts_flow_forward")
    chain(0, 0, B)
    stamp("dec_fwd_end")
    # (z in the reference's [B, C, T] layout and the log-determinants on the encoder's stream, off the chain to the log-prior, was measured in round 5 and lost:
    #  5.07 against 5.01 ms/step - the extra cross-stream edges cost the replayed graph more than the two passes take)
    an_logs = prep.keep["an_logs"].contiguous()
    z = unsqueeze_rows(cfg, buf.x[cfg.F], lengths, B, Tm)
    part = torch.empty(cfg.F * B, device=mels.device)
    logdet = torch.empty(B, device=mels.device)
    _lib.check(L.glowtts_decoder_logdet(_lib.ptr(buf.outs), R * prep.ldo, _lib.ptr(an_logs), _lib.ptr(prep.winfo),
                                        _lib.ptr(rowmask), _lib.ptr(part), _lib.ptr(logdet), cfg.F, B, T + 2 * ROW_PAD, cfg.C, prep.ldo,
                                        _lib.stream()), "glowtts_decoder_logdet")
    if pitch is not None:
        prep.set_cond(prep.cond)            # the backward addresses the conditioning gradient per utterance
    return z, logdet, buf, rowmask, T, prow


def decoder_inverse(cfg, W, z, lengths, cond=None, fill=None, prep=None, pitch=None):
    """Decoder.forward(reverse=True) (Modules.py:298-309): z [B,Cm,Tm] -> mels [B,Cm,ns*(Tm//ns)].
    prep: a _Prepared built earlier from the same weights (then W is not read; `cond`, which varies per call, is attached here).
    pitch: None or (pitches [B, Tm], pitch_w [F, L, 2H, ns], pitch_b [F, L, 2H]) - GR mode."""
    L = _L()
    if prep is None:
        prep = _Prepared(cfg, W, need_bwd=False, cond=cond)
    elif cond is not None or prep.cond is not None:
        prep.set_cond(cond)
    B, _, Tm = z.shape
    x, rowmask, T = squeeze_rows(cfg, z, lengths)
    R = x.shape[0]
    dev = z.device
    other = torch.empty_like(x)
    xmid = torch.empty_like(x)
    hs = torch.empty(2, R, cfg.H, device=dev, dtype=cfg.act_dtype)
    gates = torch.empty(R, 2 * cfg.H, device=dev, dtype=cfg.act_dtype)
    actp = torch.empty(R, cfg.H, device=dev, dtype=torch.bfloat16) if cfg.act_bf16 else None
    skipb = torch.empty(R, cfg.H, device=dev, dtype=torch.bfloat16) if cfg.act_bf16 else None
    skip = torch.empty(R, cfg.H, device=dev)
    dims = _dims(cfg, B, T)
    cur, nxt = x, other
    prow = pitch_rows(cfg, pitch[0], rowmask, B, T) if pitch is not None else None
    for f in range(cfg.F - 1, -1, -1):
        if pitch is not None:
            cr = _cond_rows(cfg, prep, f, prow, pitch[1], pitch[2], T + 2 * ROW_PAD)
            prep.set_cond_rows(f, cr)
        a = FlowActs()
        a.xout, a.xmid, a.xin = cur.data_ptr(), xmid.data_ptr(), nxt.data_ptr()
        a.hs[0], a.hs[1] = hs[0].data_ptr(), hs[1].data_ptr()
        a.gates[0], a.skip, a.rowmask = gates.data_ptr(), skip.data_ptr(), rowmask.data_ptr()
        if actp is not None:
            a.acts[0], a.skip_bf = actp.data_ptr(), skipb.data_ptr()
        a.outs = None
        _lib.check(L.glowtts_flow_inverse(ctypes.byref(dims), ctypes.byref(prep.params[f]), ctypes.byref(a), _lib.stream()),
                   "glowtts_flow_inverse")
        cur, nxt = nxt, cur
    return unsqueeze_rows(cfg, cur, lengths, B, (Tm // cfg.ns) * cfg.ns, fill=fill)


def actnorm_data_init(cfg, W, mels, lengths, cond=None, allreduce=None, pitch=None):
    """ActNorm data-dependent init (Modules.py:685-687, 698-711): runs the flows once, setting
    W['an_logs'][f] / W['an_bias'][f] from the masked batch statistics of each flow's input.
    `allreduce(stats)` (optional) sums the [2C+1] statistics over data-parallel ranks."""
    L = _L()
    with torch.no_grad():
        B, _, Tm = mels.shape
        x, rowmask, T = squeeze_rows(cfg, mels, lengths)
        R = x.shape[0]
        dev = mels.device
        dims = _dims(cfg, B, T)
        stats = torch.empty(2 * cfg.C + 1, device=dev)
        scratch = torch.empty(L.glowtts_actnorm_stats_scratch_floats(R, cfg.C), device=dev)
        bufs = [x, torch.empty_like(x)]
        xmid = torch.empty_like(x)
        hs = torch.empty(cfg.L, R, cfg.H, device=dev, dtype=cfg.act_dtype)
        gates = torch.empty(cfg.L, R, 2 * cfg.H, device=dev, dtype=cfg.act_dtype)
        actp = torch.empty(cfg.L, R, cfg.H, device=dev, dtype=torch.bfloat16) if cfg.act_bf16 else None
        skip = torch.empty(R, cfg.H, device=dev)
        for f in range(cfg.F):
            _lib.check(L.glowtts_actnorm_stats(_lib.ptr(bufs[0]), _lib.ptr(rowmask), _lib.ptr(stats), _lib.ptr(scratch), R, cfg.C,
                                               _lib.stream()), "actnorm_stats")
            if allreduce is not None:
                allreduce(stats)
            _lib.check(L.glowtts_actnorm_from_stats(_lib.ptr(stats), W["an_logs"][f].data_ptr(), W["an_bias"][f].data_ptr(), cfg.C,
                                                    _lib.stream()), "actnorm_from_stats")
            prep = _Prepared(cfg, W, need_bwd=False, cond=cond)       # re-pack is cheap relative to a one-off init
            if pitch is not None:
                prow = pitch_rows(cfg, pitch[0], rowmask, B, T)
                cr = _cond_rows(cfg, prep, f, prow, pitch[1], pitch[2], T + 2 * ROW_PAD)
                prep.set_cond_rows(f, cr)
            outs = torch.empty(R, prep.ldo, device=dev)
            a = FlowActs()
            a.xin, a.xmid, a.xout = bufs[0].data_ptr(), xmid.data_ptr(), bufs[1].data_ptr()
            for l in range(cfg.L):
                a.hs[l], a.gates[l] = hs[l].data_ptr(), gates[l].data_ptr()
                if actp is not None:
                    a.acts[l] = actp[l].data_ptr()
            a.skip, a.outs, a.rowmask = skip.data_ptr(), outs.data_ptr(), rowmask.data_ptr()
            _lib.check(L.glowtts_flow_forward(ctypes.byref(dims), ctypes.byref(prep.params[f]), ctypes.byref(a), _lib.stream()),
                       "glowtts_flow_forward(init)")
            bufs.reverse()


# The weight preparation of a training step can be issued BEFORE the text encoder is launched on its stream (modules.GlowTTS.forward): inside a
# replayed hipGraph the decoder's branch otherwise starts ~180 us after the encoder's (measured with in-graph stamps: the runtime reaches the nodes of
# the second branch late), and the 80-us preparation launch sat behind that.  `early_prepare` builds the _Prepared; DecoderFunction.forward picks it up.
EARLY = {"prep": None, "key": None}
# A stream the caller joins with its own before it consumes z / the log-determinants (modules.GlowTTS.forward: the text encoder's stream), or None
AUX = {"stream": None, "stacks": None, "seed": None}


def _leaf_grads_unset():
    """True when no leaf parameter of the decoder holds a gradient yet (zero_grad(set_to_none=True), the trainer's and torch's default): AccumulateGrad then
    takes the returned tensors over without reading them."""
    st = AUX["stacks"]() if AUX["stacks"] is not None else None
    return st is not None and all(p.grad is None for ls in st.S.values() for p in ls.leaves)



def _split_gv(weights):
    A = dict(zip(WEIGHT_KEYS_GV, [w.detach().contiguous() for w in weights]))
    GV = {k: (A.pop("g" + k[1:]), A.pop("v" + k[1:])) for k in WN_KEYS}
    return A, GV


def early_prepare(cfg, weights, mel_shape, fused_bwd_ok=True, conditioned=False):
    """weights: DecoderStacks.weights(gv=True).  Must be followed by DecoderFunction.apply on the same weights in the same forward."""
    if len(weights) != len(WEIGHT_KEYS_GV):
        return
    W, GV = _split_gv(weights)
    need_bwd = any(w.requires_grad for w in weights)
    EARLY["prep"] = _Prepared(cfg, W, need_bwd=need_bwd, cond=None, fused_bwd_ok=fused_bwd_ok,
                              rows=mel_shape[0] * (mel_shape[2] // cfg.ns + 2 * ROW_PAD), GV=GV, conditioned=conditioned)
    EARLY["key"] = (tuple(w.data_ptr() for w in weights), need_bwd, bool(fused_bwd_ok), GV, bool(conditioned))


class DecoderFunction(torch.autograd.Function):
    """z, logdet = Decoder(mels)   with autograd through the hand-written backward kernels."""

    @staticmethod
    def forward(ctx, cfg, mels, lengths, cond, drop_p, pitches, pitch_w, pitch_b, *weights):
        """pitches [B, Tm] / pitch_w [F, L, 2H, ns] / pitch_b [F, L, 2H]: the GR-mode per-frame pitch conditioning (Modules.py:867-869), else None."""
        GV = None
        if len(weights) == len(WEIGHT_KEYS_GV):
            W, GV = _split_gv(weights)
        else:
            W = dict(zip(WEIGHT_KEYS, [w.detach().contiguous() for w in weights]))
        need_bwd = any(w.requires_grad for w in weights) or mels.requires_grad or (cond is not None and cond.requires_grad) or \
            (pitch_w is not None and pitch_w.requires_grad)
        condc = cond.detach().contiguous() if cond is not None else None
        early, ekey = EARLY["prep"], EARLY["key"]
        EARLY["prep"] = EARLY["key"] = None
        if early is not None and GV is not None and ekey[0] == tuple(w.data_ptr() for w in weights) and ekey[1] == need_bwd and ekey[2] == (pitches is None) and \
                ekey[4] == (cond is not None):
            prep, GV = early, ekey[3]                           # (issued before the encoder's launches; the same weights)
            prep.set_cond(condc)
        else:
            prep = _Prepared(cfg, W, need_bwd=need_bwd, cond=condc, fused_bwd_ok=pitches is None,
                             rows=mels.shape[0] * (mels.shape[2] // cfg.ns + 2 * ROW_PAD), GV=GV)
        ctx.GV = GV
        # one random word on the device (torch's graph-safe generator); kept for the backward, which regenerates the masks
        seed = None
        if drop_p > 0:                                          # (GlowTTS.forward hands the step's word over; a direct caller draws one here)
            seed, AUX["seed"] = AUX.get("seed"), None
            if seed is None:
                seed = step_seed(mels.device)
        pitch = (pitches.detach(), pitch_w.detach().contiguous(), pitch_b.detach().contiguous()) if pitches is not None else None
        if pitch is not None and condc is None:
            raise _lib.GlowTTSHipError("per-frame pitch conditioning comes with the GR mode's speaker / prosody conditioning (Modules.py:84-90)")
        z, logdet, buf, rowmask, T, prow = _run_forward(cfg, prep, mels.detach(), lengths, drop_p, seed, pitch)
        ctx.drop = (drop_p, seed)
        ctx.prow = prow
        if need_bwd:
            ctx.cfg, ctx.prep, ctx.buf, ctx.rowmask, ctx.T = cfg, prep, buf, rowmask, T
            ctx.lengths, ctx.mel_shape = lengths, mels.shape
            ctx.want_dmel = mels.requires_grad
        # third output (not differentiable): z in the rows layout [B*(T+4), C] - one squeezed row is ns consecutive frames x Cm channels,
        # i.e. already "frames x channels" for the log-prior GEMM (alignment.log_prior_t(z_rows=...) reads it instead of transposing z)
        z_rows = buf.x[cfg.F]
        ctx.mark_non_differentiable(z_rows)
        ctx.set_materialize_grads(False)                     # (its "gradient" stays None instead of an 8-MB zero fill in front of the decoder's backward)
        return z, logdet, z_rows

    @staticmethod
    def backward(ctx, dz, dlogdet, _dz_rows=None):
        L = _L()
        cfg, prep, buf, rowmask, T = ctx.cfg, ctx.prep, ctx.buf, ctx.rowmask, ctx.T
        W = prep.keep
        B, Cm, Tm = ctx.mel_shape
        if dz is None:                                       # (only the log-determinant was differentiated: tests)
            dz = torch.zeros(ctx.mel_shape, device=dlogdet.device)
        dev = dz.device
        if getattr(prep, "bwd_side", None) is not None:
            torch.cuda.current_stream(dev).wait_stream(prep.bwd_side)
        prep.launch_bwd_images()                                 # (nobody has issued the backward-only images yet: here, on this stream)
        F_, Lw, H, C = cfg.F, cfg.L, cfg.H, cfg.C
        dx, _, _ = squeeze_rows(cfg, dz.contiguous(), ctx.lengths, want_mask=False)
        R = dx.shape[0]
        dld = dlogdet.contiguous() if dlogdet is not None else torch.zeros(B, device=dev)
        GV = ctx.GV
        # every entry is fully written below (weight-gradient launches store, not accumulate)
        # The gradients this function RETURNS live in two arenas - one for the classes of the deferrable tail (TAIL_STACKS), one for the rest -, each
        # class a gap-free view: the leaves' .grad are views of them (LeafStack), so a data-parallel step exchanges the decoder's gradients with ONE
        # collective per arena instead of one per class (distributed.FlatGradReducer reduces storages that its gradients tile; a collective costs
        # ~25 us of launch and stream hand-over whatever its size).  Weight-normalised classes return (d g, d v); their d w is an intermediate.
        from .distributed import register_arena

        def arenas(shapes):
            out = {}
            for tail in (False, True):
                ks = [k for k in shapes if (_TAIL_OF.get(k, False)) == tail]
                sizes = [int(math.prod(shapes[k])) for k in ks]
                padded = [(n + 3) & ~3 for n in sizes]                       # (16-byte aligned classes)
                flat = (torch.empty if padded == sizes else torch.zeros)(sum(padded), device=dev)
                off = 0
                for k, n, npad in zip(ks, sizes, padded):
                    out[k] = flat[off:off + n].view(shapes[k])
                    off += npad
                if ks:
                    register_arena(*[out[k] for k in ks])                    # (its pads are private and zero: the reducer may sum the whole span)
            return out
        ret_shapes = {}
        for k in (WEIGHT_KEYS_GV if GV is not None else WEIGHT_KEYS):
            wk = "w" + k[1:]
            if GV is not None and k[0] in "gv" and wk in GV:
                ret_shapes[k] = tuple(GV[wk][0 if k[0] == "g" else 1].shape)
            else:
                ret_shapes[k] = tuple(W[k].shape)
        RET = arenas(ret_shapes)
        G = {k: (RET[k] if k in RET else torch.empty_like(GV[k][1] if (GV is not None and k in GV) else W[k])) for k in WEIGHT_KEYS}
        d_an = torch.empty(F_, 2 * C + 16, device=dev)
        # every flow / layer keeps its own gradient buffers: the weight gradients of ALL flows are computed afterwards by
        # two grouped launches (k-tap problems, 1x1 problems) whose tiles fill the chip without split-K or atomics
        # (fp32 rows of d(m, logs): only where somebody reads them - with bf16-stored activations every consumer, the End conv's data and weight gradients, takes
        #  the bf16 copy below, and the coupling backward then writes that copy alone: 768 of the 4 480 bytes per row of its pass, round 6)
        h0bf_ = bool(cfg.act_bf16 and buf.xa_bf is not None and buf.skipb is not None)
        douts = None if h0bf_ else torch.empty(F_, R, prep.ldo, device=dev)           # (pad columns are zeroed by the coupling backward kernel)
        # bf16 copy of d(m, logs): the End data gradient's operand and (round 4) DY of the End conv's weight gradient - one per flow, like everything the
        # deferred weight-gradient launches read
        douts_bf = torch.empty(F_, R, prep.ldo, device=dev, dtype=torch.bfloat16) if cfg.act_bf16 else None
        fuse = TUNE["fuse_coupling_bwd"]
        dins = (torch.empty if prep.ldin == 2 * H else torch.zeros)(F_, Lw, R, prep.ldin, device=dev, dtype=cfg.act_dtype)    # only pad columns need zeros
        dskip = torch.empty(F_, R, H, device=dev, dtype=cfg.act_dtype)
        # d h0: bf16 like d x_l (round 4: the Start conv's data gradient reads bf16 rows, its weight gradient raw bf16 operands) where the
        # bf16 copy of x_a exists; else fp32
        h0bf = bool(cfg.act_bf16 and buf.xa_bf is not None and buf.skipb is not None)
        dh0 = torch.empty(F_, R, H, device=dev, dtype=cfg.act_dtype if h0bf else torch.float32)
        dhn = torch.empty(F_, max(Lw - 1, 1), R, H, device=dev, dtype=cfg.act_dtype)       # d x_l, l >= 1
        dh_ptr = lambda f, l: dh0[f].data_ptr() if l == 0 else dhn[f, l - 1].data_ptr()
        nscr = L.glowtts_actnorm_stats_scratch_floats(R, C)
        scratch = torch.empty(F_, nscr, device=dev)          # per-flow partials of the ActNorm / 1x1 parameter gradients, reduced once below
        # conditioning gradient [B, F*L*2H], accumulated by the gate-derivative epilogues; GR mode: + ns rows that collect the Pitch_l weight gradients
        npit = ctx.prow.shape[1] if ctx.prow is not None else 0
        # (int64: 2^-40 fixed-point accumulators of the kernels' integer atomic adds - the sums do not depend on the order the workgroups arrive in)
        dcond = torch.zeros((B + npit,) + tuple(prep.cond.shape[1:]), device=dev, dtype=torch.int64) if prep.cond is not None else None
        bf = cfg.act_bf16
        gk = WgradGroup(R, cfg.k, cfg.precision, io_flags=(ops.WIO_DY_BF16 | ops.WIO_X_BF16) if bf else 0)     # In_l (k taps)
        g1 = WgradGroup(R, 1, cfg.precision)                            # Start / End (1x1)
        # Res_Skip_l (1x1 on tanh*sigmoid): the stored bf16 product, or the fp32 (tanh, sigmoid) pairs through the PAIRMUL prologue
        # Row splits of the one-tap group (round 5, TUNE["wgrad_tail_splits"] = S): its 108 problems move 1.07 GB of unique operands and are bound by how many
        # bytes a workgroup keeps in flight; S x the workgroups over B / S utterances each (one tap: no halo, any row cut is exact) into S partial images,
        # summed in a fixed order into the gradient tensors by ONE launch (glowtts_sum_slices_seg)
        S_t = int(TUNE["wgrad_tail_splits"])
        tail_seg = tail_part = tail_fns = None
        if S_t > 1 and h0bf and B % S_t == 0 and int(TUNE["wgrad_split"]) == 1:
            names = ["w_end", "b_end", "w_rs_last", "b_rs_last", "w_start", "b_start"] + (["w_rs", "b_rs"] if Lw > 1 else [])
            if all(G[k].is_contiguous() and G[k].numel() % 4 == 0 and G[k].data_ptr() % 16 == 0 for k in names):
                tot, tail_seg = 0, []
                for k in names:
                    tail_seg.append((G[k].data_ptr(), G[k].numel(), tot))
                    tot += G[k].numel()
                tail_part = torch.empty(S_t, tot, device=dev)
        R_t = R // S_t if tail_seg is not None else R
        # (in row splits the group takes the LDS-DMA kernel's 192 x 192 one-tap tiles: 108 x S workgroups that read every operand byte once - 4.84 against 4.96
        # ms/step for S = 2, two alternating runs on one box; unsplit that kernel is 108 workgroups on 256 CUs and loses to the staged one: 4.99)
        gp = WgradGroup(R_t, 1, cfg.precision, ops.APRO_NONE if bf else ops.APRO_PAIRMUL, io_flags=(ops.WIO_DY_BF16 | ops.WIO_X_BF16) if bf else 0,
                        dma_k1=True if tail_seg is not None else None)
        if tail_seg is not None:
            _gp_add = gp.add

            def _split_add(dy, lddy, m, x, ldx, ca, dw, dbias, **kw):          # (bf16 operands: 2 bytes per element)
                def image(ptr, s_):
                    for p0, n, off in tail_seg:
                        if p0 <= ptr < p0 + 4 * n:
                            return tail_part.data_ptr() + 4 * (s_ * tail_part.shape[1] + off) + (ptr - p0)
                    raise AssertionError("one-tap weight gradient outside the split's segments")
                for s_ in range(S_t):
                    _gp_add(dy + 2 * s_ * R_t * lddy, lddy, m, x + 2 * s_ * R_t * ldx, ldx, ca, image(dw, s_), image(dbias, s_), **kw)
            gp.add = _split_add
        C2 = C // 2
        order = list(range(F_ - 1, -1, -1))
        for f in order:       # weight-gradient problems of every flow (autograd of Modules.py:791,861,871,793): all pointers are known up front
            # End conv: fp32 (m, logs) gradients x fp32 skip sum, in the Start conv's launch (a launch costs one tile time whatever its tile
            # count up to the CU count: a separate launch for these 72 tiles was +0.15 ms, inside the Res_Skip launch +0.10 ms)
            if h0bf:      # raw bf16 operands: the End / Start problems ride in the Res_Skip group's launch (same rows, 1 tap, 16-byte items)
                gp.add(douts_bf[f].data_ptr(), prep.ldo, prep.ldo, buf.skipb[f].data_ptr(), H, H, G["w_end"][f].data_ptr(), G["b_end"][f].data_ptr(),
                       perm=ops.PERM_PAIR, perm_h=C2)
            else:
                g1.add(douts[f].data_ptr(), prep.ldo, prep.ldo, buf.skip[f].data_ptr(), H, H, G["w_end"][f].data_ptr(), G["b_end"][f].data_ptr(),
                       perm=ops.PERM_PAIR, perm_h=C2)
            for l in range(Lw):
                gates, ldg = (buf.actp[f, l].data_ptr(), H) if bf else (buf.gates[f, l].data_ptr(), 2 * H)
                if l == Lw - 1:
                    gp.add(dskip[f].data_ptr(), H, H, gates, ldg, H, G["w_rs_last"][f].data_ptr(), G["b_rs_last"][f].data_ptr())
                else:
                    gp.add(dh_ptr(f, l + 1), H, H, gates, ldg, H, G["w_rs"][f, l].data_ptr(), G["b_rs"][f, l].data_ptr())
                    gp.add(dskip[f].data_ptr(), H, H, gates, ldg, H, G["w_rs"][f, l].data_ptr() + 4 * H * H, G["b_rs"][f, l].data_ptr() + 4 * H)
                gk.add(dins[f, l].data_ptr(), prep.ldin, prep.ldin, buf.hs[f, l].data_ptr(), H, H, G["w_in"][f, l].data_ptr(), G["b_in"][f, l].data_ptr(),
                       perm=ops.PERM_PAIR, perm_h=H)
            if h0bf:
                gp.add(dh0[f].data_ptr(), H, H, buf.xa_bf[f].data_ptr(), C2, C2, G["w_start"][f].data_ptr(), G["b_start"][f].data_ptr())
            else:
                g1.add(dh0[f].data_ptr(), H, H, buf.xmid[f].data_ptr(), C, C2, G["w_start"][f].data_ptr(), G["b_start"][f].data_ptr())
            # TUNE["wgrad_split"] = n: weight gradients in n segments, each launched on a second stream as soon as its flows' chain is
            # done (default 1: one launch per class after the chain; see DESIGN.md for the measurements)
            nseg = int(TUNE["wgrad_split"])
            if nseg > 1:
                per = -(-len(order) // nseg)
                pos = order.index(f) + 1
                if pos % per == 0 and pos < len(order):
                    for grp in (gk, g1, gp):
                        grp.end_segment()
        # Few, large launches: 216-432 k-tap tiles per launch fill the chip without split-K (per-flow launches of 36 tiles were
        # measured 3x slower in total, even on a second stream).
        # TUNE["wgrad_balance"] (round 6, OFF): 288 one-per-CU In_l tiles on 256 CUs are two rounds, the second with 32 tiles; the balanced two-phase launch
        # (WgradGroup.balance) runs them in 1 + 1/8 rounds and ends the decoder's tail 126 us earlier (dec_wgrads_done 4 311 against 4 437 us) - and the step
        # takes as long as before (4.71-4.77 against 4.66-4.76 ms/step, five alternating pairs on two boxes): the idle CUs of the second round were where the
        # encoder's backward ran, which now ends that much later (enc_wgrads_done 4 523 against 4 292).  Kept for a step whose second half is not shared.
        if TUNE["wgrad_balance"] and int(TUNE["wgrad_split"]) == 1 and bf:
            gk.balance(B, T + 2 * ROW_PAD, dev)
        for grp in (gk, g1, gp):
            grp.end_segment()
        WgradGroup.upload_all((gk, g1, gp), dev)
        halves = len(gk.segments)

        def sum_tail():
            if tail_seg is not None:                       # the one-tap group's partial images -> the gradient tensors
                segs = (SumSeg * len(tail_seg))(*[SumSeg(p0, off, n) for p0, n, off in tail_seg])
                _lib.check(L.glowtts_sum_slices_seg(tail_part.data_ptr(), S_t, tail_part.shape[1], segs, len(tail_seg), _lib.stream()), "glowtts_sum_slices_seg")
        main = torch.cuda.current_stream()
        side = _wgrad_stream(dev)
        stamp("dec_bwd_begin")
        Tp = T + 2 * ROW_PAD
        chunks = [(0, 0, B)]
        # the ActNorm / 1x1 backward leaves one row of partial sums per block of its launch: a chunk's blocks follow the previous chunk's
        blk_off, nblk_all = [], 0
        for _, _, nb in chunks:
            blk_off.append(nblk_all)
            nblk_all += L.glowtts_actnorm_bwd_blocks(nb * Tp)
        assert nblk_all * (2 * C + 16) <= nscr
        at = lambda t, row0: t.data_ptr() + row0 * t.stride(-2) * t.element_size()

        def chain(ci, b0, nb):
            r0 = b0 * Tp
            for f in order:
                g = FlowGrads()
                g.dx, g.dlogdet, g.douts, g.dskip = at(dx, r0), dld.data_ptr() + 4 * b0, (at(douts[f], r0) if douts is not None else None), at(dskip[f], r0)
                g.douts_bf = at(douts_bf[f], r0) if douts_bf is not None else None
                g.dh0_bf16 = int(h0bf)
                # the last kernel of this flow's backward also applies the coupling backward of the flow that runs next (f - 1)
                g.coupling_done = int(fuse and f != order[0])
                if fuse and f > 0:
                    g.prev_xmid, g.prev_outs, g.prev_douts = at(buf.xmid[f - 1], r0), at(buf.outs[f - 1], r0), (at(douts[f - 1], r0) if douts is not None else None)
                    g.prev_douts_bf = at(douts_bf[f - 1], r0) if douts_bf is not None else None
                g.scratch, g.d_an, g.defer_wgrad = scratch[f].data_ptr() + 4 * blk_off[ci] * (2 * C + 16), None, 1
                for l in range(Lw):
                    g.dh[l] = at(dh0[f], r0) if l == 0 else at(dhn[f, l - 1], r0)
                    g.dins[l] = at(dins[f, l], r0)
                if dcond is not None:
                    g.dcond = dcond.data_ptr() + 8 * f * Lw * 2 * H + 8 * b0 * dcond.stride(0)
                    if npit:
                        g.pitch_rows, g.pitch_ns = ctx.prow.data_ptr(), npit
                acts = buf.acts(f, Lw, rowmask, r0)
                dims = _dims(cfg, nb, T, ctx.drop[0], ctx.drop[1], f, ci)
                _lib.check(L.glowtts_flow_backward(ctypes.byref(dims), ctypes.byref(_params_at(prep, f, b0)), ctypes.byref(acts), ctypes.byref(g),
                                                   _lib.stream()), "glowtts_flow_backward")
                if ci == 0:
                    stamp(f"dec_bwd_flow{f}")
                if halves > 1:
                    per = -(-len(order) // halves)
                    pos = order.index(f) + 1
                    if pos % per == 0 and pos < len(order):
                        side.wait_stream(main)
                        with torch.cuda.stream(side):
                            for grp in (gk, gp, g1):
                                grp.launch_segment(pos // per - 1)
        chain(0, 0, B)
        if halves > 1:
            side.wait_stream(main)
            with torch.cuda.stream(side):
                for grp in (gk, gp, g1):
                    grp.launch_segment(halves - 1)
            main.wait_stream(side)
        elif TAIL["defer"]:
            gk.launch_segment(0)
            # the queued launches read the kept activations and this backward's gradient buffers through raw pointers: the closure
            # keeps them alive (also across replays when the flush is captured as its own hipGraph)
            keep = (buf, dins, dskip, dh0, dhn, douts, G)
            TAIL["pending"].append(lambda keep=keep: (gp.launch_segment(0), g1.launch_segment(0), sum_tail()))
        else:
            # Conditioned modes (SE / PE / GR): the conditioning gradient is complete when the data-gradient chain ends, and a whole backward pass hangs
            # on it (CondLinear, then the prosody encoder: ~60 launches; config 5 ran them BEHIND the decoder's ~0.9 ms of weight-gradient launches).
            # The tail - weight-gradient groups, weight-norm backward, parameter-gradient sums - needs nothing from that pass and nothing of the pass
            # needs the tail: it goes to the weight-gradient stream (`tail_fns`, issued by `release`), the caller's stream carries on with the conditioning
            # encoders' backward, and the two are joined by a callback the autograd engine runs when the whole backward has been issued
            # (TUNE["tail_aside"]; config 5 6.38 -> 6.08 ms/step, config 3 5.14 -> 5.11).  Only where what this function RETURNS for the tail's classes is
            # final - the (g, v) form, whose gradients nothing but AccumulateGrad touches, and only while those leaves hold no gradient yet (an
            # accumulating AccumulateGrad would read them on the caller's stream): with plain weights WeightNorm's own backward reads d w right away.
            # Measured and rejected, twice: holding the tail back until the prosody encoder's chain of small launches is through (they take 24-45 us each beside
            # 216 one-per-CU weight-gradient workgroups): 6.20 against 6.08 ms/step with ~45 launches in that chain, 6.00 against 5.85 with ~15 - the tail
            # (0.95 ms) is then the critical path behind them (profiles/r06_config5_timeline_hold.txt).
            if dcond is not None and GV is not None and TUNE["tail_aside"] and _engine_callbacks_ok() and _leaf_grads_unset():
                tail_fns = []
            (tail_fns.append if tail_fns is not None else (lambda f: f()))(lambda: [grp.launch_segment(0) for grp in (gk, gp, g1)])
            (tail_fns.append if tail_fns is not None else (lambda f: f()))(sum_tail)
        # weight-norm backward of the (g, v) form (Modules.py:766, 818, 825): d g, d v from d w; the classes whose d w comes from the deferrable
        # 1x1 groups are queued behind them
        GVgrad = {}
        if GV is not None:
            # one launch per destination (now / the deferred tail / the weight-gradient stream) for all classes that go there: as four dependent launches
            # (5 - 40 us each) they were the last ~85 us of the decoder's chain in front of the gradient norm
            wn_now, wn_pending, wn_keep = [], [], []
            for k in WN_KEYS:
                g_, v_ = GV[k]
                if v_.numel() == 0:
                    GVgrad[k] = (torch.zeros_like(g_), torch.zeros_like(v_))
                    continue
                cols = v_.shape[-1] * v_.shape[-2]
                rows_ = v_.numel() // cols
                dv, dg = RET["v" + k[1:]], RET["g" + k[1:]]
                GVgrad[k] = (dg, dv)
                j = WnBwdJob()
                j.dw, j.v, j.g, j.inv_norm, j.dv, j.dg = G[k].data_ptr(), v_.data_ptr(), g_.data_ptr(), prep.inv[k].data_ptr(), dv.data_ptr(), dg.data_ptr()
                j.rows, j.cols = rows_, cols
                wn_keep.append((g_, v_, dv, dg, prep.inv[k], G[k]))
                (wn_pending if (TAIL["defer"] and halves == 1 and k != "w_in") else wn_now).append(j)

            def wn_run(jobs, keep=wn_keep):
                if jobs:                                        # (4.70 / 4.72 / 4.70 against 4.75 / 4.70 / 4.76 ms/step for one launch per class, three alternating pairs)
                    L.glowtts_weightnorm_bwd_multi.argtypes = [c_void_p, c_int, c_void_p]
                    _lib.check(L.glowtts_weightnorm_bwd_multi((WnBwdJob * len(jobs))(*jobs), len(jobs), _lib.stream()), "glowtts_weightnorm_bwd_multi")
            if wn_pending:
                TAIL["pending"].append(lambda jobs=wn_pending: wn_run(jobs))
            if tail_fns is not None:
                tail_fns.append(lambda jobs=wn_now: wn_run(jobs))
            else:
                wn_run(wn_now)

        def param_sums():
            stamp("dec_wgrads_done")
            _lib.check(L.glowtts_colsum_batched(scratch.data_ptr(), d_an.data_ptr(), nblk_all, 2 * C + 16, F_, nscr, 2 * C + 16, _lib.stream()),
                       "glowtts_colsum_batched")
            # + the log-determinant terms of the parameters (Modules.py:694, 747): logdet_b += (sum logs + logdet(W) C/4) * len_b
            _lib.check(L.glowtts_decoder_param_grads(d_an.data_ptr(), dld.data_ptr(), rowmask.data_ptr(), prep.winfo.data_ptr(), G["an_logs"].data_ptr(),
                                                     G["an_bias"].data_ptr(), G["inv_w"].data_ptr(), F_, B, T + 2 * ROW_PAD, C, _lib.stream()),
                       "glowtts_decoder_param_grads")
        if tail_fns is None:
            param_sums()
        else:
            tail_fns.append(param_sums)
            # everything the tail reads or writes stays alive until the join has been queued on the caller's stream (blocks handed back to the caching
            # allocator before that could be given out on that stream while the tail still uses them)
            keep = [(buf, dins, dskip, dh0, dhn, douts, douts_bf, G, RET, GV, scratch, d_an, dld, tail_part, gk, gp, g1, prep, rowmask), tail_fns]

            def release():
                if keep[1] is None:
                    return
                fns, keep[1] = keep[1], None
                side.wait_stream(torch.cuda.current_stream(dev))          # (behind everything the releasing stream holds: the chain, or the held-back pass)
                with torch.cuda.stream(side):
                    for fn in fns:
                        fn()

            def at_end():
                release()
                main.wait_stream(side)
                keep[0] = None
            release()
            torch.autograd.Variable._execution_engine.queue_callback(at_end)
        dmel = unsqueeze_rows(cfg, dx, ctx.lengths, B, Tm) if ctx.want_dmel else None
        if dcond is not None:
            # (one launch; a non-finite gate gradient poisons its accumulator and reads NaN here - ADVICE r5)
            acc, dcond = dcond.contiguous(), torch.empty(dcond.shape, device=dcond.device)
            _L().glowtts_fx_to_float.argtypes = [c_void_p, c_void_p, ctypes.c_int64, c_void_p]
            _lib.check(_L().glowtts_fx_to_float(acc.data_ptr(), dcond.data_ptr(), acc.numel(), _lib.stream()), "glowtts_fx_to_float")
        dpw = dpb = None
        if ctx.prow is not None:
            # Pitch_l conv (Modules.py:846-852, 867-869): bias gradient = the conditioning gradient summed over utterances; weight gradient =
            # sum_r d pre[r][n] * pitch[r][j] with d pre taken BEFORE the WaveNet dropout's keep mask (the pitch term joins behind the
            # dropout, Modules.py:861-869), accumulated by the gate-derivative epilogues into the rows behind the utterances'
            dpb = dcond[:B].sum(0).view(F_, Lw, 2 * H)
            dpw = dcond[B:].view(npit, F_, Lw, 2 * H).permute(1, 2, 3, 0).contiguous()
            dcond = dcond[:B]
        if GV is not None:
            out = []
            for k in WEIGHT_KEYS_GV:
                wk = "w" + k[1:]
                if k[0] in "gv" and wk in GVgrad:
                    out.append(GVgrad[wk][0 if k[0] == "g" else 1])
                else:
                    out.append(G[k].view_as(W[k]))
            return (None, dmel, None, dcond, None, None, dpw, dpb) + tuple(out)
        return (None, dmel, None, dcond, None, None, dpw, dpb) + tuple(G[k].view_as(W[k]) for k in WEIGHT_KEYS)


def _wn(g, v):
    """Old-style torch weight_norm (Modules.py:766,818,825): w = g * v / ||v||, norm over (in, k) per output channel."""
    return g * v / v.flatten(-2).norm(dim=-1).unsqueeze(-1).unsqueeze(-1)


def stack_decoder_weights(P, cfg, prefix="layer_Dict.Decoder.layer_Dict.Flows"):
    """Builds the stacked effective weights from reference-named parameters (name -> tensor).
    Differentiable torch code on small tensors (weight-norm stays in torch so its autograd is free);
    returns the tuple in WEIGHT_KEYS order."""
    F_, L = cfg.F, cfg.L
    fl = lambda f: f"{prefix}.{f}.layers"
    wn = lambda f, name: (P[f"{fl(f)}.2.layer_Dict.{name}.weight_g"], P[f"{fl(f)}.2.layer_Dict.{name}.weight_v"],
                          P[f"{fl(f)}.2.layer_Dict.{name}.bias"])
    an_logs = torch.stack([P[f"{fl(f)}.0.logs"].reshape(-1) for f in range(F_)])
    an_bias = torch.stack([P[f"{fl(f)}.0.bias"].reshape(-1) for f in range(F_)])
    inv_w = torch.stack([P[f"{fl(f)}.1.weight"] for f in range(F_)])
    st = [wn(f, "Start") for f in range(F_)]
    w_start = _wn(torch.stack([t[0] for t in st]), torch.stack([t[1] for t in st]))
    b_start = torch.stack([t[2] for t in st])
    ins = [[wn(f, f"WaveNet.layer_Dict.In_{l}") for l in range(L)] for f in range(F_)]
    w_in = _wn(torch.stack([torch.stack([t[0] for t in row]) for row in ins]), torch.stack([torch.stack([t[1] for t in row]) for row in ins]))
    b_in = torch.stack([torch.stack([t[2] for t in row]) for row in ins])
    rs = [[wn(f, f"WaveNet.layer_Dict.Res_Skip_{l}") for l in range(L)] for f in range(F_)]
    if L > 1:
        w_rs = _wn(torch.stack([torch.stack([t[0] for t in row[:-1]]) for row in rs]), torch.stack([torch.stack([t[1] for t in row[:-1]]) for row in rs]))
        b_rs = torch.stack([torch.stack([t[2] for t in row[:-1]]) for row in rs])
    else:
        dev = w_in.device
        w_rs, b_rs = torch.zeros(F_, 0, 2 * cfg.H, cfg.H, 1, device=dev), torch.zeros(F_, 0, 2 * cfg.H, device=dev)
    w_rs_last = _wn(torch.stack([row[-1][0] for row in rs]), torch.stack([row[-1][1] for row in rs]))
    b_rs_last = torch.stack([row[-1][2] for row in rs])
    w_end = torch.stack([P[f"{fl(f)}.2.layer_Dict.End.weight"] for f in range(F_)])
    b_end = torch.stack([P[f"{fl(f)}.2.layer_Dict.End.bias"] for f in range(F_)])
    return (an_logs, an_bias, inv_w, w_start, b_start, w_in, b_in, w_rs, b_rs, w_rs_last, b_rs_last, w_end, b_end)


# --------------------------------------------------------------------------------------------------------------------
# Stacked parameter storage.  The decoder launches take every flow's / layer's weights as ONE tensor.  Building that with
# torch.stack costs ~100 copy launches per step (and a split in the backward); instead the reference-named leaf Parameters are
# re-pointed once to be views into one flat tensor per weight class (the flat-parameter technique of data-parallel wrappers):
# the stacked tensor then exists without any copy, and the backward hands every leaf a view of the stacked gradient.
# --------------------------------------------------------------------------------------------------------------------
class _StackedLeaves(torch.autograd.Function):
    @staticmethod
    def forward(ctx, flat, *leaves):
        ctx.n, ctx.leaf_shape = len(leaves), leaves[0].shape
        return flat.detach()                              # same storage as the leaves, fresh autograd identity

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous().view(ctx.n, *ctx.leaf_shape)
        return (None,) + tuple(g.unbind(0))               # views: AccumulateGrad keeps them as the leaves' .grad without a copy


class LeafStack:
    """A list of equally shaped leaf tensors (Parameters) kept as views of one flat tensor of shape lead + leaf_shape."""

    def __init__(self, leaves, lead):
        self.leaves, self.lead = list(leaves), tuple(lead)
        assert len(self.leaves) == int(torch.tensor(self.lead).prod()) and len({tuple(t.shape) for t in self.leaves}) == 1
        self.flat = None

    def _aliased(self):
        f = self.flat
        if f is None or f.device != self.leaves[0].device:
            return False
        base, step = f.data_ptr(), self.leaves[0].numel() * f.element_size()
        return all(t.data_ptr() == base + i * step and t.is_contiguous() for i, t in enumerate(self.leaves))

    def tensor(self):
        if not self._aliased():                           # first use, or the leaves were moved (model.to(...)): one-time re-pointing
            if self.leaves[0].is_cuda and torch.cuda.is_current_stream_capturing():
                raise _lib.GlowTTSHipError("run one eager step before capturing a hipGraph (parameter storage not flattened yet)")
            with torch.no_grad():
                flat = torch.stack([t.detach() for t in self.leaves]).contiguous()
                for i, t in enumerate(self.leaves):
                    t.data = flat[i]
                self.flat = flat.view(self.lead + tuple(self.leaves[0].shape))
        if torch.is_grad_enabled() and any(t.requires_grad for t in self.leaves):
            return _StackedLeaves.apply(self.flat, *self.leaves)
        return self.flat


class WeightNorm(torch.autograd.Function):
    """w = g * v / ||v||  (old-style torch weight_norm, norm over (in, k) per output channel) on stacked tensors [..., O, I, k]."""

    @staticmethod
    def forward(ctx, g, v, tail=False):
        ctx.tail = tail                                  # its backward may be queued behind the deferred 1x1 weight gradients (TAIL)
        v, g = v.contiguous(), g.contiguous()
        rows, cols = v.numel() // (v.shape[-1] * v.shape[-2]), v.shape[-1] * v.shape[-2]
        w, inv = torch.empty_like(v), torch.empty(rows, device=v.device)
        L = _L()
        _lib.check(L.glowtts_weightnorm_fwd(v.data_ptr(), g.data_ptr(), w.data_ptr(), inv.data_ptr(), rows, cols, _lib.stream()), "glowtts_weightnorm_fwd")
        ctx.save_for_backward(g, v, inv)
        ctx.dims = (rows, cols)
        return w

    @staticmethod
    def backward(ctx, dw):
        g, v, inv = ctx.saved_tensors
        rows, cols = ctx.dims
        dw = dw.contiguous()
        dv, dg = torch.empty_like(v), torch.empty_like(g)
        run = lambda: _lib.check(_L().glowtts_weightnorm_bwd(dw.data_ptr(), v.data_ptr(), g.data_ptr(), inv.data_ptr(), dv.data_ptr(), dg.data_ptr(),
                                                             rows, cols, _lib.stream()), "glowtts_weightnorm_bwd")
        if ctx.tail and TAIL["defer"]:
            TAIL["pending"].append(run)                  # dw is filled by the deferred weight-gradient launches queued before this one
        else:
            run()
        return dg, dv, None


class CondLinear(torch.autograd.Function):
    """cond [B, N] = sum over kinds of  vec_k [B, D_k] @ (g_k v_k / ||v_k||)^T + bias_k   for the N = F * L * 2H weight-normalised 1x1 conditioning
    convs of the decoder (Modules.py:832-845, 863-866).  apply(g_0, v_0, bias_0, vec_0, g_1, ...): one forward launch per kind (the second accumulates),
    one backward launch per kind producing d g, d v, d bias and - when the vectors are differentiable (LUT / prosody encoder) - their gradient by a
    deterministic two-stage sum."""

    @staticmethod
    def forward(ctx, *args):
        L = _L()
        kinds = [args[i:i + 4] for i in range(0, len(args), 4)]
        B = kinds[0][3].shape[0]
        N = kinds[0][1].numel() // kinds[0][1].shape[-2]
        dev = kinds[0][3].device
        out = torch.empty(B, N, device=dev)
        keep = []
        for i, (g, v, b, vec) in enumerate(kinds):
            g, v, b, vec = g.detach().contiguous(), v.detach().contiguous(), b.detach().contiguous(), vec.detach().contiguous()
            D = v.shape[-2]
            assert v.shape[-1] == 1 and v.numel() == N * D and g.numel() == N and b.numel() == N and tuple(vec.shape) == (B, D)
            inv = torch.empty(N, device=dev)
            _lib.check(L.glowtts_cond_linear_fwd(v.data_ptr(), g.data_ptr(), b.data_ptr(), vec.data_ptr(), out.data_ptr(), inv.data_ptr(), N, D, B, int(i > 0),
                                                 _lib.stream()), "glowtts_cond_linear_fwd")
            keep += [g, v, vec, inv]
        ctx.save_for_backward(*keep)
        ctx.shapes = [(tuple(k[0].shape), tuple(k[1].shape), tuple(k[2].shape)) for k in kinds]
        return out

    @staticmethod
    def backward(ctx, dout):
        L = _L()
        keep = ctx.saved_tensors
        dout = dout.contiguous()
        B, N = dout.shape
        grads = []
        for i, (gs, vs, bs) in enumerate(ctx.shapes):
            g, v, vec, inv = keep[4 * i:4 * i + 4]
            D = vec.shape[1]
            dv, dg, db = torch.empty_like(v), torch.empty_like(g), torch.empty(N, device=dout.device)
            want_vec = ctx.needs_input_grad[4 * i + 3]
            dvec = torch.empty_like(vec) if want_vec else None
            scratch = torch.empty(L.glowtts_cond_linear_bwd_scratch_floats(N, D, B), device=dout.device)
            _lib.check(L.glowtts_cond_linear_bwd(dout.data_ptr(), N, v.data_ptr(), g.data_ptr(), inv.data_ptr(), vec.data_ptr(), dv.data_ptr(), dg.data_ptr(),
                                                 db.data_ptr(), dvec.data_ptr() if want_vec else None, scratch.data_ptr(), N, D, B, _lib.stream()),
                       "glowtts_cond_linear_bwd")
            grads += [dg.view(gs), dv.view(vs), db.view(bs), dvec]
        return tuple(grads)


class DecoderStacks:
    """Leaf stacks of the decoder's parameters, built once per model from the reference-named parameter dict."""

    def __init__(self, P, cfg, prefix="layer_Dict.Decoder.layer_Dict.Flows"):
        F_, L = cfg.F, cfg.L
        self.cfg = cfg
        fl = lambda f: f"{prefix}.{f}.layers"
        S = {}
        S["an_logs"] = LeafStack([P[f"{fl(f)}.0.logs"] for f in range(F_)], (F_,))
        S["an_bias"] = LeafStack([P[f"{fl(f)}.0.bias"] for f in range(F_)], (F_,))
        S["inv_w"] = LeafStack([P[f"{fl(f)}.1.weight"] for f in range(F_)], (F_,))

        def wn3(tag, names, lead):
            for part, key in (("g", "weight_g"), ("v", "weight_v"), ("b", "bias")):
                S[f"{tag}_{part}"] = LeafStack([P[f"{n}.{key}"] for n in names], lead)
        wn3("start", [f"{fl(f)}.2.layer_Dict.Start" for f in range(F_)], (F_,))
        wn3("in", [f"{fl(f)}.2.layer_Dict.WaveNet.layer_Dict.In_{l}" for f in range(F_) for l in range(L)], (F_, L))
        if L > 1:
            wn3("rs", [f"{fl(f)}.2.layer_Dict.WaveNet.layer_Dict.Res_Skip_{l}" for f in range(F_) for l in range(L - 1)], (F_, L - 1))
        wn3("rsl", [f"{fl(f)}.2.layer_Dict.WaveNet.layer_Dict.Res_Skip_{L - 1}" for f in range(F_)], (F_,))
        S["end_w"] = LeafStack([P[f"{fl(f)}.2.layer_Dict.End.weight"] for f in range(F_)], (F_,))
        S["end_b"] = LeafStack([P[f"{fl(f)}.2.layer_Dict.End.bias"] for f in range(F_)], (F_,))
        for kind in ("Speaker", "Prosody", "Pitch"):
            if f"{fl(0)}.2.layer_Dict.WaveNet.layer_Dict.{kind}_0.weight_v" in P:
                wn3(kind, [f"{fl(f)}.2.layer_Dict.WaveNet.layer_Dict.{kind}_{l}" for f in range(F_) for l in range(L)], (F_ * L,))
        self.S = S

    def tail_leaves(self):
        """Parameters whose gradients are produced by the deferrable tail of the backward (see TAIL)."""
        return [p for k in TAIL_STACKS if k in self.S for p in self.S[k].leaves]

    def weights(self, gv=False):
        """The stacked effective weights in WEIGHT_KEYS order (differentiable w.r.t. the leaves); gv = True: WEIGHT_KEYS_GV order - the
        weight-normalised convs as their (weight_g, weight_v) stacks, for DecoderFunction's one-launch weight preparation."""
        S, cfg = self.S, self.cfg
        F_, L = cfg.F, cfg.L
        t = lambda k: S[k].tensor()
        if gv:
            if L > 1:
                rs = (t("rs_g"), t("rs_v"), t("rs_b"))
            else:
                dev = S["in_v"].leaves[0].device
                rs = (torch.zeros(F_, 0, 2 * cfg.H, 1, 1, device=dev), torch.zeros(F_, 0, 2 * cfg.H, cfg.H, 1, device=dev), torch.zeros(F_, 0, 2 * cfg.H, device=dev))
            return (t("an_logs").view(F_, -1), t("an_bias").view(F_, -1), t("inv_w"), t("start_g"), t("start_v"), t("start_b"),
                    t("in_g"), t("in_v"), t("in_b")) + rs + (t("rsl_g"), t("rsl_v"), t("rsl_b"), t("end_w"), t("end_b"))
        wn = lambda tag: WeightNorm.apply(t(tag + "_g"), t(tag + "_v"), (tag + "_v") in TAIL_STACKS)
        if L > 1:
            w_rs, b_rs = wn("rs"), t("rs_b")
        else:
            dev = S["in_v"].leaves[0].device
            w_rs, b_rs = torch.zeros(F_, 0, 2 * cfg.H, cfg.H, 1, device=dev), torch.zeros(F_, 0, 2 * cfg.H, device=dev)
        return (t("an_logs").view(F_, -1), t("an_bias").view(F_, -1), t("inv_w"), wn("start"), t("start_b"), wn("in"), t("in_b"),
                w_rs, b_rs, wn("rsl"), t("rsl_b"), t("end_w"), t("end_b"))

    def pitch_weights(self):
        """(pitch_w [F, L, 2H, ns], pitch_b [F, L, 2H]) of the GR-mode Pitch_l convs (Modules.py:846-852), or (None, None)."""
        if "Pitch_v" not in self.S:
            return None, None
        cfg = self.cfg
        w = WeightNorm.apply(self.S["Pitch_g"].tensor(), self.S["Pitch_v"].tensor()).squeeze(-1)
        return w.view(cfg.F, cfg.L, 2 * cfg.H, -1), self.S["Pitch_b"].tensor().view(cfg.F, cfg.L, 2 * cfg.H)

    def conditioning(self, speakers=None, prosodies=None):
        """cond[b, f, l, :] = Speaker_l(spk_b) + Prosody_l(pro_b)  (Modules.py:863-866): one launch per kind straight from the weight-norm pairs
        (CondLinear; csrc/cond_ops.hip), or - shapes that kernel does not take - weight norm + one batched matmul each."""
        kinds = [(kind, vec) for kind, vec in (("Speaker", speakers), ("Prosody", prosodies)) if vec is not None]
        if not kinds:
            return None
        # (the kernels' own bound - D in {128, 256, 384, 512}, B <= 64 AND their LDS tiles within 160 KiB: D = 512 stops at B = 40 - asked of the library,
        #  so that a shape it rejects takes the matmul path below instead of raising in the forward: ADVICE r5)
        n_out = self.cfg.F * self.cfg.L * 2 * self.cfg.H
        if TUNE["cond_hip"] and all(vec.is_cuda and vec.dtype == torch.float32 and vec.dim() == 2 and vec.shape[1] == self.S[kind + "_v"].tensor().shape[-2]
                                    and _lib.lib().glowtts_cond_linear_supported(n_out, int(vec.shape[1]), int(vec.shape[0])) for kind, vec in kinds):
            args = []
            for kind, vec in kinds:
                args += [self.S[kind + "_g"].tensor(), self.S[kind + "_v"].tensor(), self.S[kind + "_b"].tensor(), vec]
            cond = CondLinear.apply(*args)
            return cond.view(cond.shape[0], self.cfg.F, self.cfg.L, 2 * self.cfg.H)
        cond = None
        for kind, vec in kinds:
            w = WeightNorm.apply(self.S[kind + "_g"].tensor(), self.S[kind + "_v"].tensor()).squeeze(-1)
            c = torch.einsum("nod,bd->bno", w, vec) + self.S[kind + "_b"].tensor()
            cond = c if cond is None else cond + c
        if cond is None:
            return None
        return cond.view(cond.shape[0], self.cfg.F, self.cfg.L, 2 * self.cfg.H).contiguous()


def stack_cond_weights(P, cfg, kind, prefix="layer_Dict.Decoder.layer_Dict.Flows"):
    """Speaker_l / Prosody_l conditioning 1x1 convs (Modules.py:832-845) stacked as [F*L, 2H, D] (+ bias [F*L, 2H])."""
    gs, vs, bs = [], [], []
    for f in range(cfg.F):
        for l in range(cfg.L):
            q = f"{prefix}.{f}.layers.2.layer_Dict.WaveNet.layer_Dict.{kind}_{l}"
            gs.append(P[q + ".weight_g"]); vs.append(P[q + ".weight_v"]); bs.append(P[q + ".bias"])
    w = _wn(torch.stack(gs), torch.stack(vs)).squeeze(-1)
    return w, torch.stack(bs)


def conditioning(P, cfg, speakers=None, prosodies=None):
    """cond[b, f, l, :] = Speaker_l(spk_b) + Prosody_l(pro_b)  (Modules.py:863-866), one batched matmul each."""
    cond = None
    for kind, vec in (("Speaker", speakers), ("Prosody", prosodies)):
        if vec is None:
            continue
        w, b = stack_cond_weights(P, cfg, kind)
        c = torch.einsum("nod,bd->bno", w, vec) + b
        cond = c if cond is None else cond + c
    if cond is None:
        return None
    return cond.view(cond.shape[0], cfg.F, cfg.L, 2 * cfg.H).contiguous()
