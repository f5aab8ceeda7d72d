"""Thin Python wrappers over the C ABI (include/glowtts_hip.h): argument marshalling only.
Every function launches asynchronously on torch's current HIP stream."""
import ctypes

import torch

from . import _lib
from ._lib import c_int, c_void_p

F32, BF16 = 0, 1
PERM_NONE, PERM_PAIR = 0, 1
APRO_NONE, APRO_PAIRMUL, APRO_SQNEG = 0, 1, 2
EPI_LINEAR, EPI_GATE, EPI_RESSKIP, EPI_COUPLE, EPI_DGATE = 0, 1, 2, 3, 4
IO_A_BF16, IO_IN0_BF16, IO_OUT0_BF16 = 1, 2, 4      # glowtts_conv_args.io_flags
WIO_DY_BF16, WIO_X_BF16, WIO_WIDE, WIO_DMA = 1, 2, 4, 8           # glowtts_wgrad_args.io_flags (DMA: grouped launches only)
F_BIAS, F_RELU, F_ADD_IN0, F_MASK, F_ACCUM, F_FIRST, F_LAST, F_REVERSE, F_COLMASK, F_DROPOUT = 1, 2, 4, 8, 16, 32, 64, 128, 256, 512
F_GATE_IN0 = 2048


class ConvArgs(ctypes.Structure):
    """Mirror of `glowtts_conv_args` (include/glowtts_hip.h)."""
    _fields_ = [
        ("a", c_void_p), ("lda", ctypes.c_int64),
        ("a2", c_void_p), ("lda2", ctypes.c_int64),
        ("ca1", c_int), ("ca", c_int),
        ("apro", c_int),
        ("rows", c_int),
        ("w", c_void_p),
        ("n", c_int), ("npad", c_int), ("kchunks", c_int),
        ("taps", c_int), ("pad", c_int),
        ("precision", c_int),
        ("epi", c_int), ("flags", c_int),
        ("h", c_int),
        ("rows_per_utt", c_int),
        ("bias", c_void_p),
        ("rowmask", c_void_p),
        ("cond", c_void_p), ("ldcond", ctypes.c_int64),
        ("out0", c_void_p), ("ld0", ctypes.c_int64),
        ("out1", c_void_p), ("ld1", ctypes.c_int64),
        ("in0", c_void_p), ("ldi0", ctypes.c_int64),
        ("batch", c_int),
        ("a_bstride", ctypes.c_int64), ("w_bstride", ctypes.c_int64), ("bias_bstride", ctypes.c_int64),
        ("out_bstride", ctypes.c_int64), ("mask_bstride", ctypes.c_int64),
        ("ncols_valid", c_void_p),
        ("seed", ctypes.c_uint32), ("drop_p", ctypes.c_float),
        ("seed_ptr", c_void_p),
        ("io_flags", c_int),
    ]


class PackedWeight:
    """A conv weight in MFMA tile order (glowtts_pack_weight)."""
    __slots__ = ("data", "npad", "kchunks", "taps", "precision", "n")

    def __init__(self, data, npad, kchunks, taps, precision, n):
        self.data, self.npad, self.kchunks, self.taps, self.precision, self.n = data, npad, kchunks, taps, precision, n


def pack_weight(w, transpose=False, perm=PERM_NONE, perm_h=0, precision=BF16):
    """w: fp32 device tensor [O, I, taps] (torch Conv1d layout) -> PackedWeight."""
    L = _lib.lib()
    w = w.contiguous()
    O, I, taps = w.shape
    npad, kch = c_int(0), c_int(0)
    _lib.check(L.glowtts_pack_weight(None, O, I, taps, int(transpose), perm, perm_h, precision, None,
                                     ctypes.byref(npad), ctypes.byref(kch), None), "glowtts_pack_weight(size)")
    buf = torch.empty(taps * kch.value * npad.value * 64, dtype=torch.uint8, device=w.device)
    _lib.check(L.glowtts_pack_weight(_lib.ptr(w), O, I, taps, int(transpose), perm, perm_h, precision, _lib.ptr(buf),
                                     None, None, _lib.stream()), "glowtts_pack_weight")
    n = I if transpose else O
    return PackedWeight(buf, npad.value, kch.value, taps, precision, n)


class PackJob(ctypes.Structure):
    """Mirror of `glowtts_pack_job` (include/glowtts_hip.h)."""
    _fields_ = [("w", c_void_p), ("packed", c_void_p)] + [(n, c_int) for n in
                ("O", "I", "taps", "transpose", "perm", "perm_h", "N", "K", "npad", "kchunks", "block0", "reserved")]


class PackSet:
    """Conv weights of different shapes packed by ONE launch (glowtts_pack_weight_multi) into one buffer.
    items: [(key, fp32 weight [O, I, taps], transpose)].  The device job table is built once; `run()` re-packs the current values
    (call it once per forward: the optimizer changes them in place).  `matches(items)` tells whether the pointers still hold."""

    def __init__(self, items, precision=BF16):
        L = _lib.lib()
        L.glowtts_pack_job_init.argtypes = [ctypes.POINTER(PackJob), c_void_p] + [c_int] * 7 + [c_void_p, c_int,
                                            ctypes.POINTER(c_int), ctypes.POINTER(ctypes.c_int64)]
        L.glowtts_pack_weight_multi.argtypes = [c_void_p, c_int, c_int, c_int, c_void_p]
        assert items
        dev = items[0][1].device
        n = len(items)
        jobs = (PackJob * n)()
        blocks, nbytes = c_int(0), ctypes.c_int64(0)
        offs, total = [], 0
        for i, (_, w, tr) in enumerate(items):
            assert w.dim() == 3 and w.is_contiguous() and w.dtype == torch.float32
            O, I, k = w.shape
            _lib.check(L.glowtts_pack_job_init(ctypes.byref(jobs[i]), None, O, I, k, int(tr), PERM_NONE, 0, precision, None, 0,
                                               ctypes.byref(blocks), ctypes.byref(nbytes)), "glowtts_pack_job_init(size)")
            offs.append(total)
            total += (nbytes.value + 255) // 256 * 256
        self.data = torch.empty(total, dtype=torch.uint8, device=dev)
        self.packed, b0 = {}, 0
        for i, (key, w, tr) in enumerate(items):
            O, I, k = w.shape
            _lib.check(L.glowtts_pack_job_init(ctypes.byref(jobs[i]), w.data_ptr(), O, I, k, int(tr), PERM_NONE, 0, precision,
                                               self.data.data_ptr() + offs[i], b0, ctypes.byref(blocks), ctypes.byref(nbytes)), "glowtts_pack_job_init")
            b0 += blocks.value
            self.packed[(key, bool(tr))] = PackedWeight(self.data[offs[i]:offs[i] + nbytes.value], jobs[i].npad, jobs[i].kchunks, k, precision,
                                                        I if tr else O)
        self.table = torch.frombuffer(bytearray(jobs), dtype=torch.uint8).to(dev)
        self.njobs, self.blocks, self.precision = n, b0, precision
        self.sig = self.signature(items)

    @staticmethod
    def signature(items):
        return tuple((k, w.data_ptr(), tuple(w.shape), bool(tr)) for k, w, tr in items)

    def run(self):
        _lib.check(_lib.lib().glowtts_pack_weight_multi(self.table.data_ptr(), self.njobs, self.blocks, self.precision, _lib.stream()),
                   "glowtts_pack_weight_multi")

    def get(self, key):
        """(forward pack, transposed pack or None)"""
        return self.packed[(key, False)], self.packed.get((key, True))


def conv_cl(a, pw, ca, rows, *, lda=None, a2=None, lda2=0, ca1=0, apro=APRO_NONE, pad=0, epi=EPI_LINEAR, flags=0,
            n=None, h=0, rows_per_utt=1, bias=None, rowmask=None, cond=None, ldcond=0,
            out0=None, ld0=0, out1=None, ld1=0, in0=None, ldi0=0, out0_off=0, a_off=0, drop_p=0.0, seed=0, seed_t=None, io_flags=0):
    """Launches glowtts_conv_cl.  Tensors are fp32 device tensors; *_off are element offsets into them
    (to address a channel sub-range of a wider row).  io_flags (IO_*): a / in0 / out0 are bf16 tensors instead."""
    args = ConvArgs()
    args.a = a.data_ptr() + 4 * a_off
    args.lda = lda if lda is not None else a.shape[-1]
    args.a2 = a2.data_ptr() if a2 is not None else None
    args.lda2 = lda2
    args.ca1, args.ca, args.apro, args.rows = ca1, ca, apro, rows
    args.w = pw.data.data_ptr()
    args.n = n if n is not None else pw.n
    args.npad, args.kchunks, args.taps, args.pad, args.precision = pw.npad, pw.kchunks, pw.taps, pad, pw.precision
    args.epi, args.flags, args.h, args.rows_per_utt = epi, flags, h, rows_per_utt
    args.bias = bias.data_ptr() if bias is not None else None
    args.rowmask = rowmask.data_ptr() if rowmask is not None else None
    args.cond = cond.data_ptr() if cond is not None else None
    args.ldcond = ldcond
    args.out0 = out0.data_ptr() + 4 * out0_off
    args.ld0 = ld0
    args.out1 = out1.data_ptr() if out1 is not None else None
    args.ld1 = ld1
    args.in0 = in0.data_ptr() if in0 is not None else None
    args.ldi0 = ldi0
    args.drop_p, args.seed = float(drop_p), int(seed) & 0xFFFFFFFF
    args.seed_ptr = seed_t.data_ptr() if seed_t is not None else None
    args.io_flags = io_flags
    assert not (io_flags and (a_off or out0_off))
    _lib.check(_lib.lib().glowtts_conv_cl(ctypes.byref(args), _lib.stream()), "glowtts_conv_cl")
