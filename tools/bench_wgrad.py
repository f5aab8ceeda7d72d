"""The decoder's weight-gradient groups alone, staged kernel against the DMA kernel: the 5-tap group (36 problems of m = 384, ca = 192 over 12 928 rows, bf16
operands) and the one-tap group (84 Res_Skip + 12 End + 12 Start problems).   usage (GPU box): python tools/bench_wgrad.py"""
import ctypes, sys, torch
sys.path.insert(0, "/root/repo")
from glow_tts_amd import decoder as D, ops, _lib

R, M, CA, TAPS, NJ = 12928, 384, 192, 5, 36
L = D._L()
io = ops.WIO_DY_BF16 | ops.WIO_X_BF16
dys = [(torch.randn(R, M, device="cuda") * 0.5).to(torch.bfloat16) for _ in range(NJ)]
xs = [torch.randn(R, CA, device="cuda").to(torch.bfloat16) for _ in range(NJ)]
dws = [torch.empty(M, CA, TAPS, device="cuda") for _ in range(NJ)]
dbs = [torch.empty(M, device="cuda") for _ in range(NJ)]


def group(dma):
    g = D.WgradGroup(R, TAPS, ops.BF16, io_flags=io, tag=f"b{int(dma)}")
    for dy, x, dw, db in zip(dys, xs, dws, dbs):
        g.add(dy.data_ptr(), M, M, x.data_ptr(), CA, CA, dw.data_ptr(), db.data_ptr(), perm=ops.PERM_PAIR, perm_h=M // 2)
    g._dma = dma
    g.end_segment(); g.upload(torch.device("cuda"))
    return g


def run(g, n=20):
    start, nj, tiles = g.segments[0]
    flags = g.io_flags | ops.WIO_WIDE | (ops.WIO_DMA if g._dma else 0)
    def once():
        _lib.check(L.glowtts_wgrad_grouped_io(g.table.data_ptr(), nj, tiles, R, TAPS, 2, ops.APRO_NONE, ops.BF16, 1, 0, flags, _lib.stream()), "wgrad")
    for _ in range(3):
        once()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        once()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


flops = 2.0 * NJ * R * M * CA * TAPS
gs, gd = group(False), group(True)
for rep in range(2):
    t = run(gs); print(f"staged kernel      {t:8.1f} us  {flops / t / 1e6:7.1f} TFLOP/s  tiles {gs.segments[0][2]}")
    t = run(gd); print(f"dma kernel         {t:8.1f} us  {flops / t / 1e6:7.1f} TFLOP/s  tiles {gd.segments[0][2]}")


# the one-tap group of the decoder: 84 Res_Skip problems (192 x 192), 12 End (2 x 80 PAIR-packed in 192 columns, x 192), 12 Start (192 x 80)
jobs1 = [(192, 192, ops.PERM_NONE, 0)] * 84 + [(192, 192, ops.PERM_PAIR, 80)] * 12 + [(192, 80, ops.PERM_NONE, 0)] * 12
dy1 = [(torch.randn(R, m, device="cuda") * 0.5).to(torch.bfloat16) for m, _, _, _ in jobs1]
x1 = [torch.randn(R, ca, device="cuda").to(torch.bfloat16) for _, ca, _, _ in jobs1]
dw1 = [torch.empty(m, ca, 1, device="cuda") for m, ca, _, _ in jobs1]
db1 = [torch.empty(m, device="cuda") for m, _, _, _ in jobs1]


def group1(dma):
    g = D.WgradGroup(R, 1, ops.BF16, io_flags=io, tag=f"k1{int(dma)}")
    for dy, x, dw, db, (m, ca, perm, ph) in zip(dy1, x1, dw1, db1, jobs1):
        g.add(dy.data_ptr(), m, m, x.data_ptr(), ca, ca, dw.data_ptr(), db.data_ptr(), perm=perm, perm_h=ph)
    assert g._wide and g._dma
    g._dma = dma
    g.end_segment(); g.upload(torch.device("cuda"))
    return g


def run1(g, n=20):
    start, nj, tiles = g.segments[0]
    flags = g.io_flags | ops.WIO_WIDE | (ops.WIO_DMA if g._dma else 0)
    once = lambda: _lib.check(L.glowtts_wgrad_grouped_io(g.table.data_ptr(), nj, tiles, R, 1, 0, ops.APRO_NONE, ops.BF16, 1, 0, flags, _lib.stream()), "wgrad")
    for _ in range(3):
        once()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        once()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for dma in (False, True, False, True):
    g = group1(dma)
    print(f"one-tap group, {'dma   ' if dma else 'staged'} kernel {run1(g):8.1f} us  tiles {g.segments[0][2]}")
