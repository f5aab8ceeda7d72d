"""How fast does a chain of one-workgroup kernels on a second stream advance beside back-to-back fused coupling launches (249 one-per-CU workgroups)?
Stream A: N fused forward launches (B utterances).  Stream B: a chain of glowtts_debug_stamp launches (one wave each, each writes the wall clock).  Both as ONE captured
hipGraph (fork / join), replayed; prints the stamps' spacing while A is busy and while it is idle.   usage: starve_probe.py [B=32] [variant]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from glow_tts_amd import _lib, decoder as D
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
NA, NB = 12, 60
case = bench.fused_forward_case(B, 400)
run = case["run"]
L = D._L()
L.glowtts_debug_stamp.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
buf = torch.zeros(NB + 8, dtype=torch.int64, device="cuda")
small = torch.zeros(3840, device="cuda")
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
def body(kind):
    main = torch.cuda.current_stream()
    _lib.check(L.glowtts_debug_stamp(buf.data_ptr() + 8 * (NB + 2), _lib.stream()), "stamp")       # (a node in front of the fork, as in the training step)
    sb.wait_stream(main)
    with torch.cuda.stream(sb):
        for i in range(NB):
            if kind == "stamp":
                _lib.check(L.glowtts_debug_stamp(buf.data_ptr() + 8 * i, _lib.stream()), "stamp")
            else:                                   # a tiny torch elementwise launch between stamps
                small.add_(1.0)
                _lib.check(L.glowtts_debug_stamp(buf.data_ptr() + 8 * i, _lib.stream()), "stamp")
    _lib.check(L.glowtts_debug_stamp(buf.data_ptr() + 8 * NB, _lib.stream()), "stamp")
    for _ in range(NA):
        run()
    _lib.check(L.glowtts_debug_stamp(buf.data_ptr() + 8 * (NB + 1), _lib.stream()), "stamp")
    main.wait_stream(sb)
for kind in ("stamp", "torch+stamp"):
    with torch.cuda.stream(sa):
        body(kind); body(kind)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=sa):
        body(kind)
    for _ in range(8):
        g.replay()
    torch.cuda.synchronize()
    t = buf.cpu().tolist()
    a0, a1 = t[NB], t[NB + 1]
    st = t[:NB]
    inside = [x for x in st if a0 <= x <= a1]
    after = [x for x in st if x > a1]
    d = lambda xs: [(b - a) / 100.0 for a, b in zip(xs, xs[1:])]
    di, da = d(inside), d(after)
    med = lambda v: sorted(v)[len(v) // 2] if v else float("nan")
    before = [x for x in st if x < a0]
    print(f"  chain: first {(min(st) - a0) / 100.0:.0f} us, last {(max(st) - a0) / 100.0:.0f} us relative to the fused launches' begin; {len(before)} before it, spacing median {med(d(before)):.1f} us")
    print(f"B={B} chain={kind}: fused span {(a1 - a0) / 100.0:.0f} us for {NA} launches; {len(inside)} of {NB} chain launches ran inside it: spacing median "
          f"{med(di):.1f} us (max {max(di) if di else 0:.1f}); {len(after)} after it: spacing median {med(da):.1f} us")
