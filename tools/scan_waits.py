"""Lists kernels whose ISA waits for (almost) every load it issues: `s_waitcnt vmcnt(0)` count close to the load count.  That is the
signature of a loop of predicated loads (`if (row < n) v = p[...]`), which hipcc compiles to branch - load - wait per element, i.e. one
dependent round trip to L2 / HBM per element (round 3: the attention kernels' operand staging, 12 round trips per 128 x 96 operand, and the
LayerNorm rows).  Static counts only - a kernel with several epilogue variants in one body shows the sum - so a hit is a
pointer to read the loop, not a verdict.
    python tools/scan_waits.py [file.hip ...]          (default: every .hip under glow_tts_amd/csrc; cross-compiles with hipcc, no GPU needed)"""
import glob, os, re, subprocess, sys, tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(REPO, "glow_tts_amd", "csrc")
files = sys.argv[1:] or sorted(glob.glob(os.path.join(CSRC, "*.hip")))
with tempfile.TemporaryDirectory() as tmp:
    for f in files:
        out = os.path.join(tmp, os.path.basename(f) + ".s")
        r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-I" + CSRC, f, "-o", out],
                           capture_output=True, text=True)
        if r.returncode != 0 or not os.path.exists(out):
            print(f"{os.path.basename(f)}: did not compile", file=sys.stderr)
            continue
        txt = open(out).read()
        parts = re.split(r"\n(_Z[\w]+):\s+; @", txt)
        for i in range(1, len(parts), 2):
            name, body = parts[i], parts[i + 1].split(".Lfunc_end")[0]
            loads = len(re.findall(r"\b(?:global_load|buffer_load)\w*", body))
            w0 = len(re.findall(r"s_waitcnt vmcnt\(0\)", body))
            if loads >= 5 and w0 >= max(4, loads // 3):
                print(f"{os.path.basename(f):24s} loads {loads:4d}  vmcnt(0) {w0:4d}  {name[:90]}")
