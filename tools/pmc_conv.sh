#!/bin/bash
# HBM traffic of the two hottest kernels from rocprofv3 PMC counters, collected as /opt/skills/guides/MI355X_MICROARCH.md prescribes:
# FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes (no tracing options), FETCH_SIZE doubled (gfx950 tallies 128-byte requests at 64 B),
# WRITE_SIZE calibrated against a 1 GiB fill measured in the same pass (and the x2 of FETCH_SIZE checked on a 1 GiB streaming read).  Writes $OUT (default gpurun_out/conv_pmc.json), which
# is copied to profiles/rNN_conv_pmc.json; bench.py reports its `traffic` figures and names the file.
#   usage (GPU box): tools/pmc_conv.sh [out.json]
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=${1:-$REPO/gpurun_out/conv_pmc.json}
case "$OUT" in /*) ;; *) OUT="$PWD/$OUT";; esac
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  PMC_WN_BWD=1 ITERS=6 timeout 300 rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_$c -o run -- python $REPO/tools/pmc_conv.py > /tmp/pmc_$c.log 2>&1 || { echo "rocprofv3 $c failed"; tail -5 /tmp/pmc_$c.log; }
done
python - "$OUT" <<'PY'
import collections, csv, glob, json, sys
def load(counter):
    acc = collections.defaultdict(list)
    for f in glob.glob(f"/tmp/pmc_{counter}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v[-4:]) / len(v[-4:]) for k, v in acc.items()}, {k: max(v) for k, v in acc.items()}     # mean of the last launches (warm caches, like the step); max
(fetch, fetch_max), (write, write_max) = load("FETCH_SIZE"), load("WRITE_SIZE")
def pick(d, *subs):
    for k, v in d.items():
        if all(s in k for s in subs):
            return v
    return None
GiB = 1024.0 ** 3
# the counters report KiB
def pick_max(d, sub):                                    # the 1 GiB calibration launches are the largest of their kernel class
    v = [x for k, x in d.items() if sub in k]
    return max(v) if v else None
fill_w = pick_max(write_max, "FillFunctor")
mul_r, mul_w = pick_max(fetch_max, "MulFunctor"), pick_max(write_max, "MulFunctor")
wcal = GiB / (fill_w * 1024) if fill_w else None
out = {"shape": [32, 400], "precision": "bf16", "units": "bytes per launch",
       "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes (tools/pmc_conv.sh); FETCH_SIZE x 1024 x 2 (gfx950: 128-byte "
                 "requests tallied at 64 B, MI355X_MICROARCH.md); WRITE_SIZE x 1024 x the factor that makes a 1 GiB fill read 1 GiB",
       "calibration": {"fill_1GiB_WRITE_SIZE_KiB": fill_w, "write_factor": wcal, "mul_1GiB_FETCH_SIZE_KiB": mul_r,
                       "mul_fetch_bytes_after_x2_over_1GiB": (mul_r * 2048 / GiB) if mul_r else None, "mul_1GiB_WRITE_SIZE_KiB": mul_w}}
for name, subs in (("in_fwd", ("conv_dma_kernel", "Li1ELi5E")), ("in_dgrad", ("conv_dma_kernel", "Li0ELi5E")), ("wn_fwd", ("wn_fwd_kernel",)), ("wn_bwd", ("wn_bwd_kernel",))):
    f, w = pick(fetch, *subs), pick(write, *subs)
    if f is None and not name.startswith("wn_"):      # demangled names
        subs2 = ("conv_dma_kernel<1, 5", ) if name == "in_fwd" else ("conv_dma_kernel<0, 5", )
        f, w = pick(fetch, *subs2), pick(write, *subs2)
    if f is not None and w is not None:
        fb, wb = f * 1024 * 2, w * 1024 * (wcal or 1.0)
        out[name] = {"FETCH_SIZE_KiB": f, "WRITE_SIZE_KiB": w, "fetch_bytes": fb, "write_bytes": wb, "traffic": fb + wb}
out["kernels_seen"] = sorted(set(list(fetch) + list(write)))[:40]
json.dump(out, open(sys.argv[1], "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "kernels_seen"}, indent=1))
PY
