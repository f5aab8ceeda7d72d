#!/bin/bash
# Kernel ORDER of one replayed training step (rocprofv3 --kernel-trace serialises a graph's kernels: durations are real, overlaps are not):
# what runs before the decoder's first flow, between the forward and the backward, and after the last data gradient.
#   usage (GPU box): tools/step_order.sh <tag>      -> gpurun_out/<tag>_step_order.txt
REPO=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ord_$TAG
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/ord_$TAG -o run -- python $REPO/bench.py --steps 3 --warmup 2 --windows 0 --no-cpu-baseline --no-f32-key $ORDER_ARGS > /tmp/ord_$TAG.json 2> /tmp/ord_$TAG.err
python - "$TAG" "$REPO" <<'PY'
import csv, glob, sys, re
tag, repo = sys.argv[1], sys.argv[2]
f = glob.glob(f"/tmp/ord_{tag}/**/*kernel_trace.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ends = [i for i, r in enumerate(rows) if "radam_kernel" in r["Kernel_Name"]]
lo, hi = ends[-2] + 1, ends[-1] + 1          # the last complete step
step = rows[lo:hi]
t0 = int(step[0]["Start_Timestamp"])
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"^void ", "", n)
    return n[:100]
out = []
acc = 0.0
for r in step:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    acc += d
    out.append(f"{(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} {d:7.1f} {acc:8.1f}  q{r.get('Queue_Id', '?')}  {short(r['Kernel_Name'])}")
open(f"{repo}/gpurun_out/{tag}_step_order.txt", "w").write("# start_us  dur_us  cumulative_kernel_us  queue  kernel   (one replayed step, serialised by the tracer)\n" + "\n".join(out) + "\n")
print(len(step), "kernels in the step, serialised kernel time", round(acc, 1), "us")
PY
