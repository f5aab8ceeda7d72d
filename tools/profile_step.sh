#!/bin/bash
# Per-kernel time of the benchmarked training step: rocprofv3 --kernel-trace --stats around bench.py (the graph's kernels are traced one
# at a time, so the sum is the SERIALISED kernel time of both streams, not the step time).  Writes a per-step summary CSV.
#   usage (GPU box): tools/profile_step.sh <tag> [bench args]      -> gpurun_out/<tag>_kernel_stats.csv, gpurun_out/<tag>_bench.json
REPO=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-step}; shift
STEPS=20; WARM=5
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$TAG
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o run -- python $REPO/bench.py --steps $STEPS --warmup $WARM --windows 0 --no-cpu-baseline --no-f32-key --profile-run "$@" > $REPO/gpurun_out/${TAG}_bench.json 2> /tmp/prof_$TAG.err
python - "$TAG" "$REPO" <<'PY'
import csv, glob, sys, collections
tag, repo = sys.argv[1], sys.argv[2]
f = glob.glob(f"/tmp/prof_{tag}/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
# steps traced = launches of a kernel that runs exactly once per step and in none of bench.py's micro-benchmarks: the RAdam update
# (eager warm-up steps and graph replays alike; the capture pass launches nothing)
one = [r for r in rows if "radam_kernel" in r["Name"]]
steps = max(float(one[0]["Calls"]) if one else 1.0, 1.0)
out = [("Name", "CallsPerStep", "TotalDurationNsPerStep", "AverageNs", "Percentage")]
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
    out.append((r["Name"], round(float(r["Calls"]) / steps, 2), round(float(r["TotalDurationNs"]) / steps, 1), r["AverageNs"], round(100 * float(r["TotalDurationNs"]) / tot, 2)))
with open(f"{repo}/gpurun_out/{tag}_kernel_stats.csv", "w", newline="") as fo:
    csv.writer(fo).writerows(out)
print("steps traced", steps, "serialised kernel time per step (ms)", tot / steps / 1e6)
for o in out[1:16]:
    print(o)
PY
tail -c 600 $REPO/gpurun_out/${TAG}_bench.json
