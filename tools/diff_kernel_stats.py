"""Per-kernel difference of two tools/profile_step.sh summaries.   usage: python tools/diff_kernel_stats.py gpurun_out/a_kernel_stats.csv gpurun_out/b_kernel_stats.csv [n]"""
import csv, sys
a = {r["Name"]: r for r in csv.DictReader(open(sys.argv[1]))}
b = {r["Name"]: r for r in csv.DictReader(open(sys.argv[2]))}
n = int(sys.argv[3]) if len(sys.argv) > 3 else 25
rows = []
for k in set(a) | set(b):
    ta = float(a[k]["TotalDurationNsPerStep"]) if k in a else 0.0
    tb = float(b[k]["TotalDurationNsPerStep"]) if k in b else 0.0
    rows.append((tb - ta, ta, tb, a[k]["CallsPerStep"] if k in a else "0", b[k]["CallsPerStep"] if k in b else "0", k[:120]))
rows.sort(key=lambda r: -abs(r[0]))
print("serialised kernel time per step: %.0f us -> %.0f us" % (sum(r[1] for r in rows) / 1e3, sum(r[2] for r in rows) / 1e3))
for r in rows[:n]:
    print("%+8.1f us  %8.1f -> %8.1f  calls %s/%s  %s" % (r[0] / 1e3, r[1] / 1e3, r[2] / 1e3, r[3], r[4], r[5]))
