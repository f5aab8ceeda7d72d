"""Reads a rocprofv3 kernel-trace CSV and prints, per kernel name, the mean duration in the first and the second half of its dispatches."""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    d[r["Kernel_Name"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
rows = []
for k, v in d.items():
    v.sort()
    n = len(v)
    if n < 40:
        continue
    a = sum(x[1] for x in v[: n // 3]) / (n // 3)
    b = sum(x[1] for x in v[-(n // 3):]) / (n // 3)
    rows.append((a * n / 3 - b * n / 3, a, b, n, k))
rows.sort(reverse=True)
for diff, a, b, n, k in rows[:15]:
    print(f"first third {a / 1e3:9.1f} us  last third {b / 1e3:9.1f} us  n={n:5d}  {k[:90]}")
