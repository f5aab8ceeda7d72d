cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_t
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_t -o run -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --windows 0 --no-cpu-baseline --no-f32-key > /dev/null 2> /tmp/prof_t.err
python - <<'PY'
import csv, glob, collections
f = glob.glob("/tmp/prof_t/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
print(rows[0].keys())
sel = [r for r in rows if "conv_cl_kernel" in r["Kernel_Name"] and "float" not in r["Kernel_Name"].split("(")[0]]
by = collections.defaultdict(list)
for r in sel:
    by[(r["Kernel_Name"][:110], r.get("Grid_Size", r.get("Grid_Size_X")), r.get("Workgroup_Size", r.get("Workgroup_Size_X")))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
    print(k, len(v), "avg us %.1f" % (sum(v) / len(v)), "max %.1f" % max(v), "min %.1f" % min(v))
# neighbours (in start order, same stream) of the slowest 1x1 dispatches
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
slow = [i for i, r in enumerate(rows) if "2, 2, 2, 0, 1, 0, false" in r["Kernel_Name"] and int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) > 100000]
for i in slow[-2:]:
    sid = rows[i]["Stream_Id"]
    same = [j for j in range(max(0, i - 40), min(len(rows), i + 40)) if rows[j]["Stream_Id"] == sid]
    k = same.index(i)
    print("---- slow dispatch, stream", sid, {k: v for k, v in rows[i].items() if k not in ("Kernel_Name",)})
    for j in same[max(0, k - 5): k + 4]:
        r = rows[j]
        print("   %s %8.1f us  %s" % ("->" if j == i else "  ", (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Kernel_Name"].replace("(anonymous namespace)::", "")[:120]))
PY
