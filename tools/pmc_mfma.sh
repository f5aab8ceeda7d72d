#!/bin/bash
# Matrix-pipe utilisation of the two hottest kernels from rocprofv3 PMC counters: SQ_VALU_MFMA_BUSY_CYCLES and GRBM_GUI_ACTIVE in one --pmc pass
# (SQ and GRBM slots are independent, no tracing options), calibrated in the same pass against glowtts_mfma_clock_probe - a loop of
# back-to-back bf16 MFMAs with one wave per SIMD on every CU, i.e. 100 % by construction: util = (BUSY / ACTIVE)_kernel / (BUSY / ACTIVE)_probe.
#   usage (GPU box): tools/pmc_mfma.sh [out.json]     (default gpurun_out/conv_mfma_pmc.json; copied to profiles/rNN_conv_mfma_pmc.json)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=${1:-$REPO/gpurun_out/conv_mfma_pmc.json}
case "$OUT" in /*) ;; *) OUT="$PWD/$OUT";; esac
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_mfma
PMC_PROBE=1 PMC_WN_BWD=1 ITERS=6 timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_mfma -o run -- python $REPO/tools/pmc_conv.py > /tmp/pmc_mfma.log 2>&1 || { echo "rocprofv3 failed"; tail -5 /tmp/pmc_mfma.log; }
python - "$OUT" <<'PY'
import collections, csv, glob, json, sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("/tmp/pmc_mfma/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
def ratio(sub):
    for k, v in acc.items():
        if sub in k and v.get("SQ_VALU_MFMA_BUSY_CYCLES") and v.get("GRBM_GUI_ACTIVE"):
            b, a = v["SQ_VALU_MFMA_BUSY_CYCLES"][-4:], v["GRBM_GUI_ACTIVE"][-4:]
            return sum(b) / len(b), sum(a) / len(a)
    return None
probe = ratio("mfma_clock_probe")
out = {"method": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE (tools/pmc_mfma.sh), mean of the last 4 launches; utilisation relative to "
                 "glowtts_mfma_clock_probe (back-to-back bf16 MFMAs, one wave per SIMD on every CU = 100 %) measured in the same pass",
       "probe": {"mfma_busy_cycles": probe[0], "gui_active": probe[1]} if probe else None}
for name, sub in (("in_fwd", "conv_dma_kernel<1, 5"), ("in_dgrad", "conv_dma_kernel<0, 5"), ("wn_fwd", "wn_fwd_kernel"), ("wn_bwd", "wn_bwd_kernel")):
    r = ratio(sub) or ratio(sub.replace("<1, 5", "ILi1ELi5E").replace("<0, 5", "ILi0ELi5E"))
    if r and probe:
        out[name] = {"mfma_busy_cycles": r[0], "gui_active": r[1], "mfma_util": (r[0] / r[1]) / (probe[0] / probe[1])}
json.dump(out, open(sys.argv[1], "w"), indent=1)
print(json.dumps(out, indent=1))
PY
