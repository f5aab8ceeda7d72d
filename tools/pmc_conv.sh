cd /tmp && export TMPDIR=/tmp
for c in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_MFMA GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  n=$(echo $c | cut -d' ' -f1)
  ITERS=5 timeout 120 rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_$n -o run -- python /root/repo/tools/bench_conv.py > /tmp/pmc_$n.log 2>&1
  python - <<PY
import csv,collections
rows=list(csv.DictReader(open("/tmp/pmc_$n/run_counter_collection.csv")))
acc=collections.defaultdict(list)
for r in rows:
    if "conv_dma" in r["Kernel_Name"] or "conv_cl_kernel" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in acc.items(): print(k, "per launch (mean of last launches):", sum(v[-4:])/len(v[-4:]), "n", len(v))
PY
done
