#!/bin/bash
# PMC counters of one GEMM shape:  gpu_pmc.sh M K N hint
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
M=$1; K=$2; N=$3; H=$4
python $R/tools/gemm_probe.py $M $K $N $H 10
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT" "GRBM_GUI_ACTIVE GRBM_COUNT TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $O/pmc_$i -o pmc -- python $R/tools/gemm_probe.py $M $K $N $H 3 > $O/pmc_$i.log 2>&1
  f=$(find $O/pmc_$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" <<'PY'
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for r in rows:
    k=r.get("Kernel_Name","")[:60]
    if "gemm" not in k: continue
    agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[(k,r["Counter_Name"])]+=1
for k,v in agg.items():
    for c,val in v.items(): print(f"{k:60s} {c:32s} {val/cnt[(k,c)]:.4g} (avg over {cnt[(k,c)]} dispatches)")
PY
done
