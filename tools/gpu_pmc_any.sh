#!/bin/bash
# PMC counters for any probe:  gpu_pmc_any.sh <kernel-name-substring> <probe command...>
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
PAT=$1; shift
"$@"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum" "FETCH_SIZE" "WRITE_SIZE" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TA_BUSY_avr TA_TA_BUSY_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $O/pmcz_$i -o pmc -- "$@" > $O/pmcz_$i.log 2>&1
  f=$(find $O/pmcz_$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" "$PAT" <<'PY'
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for r in rows:
    k=r.get("Kernel_Name","")[:50].replace("void (anonymous namespace)::","")
    if sys.argv[2] not in r.get("Kernel_Name",""): continue
    agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[(k,r["Counter_Name"])]+=1
for k,v in agg.items():
    for c,val in v.items(): print(f"{k:42s} {c:26s} {val/cnt[(k,c)]:.4g}")
PY
done
