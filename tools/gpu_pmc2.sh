#!/bin/bash
# compare wave-cycles / busy cycles of GEMM variants:  gpu_pmc2.sh M K N hint "dbg values"
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
M=$1; K=$2; N=$3; H=$4
for d in $5; do
  export TFIMM_GEMM_DBG=$d
  rm -rf $O/pmcx
  timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmcx -o pmc -- python $R/tools/gemm_probe.py $M $K $N $H 50 > $O/pmcx.log 2>&1
  grep "TF/s" $O/pmcx.log
  f=$(find $O/pmcx -name "*counter_collection.csv" | head -1)
  python - "$f" "$d" <<'PY'
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
agg=collections.defaultdict(list)
for r in rows:
    if "gemm_stream" not in r.get("Kernel_Name",""): continue
    agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
out=[f"dbg={sys.argv[2]}"]
for c,v in sorted(agg.items()):
    v=v[len(v)//2:]   # steady-state half
    out.append(f"{c}={sum(v)/len(v):.4g}")
print(" ".join(out))
PY
  kt=$(find $O/pmcx -name "*kernel_trace.csv" | head -1)
  [ -n "$kt" ] && python - "$kt" <<'PY'
import csv,sys
rows=[r for r in csv.DictReader(open(sys.argv[1])) if "gemm_stream" in r["Kernel_Name"]]
d=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"])) for r in rows]
d=d[len(d)//2:]
print(f"  profiled kernel duration avg {sum(d)/len(d)/1e3:.1f} us over {len(d)} launches")
PY
done
