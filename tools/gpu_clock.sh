#!/bin/bash
# shader clock under load: GRBM_GUI_ACTIVE cycles of a dispatch / its duration (kernel trace), stream vs deep-ring GEMM
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
for h in 21 28; do
  timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $O/clk_$h -o c -- python $R/tools/gemm_probe.py 100864 3072 768 $h 20 > $O/clk_$h.log 2>&1
  python - $O/clk_$h <<'PY'
import csv, glob, sys, collections
d = sys.argv[1]
kt = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
cc = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
dur = {}
for r in csv.DictReader(open(kt)):
    if "gemm" in r["Kernel_Name"]:
        dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"][:40])
agg = collections.defaultdict(dict)
for r in csv.DictReader(open(cc)):
    if r["Dispatch_Id"] in dur:
        agg[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
ids = sorted(agg, key=int)[5:]
ns = sum(dur[i][0] for i in ids) / len(ids)
g = sum(agg[i]["GRBM_GUI_ACTIVE"] for i in ids) / len(ids)
mf = sum(agg[i].get("SQ_VALU_MFMA_BUSY_CYCLES", 0) for i in ids) / len(ids)
sb = sum(agg[i].get("SQ_BUSY_CYCLES", 0) for i in ids) / len(ids)
print(dur[ids[0]][1], f"avg {ns/1e3:.1f} us  GRBM_GUI_ACTIVE {g:.4g}  -> {g/ns:.3f} cycles/ns;  MFMA_BUSY {mf:.4g} SQ_BUSY {sb:.4g}")
PY
done
