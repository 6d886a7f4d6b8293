#!/bin/bash
# MFMA-busy fraction per kernel of one workload's forward (PMC pass on its own, kernel trace only):
#   gpu_mfma_busy.sh <workload>   ->  gpurun_out/mfma_busy_<workload>.txt
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
W=${1:-vit_base_patch16_224}
rm -rf $O/mfma_$W
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $O/mfma_$W -o m -- \
  python $R/bench.py --workload $W --steps 3 --warmup 1 --no-graph --no-cpu-baseline --no-kernel-events --extra "" > $O/mfma_$W.log 2>&1
python - $O/mfma_$W $W > $O/mfma_busy_$W.txt <<'PY'
import csv, sys, glob, collections
d, w = sys.argv[1], sys.argv[2]
cc = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
kt = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
dur = {}
for r in csv.DictReader(open(kt)):
    dur[r["Dispatch_Id"]] = (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) * 1e-3
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); us = collections.Counter()
seen = set()
for r in csv.DictReader(open(cc)):
    k = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "")[:70]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if (r["Dispatch_Id"]) not in seen:
        seen.add(r["Dispatch_Id"]); n[k] += 1; us[k] += dur.get(r["Dispatch_Id"], 0.0)
print(f"# {w}: per kernel -- launches, total us, MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs), effective clock")
for k, tot in sorted(us.items(), key=lambda kv: -kv[1])[:12]:
    a = agg[k]
    gui = a.get("GRBM_GUI_ACTIVE", 0.0) / 8.0          # summed over 8 XCDs
    mf = a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / 1024.0
    busy = mf / gui if gui else 0.0
    clk = gui / (tot * 1e-6) / 1e9 if tot else 0.0
    print(f"{k:72s} n={n[k]:4d} {tot:10.1f} us  MFMA busy {100*busy:5.1f} %  clock {clk:4.2f} GHz")
PY
cat $O/mfma_busy_$W.txt | cut -c1-170
rm -rf $O/mfma_$W
