#!/bin/bash
# Where do a workload's kernels spend their wave cycles?  One PMC pass (kernel trace only) with the SQ wave-state counters:
#   gpu_pipe_busy.sh <workload>   ->  gpurun_out/pipe_busy_<workload>.txt
# Per kernel: share of SQ_WAVE_CYCLES parked (SQ_WAIT_ANY: s_waitcnt / barrier), issue-stalled (SQ_WAIT_INST_ANY) and issuing
# (SQ_ACTIVE_INST_ANY), the issuing share split into VALU / LDS / VMEM, and VALU issue cycles per SIMD cycle
# (4 x SQ_ACTIVE_INST_VALU quad-cycles over 1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
W=${1:-efficientnet_b4}
rm -rf $O/pipe_$W
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAVES \
  --kernel-trace --output-format csv -d $O/pipe_$W -o m -- \
  python $R/bench.py --workload $W --steps 3 --warmup 1 --no-graph --no-cpu-baseline --no-kernel-events --extra "" > $O/pipe_$W.log 2>&1
echo "rocprof rc=$?"; tail -3 $O/pipe_$W.log | cut -c1-300
python - $O/pipe_$W $W > $O/pipe_busy_$W.txt <<'PY'
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
    k = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").replace("tfimm_gemm::", "")[:64]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Dispatch_Id"] not in seen:
        seen.add(r["Dispatch_Id"]); n[k] += 1; us[k] += dur.get(r["Dispatch_Id"], 0.0)
print(f"# {w}: per kernel -- launches, total us | of SQ_WAVE_CYCLES: parked (s_waitcnt / barrier), issue-stalled, issuing | of issuing: VALU, LDS, VMEM | "
      f"VALU issue cycles per SIMD cycle | resident waves per SIMD (SQ_WAVE_CYCLES x 4 / (1024 x GRBM_GUI_ACTIVE / 8))")
for k, tot in sorted(us.items(), key=lambda kv: -kv[1])[:14]:
    a = agg[k]
    wc = a.get("SQ_WAVE_CYCLES", 0.0) or 1.0
    gui = a.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    act = a.get("SQ_ACTIVE_INST_ANY", 0.0) or 1.0
    simd = 1024.0 * gui or 1.0
    print(f"{k:66s} n={n[k]:4d} {tot:9.1f} us | parked {100*a.get('SQ_WAIT_ANY',0)/wc:5.1f} % stalled {100*a.get('SQ_WAIT_INST_ANY',0)/wc:5.1f} % issuing {100*act/wc:5.1f} % | "
          f"VALU {100*a.get('SQ_ACTIVE_INST_VALU',0)/act:5.1f} % LDS {100*a.get('SQ_ACTIVE_INST_LDS',0)/act:5.1f} % VMEM {100*a.get('SQ_ACTIVE_INST_VMEM',0)/act:5.1f} % | "
          f"VALU/SIMD {4*a.get('SQ_ACTIVE_INST_VALU',0)/simd:5.2f} | waves/SIMD {4*wc/simd:4.1f}")
PY
cat $O/pipe_busy_$W.txt | cut -c1-260
rm -rf $O/pipe_$W
