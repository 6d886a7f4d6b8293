#!/bin/bash
# HBM traffic of every scored workload from rocprofv3 PMC counters (MI355X_MICROARCH.md §HBM): FETCH_SIZE and WRITE_SIZE in
# SEPARATE --pmc passes (TCC slots) over an eager (no hipGraph) run of bench.py; per-family sums per step go to
# gpurun_out/traffic.json (copy to profiles/rNN_traffic.json).  Units: counters are KiB; FETCH_SIZE is doubled for the
# 16 B/lane reads of gfx950 (guide, HBM section), WRITE_SIZE taken as is; kernels with known byte counts (input cast)
# are kept in the output as a calibration check.
#   tools/gpu_traffic.sh [workload ...]      (default: the four BASELINE.json GPU configurations)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
STEPS=3; WARM=1
WL=${@:-resnet50 vit_base_patch16_224 swin_base_patch4_window7_224 efficientnet_b4}
for w in $WL; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $O/traffic_${w}_$c
    timeout 900 rocprofv3 --pmc $c --output-format csv -d $O/traffic_${w}_$c -o t -- python $R/bench.py --workload $w --steps $STEPS --warmup $WARM --no-graph --no-cpu-baseline --no-kernel-events --extra "" > $O/traffic_${w}_$c.log 2>&1
    echo "$w $c rc=$?"
  done
done
python - "$O" $STEPS $WARM $WL <<'PY'
import csv, glob, json, sys, collections
O, steps, warm = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
FAMILIES = {"resnet50": ("gemm",), "vit_base_patch16_224": ("gemm",), "swin_base_patch4_window7_224": ("gemm", "attention"),
            "efficientnet_b4": ("gemm", "dwconv")}
def family(k):
    if "tfimm_gemm" in k or "stem_pool" in k or "mlp_fused" in k: return "gemm"
    if "attn_" in k or "tha_" in k: return "attention"
    if "dwconv" in k or "expand_dw" in k: return "dwconv"
    if "cast_" in k or "preprocess" in k: return "cast_input"
    return "other"
res = {"_doc": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over eager bench.py steps; bytes per step and per "
               "launch of the workload's roofline family; FETCH_SIZE x 2 (gfx950 wide-read correction), counters in KiB",
       "steps_profiled": warm + 2 * steps, "workloads": {}}
n = warm + 2 * steps      # bench.py --no-graph launches W warm-up + K timed steps + K more for the per-step medians (a pass of its own since round 5)
for w in sys.argv[4:]:
    out = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        f = glob.glob(f"{O}/traffic_{w}_{c}/**/*counter_collection.csv", recursive=True)
        if not f:
            continue
        agg = collections.defaultdict(float); cnt = collections.Counter()
        for r in csv.DictReader(open(f[0])):
            if r["Counter_Name"] != c:
                continue
            fam = family(r["Kernel_Name"])
            agg[fam] += float(r["Counter_Value"]); cnt[fam] += 1
        out[c] = {k: {"sum_kib": v, "dispatches": cnt[k]} for k, v in agg.items()}
    try:
        fams = FAMILIES.get(w, ("gemm",))
        fetch = sum(out["FETCH_SIZE"].get(k, {"sum_kib": 0})["sum_kib"] for k in fams) * 1024 * 2 / n
        write = sum(out["WRITE_SIZE"].get(k, {"sum_kib": 0})["sum_kib"] for k in fams) * 1024 / n
        launches = sum(out["FETCH_SIZE"].get(k, {"dispatches": 0})["dispatches"] for k in fams) / n
        out.update(family=list(fams), family_launches_per_step=launches, fetch_bytes_per_step_corrected=fetch,
                   write_bytes_per_step=write, hbm_bytes_per_step=fetch + write, hbm_bytes_per_launch=(fetch + write) / launches)
    except (KeyError, ZeroDivisionError) as e:
        out["error"] = repr(e)
    res["workloads"][w] = out
json.dump(res, open(f"{O}/traffic.json", "w"), indent=1)
print(json.dumps({w: {k: v for k, v in o.items() if not isinstance(v, dict)} for w, o in res["workloads"].items()}, indent=1))
PY
