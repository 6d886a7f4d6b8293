#!/bin/bash
# Per-kernel L2 read requests against HBM fetch bytes: which kernels re-read through L2 far more than they bring from HBM
# (how the bias-tile reloads of the Swin window kernel and the gate reads of the SE GEMMs show up).  Two --pmc passes per workload
# (TCP_TCC_READ_REQ_sum + TCC_HIT/MISS; FETCH_SIZE), eager bench.py steps.   tools/gpu_l2_audit.sh [workload ...] -> gpurun_out/l2_audit_<w>.txt
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
WL=${@:-resnet50 vit_base_patch16_224 swin_base_patch4_window7_224 efficientnet_b4}
for w in $WL; do
  i=0
  for set in "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE"; do
    i=$((i+1)); rm -rf $O/l2a_${w}_$i
    timeout 600 rocprofv3 --pmc $set --output-format csv -d $O/l2a_${w}_$i -o p -- python $R/bench.py --workload $w --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-kernel-events --branches 1 --extra "" > $O/l2a_${w}_$i.log 2>&1
    echo "$w pass $i rc=$?"
  done
  python - "$O" "$w" <<'PY' > $O/l2_audit_$w.txt
import csv, glob, sys, collections
O, w = sys.argv[1], sys.argv[2]
def load(i):
    f = glob.glob(f"{O}/l2a_{w}_{i}/**/*counter_collection.csv", recursive=True)
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").replace("tfimm_gemm::", "")[:64]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] in ("TCP_TCC_READ_REQ_sum", "FETCH_SIZE"): cnt[k] += 1
    return agg, cnt
a1, c1 = load(1); a2, c2 = load(2)
print(f"# {w}: per kernel (sum over the profiled eager steps) -- launches, L2 read requests x 64 B, HBM fetch (FETCH_SIZE KiB x 2: gfx950 wide reads), their ratio, L2 hit rate")
rows = []
for k in a1:
    req = a1[k].get("TCP_TCC_READ_REQ_sum", 0.0) * 64
    fetch = a2.get(k, {}).get("FETCH_SIZE", 0.0) * 1024 * 2
    hit, miss = a1[k].get("TCC_HIT_sum", 0.0), a1[k].get("TCC_MISS_sum", 0.0)
    rows.append((req, k, c1[k], fetch, hit / max(hit + miss, 1.0)))
for req, k, n, fetch, hr in sorted(rows, reverse=True)[:24]:
    print(f"{k:64s} n={n:5d}  L2 reads {req / 1e9:8.2f} GB  HBM fetch {fetch / 1e9:8.2f} GB  x{req / max(fetch, 1):5.1f}  hit {hr * 100:4.1f} %")
PY
  head -16 $O/l2_audit_$w.txt | cut -c1-170
  rm -rf $O/l2a_${w}_1 $O/l2a_${w}_2
done
