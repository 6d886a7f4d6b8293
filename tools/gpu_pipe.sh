#!/bin/bash
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "tile28 or t28 or t21_ or tile21" 2>&1 | tail -8
for shp in "100864 768 2304" "100864 768 768" "100864 3072 768" "100864 768 3072" "50176 512 2048" "200704 256 1024" "50176 2304 256" "12544 2048 512"; do
  for h in 21 28; do
    timeout 120 python tools/gemm_probe.py $shp $h 50 2>&1 | tail -1
  done
done
