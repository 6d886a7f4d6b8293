#!/bin/bash
# Round-2 GPU session A: all gpu tests + the default bench line.  Summaries land under gpurun_out/.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q --timeout=900 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -25 $O/pytest_gpu.log | cut -c1-400
timeout 1200 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
cut -c1-6000 $O/bench.json; tail -5 $O/bench.err
