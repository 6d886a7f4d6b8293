#!/bin/bash
# Probe builds of the library (never shipped as the product): tools/probes/bin/libtfimm_hip_thadbg.so = -DTFIMM_THA_DBG
# (talking-heads kernel with LDS integrity checks), libtfimm_hip_streamdbg.so = -DTFIMM_STREAM_DBG (probe bits of the persistent
# GEMM), libneighbour.so = synthetic neighbours.  Run after `make` in csrc (links the product objects for everything else).
set -eu
R=$(cd "$(dirname "$0")/../.." && pwd); S=$R/tensorflow-image-models_amd/csrc; O=$R/tools/probes/bin; mkdir -p $O
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wall -Wno-unused-function"
cd $S
hipcc $F -DTFIMM_THA_DBG -c cait.hip -o $O/cait_dbg.o
objs=$(ls build/*.o | grep -v plan_host)
hipcc --offload-arch=gfx950 -shared -fPIC $(echo $objs | tr ' ' '\n' | grep -v '/cait.o') $O/cait_dbg.o -o $O/libtfimm_hip_thadbg.so
for t in 0 1 2 3 4 5 6 8; do hipcc $F -DTFIMM_STREAM_DBG -DTILE_ID=$t -c gemm_stream_inst.hip -o $O/stream_dbg_$t.o & done; wait
hipcc --offload-arch=gfx950 -shared -fPIC $(echo $objs | tr ' ' '\n' | grep -v '/gemm_stream_inst_') $O/stream_dbg_*.o -o $O/libtfimm_hip_streamdbg.so
hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC $R/tools/probes/neighbour_kernels.hip -o $O/libneighbour.so
if [ "${1:-}" = all ]; then
  # libtfimm_hip_probe.so: EVERY translation unit with -DTFIMM_PROBE_HOOKS -- the library the ablation / stamp tools need
  # (TFIMM_HIP_LIB=tools/probes/bin/libtfimm_hip_probe.so tools/chain_ablate.sh, gemm_stamps.py, mbconv_ablate.sh, attn_probe.py)
  make -C $S -j8 OBJDIR=$O/probe_build OUT=$O/libtfimm_hip_probe.so CXXFLAGS="$F -DTFIMM_PROBE_HOOKS" $O/libtfimm_hip_probe.so
fi
ls -la $O/*.so
