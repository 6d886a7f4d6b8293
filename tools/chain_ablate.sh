for d in 0 1 2 3 4 5 7; do echo "dbg=$d"; TFIMM_CHAIN_DBG=$d python tools/chain_probe.py 256 2>&1 | grep "B=256" | cut -c1-125; done
# NEEDS A PROBE BUILD: the *_DBG switches exist only with -DTFIMM_PROBE_HOOKS (tools/probes/build_dbg_libs.sh all; export TFIMM_HIP_LIB=tools/probes/bin/libtfimm_hip_probe.so)
