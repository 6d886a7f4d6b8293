for d in 0 1 2 3 4 5 7; do echo "dbg=$d"; TFIMM_CHAIN_DBG=$d python tools/chain_probe.py 256 2>&1 | grep "B=256" | cut -c1-125; done
