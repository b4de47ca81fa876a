for d in 0 1 2 4 6 7; do echo "dbg=$d"; DM_DCN_FUSED_DBG=$d timeout -k 10 100 python tools/dcn_fused_probe.py time 2>&1 | grep "sigma=0.0"; done
