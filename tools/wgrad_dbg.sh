for d in 0 1 2 3 4 7 8 10 14; do echo -n "DBG=$d: "; LVL_WGRAD_DBG=$d timeout 100 python tools/probe_wgrad_mfma.py 2>&1 | grep "^fc1" | cut -c1-60; done
