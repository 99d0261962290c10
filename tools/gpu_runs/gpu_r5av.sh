#!/bin/bash
# round 5, call av: the edited poisoned-replay tests once (last seconds of the round's GPU budget)
O=gpurun_out/r5av
mkdir -p $O
timeout 40 python -m pytest tests/test_gpu_graph_step.py -q -k "free_device_memory" 2>&1 | grep -E "^E  |passed|failed|^FAILED" | cut -c1-600 | head -8 > $O/tests.txt
