#!/bin/bash
set -u
O=gpurun_out/r3p
mkdir -p $O
(timeout 600 python tools/probe_skinny.py --out $O/skinny_variants.json 2>&1 | tail -30) > $O/probe.log
echo done > $O/finished
