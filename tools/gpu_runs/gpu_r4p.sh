#!/bin/bash
# round 4, call p: (1) what the dbias rider of lvl_linear_wgrad costs at the qkv shapes (would replace lvl_qkv_bias_grad);
# (2) streaming space kernels against the resident ones at the TSF-B shape (197 keys)
set -u
O=gpurun_out/r4p
mkdir -p $O
export TMPDIR=/tmp
timeout 200 python tools/probe_wgrad_mfma.py 200960 qkv tqkv > $O/wgrad_dbias.txt 2>&1
for st in -1 1; do
  for w in fwd bwd; do
    (PROBE_STREAM=$st timeout 120 python tools/probe_attn.py space $w 256 30 2>&1 | tail -1) >> $O/space_tsfb_stream$st.txt
  done
done
timeout 120 python tools/probe_qkv_bias.py > $O/qkv_bias.txt 2>&1
echo done > $O/finished
