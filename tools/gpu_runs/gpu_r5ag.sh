#!/bin/bash
# round 5, call ag: which part of the captured iteration reads the culprit address (12 small poison tensors behind lr)
set -u
O=gpurun_out/r5ag
mkdir -p $O
export TMPDIR=/tmp
for v in base nostep noclamp fulltext nostep,noclamp nostep,noclamp,fulltext; do
  PROBE_VARIANT=$v PROBE_POISON_ITS=2,3,4 PROBE_POISON_STREAMS=cur PROBE_FILL_SET=0:12 timeout 200 python tools/probe_graph_step_poison.py 2>&1 | grep -v amdgpu.ids | tail -3 | cut -c1-250 | sed "s/^/[$v] /" >> $O/variants.txt
done
echo done > $O/finished
