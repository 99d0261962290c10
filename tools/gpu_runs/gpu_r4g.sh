#!/bin/bash
# round 4, call g: PMC passes of the streaming space kernels at the config-4 shape (what bounds them?)
set -u
export TMPDIR=/tmp
export PROBE_F=16 PROBE_N=576 PROBE_H=16
bash tools/pmc_probe.sh stream_fwd space fwd 8 3 > gpurun_out/pmc_stream_fwd.log 2>&1
bash tools/pmc_probe.sh stream_bwd space bwd 8 3 > gpurun_out/pmc_stream_bwd.log 2>&1
rm -rf gpurun_out/pmc_stream_fwd/p? gpurun_out/pmc_stream_bwd/p?
echo done > gpurun_out/pmc_stream_finished
