#!/bin/bash
# round 4, call e: everything after the permlane-swap reductions / hidden-visibility build: smoke, full GPU suite, default
# bench (with cpu baseline), attention probes (resident TSF-B kernels; streaming config-4 kernels, both register budgets),
# rocprofv3 kernel trace of the bench, PMC traffic passes
set -u
O=gpurun_out/r4e
mkdir -p $O
export TMPDIR=/tmp
(timeout 300 python __graft_entry__.py smoke 2>&1 | tail -4) > $O/smoke.log
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -E "^E  |passed|failed|^FAILED|error" | cut -c1-300 | head -40) > $O/pytest.log
(timeout 600 python bench.py 2>/dev/null | tail -1) > $O/bench.json
for w in fwd bwd; do (timeout 300 python tools/probe_attn.py space $w 256 20 2>&1 | tail -1) >> $O/probe_tsfb.txt; done
for v in 0 1; do for w in fwd bwd; do
  (PROBE_STREAM_VARIANT=$v PROBE_F=16 PROBE_N=576 PROBE_H=16 timeout 300 python tools/probe_attn.py space $w 8 20 2>&1 | tail -1) >> $O/probe_config4_variant$v.txt
done; done
for mode in default serial; do
  cd /tmp
  if [ $mode = serial ]; then export LAVILA_TEXT_STREAM=0; else unset LAVILA_TEXT_STREAM; fi
  timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_$mode -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-events > $GRAFT_REPO_ROOT/$O/prof_$mode.log 2>&1
  cd $GRAFT_REPO_ROOT
  DB=$(find $O/prof_$mode -name "*.db" | head -1)
  [ -n "$DB" ] && python tools/kernel_stats.py $DB 7 > $O/bench_kernel_stats_$mode.csv 2>$O/kernel_stats_$mode.err
  rm -rf $O/prof_$mode
done
unset LAVILA_TEXT_STREAM
bash tools/pmc_bench_traffic.sh > $O/traffic.log 2>&1
cp gpurun_out/traffic/*.json $O/ 2>/dev/null
rm -rf gpurun_out/traffic/FETCH_SIZE gpurun_out/traffic/WRITE_SIZE
echo done > $O/finished
