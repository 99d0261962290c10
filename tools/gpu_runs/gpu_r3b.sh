#!/bin/bash
# round-3 GPU pass B: all GPU tests (no -x), bench, kernel trace, GEMM / wgrad probes dyn on/off, contention probe
set -u
O=gpurun_out/r3b
mkdir -p $O
export TMPDIR=/tmp
(timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -60) > $O/pytest.log
timeout 300 python bench.py --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.err
for dyn in 1 0; do
  echo "== LAVILA_DYNAMIC_TILES=$dyn" >> $O/gemm_probe.txt
  LAVILA_DYNAMIC_TILES=$dyn timeout 300 python tools/probe_gemm_tn.py 2>&1 | grep -v "^check" >> $O/gemm_probe.txt
done
timeout 100 python tools/probe_cu_contention.py --build > $O/contention.txt 2>&1
for spec in "0 1" "16 1" "16 0" "32 1" "0 0"; do
  set -- $spec
  echo "== spin_wgs=$1 LAVILA_DYNAMIC_TILES=$2" >> $O/contention.txt
  LAVILA_DYNAMIC_TILES=$2 timeout 300 python tools/probe_cu_contention.py $1 --steps 6 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" >> $O/contention.txt 2>&1
done
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-events > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT
DB=$(find $O/prof -name "*.db" | head -1)
[ -n "$DB" ] && python tools/kernel_stats.py $DB 5 > $O/kernel_stats.csv 2>$O/kernel_stats.err
rm -rf $O/prof
echo done > $O/finished
