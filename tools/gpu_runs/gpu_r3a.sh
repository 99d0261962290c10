#!/bin/bash
# round-3 GPU pass A: tests, bench, N>1 rehearsal, loss timing, contention probe, GEMM probe
set -u
O=gpurun_out/r3a
mkdir -p $O
export TMPDIR=/tmp
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -40) > $O/pytest.log
echo "pytest rc=$?" >> $O/pytest.log
timeout 300 python bench.py --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.err
timeout 300 python bench.py --gpus 2 --batch 32 --steps 4 --warmup 2 > $O/bench_2rank_gloo.json 2> $O/bench_2rank.err
timeout 200 python tools/probe_clip_loss.py > $O/clip_loss_timing.txt 2>&1
timeout 100 python tools/probe_cu_contention.py --build > $O/contention.txt 2>&1
for spec in "0 1" "16 1" "16 0" "32 1"; do
  set -- $spec
  echo "== spin_wgs=$1 LAVILA_DYNAMIC_TILES=$2" >> $O/contention.txt
  LAVILA_DYNAMIC_TILES=$2 timeout 300 python tools/probe_cu_contention.py $1 --steps 6 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" >> $O/contention.txt 2>&1
done
for dyn in 1 0; do
  echo "== LAVILA_DYNAMIC_TILES=$dyn" >> $O/gemm_probe.txt
  LAVILA_DYNAMIC_TILES=$dyn timeout 300 python tools/probe_gemm_tn.py 2>&1 | grep -v "^check" >> $O/gemm_probe.txt
done
echo done > $O/finished
for dyn in 1 0; do
  echo "== wgrad LAVILA_DYNAMIC_TILES=$dyn" >> $O/gemm_probe.txt
  LAVILA_DYNAMIC_TILES=$dyn timeout 200 python tools/probe_wgrad_mfma.py 2>&1 >> $O/gemm_probe.txt
done
echo done2 > $O/finished
