#!/bin/bash
# usage: tools/pmc_gemm.sh <tag> <command...>  -- PMC passes incl. cache counters into gpurun_out/pmc_<tag>/
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/pmc_$tag
mkdir -p $out
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCC_BUSY_sum TCP_GATE_EN1_sum TCP_TA_TCP_STATE_READ_sum" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $out/p$i -o p$i -- "$@" > $out/p$i.log 2>&1
done
python - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + '/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r['Kernel_Name'][:70]][r['Counter_Name']].append(float(r['Counter_Value']))
with open(out + '/summary.txt', 'w') as fo:
    for k, d in agg.items():
        if 'elementwise' in k or 'fill' in k.lower() or 'copy' in k.lower() or 'distribution' in k:
            continue
        line = k + ' | ' + ' '.join(f'{c}={sum(v)/len(v):.4g}(n={len(v)})' for c, v in sorted(d.items()))
        print(line); fo.write(line + '\n')
PY
tail -3 $out/p3.log $out/p4.log
