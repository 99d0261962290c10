#!/bin/bash
# HBM traffic of the dominant kernels INSIDE the bench step: two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; counters only,
# no tracing) of `bench.py --steps 1 --warmup 1`, aggregated per kernel family into gpurun_out/traffic/*.json
# (copy to profiles/rNN_traffic_*.json). Units / gfx950 correction: MI355X_MICROARCH.md section HBM (FETCH_SIZE counts the
# 128-byte requests of wide coalesced reads at 64 B: x2; both counters are in KiB).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/traffic
mkdir -p $out
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $out/$c -o p -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-events > $out/$c.log 2>&1
done
python - "$out" <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
rows = collections.defaultdict(dict)          # (dispatch id) -> {name, FETCH, WRITE}
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    for f in glob.glob(f'{out}/{c}/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            d = rows[(r['Dispatch_Id'])]
            d['name'] = r['Kernel_Name']
            d[r['Counter_Name']] = d.get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
# dispatch ids differ between the two passes but the launch ORDER is identical: pair the k-th launch of a kernel name
per = collections.defaultdict(lambda: {'FETCH_SIZE': [], 'WRITE_SIZE': []})
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    seq = sorted(((int(k), v) for k, v in rows.items() if c in v), key=lambda t: t[0])
    for _, v in seq:
        per[v['name']][c].append(v[c])
def family(sub, min_bytes, tag, note):
    f, w = [], []
    for name, d in per.items():
        if sub in name:
            n = min(len(d['FETCH_SIZE']), len(d['WRITE_SIZE']))
            f += d['FETCH_SIZE'][:n]; w += d['WRITE_SIZE'][:n]
    tr = [2 * 1024 * a + 1024 * b for a, b in zip(f, w)]
    tr = [t for t in tr if t >= min_bytes]
    if not tr:
        return
    j = {'kernel': sub, 'launches': len(tr), 'traffic_bytes_per_launch': round(sum(tr) / len(tr)),
         'min': round(min(tr)), 'max': round(max(tr)),
         'note': note + ' -- rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py --steps 1 --warmup 1 (both steps counted), '
                 'bytes = 2*1024*FETCH_SIZE + 1024*WRITE_SIZE (gfx950: wide reads are tallied at half), averaged over the launches'}
    json.dump(j, open(f'{out}/{tag}.json', 'w'), indent=1)
    print(tag, j['launches'], 'launches, avg', j['traffic_bytes_per_launch'] / 1e9, 'GB (min', j['min'] / 1e9, 'max', j['max'] / 1e9, ')')
family('gemm_tn_kernel', 2e8, 'r06_traffic_gemm_tn', 'all video-tower lvl_linear_tn launches (forward + input gradient, shape mix of the step)')
family('wgrad_kernel<4, 2, 6, 6', 2e8, 'r06_traffic_wgrad', 'all video-tower lvl_linear_wgrad launches (shape mix of the step)')
family('13, false, 8, false', 1e8, 'r06_traffic_space_fwd', 'space-mode lvl_divided_attn_fwd launches')
family("space_bwd_fused_kernel", 1e8, "r06_traffic_space_bwd", "space-mode lvl_divided_attn_bwd launches (fused kernel)")
PY
