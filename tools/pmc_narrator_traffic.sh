#!/bin/bash
# HBM-side traffic of the narrator's decode kernels: two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; counters only, no
# tracing) of tools/probe_narrator.py (eager launches, 64 captions, 13 tokens), aggregated per kernel family into
# gpurun_out/narrator_traffic/r03_narrator_traffic.json. Units / gfx950 correction as in tools/pmc_bench_traffic.sh
# (MI355X_MICROARCH.md, HBM section): bytes = 2 * 1024 * FETCH_SIZE + 1024 * WRITE_SIZE.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/narrator_traffic
mkdir -p $out
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $out/$c -o p -- python tools/probe_narrator.py --batch 64 --length 13 --half --reps 1 --modes eager > $out/$c.log 2>&1
done
python - "$out" <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
per = collections.defaultdict(lambda: {'FETCH_SIZE': [], 'WRITE_SIZE': []})
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    rows = []
    for f in glob.glob(f'{out}/{c}/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r['Dispatch_Id']), r['Kernel_Name'], float(r['Counter_Value'])))
    agg = collections.OrderedDict()
    for d, n, v in sorted(rows):
        agg[(d, n)] = agg.get((d, n), 0.0) + v
    for (d, n), v in agg.items():
        per[n][c].append(v)
steps = 2 * 12          # warm-up call + timed call of generate(), 12 token steps each
fam = {'skinny_kernel': 'Conv1Ds on lvl_linear_skinny (97 per step incl. none of lm_head)', 'gated_add_ln_kernel': 'fused add + LayerNorm',
       'cls_attn_fwd_kernel': 'cross attention (algorithmic: 50.3 MB of image keys / values per launch)',
       'decode_self_attn_kernel': 'self attention step', 'sample_kernel': 'sampler (algorithmic: 64 x 100 KB)',
       'gpt2_embed_kernel': 'embedding'}
res = {}
total = 0.0
for sub, note in fam.items():
    f, w = [], []
    for name, d in per.items():
        if sub in name:
            n = min(len(d['FETCH_SIZE']), len(d['WRITE_SIZE']))
            f += d['FETCH_SIZE'][:n]; w += d['WRITE_SIZE'][:n]
    if not f:
        continue
    tr = [2 * 1024 * a + 1024 * b for a, b in zip(f, w)]
    res[sub] = {'launches': len(tr), 'bytes_per_launch': round(sum(tr) / len(tr)), 'bytes_per_token_step': round(sum(tr) / steps), 'what': note}
    total += sum(tr) / steps
# gemm_tn launches of 60-120 MB, in launch order: per generate() call first the 12 per-clip image key / value projections
# ([16384 x 768] -> 1536: 78 MB algorithmic), then one lm_head per token step (77.5 MB table + 6.5 MB logits)
name = [n for n in per if 'gemm_tn_kernel<0>' in n]
seq = []
for n in name:
    m = min(len(per[n]['FETCH_SIZE']), len(per[n]['WRITE_SIZE']))
    seq += [2 * 1024 * a + 1024 * b for a, b in zip(per[n]['FETCH_SIZE'][:m], per[n]['WRITE_SIZE'][:m])]
seq = [t for t in seq if 6e7 <= t <= 1.2e8]
calls = 2
per_call = len(seq) // calls
kv, lm = [], []
for c in range(calls):
    blk = seq[c * per_call:(c + 1) * per_call]
    kv += blk[:12]
    lm += blk[12:]
if lm:
    res['lm_head (gemm_tn_kernel<0>)'] = {'launches': len(lm), 'bytes_per_launch': round(sum(lm) / len(lm)),
                                           'bytes_per_token_step': round(sum(lm) / steps), 'what': 'algorithmic 84 MB'}
    total += sum(lm) / steps
if kv:
    res['image key/value projections (gemm_tn_kernel<0>, once per generate call)'] = {
        'launches': len(kv), 'bytes_per_launch': round(sum(kv) / len(kv)), 'what': 'algorithmic 78 MB; not part of a token step'}
res['total_bytes_per_token_step'] = round(total)
res['algorithmic_bytes_per_token_step'] = {'weights_bf16': 2 * 210e6, 'image_keys_values': 12 * 64 * 256 * 1536 * 2, 'note': 'plus <= 15 MB of self-attention cache and the activations'}
json.dump(res, open(f'{out}/r03_narrator_traffic.json', 'w'), indent=1)
print(json.dumps(res, indent=1))
PY
rm -rf $out/FETCH_SIZE $out/WRITE_SIZE
