#!/bin/bash
# HBM-side traffic of the narrator's decode kernels: two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; counters only, no
# tracing) of tools/probe_narrator.py (eager launches, 64 clips x RET captions, 13 tokens), aggregated per kernel family into
# gpurun_out/narrator_traffic/r03_narrator_traffic_n<RET>.json. Units / gfx950 correction as in tools/pmc_bench_traffic.sh
# (MI355X_MICROARCH.md, HBM section): bytes = 2 * 1024 * FETCH_SIZE + 1024 * WRITE_SIZE.   usage: pmc_narrator_traffic.sh [RET]
RET=${1:-1}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/narrator_traffic
mkdir -p $out
EXTRA=""; [ "$RET" != "1" ] && EXTRA="--returns $RET --sample"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $out/$c -o p -- python tools/probe_narrator.py --batch 64 --length 13 --half --reps 1 --modes eager $EXTRA > $out/$c.log 2>&1
done
python - "$out" "$RET" <<'PY'
import csv, glob, json, sys, collections
out, ret = sys.argv[1], int(sys.argv[2])
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
rows = 64 * ret
def launches(sub):
    tr = []
    for name, d in per.items():
        if sub in name:
            n = min(len(d['FETCH_SIZE']), len(d['WRITE_SIZE']))
            tr += [2 * 1024 * a + 1024 * b for a, b in zip(d['FETCH_SIZE'][:n], d['WRITE_SIZE'][:n])]
    return tr
res, total = {}, 0.0
def add(label, tr, note, in_step=True):
    global total
    if not tr:
        return
    res[label] = {'launches': len(tr), 'bytes_per_launch': round(sum(tr) / len(tr)), 'what': note}
    if in_step:
        res[label]['bytes_per_token_step'] = round(sum(tr) / steps)
        total += sum(tr) / steps
strips = launches('skinny_kernel') + launches('skinny_ln_kernel')
tiles = launches('mid_kernel')
lm_tiles = [t for t in tiles if t >= 6e7]                       # lm_head on the tile kernel (<= 128 rows): 77.5 MB table
add('Conv1Ds (skinny_kernel / skinny_ln_kernel / mid_kernel below 60 MB)', strips + [t for t in tiles if t < 6e7],
    'algorithmic: 343 MB of weights per step + activations')
add('lm_head on mid_kernel', lm_tiles, 'algorithmic 84 MB')
# gemm_tn launches in order: the encoder passes first (probe: 2 x encode_image), then per generate() call the 12 image k|v
# projections followed by one lm_head per token step when that runs on the panel kernel (more than 128 rows)
tn = launches('gemm_tn_kernel<0>')
lm_per_call = 12 if rows > 128 else 0
tail = tn[len(tn) - 2 * (12 + lm_per_call):]
kv, lm = [], []
for c in range(2):
    blk = tail[c * (12 + lm_per_call):(c + 1) * (12 + lm_per_call)]
    kv += blk[:12]
    lm += blk[12:]
add('lm_head on gemm_tn_kernel<0>', lm, f'algorithmic {(77.5e6 + rows * 50432 * 2 + rows * 1536) / 1e6:.0f} MB')
add('image key/value projections (gemm_tn_kernel<0>, once per generate call)', kv, 'algorithmic 78 MB; not part of a token step', False)
add('fused add + LayerNorm (gated_add_ln_kernel)', launches('gated_add_ln_kernel'), '')
add('cross attention (cls_attn_fwd_kernel / cross_attn_mfma_kernel)', launches('cls_attn_fwd_kernel') + launches('cross_attn_mfma_kernel'),
    'algorithmic: 50.3 MB of image keys / values per launch')
add('self attention step (decode_self_attn_kernel)', launches('decode_self_attn_kernel'), '')
add('sampler (sample_kernel)', launches('sample_kernel'), f'algorithmic: {rows} x 100 KB')
add('embedding (gpt2_embed_kernel)', launches('gpt2_embed_kernel'), '')
res['total_bytes_per_token_step'] = round(total)
res['algorithmic_bytes_per_token_step'] = {'weights_bf16': 2 * 210e6, 'image_keys_values': 12 * 64 * 256 * 1536 * 2,
                                           'logits': rows * 50432 * 2, 'note': 'plus the self-attention caches (rows x position x 3 KB x 12) and the activations'}
res['captions'] = rows
json.dump(res, open(f'{out}/r03_narrator_traffic_n{ret}.json', 'w'), indent=1)
print(json.dumps(res, indent=1))
PY
rm -rf $out/FETCH_SIZE $out/WRITE_SIZE
