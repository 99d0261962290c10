"""Does a GraphedTrainStep replay read memory it does not own? Two identical runs of the same five steps, one of them NaN-filling
every cached free block of the ordinary allocator pools (current stream, step stream) between the calls: any difference /
NaN means a captured kernel (or a recorded closure) holds a pointer to memory the allocator considers free.
PROBE_GROUP=0: no process group (one graph); PROBE_GROUP=gloo|nccl: one-rank group (graph segments)."""
import os
import sys
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
os.environ.setdefault('MASTER_PORT', '29656')
os.environ['LAVILA_TEXT_STREAM'] = os.environ.get('LAVILA_TEXT_STREAM', '0')
from helpers import build_model                                     # noqa: E402
from lavila.models.loss import CLIPLoss                             # noqa: E402
from lavila_amd.graph_step import GraphedTrainStep                  # noqa: E402
from oracle import oracle as O                                      # noqa: E402

CFG = dict(img=32, patch=16, frames=2, dim=256, depth=2, heads=4, t_width=256, t_heads=4, t_layers=2, vocab=512,
           embed=64, batch=3, gated=False)
if os.environ.get('PROBE_CFG') == 'tsfb':        # every GEMM of this geometry is tiled by lavila_amd's own kernels
    CFG = dict(img=224, patch=16, frames=4, dim=768, depth=1, heads=12, t_width=512, t_heads=8, t_layers=1, vocab=512,
               embed=256, batch=4, gated=False)
if os.environ.get('PROBE_CFG') == 'tiny256':
    CFG = dict(CFG, embed=256)
LIBRARY_GEMMS = {}


def count_library_gemms():
    import torch.nn.functional as F
    for o, n in [(F, 'linear'), (torch, 'matmul'), (torch, 'mm'), (torch, 'bmm'), (torch, 'addmm'), (torch, 'einsum'),
                 (torch.Tensor, 'matmul'), (torch.Tensor, '__matmul__'), (torch.Tensor, '__rmatmul__'), (torch.Tensor, 'mm')]:
        f = getattr(o, n)

        def wrapped(*a, _f=f, _n=n, **k):
            if any(torch.is_tensor(x) and x.is_cuda for x in a):
                key = (_n, tuple(tuple(x.shape) for x in a if torch.is_tensor(x)))
                LIBRARY_GEMMS[key] = LIBRARY_GEMMS.get(key, 0) + 1
            return _f(*a, **k)
        setattr(o, n, wrapped)


if os.environ.get('PROBE_COUNT_GEMM') == '1':
    count_library_gemms()
torch.cuda.set_device(0)
dev = torch.device('cuda', 0)
group = os.environ.get('PROBE_GROUP', '0')
if group != '0':
    dist.init_process_group(group, rank=0, world_size=1)


def run(poison):
    torch.manual_seed(0)
    if poison and (os.environ.get('PROBE_OWNERS') == '1' or os.environ.get('PROBE_DESCRIBE') == '1'):
        torch.cuda.memory._record_memory_history(enabled='all', context='all', stacks='python', max_entries=600000)
    model = build_model(CFG)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    model.load_state_dict(O.procedural_weights(shapes, seed=5))
    model.cuda().train()
    crit = CLIPLoss(use_vissl=True, cache_labels=True, rank=0, world_size=1)
    fused = os.environ.get('PROBE_FUSED', '1') == '1'
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3, eps=1e-3, fused=fused, foreach=None if fused else True, capturable=True)
    B = CFG['batch']
    variant = os.environ.get('PROBE_VARIANT', '')
    if 'nostep' in variant:                       # the graph without the optimizer: forward, loss, backward, clamp
        opt.step = lambda *a, **k: None
    kw = {}
    if 'noclamp' in variant:
        kw['clamp_logit_scale'] = None
    if 'fulltext' in variant:                     # all 77 positions: the token slice is the whole (contiguous) tensor
        kw['text_bucket'] = 77
    step = GraphedTrainStep(model, crit, opt, (B, 3, CFG['frames'], CFG['img'], CFG['img']), (B, 77), dev, **kw)
    losses = []
    saves = {}
    if os.environ.get('PROBE_SAVE') == '1':          # preallocated: looking must not change what the allocator hands out
        names = [n for n, _ in model.named_parameters()]
        plist = [q for _, q in model.named_parameters()]
        for it_ in (1, 2, 3):
            saves[it_] = {k: [torch.empty_like(q) for q in plist] for k in ('param', 'grad', 'exp_avg', 'exp_avg_sq')}
            saves[it_]['step'] = [torch.empty((), device=dev) for _ in plist]
            saves[it_]['names'] = names
    for it in range(5):
        video, tokens = O.synthetic_batch(B, CFG['frames'], CFG['img'], seed=40 + it)
        tokens = tokens.clone()
        tokens[:, 1:31] = tokens[:, 1:31] % 510 + 1
        tokens[:, 0], tokens[:, 31] = 510, 511
        if poison and (os.environ.get('PROBE_POISON_ITS') is None or str(it) in os.environ['PROBE_POISON_ITS'].split(',')):
            torch.cuda.synchronize()
            streams = {'cur': torch.cuda.current_stream(), 'step': step._stream}
            if getattr(step, '_comm', None) is not None:
                streams['comm'] = step._comm
            which = os.environ.get('PROBE_POISON_STREAMS', 'cur,step,comm').split(',')
            nbytes = [int(x) for x in os.environ.get('PROBE_POISON_BYTES', '256,2048,16384,131072,1048576,4194304,16777216,67108864').split(',')]
            for name, st in streams.items():
                if name not in which:
                    continue
                with torch.cuda.stream(st):
                    fill = os.environ.get('PROBE_FILL_SET')           # 'lo:hi': NaN into these poison tensors only (all are allocated)
                    if fill is None:
                        junk = [torch.full((n // 4,), float('nan'), device=dev) for n in nbytes for _ in range(12)]
                    else:
                        lo_, hi_ = (int(x) for x in fill.split(':'))
                        junk = [torch.empty((n // 4,), device=dev) for n in nbytes for _ in range(12)]
                        for j in junk[lo_:hi_]:
                            j.fill_(float('nan'))
                        if os.environ.get('PROBE_DESCRIBE') == '1':
                            snap = torch.cuda.memory._snapshot()
                            segs = sorted(snap['segments'], key=lambda sg: sg['address'])
                            for j in junk[lo_:hi_]:
                                a = j.data_ptr()
                                print(f'[describe] poison tensor at {a:#x}, {j.numel() * 4} B', flush=True)
                                known = [(f'param {n_}', q) for n_, q in model.named_parameters()]
                                known += [(f'buffer {n_}', q) for n_, q in model.named_buffers()]
                                known += [(f'grad {n_}', q.grad) for n_, q in model.named_parameters() if q.grad is not None]
                                for n_, q in model.named_parameters():
                                    known += [(f'state[{k_}] {n_}', v_) for k_, v_ in opt.state.get(q, {}).items() if torch.is_tensor(v_)]
                                known += [('step.tokens', step.tokens), ('step.video', step.video)]
                                known += [(f'lr {gi}', g_['lr']) for gi, g_ in enumerate(opt.param_groups) if torch.is_tensor(g_['lr'])]
                                near = sorted((t_.data_ptr(), t_.numel() * t_.element_size(), n_, tuple(t_.shape)) for n_, t_ in known
                                              if t_.is_cuda and -(1 << 20) < t_.data_ptr() - a < (1 << 16))
                                for b0, nb, n_, shp in near:
                                    print(f'[describe]   live {b0:#x} (+{a - b0:>8d} B to the poison) {nb:>8d} B  {n_} {shp}', flush=True)
                                for si, sg in enumerate(segs):
                                    if sg['address'] <= a < sg['address'] + sg['total_size']:
                                        for sj in range(max(0, si - 2), min(len(segs), si + 3)):
                                            t = segs[sj]
                                            print(f"[describe]   {'>>' if sj == si else '  '} segment {t['address']:#x}..{t['address'] + t['total_size']:#x} "
                                                  f"{t['total_size']:>10d} B pool {tuple(t.get('segment_pool_id', (0, 0)))} stream {t.get('stream')} "
                                                  f"{t.get('segment_type')} active {t.get('allocated_size')}", flush=True)
                                        off = sg['address']
                                        for b in sg['blocks']:
                                            fr = [f for f in b.get('frames', []) if '/lavila_amd/' in f['filename'] or '/tools/' in f['filename'] or '/optim/' in f['filename']]
                                            mark = '<< poison' if off <= a < off + b['size'] else ''
                                            if abs(off - a) < (1 << 16) or mark:
                                                print(f"[describe]      block {off:#x} {b['size']:>9d} B {b['state']:22s} "
                                                      f"{' <- '.join(os.path.basename(f['filename']) + ':' + str(f['line']) + ' ' + f['name'] for f in fr[:4])} {mark}", flush=True)
                                            off += b['size']
                    if os.environ.get('PROBE_OWNERS') == '1':
                        # who owned the memory under each poison tensor last? (device trace: the most recent earlier
                        # allocation overlapping its range)
                        snap = torch.cuda.memory._snapshot()
                        tr = snap['device_traces'][0]
                        mine = {j.data_ptr() for j in junk}
                        allocs = [(i, e) for i, e in enumerate(tr) if e['action'] == 'alloc']
                        own = {}
                        for j in junk:
                            a0, a1 = j.data_ptr(), j.data_ptr() + j.numel() * 4
                            prev = [(i, e) for i, e in allocs if e['addr'] < a1 and a0 < e['addr'] + e['size'] and
                                    not (e['addr'] in mine and e['size'] >= j.numel() * 4 and i >= allocs[-len(junk)][0])]
                            for i, e in prev[-3:]:
                                fr = [f for f in e.get('frames', []) if '/lavila_amd/' in f['filename'] or '/tools/' in f['filename']
                                      or '/optim/' in f['filename'] or 'autograd' in f['filename']]
                                key = (e['size'], e['stream'], ' <- '.join(f"{os.path.basename(f['filename'])}:{f['line']} {f['name']}" for f in fr[:5]))
                                own[key] = own.get(key, 0) + 1
                        print(f'[poison it={it} stream={name}] last owners of the poisoned memory:', flush=True)
                        for (sz, st_, key), n in sorted(own.items(), key=lambda kv: -kv[1])[:40]:
                            print(f'    {n:3d} x {sz:10d} B stream {st_}  {key}', flush=True)
                    if os.environ.get('PROBE_OVERLAP') == '1':
                        live = []
                        for n_, q in model.named_parameters():
                            live.append((f'param {n_}', q))
                            if q.grad is not None:
                                live.append((f'grad {n_}', q.grad))
                            for k_, v_ in opt.state.get(q, {}).items():
                                if torch.is_tensor(v_):
                                    live.append((f'state[{k_}] {n_}', v_))
                        for gi, grp in enumerate(opt.param_groups):
                            if torch.is_tensor(grp['lr']):
                                live.append((f'lr group {gi}', grp['lr']))
                        live += [('step.video', step.video), ('step.tokens', step.tokens)]
                        hits = {}
                        for j in junk:
                            a0, a1 = j.data_ptr(), j.data_ptr() + j.numel() * j.element_size()
                            for name_, t_ in live:
                                if not t_.is_cuda:
                                    continue
                                b0 = t_.untyped_storage().data_ptr()
                                b1 = b0 + t_.untyped_storage().nbytes()
                                if a0 < b1 and b0 < a1:
                                    kind = name_.split(' ')[0]
                                    hits.setdefault(kind, []).append(name_)
                        print(f'[poison it={it} stream={name}] junk tensors overlapping LIVE tensors: '
                              f'{ {k: (len(v), v[:3]) for k, v in hits.items()} }', flush=True)
                    if os.environ.get('PROBE_CLASSIFY') == '1' and it >= 2:
                        segs = [(sg['address'], sg['address'] + sg['total_size'], tuple(sg.get('segment_pool_id', (0, 0))),
                                 sg.get('stream')) for sg in torch.cuda.memory._snapshot()['segments']]
                        inside = {}
                        for j in junk:
                            a = j.data_ptr()
                            hit = [(pid, sst) for lo, hi, pid, sst in segs if lo <= a < hi]
                            key = ('private' if hit and hit[0][0] != (0, 0) else 'ordinary') if hit else 'no-segment'
                            inside[key] = inside.get(key, 0) + 1
                        print(f'[poison it={it} stream={name}] junk tensors by segment kind: {inside}; segments: '
                              f'{sum(1 for x in segs if x[2] != (0, 0))} private / {sum(1 for x in segs if x[2] == (0, 0))} ordinary',
                              flush=True)
                    del junk
            torch.cuda.synchronize()
        if os.environ.get('PROBE_REPLAY_ON_STEP_STREAM') == '1':
            step._stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(step._stream):
                out_ = step(video, tokens)
            torch.cuda.current_stream().wait_stream(step._stream)
            losses.append(float(out_['loss']))
        else:
            losses.append(float(step(video, tokens)['loss']))
        if it in saves:
            sv = saves[it]
            torch._foreach_copy_(sv['param'], [q.detach() for q in plist])
            torch._foreach_copy_(sv['grad'], [q.grad for q in plist])
            for k in ('exp_avg', 'exp_avg_sq', 'step'):
                torch._foreach_copy_(sv[k], [opt.state[q][k] for q in plist])
        if poison and os.environ.get('PROBE_WHERE') == '1':
            torch.cuda.synchronize()
            bad_g = [n for n, q in model.named_parameters() if q.grad is not None and not torch.isfinite(q.grad).all()]
            bad_p = [n for n, q in model.named_parameters() if not torch.isfinite(q).all()]
            bad_s = [(n, [(k, float(v.float().abs().max())) for k, v in opt.state[q].items() if torch.is_tensor(v) and not torch.isfinite(v).all()])
                     for n, q in model.named_parameters() if q in opt.state and any(
                torch.is_tensor(v) and not torch.isfinite(v).all() for v in opt.state[q].values())]
            print(f'[it={it}] loss {losses[-1]:.5f}  non-finite grads {len(bad_g)} {bad_g[:6]}  params {len(bad_p)} {bad_p[:6]}  '
                  f'optimizer state {len(bad_s)} {bad_s[:20]}', flush=True)
    torch.cuda.synchronize()
    if LIBRARY_GEMMS:
        print('library GEMM entry points reached:', dict(sorted(LIBRARY_GEMMS.items(), key=lambda kv: -kv[1])), flush=True)
        LIBRARY_GEMMS.clear()
    if poison:
        badg = [(n, int((~torch.isfinite(q.grad)).sum()), q.grad.numel()) for n, q in model.named_parameters()
                if q.grad is not None and not torch.isfinite(q.grad).all()]
        print(f'non-finite gradients after the last call: {len(badg)}: {badg[:40]}', flush=True)

        def runs(ix):
            ix = sorted(int(i) for i in ix)
            out, start, prev = [], None, None
            for i in ix:
                if start is None:
                    start = prev = i
                elif i == prev + 1:
                    prev = i
                else:
                    out.append((start, prev)); start = prev = i
            if start is not None:
                out.append((start, prev))
            return out[:8]
        for n, q in model.named_parameters():
            if q.grad is None or torch.isfinite(q.grad).all() or not any(t in n for t in ('blocks.1.', 'transformer.', 'ln_final', 'visual.norm')):
                continue
            bad = ~torch.isfinite(q.grad)
            if bad.dim() == 2:
                print(f'   {n}: rows {runs(bad.any(1).nonzero().flatten().tolist())} cols {runs(bad.any(0).nonzero().flatten().tolist())}', flush=True)
            else:
                print(f'   {n}: elements {runs(bad.flatten().nonzero().flatten().tolist())}', flush=True)
    p = torch.cat([q.detach().flatten().float() for q in model.parameters()] +
                  [q.grad.detach().flatten().float() for q in model.parameters() if q.grad is not None])
    if saves:
        return losses, p, {it_: {k: ([t.float().cpu() for t in v] if k != 'names' else v) for k, v in sv.items()} for it_, sv in saves.items()}
    return losses, p


if os.environ.get('PROBE_SAVE') == '1':
    la, pa, sa = run(False)
    lb, pb, sb = run(True)
    for it_ in sorted(sa):
        for k in ('grad', 'exp_avg', 'exp_avg_sq', 'step', 'param'):
            rows = []
            for n, a, b in zip(sa[it_]['names'], sa[it_][k], sb[it_][k]):
                fin = bool(torch.isfinite(b).all())
                d = float((a - b).abs().max() / (a.abs().max() + 1e-30)) if fin else float('nan')
                if not fin or d > 1e-3:
                    rows.append(f'{n} ({"non-finite: " + str(int((~torch.isfinite(b)).sum())) + " of " + str(b.numel()) if not fin else f"{d:.1e}"})')
            print(f'[after call {it_}] {k:10s}: {len(rows)} of {len(sa[it_]["names"])} tensors differ from the clean run: {rows[:8]}', flush=True)
else:
    la, pa = run(False)
    lb, pb = run(True)
print('group', group, 'clean   ', la, float(pa.double().sum()))
print('group', group, 'poisoned', lb, float(pb.double().sum()))
print('max |dp|', float((pa - pb).abs().max()), 'nan in poisoned params', bool(torch.isnan(pb).any()))
if group != '0':
    dist.destroy_process_group()
