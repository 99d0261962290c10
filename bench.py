#!/usr/bin/env python
"""bench.py -- clip-text pairs/s of the dual-encoder pretraining step on N MI355X GPUs of one node.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W
    python bench.py --gpus N ...          (no launcher: bench.py re-executes itself under torch.distributed.run, one
                                           rank per GPU, rendezvous on 127.0.0.1 and a free port)

One step = forward (TimeSformer-B/16 video tower on 4x224^2 clips + CLIP text tower on 32-token captions padded to
77) + all-gathered InfoNCE loss + backward + AdamW update, bf16 autocast over f32 master weights, local batch
256 per GPU (BASELINE.json configs[1]; weak scaling). Synthetic data, seeded random
weights with the temporal attention randomised so it is live. Rank 0 prints ONE JSON line.

Extra objects in the line:
  roofline     -- the dominant hand-written kernel: the MFMA forward / input-gradient GEMM of the Linear layers
                  (lvl_linear_tn, ~55 % of the step), aggregated over all its video-tower launches: algorithmic flops
                  (2*M*N*K per launch) / duration, measured with HIP events on the launch stream inside the timed
                  region, against the 2.5 PFLOP/s dense bf16 MFMA peak. `traffic` = HBM bytes per launch from rocprofv3
                  PMC passes of the same command (profiles/r02_traffic_gemm_tn.json: FETCH_SIZE x 2 + WRITE_SIZE averaged
                  over the step's launches), next to the algorithmic M*(K+N)*2 + N*K*2.
  roofline_wgrad -- the same for the MFMA weight-gradient GEMM (lvl_linear_wgrad, ~20 % of the step).
  roofline_hbm -- the dominant HBM-bound hand-written kernel (space-mode divided attention forward,
                  lvl_divided_attn_fwd): algorithmic bytes per launch / average duration against 8 TB/s.
  cpu_baseline -- the CPU oracle (oracle/oracle.py, kind "port") timed on this box's host cores on a bounded
                  sample (one fwd+loss+bwd of a small batch of the same shapes), rank 0 at N=1 only.

`--workload narrator` (not part of the driver contract; default stays the metric above) prints the same kind of line for
BASELINE configs[4]: captions/s of VCLM_OPENAI_TIMESFORMER_BASE_GPT2 (encode_image + generate), with the decode step's
HBM roofline and the CPU oracle's narrator as cpu_baseline.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)



import warnings  # noqa: E402

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

# under DDP the text tower's gradients are produced on the side stream while DDP keeps their AccumulateGrad nodes on
# the main stream: autograd joins the two (correct, see tests/test_gpu_ddp.py) and says so on every backward
warnings.filterwarnings('ignore', message="The AccumulateGrad node's stream does not match")

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy ceiling)
MFMA_PEAK_TFLOPS = 2500.0  # dense bf16 (the 2:1-sparsity headline figure is never used)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument('--gpus', type=int, default=1)
    p.add_argument('--steps', type=int, default=8)
    p.add_argument('--warmup', type=int, default=2)
    p.add_argument('--batch', type=int, default=256, help='local batch per GPU')
    p.add_argument('--frames', type=int, default=4)
    p.add_argument('--model', default='CLIP_OPENAI_TIMESFORMER_BASE')
    p.add_argument('--dtype', default='bf16', choices=['bf16', 'f32'])
    p.add_argument('--no-cpu-baseline', action='store_true')
    p.add_argument('--cpu-batch', type=int, default=8)
    p.add_argument('--event-stride', type=int, default=4,
                   help='HIP-event pairs around the timed kernel families on every Nth step of the timed region (1 = all)')
    p.add_argument('--no-events', action='store_true', help='skip per-launch HIP events (A/B their overhead)')
    p.add_argument('--workload', default='pretrain', choices=['pretrain', 'narrator'],
                   help='pretrain (default) = BASELINE.json\'s metric; narrator = captions/s of BASELINE configs[4] '
                        '(VCLM_OPENAI_TIMESFORMER_BASE_GPT2 inference), an extra line outside the driver contract')
    p.add_argument('--returns', type=int, default=10, help='narrator: captions sampled per clip')
    p.add_argument('--length', type=int, default=77, help='narrator: caption length in tokens')
    p.add_argument('--graph-only', action='store_true',
                   help='internal: run only the graphed-step variant (lavila_amd/graph_step.py) and print its record; the '
                        'default run spawns this in a child process so that nothing in it can touch the headline')
    p.add_argument('--checkpoint', action='store_true',
                   help='use_checkpoint=True (main_pretrain.py:100,491-495: activation checkpointing per block, '
                        'timesformer.py:175-187) -- for shapes whose activations do not fit, e.g. 16 frames at local batch 256')
    p.add_argument('--reuse-tokens', action='store_true',
                   help='feed the SAME token tensor object every step (A/B only: the caption-length read-back of the '
                        'text tower is memoised per tensor object; the default hands over a new tensor per step, as a '
                        'data loader does)')
    return p.parse_args()


class KernelTimer:
    """HIP-event pairs around every launch of one C-ABI op on the current (launch) stream."""

    def __init__(self):
        self.pairs = []
        self.work = []           # algorithmic work (bytes or flops) of each timed launch
        self.tags = []           # optional label of each timed launch (which epilogue, ...)
        self.enabled = False

    def wrap(self, fn, select, work=None, tag=None):
        def timed(*a, **k):
            if not (self.enabled and select(*a, **k)):
                return fn(*a, **k)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            out = fn(*a, **k)
            e.record()
            self.pairs.append((s, e))
            if work is not None:
                self.work.append(work(*a, **k))
            self.tags.append(tag(*a, **k) if tag is not None else None)
            return out
        return timed

    def total_ms(self):
        return sum(s.elapsed_time(e) for s, e in self.pairs)

    def mean_ms(self):
        if not self.pairs:
            return None
        return sum(s.elapsed_time(e) for s, e in self.pairs) / len(self.pairs)


def _traffic_file(stem):
    """Newest committed PMC traffic summary of a kernel family (profiles/rNN_<stem>, tools/pmc_bench_traffic.sh)."""
    for rnd in ('r06', 'r05', 'r04', 'r03', 'r02'):
        path = os.path.join(ROOT, 'profiles', f'{rnd}_{stem}')
        if os.path.isfile(path):
            return path
    return os.path.join(ROOT, 'profiles', 'r06_' + stem)


def _traffic_source(path):
    """`traffic` in a roofline object is NOT measured by this run (PMC counters need their own rocprofv3 passes,
    tools/pmc_bench_traffic.sh): it is read from the committed summary of those passes, and says so."""
    return 'committed PMC passes of this command, not this run: ' + os.path.relpath(path, ROOT)


def build_model(args, device):
    import contextlib
    import io
    from lavila.models import models
    torch.manual_seed(0)
    with contextlib.redirect_stdout(io.StringIO()):
        model = getattr(models, args.model)(num_frames=args.frames, pretrained=False, project_embed_dim=256,
                                            temperature_init=0.07, drop_path_rate=0.0)
    g = torch.Generator().manual_seed(1)
    with torch.no_grad():       # shipped init leaves temporal attention an exact no-op: randomise it (std .02)
        for n, p in model.named_parameters():
            if 'timeattn' in n or n.endswith('temporal_embed'):
                p.copy_(torch.randn(p.shape, generator=g) * 0.02)
    return model.to(device)


def synthetic_batch(batch, frames, img, seed=1234, real_tokens=32, ctx=77):
    """frames randn [B,3,F,H,W] f32 (stand for mean/std-normalised pixels); tokens [B,77] = SOT,
    30 random ids, EOT (49407 = the largest id, so argmax finds it: models.py:160) at position 31, zero padding. The
    bench's own generator (tests/test_oracle_golden.py pins it to the oracle's copy, which the parity tests use)."""
    g = torch.Generator().manual_seed(seed)
    video = torch.randn(batch, 3, frames, img, img, generator=g)
    tokens = torch.zeros(batch, ctx, dtype=torch.long)
    tokens[:, 0] = 49406
    tokens[:, 1:real_tokens - 1] = torch.randint(1, 49406, (batch, real_tokens - 2), generator=g)
    tokens[:, real_tokens - 1] = 49407
    return video, tokens


def synthetic(args, rank, device, img):
    video, tokens = synthetic_batch(args.batch, args.frames, img, seed=1234 + rank)
    return video.to(device), tokens.to(device)


def _cpu_model():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def cpu_baseline(args, model, img):
    """Times the CPU oracle (port of the reference path, float32) on a bounded sample: fwd + loss + bwd of one batch
    of `--cpu-batch` clips of the same shapes, 1 warm-up + 3 timed iterations."""
    from oracle import oracle as O
    cores = min(32, os.cpu_count() or 1)     # torch intra-op threads actually used (more oversubscribes)
    torch.set_num_threads(cores)
    w = {k: v.detach().float().cpu().clone().requires_grad_(v.is_floating_point())
         for k, v in model.state_dict().items()}
    video, tokens = synthetic_batch(args.cpu_batch, args.frames, img, seed=99)
    vis_heads = model.visual.blocks[0].attn.num_heads
    txt_heads = model.transformer.resblocks[0].attn.num_heads
    times = []
    budget = time.perf_counter() + 45.0
    for it in range(4):
        t0 = time.perf_counter()
        out = O.clip_forward(video, tokens, w, vis_heads, txt_heads, norm_embed=True)
        O.clip_loss(out['image_embed'], out['text_embed'], out['logit_scale'])['loss'].backward()
        dt = time.perf_counter() - t0
        if it > 0:
            times.append(dt)
        for v in w.values():
            v.grad = None
        if time.perf_counter() > budget and times:
            break
    mean = sum(times) / len(times)
    return {'value': round(args.cpu_batch / mean, 4), 'unit': 'clip-text pairs/s', 'cores': cores, 'kind': 'port',
            'cpu': f'{_cpu_model()} ({os.cpu_count()} logical cores on the box)',
            'port_vs_reference': 'the port runs at 0.79x (mean) / 0.94x (best iteration) of the reference\'s own modules on '
                                 'the same host and batch (build container, profiles/r04_cpu_port_vs_reference.json)',
            'sample': f'oracle/oracle.py fwd+loss+bwd (no optimizer), f32, batch {args.cpu_batch}, '
                      f'{args.frames}x{img}^2 clips + 77-token captions, 1 warm-up + {len(times)} timed iterations '
                      f'(mean {mean:.2f} s, best {min(times):.2f} s)'}


def narrator_cpu_baseline(args, model, tok):
    """The CPU oracle's narrator (port of the reference: encode_image + generate with the reference's schedule, the whole
    prefix and the image keys / values recomputed for every token) on ONE clip x `--returns` captions of `--length`
    tokens, float32 (greedy: the oracle has no sampler; the arithmetic per caption is the same)."""
    from oracle import oracle as O
    cores = min(32, os.cpu_count() or 1)
    torch.set_num_threads(cores)
    w = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}
    video, _ = synthetic_batch(1, args.frames, 224, seed=99)
    t0 = time.perf_counter()
    with torch.no_grad():
        img = O.narrator_encode_image(video, w, 12, 12).repeat_interleave(args.returns, dim=0)
        O.narrator_generate_greedy(img, w, 12, tok.bos_token_id, -1, tok.pad_token_id, args.length, use_cache=False)
    dt = time.perf_counter() - t0
    return {'value': round(args.returns / dt, 4), 'unit': 'captions/s', 'cores': cores, 'kind': 'port',
            'cpu': f'{_cpu_model()} ({os.cpu_count()} logical cores on the box)',
            'sample': f'oracle/oracle.py narrator_encode_image + narrator_generate_greedy(use_cache=False), f32: 1 clip of '
                      f'{args.frames}x224^2, {args.returns} captions x {args.length} tokens, one run ({dt:.1f} s)'}


def narrator_main(args, world, rank, device):
    """captions/s of the narrator (BASELINE configs[4]): one step = encode_image of `--batch` clips + generate() of
    `--returns` nucleus samples per clip (top_p 0.95, temperature 0.7: main_infer_narrator.py:53-62) of `--length` tokens,
    fp16 parameters / clips (`--use-half`), bf16 inside; random-init weights with live gates, synthetic clips. Ranks are
    independent replicas (the captioning drivers shard clips over ranks and never communicate)."""
    import contextlib
    import io
    import types
    from lavila.models import models
    with warnings.catch_warnings(), contextlib.redirect_stdout(io.StringIO()):
        warnings.simplefilter('ignore')
        torch.manual_seed(0)
        model = models.VCLM_OPENAI_TIMESFORMER_BASE_GPT2(gated_xattn=True, num_frames=args.frames)
    with torch.no_grad():
        for b in model.text_decoder.transformer.h:          # the shipped zeros would switch the image path off
            b.alpha_cattn.fill_(0.5)
            b.alpha_dense.fill_(0.5)
    tok = types.SimpleNamespace(bos_token_id=50256, eos_token_id=50256, pad_token_id=50256)
    cpu = None
    if world == 1 and rank == 0 and not args.no_cpu_baseline:
        cpu = narrator_cpu_baseline(args, model, tok)
    model = model.to(device).eval().half()
    clips = torch.randn(args.batch, 3, args.frames, 224, 224, device=device, generator=None).half()
    kw = dict(max_text_length=args.length, top_k=None, top_p=0.95, temperature=0.7, num_return_sequences=args.returns)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    decode_ms = []

    def step():
        with torch.no_grad():
            img = model.encode_image(clips)
            ev[0].record()
            out = model.generate(img, tok, **kw)
            ev[1].record()
        return out

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ids, ppl = step()
        torch.cuda.synchronize()
        decode_ms.append(ev[0].elapsed_time(ev[1]))
    fence()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], device=device, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    if rank == 0:
        rows, steps_tok = args.batch * args.returns, args.length - 1
        dec = model.text_decoder
        n_dec = sum(p.numel() for n, p in dec.named_parameters() if not n.startswith('lm_head'))
        alg = 2 * n_dec + 12 * args.batch * 256 * 1536 * 2 + rows * 12 * (steps_tok / 2) * 1536 * 2 + rows * 50432 * 2
        step_ms = sum(decode_ms) / len(decode_ms) / steps_tok
        ach = alg / (step_ms * 1e-3) / 1e9
        line = {
            'metric': 'narrator captions/s (whole node), VCLM_OPENAI_TIMESFORMER_BASE_GPT2 4x224^2, encode_image + generate',
            'value': round(world * rows * args.steps / elapsed, 2), 'unit': 'captions/s', 'n_gpus': world,
            'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(1e3 * elapsed / args.steps, 3),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'bf16', 'data': 'synthetic',
            'config': {'workload': f'BASELINE configs[4]: {args.batch} clips x {args.returns} nucleus samples (top_p 0.95, '
                                   f'temperature 0.7) x {args.length} tokens per step, fp16 parameters / clips, bf16 inside, '
                                   'key/value-cached decode replayed as one hipGraph per token',
                       'parallelism': f'{world} independent replicas', 'decode_ms_per_token_step': round(step_ms, 4),
                       'finite_perplexities': bool(torch.isfinite(ppl).all()), 'caption_shape': list(ids.shape)},
            'roofline': {'bound': 'hbm', 'kernel': 'one decode step = hipGraph replay of 172 kernels (Conv1Ds on '
                                                   'lvl_linear_skinny, cross / self attention, fused add+LayerNorm, lm_head) '
                                                   '+ lvl_sample_next_token',
                         'achieved': round(ach, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(ach / HBM_PEAK_GBS, 4),
                         'traffic': json.load(open(_traffic_file('narrator_traffic.json')))['total_bytes_per_token_step']
                         if (args.batch, args.returns) == (64, 1) and os.path.isfile(_traffic_file('narrator_traffic.json'))
                         else None,
                         'avg_ms': round(step_ms, 4), 'launches': len(decode_ms) * steps_tok,
                         'alg_bytes_per_launch': int(alg),
                         'note': 'algorithmic bytes of a token step: decoder weights once (bf16), the image keys / values of '
                                 'the 12 cross-attentions, the self-attention caches at half fill, the logits; the step is '
                                 'bound by its dependent launches, not by HBM (DESIGN.md section 4)'},
        }
        if cpu is not None:
            line['cpu_baseline'] = cpu
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _self_launch(args):
    """`python bench.py --gpus N` without a launcher: run the same command line under torch.distributed.run (one process
    per GPU, rendezvous on 127.0.0.1 and a free port) and hand its exit status back. Rank 0 of the child job prints the
    JSON line on the inherited stdout."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}',
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # dmabuf IPC: what RCCL needs on this driver
    env.setdefault('OMP_NUM_THREADS', '8')
    return subprocess.call(cmd, env=env)


def _graphed_child(args):
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), '--graph-only', '--batch', str(args.batch), '--frames',
           str(args.frames), '--model', args.model, '--steps', str(args.steps), '--no-cpu-baseline']
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    except subprocess.TimeoutExpired:
        return {'error': 'child timed out'}
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{"graphed_step"')]
    if r.returncode != 0 or not lines:
        return {'error': f'child rc={r.returncode}', 'stderr_tail': r.stderr[-300:]}
    return json.loads(lines[-1])['graphed_step']


def graph_only_main(args, device):
    """The pretraining iteration as one replayed hipGraph per caption-length bucket: the host's share of a step drops from
    ~1100 Python-issued launches to one graph launch. The caption bound comes from the host side, as a DataLoader's token
    tensor would give it (here: read once, outside the timed region)."""
    from lavila.models.loss import CLIPLoss
    from lavila_amd.graph_step import GraphedTrainStep
    model = build_model(args, device)
    img = model.visual.patch_embed.img_size[0]
    crit = CLIPLoss(use_vissl=True, cache_labels=True, rank=0, world_size=1)
    decay = [p for n, p in model.named_parameters() if not (p.ndim < 2 or 'bias' in n or 'ln' in n or 'bn' in n)]
    no_decay = [p for n, p in model.named_parameters() if (p.ndim < 2 or 'bias' in n or 'ln' in n or 'bn' in n)]
    opt = torch.optim.AdamW([{'params': decay, 'weight_decay': 0.01}, {'params': no_decay, 'weight_decay': 0.0}],
                            lr=3e-5, betas=(0.9, 0.999), eps=1e-8, fused=True, capturable=True)
    video, tokens = synthetic(args, 0, device, img)
    gstep = GraphedTrainStep(model, crit, opt, tuple(video.shape), tuple(tokens.shape), device, video_dtype=video.dtype)
    bound = int(tokens.argmax(dim=-1).max().item()) + 1
    gstep.video = video                       # resident input, as in the eager loop (no per-step copy of the frames)
    for _ in range(4):                        # eager initialisation step, capture + first replay, two more replays
        out = gstep(video, tokens.clone(), text_len=bound)
    torch.cuda.synchronize()
    n, host, t0 = max(2, args.steps), 0.0, time.perf_counter()
    for _ in range(n):
        h0 = time.perf_counter()
        out = gstep(video, tokens.clone(), text_len=bound)
        host += time.perf_counter() - h0
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    # the replay call alone with the device idle (no queue back-pressure): what hipGraphLaunch itself costs the host
    idle = []
    for _ in range(3):
        toks = tokens.clone()
        torch.cuda.synchronize()
        h0 = time.perf_counter()
        out = gstep(video, toks, text_len=bound)
        idle.append(time.perf_counter() - h0)
        torch.cuda.synchronize()
    rec = {'ms_per_step': round(1e3 * el / n, 3), 'pairs_per_s': round(args.batch * n / el, 1),
           'host_ms_in_step_call': round(1e3 * host / n, 2),
           'replay_call_ms_device_idle': round(1e3 * min(idle), 2), 'steps': n, 'caption_bucket': gstep.buckets,
           'final_loss': round(float(out['loss'].item()), 4)}
    print(json.dumps({'graphed_step': rec}), flush=True)


def main():
    args = parse()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(_self_launch(args))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1 or os.environ.get('LAVILA_BENCH_ONE_RANK_RCCL') == '1':
        # A rank of a multi-GPU job: the RCCL communicator creates streams of its own, and HIP maps all streams of a process
        # onto 4 hardware queues by default -- the text tower's side stream then shares a queue with the main stream and its
        # overlap (worth 4 % of the step) is gone: +4.4 % step time with a communicator alive and nothing on the wire, 0.0 %
        # with 8 queues (same-box A/B, profiles/r05_one_rank_group_bisect.txt). Read by the HIP runtime at its
        # initialisation, so it is set before the first device call of the process. INTEGRATION.md section 4.
        os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
    if world != args.gpus:
        raise SystemExit(f'bench.py: --gpus {args.gpus} but the launcher set WORLD_SIZE={world}')
    assert torch.cuda.is_available(), 'bench.py needs MI355X GPUs (there is no CPU path)'
    ndev = torch.cuda.device_count()
    dev_index = local_rank % ndev          # = local_rank on a full node; lets a 1-GPU box rehearse N>1
    torch.cuda.set_device(dev_index)
    device = torch.device('cuda', dev_index)
    rehearsal = None
    force_group = world == 1 and os.environ.get('LAVILA_BENCH_ONE_RANK_RCCL') == '1'
    if os.environ.get('LAVILA_BENCH_EARLY_TEXT_STREAM') == '1':
        # bisect switch: create the text tower's side stream BEFORE the communicator creates its own streams (HIP maps
        # streams onto a small number of hardware queues round-robin in creation order)
        from lavila_amd import models as _m0
        _m0._text_stream(device)
    if force_group:
        # A/B on a single GPU: the multi-GPU code path (RCCL process group, DistributedDataParallel with its bucket
        # all-reduce, device tile counters) with ONE rank -- the collectives run on the real library, their data path is a
        # device copy. Not the default and not a scaling measurement: config.note says so.
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        os.environ.setdefault('LAVILA_DYNAMIC_TILES', '1')
        # bisect switches of the group's own cost (profiles/r05_one_rank_group_bisect.txt): another backend, lazy
        # communicator creation (no device_id: the communicator is only built by the first collective)
        backend1 = os.environ.get('LAVILA_BENCH_GROUP_BACKEND', 'nccl')
        if backend1 != 'nccl' or os.environ.get('LAVILA_BENCH_LAZY_INIT') == '1':
            dist.init_process_group(backend1, rank=0, world_size=1)
        else:
            dist.init_process_group('nccl', rank=0, world_size=1, device_id=device)
        rehearsal = 'ONE-RANK RCCL GROUP (LAVILA_BENCH_ONE_RANK_RCCL=1): DDP + RCCL call sites + tile counters on one GPU'
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        # 'nccl' = RCCL over xGMI (one rank per GPU). Fewer devices than ranks (a 1-GPU box rehearsing the N>1 path):
        # RCCL cannot put two ranks on one device, gloo carries the tensors through the host instead.
        backend = os.environ.get('LAVILA_DIST_BACKEND', 'nccl' if ndev >= world else 'gloo')
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=device)
        else:
            dist.init_process_group(backend)
            rehearsal = f'REHEARSAL: {world} ranks on {ndev} device(s) over {backend} -- not a scaling measurement'
            # Ranks that SHARE a device keep both towers on one stream: with the text tower on a second HIP stream each
            # process owns two compute queues, and a queue waiting for an event of its sibling burns whole time slices of
            # the hardware scheduler while the other process holds the device -- single steps then take 3-20 s at
            # random (profiles/r04_2rank_rehearsal_bisect.json: 8523 ms / step with, 185 ms without; the caption trim and
            # the tile counters make no difference). One process per device (every real run) is unaffected.
            os.environ.setdefault('LAVILA_TEXT_STREAM', '0')
            rehearsal += '; ranks share a device: both towers on one stream (LAVILA_TEXT_STREAM=0)'

    if args.workload == 'narrator':
        if args.batch == 256:
            args.batch = 64                # main_infer_narrator.py:48
        return narrator_main(args, world, rank, device)

    if args.graph_only:
        return graph_only_main(args, device)

    from lavila_amd import ops
    from lavila.models.loss import CLIPLoss
    model = build_model(args, device)
    img = model.visual.patch_embed.img_size[0]
    timer = KernelTimer()
    ops.divided_attn_fwd_raw = timer.wrap(ops.divided_attn_fwd_raw, lambda qkv, f, n, h, mode: mode == 0)   # space
    # the space-mode attention BACKWARD call (fused kernel + cls-gradient finalize + the bias-gradient column sums it
    # carries): the kernel furthest below its roof on this configuration, reported as `roofline_hbm_bwd`
    btimer = KernelTimer()
    ops._DividedAttnFn.backward = staticmethod(btimer.wrap(ops._DividedAttnFn.backward,
                                                           lambda ctx, *g: ctx.cfg[4] == 0 and ctx.cfg[0] == args.batch))
    # every all-token video-tower launch (M = local batch x tokens per clip rows, whatever the local batch is) of the
    # two hand-written MFMA GEMM families; the text tower's launches (<= 77 rows per caption, on the second stream) and
    # the cls-row launches of the last block (one row per clip) are not in the aggregate
    video_rows = args.batch * (1 + args.frames * model.visual.patches_per_frame)
    gtimer = KernelTimer()
    def _epilogue_of(x, w, bias=None, epilogue=0, *a, **k):
        return epilogue
    ops.linear_tn_raw = gtimer.wrap(ops.linear_tn_raw, lambda x, w, *a, **k: x.shape[0] == video_rows,
                                    work=lambda x, w, *a, **k: 2.0 * x.shape[0] * w.shape[0] * w.shape[1],
                                    tag=_epilogue_of)
    wtimer = KernelTimer()
    ops.linear_wgrad_raw = wtimer.wrap(ops.linear_wgrad_raw, lambda dy, x, *a, **k: dy.shape[0] == video_rows,
                                       work=lambda dy, x, *a, **k: 2.0 * dy.shape[0] * dy.shape[1] * x.shape[1])
    net = model
    if (world > 1 or force_group) and os.environ.get('LAVILA_BENCH_NO_DDP') != '1':     # NO_DDP: bisecting the wrapper's cost
        net = torch.nn.parallel.DistributedDataParallel(model, device_ids=[dev_index], bucket_cap_mb=200,
                                                        gradient_as_bucket_view=os.environ.get('LAVILA_BENCH_NO_BUCKET_VIEW') != '1')
    if net is not model and os.environ.get('LAVILA_BENCH_DDP_HOOK', 'none') != 'none':
        # DDP's default path divides every gradient by the world size in its own elementwise launch from the autograd hook
        # (180 four-microsecond launches per step here: profiles/r05_bench_kernel_stats_ddp_serial.csv); with a communication
        # hook registered DDP leaves the scaling to the hook, which divides the BUCKET once (4 launches per step).
        # allreduce_hook is torch's own restatement of the default all-reduce; bf16_compress_hook also halves the bytes on
        # xGMI (gradients cross the links in bf16, are accumulated into the float32 buckets on arrival). INTEGRATION.md section 4.
        which = os.environ['LAVILA_BENCH_DDP_HOOK']
        if which == 'builtin':
            # the reducer's own C++ all-reduce hook: one division per BUCKET inside the reducer, no Python callback per
            # bucket (the Python allreduce_hook measured slower than the default path: 191 vs 172 ms, round 5)
            net._register_builtin_comm_hook(dist.BuiltinCommHookType.ALLREDUCE)
        else:
            from torch.distributed.algorithms.ddp_comm_hooks import default_hooks as _hooks
            net.register_comm_hook(None, {'allreduce': _hooks.allreduce_hook, 'bf16': _hooks.bf16_compress_hook}[which])
    crit = CLIPLoss(use_vissl=True, cache_labels=True, rank=rank, world_size=world)
    decay = [p for n, p in model.named_parameters() if not (p.ndim < 2 or 'bias' in n or 'ln' in n or 'bn' in n)]
    no_decay = [p for n, p in model.named_parameters() if (p.ndim < 2 or 'bias' in n or 'ln' in n or 'bn' in n)]
    opt = torch.optim.AdamW([{'params': decay, 'weight_decay': 0.01}, {'params': no_decay, 'weight_decay': 0.0}],
                            lr=3e-5, betas=(0.9, 0.999), eps=1e-8, fused=True)
    video, tokens = synthetic(args, rank, device, img)
    amp = torch.bfloat16 if args.dtype == 'bf16' else None

    # The caption bound of the text tower's trim comes from the HOST tokens (the loader has them
    # on the host before the upload, main_pretrain.py:489-498) -- computed inside every timed step, as a training loop
    # would. LAVILA_BENCH_DEVICE_READBACK=1 (and the `device_readback` record of every default run) is the unmodified
    # reference loop: the tower reads the longest caption back from the device, one host sync per step.
    import contextlib
    from lavila_amd import models as _mm
    tokens_host = tokens.cpu()
    host_bound_default = (os.environ.get('LAVILA_BENCH_DEVICE_READBACK') != '1' and
                          os.environ.get('LAVILA_TEXT_TRIM', '1') != '0')

    def step(host_bound=None):
        # a training loop receives a NEW token tensor every iteration (main_pretrain.py:489-498)
        toks = tokens if args.reuse_tokens else tokens.clone()
        use_bound = host_bound_default if host_bound is None else host_bound
        with (_mm.fixed_text_length(_mm.caption_bound(tokens_host)) if use_bound else contextlib.nullcontext()):
            with torch.autocast('cuda', dtype=amp, enabled=amp is not None):
                out = net(video, toks, use_checkpoint=args.checkpoint, norm_embed=True)
                loss = crit(out)['loss']
        loss.backward()
        opt.step()
        opt.zero_grad(set_to_none=True)
        model.logit_scale.data.clamp_(0, 4.6052)       # main_pretrain.py:527-528
        return loss

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    verbose = os.environ.get('LAVILA_BENCH_VERBOSE') == '1'      # per-step progress on stderr (diagnosing a slow rank)
    for i in range(args.warmup):
        h0 = time.perf_counter()
        step()
        if verbose:
            torch.cuda.synchronize()
            print(f'[bench rank {rank}] warm-up step {i}: {1e3 * (time.perf_counter() - h0):.1f} ms', file=sys.stderr, flush=True)
    fence()
    torch.cuda.reset_peak_memory_stats(device)
    # HIP-event pairs around the three timed kernel families on every `event_stride`-th step of the timed region (an event
    # is a marker packet between two kernels: ~460 of them cost a step about 1 ms -- measured: 169.3-169.7 ms with events
    # on every step against 168.2-168.8 for the same loop without -- so the instrumentation runs on a sample of the steps;
    # rocprofv3's per-kernel averages of the same command are in profiles/)
    stride = max(1, int(args.event_stride))
    timed_steps = 0
    t0 = time.perf_counter()
    host_s = 0.0
    host_steps = []                # per-step host time inside step(): an outlier step shows here (no extra syncs)
    for i_step in range(args.steps):
        on = (not args.no_events) and i_step % stride == 0
        timer.enabled = wtimer.enabled = gtimer.enabled = btimer.enabled = on
        timed_steps += int(on)
        h0 = time.perf_counter()
        loss = step()
        host_steps.append(time.perf_counter() - h0)
        host_s += host_steps[-1]
        if verbose:
            print(f'[bench rank {rank}] step: host {1e3 * host_steps[-1]:.1f} ms', file=sys.stderr, flush=True)
    fence()
    elapsed = time.perf_counter() - t0
    timer.enabled = wtimer.enabled = gtimer.enabled = btimer.enabled = False
    t = torch.tensor([elapsed], device=device, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    final_loss = float(loss.item())
    # device memory of the timed region (the caching allocator's own high-water marks; HBM3E per MI355X: 288 GB)
    peak_alloc_gb = round(torch.cuda.max_memory_allocated(device) / 1e9, 2)
    peak_reserved_gb = round(torch.cuda.max_memory_reserved(device) / 1e9, 2)

    # the cost of the text tower's per-step caption-length read-back, made visible: the same step with the trim (and
    # with it the host read) switched off -- all 77 positions computed, the host free to run ahead
    no_trim = None
    if os.environ.get('LAVILA_TEXT_TRIM', '1') != '0' and not args.no_events:
        from lavila_amd import models as _m
        _m._TEXT_TRIM = False
        step(host_bound=False)
        fence()
        n2, h2, t2 = max(2, min(4, args.steps)), 0.0, time.perf_counter()
        for _ in range(n2):
            h0 = time.perf_counter()
            step(host_bound=False)
            h2 += time.perf_counter() - h0
        fence()
        no_trim = {'ms_per_step': round(1e3 * (time.perf_counter() - t2) / n2, 3),
                   'host_ms_in_step_call': round(1e3 * h2 / n2, 1), 'steps': n2}
        _m._TEXT_TRIM = True

    # the unmodified reference loop: the caption length read back from the device inside the text tower, every step
    readback = None
    if host_bound_default and not args.no_events:
        step(host_bound=False)
        fence()
        n4, h4, t4 = max(2, min(4, args.steps)), 0.0, time.perf_counter()
        for _ in range(n4):
            h0 = time.perf_counter()
            step(host_bound=False)
            h4 += time.perf_counter() - h0
        fence()
        readback = {'ms_per_step': round(1e3 * (time.perf_counter() - t4) / n4, 3),
                    'host_ms_in_step_call': round(1e3 * h4 / n4, 1), 'steps': n4}

    # the same step with the LAST block of both towers computed on every row, as the reference does (the default computes
    # only the rows that reach the output -- cls / EOT -- which is exact: DESIGN.md section 4, "Last block")
    full_last = None
    if os.environ.get('LAVILA_CLS_LAST', '1') != '0' and not args.no_events:
        from lavila_amd import timesformer as _t
        _t.CLS_ONLY_LAST_BLOCK = False
        step()
        fence()
        n3, t3 = max(2, min(4, args.steps)), time.perf_counter()
        for _ in range(n3):
            step()
        fence()
        full_last = {'ms_per_step': round(1e3 * (time.perf_counter() - t3) / n3, 3), 'steps': n3}
        _t.CLS_ONLY_LAST_BLOCK = True

    # the same iteration captured as one hipGraph per caption-length bucket and replayed (lavila_amd/graph_step.py), in a
    # child process (its own model, optimizer and allocator: nothing in it can touch the numbers above)
    graphed = None
    if (os.environ.get('LAVILA_BENCH_GRAPH', '1') != '0' and world == 1 and not force_group and not args.no_events
            and amp is not None and rank == 0):
        try:
            graphed = _graphed_child(args)
        except Exception as exc:                # a variant record must never take the headline down
            graphed = {'error': f'{type(exc).__name__}: {exc}'[:300]}

    if rank == 0:
        B, Fr = args.batch, args.frames
        N = model.visual.patches_per_frame
        T = 1 + Fr * N
        D = model.visual.embed_dim
        esize = 2 if amp is not None else 4
        alg_bytes = B * (T * 3 * D + T * D) * esize            # read packed qkv once, write out once
        kms = timer.mean_ms()
        roofline_hbm = None
        traffic = None
        tfile = _traffic_file('traffic_space_fwd.json')   # PMC pass of the same kernel/shape
        if os.path.isfile(tfile) and (B, Fr, N, D) == (256, 4, 196, 768) and amp is not None:
            traffic = json.load(open(tfile))['traffic_bytes_per_launch']
        if kms:
            achieved = alg_bytes / (kms * 1e-3) / 1e9
            roofline_hbm = {'bound': 'hbm', 'kernel': 'lvl_divided_attn_fwd[space]', 'achieved': round(achieved, 1),
                            'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(achieved / HBM_PEAK_GBS, 4),
                            'traffic': traffic, 'traffic_source': _traffic_source(tfile) if traffic else None,
                            'avg_ms': round(kms, 4), 'launches': len(timer.pairs),
                            'timed_steps': f'{timed_steps} of {args.steps} (every {stride}th step of the timed region)',
                            'alg_bytes_per_launch': alg_bytes}
        roofline_hbm_bwd = None
        bms = btimer.mean_ms()
        if bms:
            balg = B * (2 * T * 3 * D + 2 * T * D) * esize      # read qkv, out, dout; write dqkv
            btraffic = None
            bfile = _traffic_file('traffic_space_bwd.json')
            if os.path.isfile(bfile) and (B, Fr, N, D) == (256, 4, 196, 768) and amp is not None:
                btraffic = json.load(open(bfile))['traffic_bytes_per_launch']
            bach = balg / (bms * 1e-3) / 1e9
            roofline_hbm_bwd = {'bound': 'hbm', 'kernel': 'lvl_divided_attn_bwd_bias[space] (space_bwd_fused_kernel + cls-gradient '
                                                          'finalize + the bias-gradient column sums of the same call)',
                                'achieved': round(bach, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                                'frac': round(bach / HBM_PEAK_GBS, 4), 'traffic': btraffic,
                                'traffic_source': _traffic_source(bfile) if btraffic else None,
                                'avg_ms': round(bms, 4), 'launches': len(btimer.pairs), 'alg_bytes_per_launch': balg}

        def _by_epilogue(tm):
            """The launches priced per epilogue of lvl_linear_tn (include/lavila_hip.h: 0 bias, 3 bias + residual -- +1 read
            of a [rows, N] tensor, the LayerNorm pass it replaces is gone from the step --, 4 bias + QuickGELU writing the
            activation AND its derivative, 5 multiply by the stored derivative + column sums)."""
            names = {0: 'bias', 1: 'bias_quickgelu_preact', 2: 'quickgelu_bwd_preact', 3: 'bias_residual',
                     4: 'bias_quickgelu_deriv', 5: 'mul_aux_colsum'}
            out = {}
            for code in sorted(set(tm.tags)):
                idx = [i for i, t in enumerate(tm.tags) if t == code]
                ms = sum(tm.pairs[i][0].elapsed_time(tm.pairs[i][1]) for i in idx)
                fl = sum(tm.work[i] for i in idx)
                if idx and ms > 0:
                    out[names.get(code, str(code))] = {'launches': len(idx), 'avg_ms': round(ms / len(idx), 4),
                                                       'frac': round(fl / (ms * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS, 4)}
            return {'by_epilogue': out} if len(out) > 1 else {}

        def mfma_roofline(tm, kernel, tfile):
            if not tm.pairs:
                return None
            traffic = note = None
            path = _traffic_file(tfile)
            if os.path.isfile(path) and (B, Fr, N, D) == (256, 4, 196, 768):
                tj = json.load(open(path))
                traffic, note = tj['traffic_bytes_per_launch'], tj.get('note')
            tot_ms, tot_fl = tm.total_ms(), sum(tm.work)
            ach = tot_fl / (tot_ms * 1e-3) / 1e12
            return {'bound': 'mfma', 'kernel': kernel, 'achieved': round(ach, 1), 'peak': MFMA_PEAK_TFLOPS,
                    'unit': 'TFLOP/s', 'frac': round(ach / MFMA_PEAK_TFLOPS, 4), 'traffic': traffic,
                    'traffic_source': _traffic_source(path) if traffic else None,
                    'avg_ms': round(tot_ms / len(tm.pairs), 4), 'launches': len(tm.pairs),
                    'timed_steps': f'{timed_steps} of {args.steps} (every {stride}th step of the timed region)',
                    'alg_flops_per_launch': round(tot_fl / len(tm.pairs)), 'traffic_note': note,
                    # not the roofline peak: what an MFMA + fragment-read + barrier + LDS-DMA loop at this kernel's
                    # rates sustains on random bf16 data on this part (power-limited), measured by
                    # tools/probes/mfma_ceiling.hip
                    'measured_loop_ceiling': {'value': 1434.0, 'unit': 'TFLOP/s',
                                              'source': 'profiles/r02_mfma_ceiling_microbench.txt'},
                    **_by_epilogue(tm)}
        # dominant kernel: the forward / input-gradient GEMM, aggregated over all its launches in the timed region
        roofline = mfma_roofline(gtimer, 'lvl_linear_tn (gemm_tn_kernel<0|1|2|3>, all video-tower forward and '
                                 'input-gradient GEMMs incl. their fused epilogues)', 'traffic_gemm_tn.json') \
            or roofline_hbm
        roofline_wgrad = mfma_roofline(wtimer, 'lvl_linear_wgrad (wgrad_kernel<4,2,6,6,false> + its partial-tile '
                                       'reduction, all video-tower weight gradients)', 'traffic_wgrad.json')
        tower = 'TSF-L/14' if 'LARGE' in args.model else 'TSF-B/16'
        line = {
            'metric': f'clip-text pairs/s (whole node), {tower} {Fr}x{img}^2 + CLIP text tower, fwd+loss+bwd+AdamW'
                      + (' (caption bound from the host tokens: INTEGRATION.md 1c one-line driver change; unmodified loop = '
                         'config.device_readback)' if host_bound_default else ''),
            'value': round(world * B * args.steps / elapsed, 2), 'unit': 'clip-text pairs/s', 'n_gpus': world,
            'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(1e3 * elapsed / args.steps, 3),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'bf16' if amp is not None else 'f32', 'data': 'synthetic',
            'config': {'workload': f'{args.model}: {tower} {Fr}x{img}^2 clips + 32-token captions (77 ctx), '
                                   f'local batch {B}, global batch {world * B}, contrastive all-gather over RCCL',
                       'global_batch': world * B, 'parallelism': f'dp{world}', 'final_loss': round(final_loss, 4),
                       'linear_gemms': 'lvl_linear_tn / lvl_linear_wgrad (hand-written MFMA; no library GEMM on the path)',
                       'peak_mem_gb': peak_alloc_gb, 'peak_mem_reserved_gb': peak_reserved_gb,
                       'use_checkpoint': bool(args.checkpoint),
                       # wall time the host spends INSIDE step() -- not host work: once the launch queue is full every call
                       # blocks until the device has drained a step's worth (back-pressure; the `device_readback` record's
                       # also includes the text tower's caption-length read-back). `host_work_ms_per_step` is the smallest
                       # step() call of the region, i.e. what the enqueue itself costs when nothing blocks it
                       'host_ms_in_step_call': round(1e3 * host_s / args.steps, 1),
                       'host_work_ms_per_step': round(1e3 * min(host_steps), 1) if host_steps else None,
                       'host_ms_of_each_step': [round(1e3 * h, 1) for h in host_steps[:32]],
                       'text_trim_off': no_trim,
                       'caption_bound': ('models.caption_bound(host tokens) + fixed_text_length inside every timed step '
                                         f'(bound {_mm.caption_bound(tokens_host)} of 77 positions; INTEGRATION.md section 1c)')
                                        if host_bound_default else 'read back from the device by the text tower',
                       'device_readback': readback,
                       'full_last_block': full_last,
                       # the same iteration as one replayed hipGraph per caption-length bucket (lavila_amd/graph_step.py)
                       'graphed_step': graphed,
                       'exact_work_elimination': 'text positions behind the longest caption (causal: unread) and, in the LAST '
                                                 'block of each tower, the projection / LayerNorm / MLP rows that do not reach '
                                                 'the output (only norm(x)[:,0] / the EOT row leave the towers) are not computed; '
                                                 'outputs and every parameter gradient equal the reference (tests/golden/model_*.pt). '
                                                 'text_trim_off / full_last_block give the step time without each',
                       'tile_schedule': 'dynamic (device tile / chunk counters)' if ops.dynamic_tiles()
                                        else 'static (single GPU; the counters switch on inside a process group)',
                       'parity_note': 'benched path = bf16 MFMA kernels: bit-exact on integer / one-hot operands '
                                      '(tests/test_gpu_parity_bf16.py); TSF-B step vs the f32 oracle: max |d logit| 0.015, '
                                      'embeddings 1.0e-2, aggregate gradient 3.7e-2 relative L2 (bf16-inherent; 0.6e-2 / '
                                      '2.0e-2 with LAVILA_RESIDUAL_F32=1); labels / argmax exact. The 1e-3 f32 bar is met '
                                      'by the SAME kernels in f32-class mode (bf16 hi/lo operand images, 3 MFMAs per product, '
                                      'f32 results; no library GEMM, no generic attention): max |d logit| 2.3e-5 (config 1), '
                                      '3.3e-5 (TSF-B 4x224^2 B=8), 1.6e-5 / 1.0e-5 (TSF-L/14 224 / 336) against the '
                                      'reference, forward + backward (tests/test_gpu_f32_class.py)'},
            'roofline': roofline,
            'roofline_wgrad': roofline_wgrad,
            'roofline_hbm': roofline_hbm,
            'roofline_hbm_bwd': roofline_hbm_bwd,
        }
        if rehearsal:
            line['config']['note'] = rehearsal
        if world == 1 and not args.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline(args, model, img)
        print(json.dumps(line), flush=True)
    if world > 1 or force_group:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
