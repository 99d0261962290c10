"""The pretraining iteration as ONE hipGraph: forward, loss, backward, optimizer step and the logit-scale clamp of
`main_pretrain.py:482-530` captured once per caption-length bucket and replayed, instead of ~1100 launches enqueued from
Python every step (61-73 ms of host work per 172 ms step at TSF-B, local batch 256).

    step = GraphedTrainStep(model, criterion, optimizer, video_shape, tokens_shape, device)
    for it, (frames, tokens) in enumerate(loader):          # host tensors, as the reference's DataLoader yields them
        step.set_lr(lr_schedule[it])                        # main_pretrain.py:476-480 (a device scalar: no re-capture)
        out = step(frames, tokens)                          # H2D copy into the static buffers + one graph launch
        # out['loss'], out['clip_acc'] ...: device tensors owned by the graph, valid until the next call

What has to be static for a replay, and how each piece of the eager step gets there:
  * inputs: `step.video` / `step.tokens` are the buffers the graph reads; `__call__` copies into them (for host tensors
    that IS the `.cuda(non_blocking=True)` of main_pretrain.py:497-498, not an extra pass);
  * the text tower's caption trim needs the longest caption on the host. The eager path reads it back from the device
    (one sync per step); here it is taken from the HOST token tensor before the upload (free), rounded up to
    `text_bucket` positions, and one graph is kept per rounded length -- any length >= the longest caption gives
    bit-identical rows (causal mask; models.CLIP.encode_text);
  * the optimizer must be `capturable=True` (torch's fused AdamW then keeps `step` on the device); its learning rates
    become device scalars so that the per-iteration schedule is a copy, not a re-capture;
  * gradients are graph-owned: allocated inside the capture (`zero_grad(set_to_none=True)` before it), rewritten by every
    replay; the weight copies of the GEMMs are cast inside the graph (ops.weight_copies bypasses its cache under capture).
The first call is a real EAGER step -- optimizer state and kernel attributes are initialised there -- so no iteration is
lost or repeated: call k performs training step k.

A caller that ran eager iterations of the same model before (on another stream) must drop what it kept of them (the loss
tensor holds the autograd graph) before the first capture; torch warns about the stream mismatch when it has not.

What a replay may read: its own pool, the parameters / optimizer state / static inputs -- nothing else. That is tested by
filling every free block of the allocator with NaN between replays (tests/test_gpu_graph_step.py, tests/test_gpu_ddp.py).
The test found, in round 5, that hipMemsetAsync NODES are not reliable in a replayed graph on this ROCm build (their fill
pattern came back from recycled memory: "zeroed" accumulators held {0, NaN, 0, 0} repeated, and a training run drifted or
went NaN depending on what the allocator handed out next to the graph); the C-ABI zeroes with a kernel since then
(csrc/common.h: lvl_zero_f32; tests/test_boundary_cpu.py keeps memset / memcpy calls out of csrc).

Limits, loudly: `update_freq` (gradient accumulation) other than 1, stochastic depth / dropout (the RNG offset of a
replay is not advanced here), a changing batch shape, activation checkpointing, and DistributedDataParallel WRAPPERS:
DDP's bucket all-reduces would be launched from inside the captured backward, and with collectives pending torch's NCCL
watchdog thread polls their events while the stream is capturing -- this ROCm build answers hipErrorStreamCaptureUnsupported
and aborts (tried on a one-rank RCCL group: tools/gpu_runs/gpu_r4o.sh).

Multi-GPU jobs (round 5): hand over the BARE module with a process group initialised. The step then keeps every
collective OUTSIDE the captures: the iteration is recorded as a chain  graph | collective | graph | ... | graph  -- the
capture is ended in front of each collective (the two all-gathers of the contrastive loss, lavila_amd/loss.py, and the
gradient all-reduce that DDP would have issued), the collective runs eagerly between two replays on the step's stream, and
the next segment is captured into the same memory pool behind it (_Segments). Gradients are averaged over the ranks as DDP
does; parameters are broadcast from rank 0 once. What a rank loses against DDP is the overlap of the gradient all-reduce
with the backward (one coalesced all-reduce after it: ~4 ms of 172 on an 8-GPU xGMI ring); what it
gains is a host that issues ~5 launches per step instead of ~1100.
"""
import os
import threading

import torch
import torch.distributed as dist

from . import models as _models
from . import ops

_active = threading.local()

# hipGraphNodeType, in the runtime's order (hip_runtime_api.h)
NODE_TYPES = ('kernel', 'memcpy', 'memset', 'host', 'graph', 'empty', 'wait_event', 'event_record', 'sem_signal', 'sem_wait',
              'mem_alloc', 'mem_free', 'memcpy_from_symbol', 'memcpy_to_symbol')
_hip_rt = None


def _hip_runtime():
    """The libamdhip64 this process already runs on (torch's), through ctypes: graph introspection only."""
    global _hip_rt
    if _hip_rt is None:
        import ctypes
        path = None
        with open('/proc/self/maps') as f:
            for line in f:
                if 'libamdhip64' in line:
                    path = line.split()[-1]
                    break
        if path is None:
            raise RuntimeError('libamdhip64 is not mapped into this process')
        _hip_rt = ctypes.CDLL(path)
    return _hip_rt


def graph_node_types(graph):
    """{node type: count} of a captured torch.cuda.CUDAGraph built with keep_graph=True (hipGraphGetNodes +
    hipGraphNodeGetType). What it is for: a hipMemsetAsync recorded as a memset NODE is not reliable under replay on this
    ROCm build (module docstring) -- the C-ABI has none, but a library call inside the captured iteration may (torch's
    embedding backward above 3072 rows: rocPRIM's radix sort zeroes its histograms that way)."""
    import ctypes
    hip = _hip_runtime()
    raw = ctypes.c_void_p(int(graph.raw_cuda_graph()))
    n = ctypes.c_size_t(0)
    if hip.hipGraphGetNodes(raw, None, ctypes.byref(n)) != 0:
        raise RuntimeError('hipGraphGetNodes failed')
    nodes = (ctypes.c_void_p * max(1, n.value))()
    if hip.hipGraphGetNodes(raw, nodes, ctypes.byref(n)) != 0:
        raise RuntimeError('hipGraphGetNodes failed')
    counts = {}
    for i in range(n.value):
        t = ctypes.c_int(-1)
        if hip.hipGraphNodeGetType(ctypes.c_void_p(nodes[i]), ctypes.byref(t)) != 0:
            raise RuntimeError('hipGraphNodeGetType failed')
        name = NODE_TYPES[t.value] if 0 <= t.value < len(NODE_TYPES) else f'type{t.value}'
        counts[name] = counts.get(name, 0) + 1
    return counts


def active_segments():
    """The _Segments object of the capture in progress on this thread (None outside GraphedTrainStep captures): what
    distributed_utils.all_gather_rows asks before it launches a collective."""
    return getattr(_active, 'seg', None)


def collective_stream():
    """The side stream a GraphedTrainStep iteration on this thread wants its collectives on (None otherwise)."""
    return getattr(_active, 'comm', None)


def run_on_collective_stream(fn):
    """Run the collective `fn` on the step's COMMUNICATION stream, ordered behind and in front of the current stream.
    Why not on the step's own stream: ProcessGroupNCCL records the completion event of a synchronous collective on the
    stream it ran on, its watchdog thread polls that event until it has seen it complete -- and HIP refuses to query an
    event whose stream is capturing at that moment (hipErrorCapturedEvent -> the watchdog throws -> the process aborts;
    seen once in three runs of the suite with the collectives on the capture stream). A stream that never captures makes
    the poll harmless whenever it happens."""
    comm = collective_stream()
    if comm is None:
        return fn()
    cur = torch.cuda.current_stream()
    comm.wait_stream(cur)
    with torch.cuda.stream(comm):
        fn()
    cur.wait_stream(comm)


# Census of the captured graphs' node types after every capture (graph_node_types); LAVILA_GRAPH_AUDIT=0 switches it off
# (the graphs are then built without keep_graph).
AUDIT_NODES = os.environ.get('LAVILA_GRAPH_AUDIT', '1') != '0'


class _Segments:
    """graph | eager op | graph | ... recorded while ONE iteration runs under capture; replay() runs them in order.
    Every segment allocates from the same private pool (tensors that cross a segment boundary -- activations saved for the
    backward, the collectives' inputs and outputs -- stay alive because the autograd graph / the recorded closures hold
    them). Captures use the thread-local error mode when a process group exists: its watchdog thread may query the events
    of the collective that just ran while the next segment is already capturing."""

    def __init__(self, pool, error_mode):
        self.pool, self.error_mode = pool, error_mode
        self.items = []
        self._cur = None

    def begin(self):
        self._cur = torch.cuda.CUDAGraph(keep_graph=True) if AUDIT_NODES else torch.cuda.CUDAGraph()
        self._cur.capture_begin(pool=self.pool, capture_error_mode=self.error_mode)

    def end(self):
        self._cur.capture_end()
        self.items.append(self._cur)
        self._cur = None

    def eager(self, fn):
        """End the running segment, run `fn` now (the values behind it are real during the capture pass too) and at every
        replay at this point of the chain, start the next segment."""
        self.end()
        fn()
        torch.cuda.synchronize()                       # nothing of the collective is pending when the capture resumes
        self.items.append(fn)
        self.begin()

    def replay(self):
        for it in self.items:
            if isinstance(it, torch.cuda.CUDAGraph):
                it.replay()
            else:
                it()

    @property
    def graphs(self):
        return sum(isinstance(it, torch.cuda.CUDAGraph) for it in self.items)


class GraphedTrainStep:
    def __init__(self, net, criterion, optimizer, video_shape, tokens_shape, device, amp_dtype=torch.bfloat16,
                 video_dtype=torch.float32, forward_kwargs=None, text_bucket=8, clamp_logit_scale=(0.0, 4.6052),
                 loss_key='loss', stream=None, eager_calls=None, broadcast_parameters=True):
        if isinstance(net, torch.nn.parallel.DistributedDataParallel):
            # measured on this build (torch 2.10 + ROCm 7, one-rank RCCL group, wrapper constructed on the capture stream,
            # 11 eager iterations first): ProcessGroupNCCL's watchdog thread polls its work events with hipEventQuery
            # while the stream is capturing -> hipErrorStreamCaptureUnsupported -> the process aborts
            raise NotImplementedError('GraphedTrainStep: a DistributedDataParallel wrapper cannot be captured on this '
                                      'torch / ROCm build (its bucket all-reduces would be launched inside the captured '
                                      'backward; the process-group watchdog then queries events during the capture and '
                                      'aborts the process). Hand over the bare module: with a process group initialised '
                                      'the step keeps the collectives between its graph segments and averages the '
                                      'gradients itself')
        if not torch.cuda.is_available():
            raise RuntimeError('GraphedTrainStep needs a HIP device (hipGraph capture)')
        for grp in optimizer.param_groups:
            if not grp.get('capturable', False):
                raise ValueError('GraphedTrainStep: construct the optimizer with capturable=True (its step counter has to '
                                 'live on the device to be replayed)')
        self.net, self.criterion, self.optimizer = net, criterion, optimizer
        self.model = net.module if hasattr(net, 'module') else net
        for m in self.model.modules():
            if isinstance(m, torch.nn.Dropout) and m.p > 0 and self.model.training:
                raise NotImplementedError('GraphedTrainStep: dropout inside a replayed graph (RNG state is not advanced)')
            if type(m).__name__ == 'DropPath' and getattr(m, 'drop_prob', 0.) > 0 and self.model.training:
                raise NotImplementedError('GraphedTrainStep: stochastic depth inside a replayed graph')
        self.device = torch.device(device)
        self.amp_dtype = amp_dtype
        self.kwargs = dict(use_checkpoint=False, norm_embed=True)
        self.kwargs.update(forward_kwargs or {})
        if self.kwargs.get('use_checkpoint'):
            # torch.utils.checkpoint saves / restores the RNG state around the recomputation (preserve_rng_state): a
            # device-state read that is illegal while the stream is capturing. With every activation of the
            # bench configuration resident the graphed step has no use for checkpointing.
            raise NotImplementedError('GraphedTrainStep: use_checkpoint=True is not capturable (activation checkpointing '
                                      'reads the RNG state during the capture); run checkpointed models with the eager loop')
        self._packed = [m for m in self.model.modules() if hasattr(m, 'invalidate_packed_weights')]
        self.text_bucket = max(1, int(text_bucket))
        self.clamp = clamp_logit_scale
        self.loss_key = loss_key
        self.video = torch.zeros(video_shape, dtype=video_dtype, device=self.device)
        self.tokens = torch.zeros(tokens_shape, dtype=torch.long, device=self.device)
        self.context = int(tokens_shape[-1])
        # learning rates as device scalars (torch's capturable AdamW reads them on the device)
        for grp in optimizer.param_groups:
            if not torch.is_tensor(grp['lr']):
                grp['lr'] = torch.tensor(float(grp['lr']), dtype=torch.float32, device=self.device)
        self._graphs = {}            # rounded caption length -> (segments, outputs)
        self.node_types = {}         # rounded caption length -> {hipGraph node type: count} (LAVILA_GRAPH_AUDIT)
        self._pool = None
        # data parallel without the DDP wrapper: collectives between graph segments, gradients averaged here
        self.distributed = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size() if self.distributed else 1
        self._comm = None            # the collectives' own stream (run_on_collective_stream), created with the step stream
        if self.distributed and broadcast_parameters:
            with torch.no_grad():
                for t in list(self.model.parameters()) + list(self.model.buffers()):
                    dist.broadcast(t.data, src=0)
        # The eager first iteration and every capture run on ONE dedicated stream: autograd pins a parameter's
        # AccumulateGrad node to the stream it was created on, and a node that outlives its iteration (kept alive by
        # anything that still references that iteration's autograd graph) would otherwise run on a stream outside the
        # capture -- which ends the capture with a crash inside hipStreamEndCapture, not with an error.
        self._stream = stream if stream is not None else torch.cuda.Stream(device=self.device)
        if self.distributed and os.environ.get('LAVILA_GRAPH_COMM_STREAM', '1') != '0':
            self._comm = torch.cuda.Stream(device=self.device)
        self._eager_left = int(eager_calls) if eager_calls is not None else 1
        self.replays = 0

    # ---- host side of one iteration -------------------------------------------------------------------------------
    def set_lr(self, lr):
        """One value for every group or one per group (main_pretrain.py:476-480: `param_group['lr'] = lr_schedule[it]`)."""
        groups = self.optimizer.param_groups
        values = [lr] * len(groups) if not isinstance(lr, (list, tuple)) else list(lr)
        if len(values) != len(groups):
            raise ValueError(f'set_lr: {len(values)} values for {len(groups)} parameter groups')
        for grp, v in zip(groups, values):
            grp['lr'].fill_(float(v))

    def caption_length(self, tokens):
        """1 + the largest EOT position (the EOT token has the highest id: models.py:158-160) of a HOST token tensor,
        rounded up to the bucket; the full context for device tensors (reading them would be the sync this class removes)."""
        if tokens.is_cuda:
            return self.context
        longest = int(tokens.argmax(dim=-1).max()) + 1
        b = self.text_bucket
        return min(self.context, (longest + b - 1) // b * b)

    def __call__(self, video, tokens, text_len=None):
        """One training iteration on (video, tokens) -- host or device tensors of the static shapes. `text_len`: a caller's
        own bound on the caption length (>= 1 + the largest EOT position), for token tensors already on the device."""
        if tuple(video.shape) != tuple(self.video.shape) or tuple(tokens.shape) != tuple(self.tokens.shape):
            raise ValueError(f'GraphedTrainStep was built for {tuple(self.video.shape)} / {tuple(self.tokens.shape)}, got '
                             f'{tuple(video.shape)} / {tuple(tokens.shape)} (a last, smaller batch runs eagerly)')
        if text_len is None:
            L = self.caption_length(tokens)
        else:
            b = self.text_bucket
            L = min(self.context, (int(text_len) + b - 1) // b * b)
        if video.data_ptr() != self.video.data_ptr():
            self.video.copy_(video, non_blocking=True)
        if tokens.data_ptr() != self.tokens.data_ptr():
            self.tokens.copy_(tokens, non_blocking=True)
        if self._eager_left > 0:
            self._eager_left -= 1
            return self._eager(L)                 # lazy initialisation happens in real steps
        entry = self._graphs.get(L)
        if entry is None:
            entry = self._graphs[L] = self._capture(L)
        if self.distributed:
            cur = torch.cuda.current_stream(self.device)
            self._stream.wait_stream(cur)
            with torch.cuda.stream(self._stream):          # graph segments on the step's stream, collectives on its side stream
                _active.comm = self._comm
                try:
                    entry[0].replay()
                finally:
                    _active.comm = None
            cur.wait_stream(self._stream)
        else:
            entry[0].replay()
        self.replays += 1
        self._parameters_changed()
        return entry[1]

    def _parameters_changed(self):
        """A replay updates the parameters on the device behind Python's back: no version counter moves and the optimizer's
        post-step hook (which bumps ops' weight-copy generation in an eager step) only ran at capture time. Without this an
        evaluation between two replays would hit ops.weight_copies' cache and run every Linear on the bf16 copies of an
        EARLIER parameter state while LayerNorms / embeddings are current. One integer increment per replay;
        packed decoder images (narrator models) are dropped the same way."""
        ops.invalidate_weight_cache()
        for m in self._packed:
            m.invalidate_packed_weights()

    # ---- the iteration itself -------------------------------------------------------------------------------------
    def _average_gradients(self):
        """What DistributedDataParallel does with its buckets, as ONE coalesced all-reduce behind the backward (the
        gradients are graph-owned tensors at fixed addresses: the recorded closure reduces them in place at every replay)."""
        # every rank must reduce the SAME list: a parameter that got no gradient on this rank only (unused on some ranks)
        # would otherwise give mismatched all-reduce sizes -- a hang, not an error
        missing = [n for n, p in self.model.named_parameters() if p.requires_grad and p.grad is None]
        if missing:
            raise RuntimeError(f'GraphedTrainStep: {len(missing)} trainable parameter(s) received no gradient in the captured '
                               f'iteration (e.g. {missing[0]}): data-parallel replay needs every rank to reduce the same '
                               'tensors -- freeze them (requires_grad=False) or use the eager DistributedDataParallel loop')
        grads = [p.grad for p in self.model.parameters() if p.requires_grad]

        def reduce():
            if dist.get_backend() == 'nccl':               # RCCL: one grouped launch over the gradients in place
                dist.all_reduce_coalesced(grads, op=dist.ReduceOp.SUM)
            else:                                          # gloo (tests on one device): through one flat buffer
                flat = torch._utils._flatten_dense_tensors(grads)
                dist.all_reduce(flat, op=dist.ReduceOp.SUM)
                torch._foreach_copy_(grads, torch._utils._unflatten_dense_tensors(flat, grads))
            if self.world > 1:
                torch._foreach_div_(grads, float(self.world))
        op = lambda: run_on_collective_stream(reduce)      # noqa: E731
        seg = active_segments()
        if seg is not None:
            seg.eager(op)
        else:
            op()

    def _iteration(self, L):
        with _models.fixed_text_length(L):
            with torch.autocast('cuda', dtype=self.amp_dtype, enabled=self.amp_dtype is not None):
                out = self.net(self.video, self.tokens, **self.kwargs)
                losses = self.criterion(out)
            losses[self.loss_key].backward()
        if self.distributed:
            self._average_gradients()
        self.optimizer.step()
        if self.clamp is not None and hasattr(self.model, 'logit_scale'):
            self.model.logit_scale.data.clamp_(*self.clamp)            # main_pretrain.py:527-528
        # detached: nothing handed out may keep this iteration's autograd graph (and its AccumulateGrad nodes) alive
        return {k: v.detach() for k, v in losses.items() if torch.is_tensor(v)}

    def _eager(self, L):
        cur = torch.cuda.current_stream(self.device)
        self._stream.wait_stream(cur)
        with torch.cuda.stream(self._stream):
            self.optimizer.zero_grad(set_to_none=True)
            _active.comm = self._comm
            try:
                out = self._iteration(L)
            finally:
                _active.comm = None
            self.optimizer.zero_grad(set_to_none=True)
        cur.wait_stream(self._stream)
        for v in out.values():
            v.record_stream(cur)
        return out

    def _capture(self, L):
        self.optimizer.zero_grad(set_to_none=True)
        torch.cuda.synchronize(self.device)
        if self._pool is None:
            self._pool = torch.cuda.graph_pool_handle()
        # one pool for every bucket: nothing a graph allocates is read after the next call (outputs are per call, the
        # gradients are rewritten by each replay before its optimizer step reads them)
        if not self.distributed:
            graph = torch.cuda.CUDAGraph(keep_graph=True) if AUDIT_NODES else torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, pool=self._pool, stream=self._stream):
                out = self._iteration(L)
            self._audit(L, [graph])
            return graph, out
        # with a process group: the iteration as a chain of graph segments with the collectives between them
        import gc
        gc.collect()
        seg = _Segments(self._pool, 'thread_local')
        cur = torch.cuda.current_stream(self.device)
        self._stream.wait_stream(cur)
        with torch.cuda.stream(self._stream):
            _active.seg, _active.comm = seg, self._comm
            try:
                seg.begin()
                out = self._iteration(L)
                seg.end()
            finally:
                _active.seg = _active.comm = None
        cur.wait_stream(self._stream)
        self._audit(L, [g for g in seg.items if isinstance(g, torch.cuda.CUDAGraph)])
        return seg, out

    def _audit(self, L, graphs):
        """Node census of what was just captured (self.node_types[L]); memset nodes are refused loudly: their fill pattern
        is read from recycled memory at replay on this ROCm build (module docstring), i.e. a silently wrong training run."""
        if not AUDIT_NODES:
            return
        total = {}
        for g in graphs:
            for k, v in graph_node_types(g).items():
                total[k] = total.get(k, 0) + v
        self.node_types[L] = total
        if total.get('memset', 0):
            msg = (f'GraphedTrainStep: the captured iteration (caption bucket {L}) holds {total["memset"]} memset node(s) '
                   '-- a library call inside the step zeroes with hipMemsetAsync; under replay their fill pattern is read '
                   'from recycled memory on this ROCm build. lavila_amd itself records none (tests/test_boundary_cpu.py)')
            if os.environ.get('LAVILA_GRAPH_MEMSET_NODES', 'error') == 'warn':
                import warnings
                warnings.warn(msg)
            else:
                raise RuntimeError(msg + '; LAVILA_GRAPH_MEMSET_NODES=warn lets the replay run anyway')

    @property
    def buckets(self):
        return sorted(self._graphs)

    @property
    def segments(self):
        """Graph segments per captured iteration (1 without a process group; with one: 2 + the collectives of the loss)."""
        return {L: (e[0].graphs if isinstance(e[0], _Segments) else 1) for L, e in self._graphs.items()}
