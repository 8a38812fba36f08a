"""Bucketed gradient all-reduce over flat buffers, launched from autograd hooks.

The data-parallel exchange of the step: gradients live in a few flat fp32 slabs;
a bucket is a contiguous slice; when the last gradient of a bucket has been accumulated its
all-reduce (SUM — the 1/W average is folded into the optimizer kernel) is issued at once, on a
side HIP stream so RCCL traffic over xGMI overlaps the rest of backward.  Works on CPU tensors
with gloo as well (no streams), which is how the N>1 path is tested without GPUs."""
import torch
import torch.distributed as dist

from .dist import collectives_active


class BucketedGradReducer:
    def __init__(self, slabs, bucket_bytes=64 << 20, split_key=None, wire=None):
        """slabs: list of (flat_grad, params, offsets) with params[i].grad a view of
        flat_grad[offsets[i]:offsets[i+1]] (in the order backward is expected to fill them).
        split_key(param): buckets never span a change of key (the engine's graph mode launches buckets per
        gradient class — heads / text encoder / late / early video stages — as each class completes)."""
        # wire: optional list of bf16 tensors parallel to the slabs' flat gradients.  With it a bucket travels as bf16 —
        # its fp32 slice is packed into the wire slab on the comm stream right before its all-reduce, and the optimizer
        # reads the reduced bf16 copy (fp32 update) — half the bytes over xGMI; the reference all-reduces fp16 gradients
        # (mmcv_Fp16OptimizerHook.py:119-122).
        self.wire = wire
        self.launch_log = []        # (bucket, split key, bytes on the wire, 'hook' | 'where' | 'finish') of the last step
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.enabled = True         # False: hooks are inert (hipGraph capture) and finish() sends everything
        self.buckets = []           # [flat_grad, start, end, n_params]
        self.handles = []
        self.pending = []
        # A parameter may receive SEVERAL gradient contributions per backward (any weight applied twice, e.g. a tied
        # embedding; token_type_embeddings is indexed at two places in the fusion encoder, cross_transformer.py:84-94).
        # The first backward only counts them (its buckets all go out in finish()); from then on a parameter reports
        # ready on its LAST contribution.
        self.expect = {}            # id(param) -> contributions per backward (known after the first finish())
        self.seen = {}
        self.calibrated = False
        self.comm_stream = None
        self._producers = []        # per bucket: the streams its gradients were produced on (eager mode)
        self.active = collectives_active()
        self._bucket_params = []
        if not self.active:
            for _, params, _ in slabs:
                for q in params:
                    q._clv_ready = (lambda: None)
            return
        cap = max(1, bucket_bytes // 4)
        self._bucket_key, self._bucket_slab = [], []
        for si, (flat, params, offsets) in enumerate(slabs):
            start, count = 0, 0
            for i, p in enumerate(params):
                count += 1
                end = offsets[i + 1]
                boundary = split_key is not None and i + 1 < len(params) and split_key(params[i + 1]) != split_key(p)
                if end - start >= cap or i == len(params) - 1 or boundary:
                    self.buckets.append([flat, start, end, count])
                    self._bucket_slab.append(si)
                    self._bucket_key.append(split_key(p) if split_key is not None else None)
                    self._bucket_params.append(list(params[i + 1 - count:i + 1]))
                    b = len(self.buckets) - 1
                    for q in params[i + 1 - count:i + 1]:
                        hook = self._make_hook(b)
                        q.register_post_accumulate_grad_hook(hook)
                        q._clv_ready = (lambda h=hook, t=q: h(t))       # for ops that fill .grad themselves
                    start, count = end, 0
            if flat.is_cuda and self.comm_stream is None:
                self.comm_stream = torch.cuda.Stream(device=flat.device)
        self.reset()

    def begin_step(self):
        """Forget the previous step's launch log (the engine calls this when a step starts)."""
        self.launch_log = []

    def reset(self):
        self.pending = [b[3] for b in self.buckets]
        self.handles = []
        self.seen = {}
        self._producers = [[] for _ in self.buckets]

    prepacked = frozenset()      # buckets whose bf16 wire copy the replayed backward graphs have already written (engine)

    def pack_where(self, ready, skip=()):
        """Pack (fp32 slab slice -> bf16 wire slab, no collective) every bucket all of whose parameters satisfy `ready` —
        called by the engine INSIDE a backward-graph capture, right after the segment that completes those gradients: the
        pack kernels then replay with the graph instead of being launched one by one, host-paced, between the replays
        (17 launches, 0.28 ms at config 2).  skip: buckets already packed by an earlier graph.  Returns the set packed."""
        done = set()
        if not self.active or self.wire is None:
            return done
        from .. import ops
        for b, ps in enumerate(self._bucket_params):
            flat, s, e, _ = self.buckets[b]
            if b not in skip and flat.is_cuda and all(ready(q) for q in ps):
                ops.pack_bf16(flat[s:e], self.wire[self._bucket_slab[b]][s:e])
                done.add(b)
        return done

    def _payload(self, b):
        """The tensor bucket b puts on the wire: the fp32 slab slice, or its freshly packed bf16 copy."""
        flat, s, e, _ = self.buckets[b]
        if self.wire is None:
            return flat[s:e]
        w = self.wire[self._bucket_slab[b]][s:e]
        if b in self.prepacked:
            return w
        if flat.is_cuda:
            from .. import ops
            ops.pack_bf16(flat[s:e], w)
        else:
            w.copy_(flat[s:e])
        return w

    def _launch(self, b, phase='hook'):
        flat, s, e, _ = self.buckets[b]
        self.launch_log.append((b, self._bucket_key[b], (e - s) * (2 if self.wire is not None else 4), phase))
        if self.comm_stream is not None:
            # The hook countdown is HOST-ordered: the bucket's other gradients may have been produced on another
            # stream than the last hook's (the text tower runs on a side stream) — wait for every producer stream.
            self.comm_stream.wait_stream(torch.cuda.current_stream())
            for st in self._producers[b]:
                self.comm_stream.wait_stream(st)
            with torch.cuda.stream(self.comm_stream):
                self.handles.append(dist.all_reduce(self._payload(b), async_op=True))
        else:
            self.handles.append(dist.all_reduce(self._payload(b), async_op=True))

    def _make_hook(self, b):
        def hook(param):
            if not self.enabled:
                return
            k = id(param)
            if self.comm_stream is not None:
                cur = torch.cuda.current_stream()
                if all(cur != st for st in self._producers[b]):
                    self._producers[b].append(cur)
            self.seen[k] = self.seen.get(k, 0) + 1
            if not self.calibrated or self.seen[k] != self.expect.get(k, 1):
                return
            self.pending[b] -= 1
            if self.pending[b] == 0:
                self._launch(b)
        return hook

    def launch_where(self, ready):
        """Launch (asynchronously, on the comm stream) every not-yet-launched bucket all of whose parameters satisfy
        `ready(param)` — the engine's split-backward graph mode calls this between the two backward replays, so the
        all-reduce of the text / fusion / head gradients overlaps the video encoder's backward."""
        if not self.active:
            return
        if not hasattr(self, '_bucket_params'):
            raise RuntimeError('launch_where needs the bucket -> parameter map')
        for b, ps in enumerate(self._bucket_params):
            if self.pending[b] > 0 and all(ready(q) for q in ps):
                self._launch(b, 'where')
                self.pending[b] = 0

    def finish(self):
        """Wait for every bucket (launching any whose hooks did not all fire, e.g. a parameter that
        received no gradient this step), then make the compute stream wait for the comm stream."""
        if not self.active:
            return
        if not self.calibrated and self.enabled and self.seen:
            self.expect = dict(self.seen)
            self.calibrated = True
        for b, left in enumerate(self.pending):
            if left > 0:
                self._launch(b, 'finish')
        for h in self.handles:
            h.wait()
        if self.comm_stream is not None:
            cur = torch.cuda.current_stream()
            if self.time_exposed:
                # GPU-side stall of the compute stream on the comm stream = the exposed (not overlapped) part of the
                # gradient exchange: two events around the cross-stream wait, read by exposed_ms() after a sync
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(cur)
                cur.wait_stream(self.comm_stream)
                e1.record(cur)
                self._exposed.append((e0, e1))
            else:
                cur.wait_stream(self.comm_stream)
        self.reset()

    time_exposed = False
    _exposed = ()

    def start_timing(self):
        self.time_exposed, self._exposed = True, []

    def exposed_ms(self):
        """Mean per-step stall of the compute stream waiting for the gradient all-reduces (call after a device sync)."""
        ev, self._exposed, self.time_exposed = self._exposed, [], False
        return sum(a.elapsed_time(b) for a, b in ev) / len(ev) if ev else None
