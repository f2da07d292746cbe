"""Data-parallel gradient averaging: bucketed RCCL all-reduce over xGMI, overlapped with backward.

Replaces Lightning's ``strategy='ddp_find_unused_parameters_true'`` (``baseline_code/train_se.py:74-83``).
Gradients live in ONE flat f32 buffer in backward-completion order, so a bucket is a contiguous slice:
when the last parameter group of a bucket is final (``BSRNNCore.grad_ready_hook``) the slice is summed
across ranks on a dedicated HIP stream while backward keeps running on the compute stream.  Parameters
that got no gradient on a rank (bands above fs/2) are zeros in the flat buffer, which is what
``find_unused_parameters`` amounts to.  The 1/world scaling is folded into the optimizer kernel.
"""
import torch
import torch.distributed as dist

from . import ops


class GradBucketReducer:
    def __init__(self, core, bucket_bytes=25 * 1024 * 1024, group=None, force=False):
        """``force``: issue the collectives even on a communicator of size 1 (functional check of the RCCL path on a 1-GPU box)."""
        self.core = core
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.active = self.world > 1 or (force and dist.is_initialized())
        self.launched = 0                      # collectives issued since construction
        self.flat = core.flat_grads
        self.on_gpu = self.flat.is_cuda
        self.comm_stream = torch.cuda.Stream() if self.on_gpu else None
        groups = core.grad_groups()           # [(tag, off, numel)] in backward order
        self.buckets = []                      # [(set(tags), lo, hi)]
        cur_tags, lo, hi = set(), None, None
        for tag, off, n in groups:
            cur_tags.add(tag)
            lo = off if lo is None else min(lo, off)
            hi = off + n if hi is None else max(hi, off + n)
            if (hi - lo) * 4 >= bucket_bytes:
                self.buckets.append((cur_tags, lo, hi))
                cur_tags, lo, hi = set(), None, None
        if cur_tags:
            self.buckets.append((cur_tags, lo, hi))
        self._pending = [set(b[0]) for b in self.buckets]
        self._works = []
        core.grad_ready_hook = self._on_ready

    def _launch(self, lo, hi):
        view = self.flat[lo:hi]
        if not self.active:
            return
        self.launched += 1
        if self.on_gpu:
            # from the first bucket until finish() an all-reduce kernel may sit on up to NCCL_MAX_NCHANNELS CUs: cooperative recurrences
            # launched meanwhile (flow model: split BPTT) are planned on the rest, or refused (ops.reserved_cus)
            ops.COMM_RESERVED_CUS = ops.rccl_reserved_cus()      # (the effective NCCL_MAX_NCHANNELS, not our default)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            self.comm_stream.wait_event(ev)
            with torch.cuda.stream(self.comm_stream):
                self._works.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        else:
            self._works.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def _on_ready(self, tag):
        for i, (tags, lo, hi) in enumerate(self.buckets):
            if tag in self._pending[i]:
                self._pending[i].discard(tag)
                if not self._pending[i]:
                    self._launch(lo, hi)

    def finish(self):
        """Block the compute stream until every bucket is reduced; returns the 1/world scale for the optimizer."""
        try:
            for w in self._works:
                w.wait()
            if self.on_gpu and self.active:
                torch.cuda.current_stream().wait_stream(self.comm_stream)
        finally:
            self._works = []
            ops.COMM_RESERVED_CUS = 0            # (the compute stream waits above: no all-reduce kernel is resident beside what follows)
        # every bucket must have been reduced by now: a bucket with tags still pending means this rank would step on
        # local, un-averaged gradients while its peers wait in an all-reduce - fail loudly instead
        for i, (tags, lo, hi) in enumerate(self.buckets):
            if self._pending[i] and self.active:
                raise RuntimeError("backward finished without reducing bucket %d (pending parameter groups: %s)"
                                   % (i, sorted(self._pending[i])))
            self._pending[i] = set(tags)
        return 1.0 / self.world
