"""Multi-process plumbing.

Captioning shards by image ("replicas only"): every rank runs its own sub-batch, there is NO
data-path collective; torch.distributed (RCCL on the GPU box, gloo in CPU tests) is used only for barriers and the
max-over-ranks of the elapsed time.

Training has ONE exchange step, the gradient mean (DistributedDataParallel in the reference, trainer.py:119 via
uni_pipeline.py:913-927): `BucketedAllReduce` launches it bucket by bucket behind the backward pass."""
import os

import torch


def env_rank_world():
    return (int(os.environ.get('RANK', '0')), int(os.environ.get('WORLD_SIZE', '1')),
            int(os.environ.get('LOCAL_RANK', '0')))


def init(backend, device=None):
    import torch.distributed as dist
    if dist.is_initialized():
        return dist
    kw = {}
    if backend == 'nccl' and device is not None:
        kw['device_id'] = device
    dist.init_process_group(backend, **kw)
    return dist


def shard_seed(base_seed, rank):
    """Each rank captions different synthetic images (the reference shards with DistributedSampler(shuffle=False),
    uni_pipeline.py:782-850)."""
    return base_seed + rank


def max_over_ranks(value, dist, device='cpu'):
    if dist is None:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def whole_job_rate(units_per_rank_per_step, steps, world, elapsed_max):
    """images/sec of the whole job: all ranks' units over the slowest rank's time."""
    return units_per_rank_per_step * world * steps / elapsed_max


class BucketedAllReduce(object):
    """Gradient mean over ranks, overlapped with backward.

    `flat` is the flat gradient buffer; `buckets[stage]` lists the (start, end) ranges whose gradients are final once
    backward stage `stage` has run; `stages` is the order in which stages complete.  `stage_done(stage)` -- called by
    the training engine right after it enqueued the stage's last kernel -- records an event on the compute stream and
    starts the bucket's all-reduce (+ 1/world scaling) on a side stream, so the ring all-reduce of stage k travels over
    xGMI while the GPU computes stage k+1.  `finish()` makes the compute stream wait for the side stream.  Buckets are
    tens of MB (two layers each): xGMI rings are per-link bound, so few large messages beat many small ones.
    On CPU tensors (gloo tests) the same calls run without streams."""

    def __init__(self, flat, buckets, stages, dist):
        self.flat, self.buckets, self.stages, self.dist = flat, buckets, list(stages), dist
        self.world = dist.get_world_size() if (dist is not None and dist.is_initialized()) else 1
        self.cuda = flat.is_cuda
        self.comm = torch.cuda.Stream(device=flat.device) if (self.cuda and self.world > 1) else None
        self._works = []
        self._done = set()
        self.launched_bytes = 0

    def begin(self):
        if self._works:
            raise RuntimeError('begin() before the previous step\'s finish()')
        self._done = set()
        self.launched_bytes = 0

    def stage_done(self, stage):
        if stage not in self.buckets:
            raise KeyError(stage)
        if stage in self._done:
            raise RuntimeError('stage %s reported twice' % stage)
        expect = self.stages[len(self._done)]
        if stage != expect:
            raise RuntimeError('backward stages out of order: got %s, expected %s' % (stage, expect))
        self._done.add(stage)
        if self.world == 1:
            return
        scale = 1.0 / self.world
        if self.cuda:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.flat.device))
            with torch.cuda.stream(self.comm):
                self.comm.wait_event(ev)
                for a, b in self.buckets[stage]:
                    t = self.flat[a:b]
                    work = self.dist.all_reduce(t, async_op=True)
                    work.wait()                     # the side stream (not the host) waits for the collective
                    t.mul_(scale)
                    self.launched_bytes += (b - a) * t.element_size()
        else:
            for a, b in self.buckets[stage]:
                t = self.flat[a:b]
                self._works.append((self.dist.all_reduce(t, async_op=True), t))
                self.launched_bytes += (b - a) * t.element_size()

    def finish(self):
        if len(self._done) != len(self.stages) and self.world > 1:
            missing = [s for s in self.stages if s not in self._done]
            raise RuntimeError('finish() before stages %s completed' % missing)
        if self.world == 1:
            return
        if self.cuda:
            torch.cuda.current_stream(self.flat.device).wait_stream(self.comm)
        else:
            scale = 1.0 / self.world
            for work, t in self._works:
                work.wait()
                t.mul_(scale)
            self._works = []
