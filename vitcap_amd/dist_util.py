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


def host_cpu_budget():
    """Cores this process (with its children) may use: the cgroup-v2 CPU bandwidth (cpu.max) where one is set, else the scheduler
    affinity mask, else the machine.  On the GPU pool a box shows 256 hardware threads and grants 16 cores: a library that sizes
    its thread pool by cpu_count() (torch's intra-op pool: 128 threads) burns the quota of a whole 100 ms period in a few
    milliseconds, and the kernel then stops EVERY thread of the job -- launcher and decode workers included -- until the period ends."""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    try:
        q, per = open('/sys/fs/cgroup/cpu.max').read().split()
        if q != 'max':
            n = min(n, max(1, int(int(q) / int(per))))
    except (OSError, ValueError):
        pass
    return n


def cap_host_threads(reserved=0):
    """Lowers (never raises) torch's intra-op thread count to the CPU budget minus `reserved` cores (decode workers, launch threads).
    With WORLD_SIZE ranks on one host the budget is shared: each rank takes its share (host budget)."""
    world_local = int(os.environ.get('LOCAL_WORLD_SIZE', os.environ.get('WORLD_SIZE', 1)) or 1)
    n = max(1, host_cpu_budget() // max(1, world_local) - int(reserved))
    if n < torch.get_num_threads():
        torch.set_num_threads(n)
    return torch.get_num_threads()


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


def broadcast_from_rank0(tensors, dist, chunk_elems=64 << 20):
    """Every rank ends with rank 0's values of `tensors` (in place) -- what DistributedDataParallel does to a module's parameters and
    buffers when it wraps it (uni_pipeline.py:497-505 -> torch DDP's _sync_module_states): replicas that were built from different
    files, seeds or a stale snapshot cannot train as if they were one model.  Large flat buffers travel in `chunk_elems` pieces (the
    0.87 GB parameter vector as 4 messages; gloo stages CUDA tensors through the host).  Returns the number of bytes broadcast; a
    no-op (0) without an initialised process group or with one rank."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() < 2:
        return 0
    sent = 0
    for t in tensors:
        flat = t.view(-1)
        for a in range(0, flat.numel(), chunk_elems):
            piece = flat[a:a + chunk_elems]
            dist.broadcast(piece, src=0)
            sent += piece.numel() * piece.element_size()
    return sent


class BucketedAllReduce(object):
    """Gradient mean over ranks, overlapped with backward.

    `flat` is the flat gradient buffer; `buckets[stage]` lists the (start, end) ranges whose gradients are final once
    backward stage `stage` has run; `stages` is the order in which stages complete.  `stage_done(stage)` -- called by
    the training engine right after it enqueued the stage's last kernel -- records an event on the compute stream and
    starts the bucket's exchange (+ 1/world scaling) on a side stream, so the collective of stage k travels over
    xGMI while the GPU computes stage k+1.  `finish()` makes the compute stream wait for the side stream.  Buckets are
    tens of MB (two layers each): xGMI rings are per-link bound, so few large messages beat many small ones.
    On CPU tensors (gloo tests) the same calls run without streams.

    Two exchange algorithms behind the same interface (`algo`, default from VITCAP_DP_REDUCE):
      'all_reduce'  one in-place all-reduce per range, then the 1/world scaling of the whole range;
      'rs_ag'       reduce-scatter (every rank ends with the sum of ITS 1/world slice of the range), scaling of that slice
                    only, all-gather of the slices (what a ring all-reduce does internally, made explicit so that
                    the two halves of consecutive buckets overlap on xGMI's point-to-point links and the scaling touches
                    1/world of the bytes).  Range tails that do not divide by world go through a small all-reduce.
    Both give the same mean (bit-identical on 2 ranks; summation order may differ with more).

    `wire` (default from VITCAP_DP_WIRE, 'f32'): 'bf16' sends the gradients as bf16 and accumulates in fp32 -- every range is cut in
    world slices, rank r receives everybody's bf16 slice r (ONE all-to-all), sums them in fp32 in rank order, scales, and the
    bf16-rounded means are all-gathered: half the bytes on xGMI (0.33 instead of 0.67 GB per step for the 167 M gradient elements),
    no bf16 partial sums (a ring all-reduce on bf16 tensors would round after every hop), and every rank ends with the SAME bits
    (the replicas cannot drift apart).  Error per element <= 2^-9 relative on each rank's contribution + 2^-9 on the mean.

    `reserve_cus` (default from VITCAP_DP_RESERVE_CUS, 16): CUs the persistent large-GEMM grids leave free between the first
    bucket's launch and finish() (vitcap_gemm_reserve_cus).  A persistent grid is one 512-register workgroup per CU for the whole
    GEMM; the side stream's priority orders DISPATCH, it cannot evict resident workgroups, so without free CUs a collective's
    kernel starts only when a whole GEMM has drained.  Costs the GEMMs 16 / 256 of the chip while an exchange is
    in flight and nothing outside that window.

    `force_exchange` (default from VITCAP_DP_FORCE=1): run the exchange also in a process group of ONE rank -- every collective is
    then an identity and the scaling a multiplication by 1.0, so the step's parameters equal the no-dist step bit for bit, while the
    whole RCCL code path (communicator on `device_id`, collectives enqueued on the side stream behind an event, the in-place
    reduce-scatter's aliasing, finish()'s stream join) executes on the one GPU a test box has."""

    def __init__(self, flat, buckets, stages, dist, algo=None, force_exchange=None, wire=None, reserve_cus=None):
        self.flat, self.buckets, self.stages, self.dist = flat, buckets, list(stages), dist
        self.world = dist.get_world_size() if (dist is not None and dist.is_initialized()) else 1
        if force_exchange is None:
            force_exchange = os.environ.get('VITCAP_DP_FORCE', '0') == '1'
        self.exchange = self.world > 1 or (bool(force_exchange) and dist is not None and dist.is_initialized())
        self.rank = dist.get_rank() if self.exchange else 0
        self.cuda = flat.is_cuda
        # high priority: the collective's few workgroups (RCCL channels) must not queue behind GEMM grids that fill every CU --
        # the dispatcher serves a higher-priority queue first whenever a workgroup slot frees (docs/LAB_r01_r04.md section 7)
        self.comm = torch.cuda.Stream(device=flat.device, priority=-1) if (self.cuda and self.exchange) else None
        self.algo = algo or os.environ.get('VITCAP_DP_REDUCE', 'all_reduce')
        if self.algo not in ('all_reduce', 'rs_ag'):
            raise ValueError('unknown gradient exchange %r (all_reduce | rs_ag)' % self.algo)
        self._native_rs = self.exchange and dist.get_backend() == 'nccl'     # gloo has no reduce_scatter: emulate per slice
        self.wire = wire or os.environ.get('VITCAP_DP_WIRE', 'f32')
        if self.wire not in ('f32', 'bf16'):
            raise ValueError('unknown gradient wire format %r (f32 | bf16)' % self.wire)
        if reserve_cus is None:
            reserve_cus = int(os.environ.get('VITCAP_DP_RESERVE_CUS', '16'))
        self.reserve_cus = int(reserve_cus) if (self.cuda and self.exchange) else 0
        self._reserved = False
        self._stage16 = {}                    # bf16 staging buffers by element count (send, recv / gathered)
        self._works = []
        self._done = set()
        self.launched_bytes = 0

    def begin(self):
        if self._works:
            raise RuntimeError('begin() before the previous step\'s finish()')
        self._done = set()
        self.launched_bytes = 0

    def _reserve(self, on):
        """Persistent GEMM grids keep `reserve_cus` CUs free while buckets are in flight (see the class docstring)."""
        if not self.reserve_cus or on == self._reserved:
            return
        from ._lib import lib
        lib.vitcap_gemm_reserve_cus(self.reserve_cus if on else 0)
        self._reserved = on

    def _exchange_bf16(self, a, b):
        t = self.flat[a:b]
        n = b - a
        per = n // self.world
        m = per * self.world
        if per:
            if m not in self._stage16:
                self._stage16[m] = (torch.empty(m, dtype=torch.bfloat16, device=t.device), torch.empty(m, dtype=torch.bfloat16, device=t.device))
            send, recv = self._stage16[m]
            send.copy_(t[:m])                                         # fp32 -> bf16 (round to nearest even)
            self.dist.all_to_all_single(recv, send)                   # recv[r * per : (r + 1) * per] = rank r's slice `self.rank`
            mean = recv.view(self.world, per).to(torch.float32).sum(0).mul_(1.0 / self.world)       # fp32 accumulate, rank order
            mine = send[self.rank * per:(self.rank + 1) * per]
            mine.copy_(mean)
            if self._native_rs:
                self.dist.all_gather_into_tensor(recv, mine)
            else:
                self.dist.all_gather([recv[r * per:(r + 1) * per] for r in range(self.world)], mine.clone())
            t[:m].copy_(recv)
        if m < n:
            tail = t[m:]
            self.dist.all_reduce(tail)
            tail.mul_(1.0 / self.world)

    # ---- one range [a, b) of the flat buffer, synchronous with respect to the calling (side) stream
    def _exchange(self, a, b):
        if self.wire == 'bf16':
            return self._exchange_bf16(a, b)
        t = self.flat[a:b]
        scale = 1.0 / self.world
        if self.algo == 'all_reduce':
            self.dist.all_reduce(t)
            t.mul_(scale)
            return
        n = b - a
        per = n // self.world
        m = per * self.world
        if per:
            body = t[:m]
            mine = body[self.rank * per:(self.rank + 1) * per]
            if self._native_rs:
                self.dist.reduce_scatter_tensor(mine, body)          # in place: the output is this rank's slice of the input
                mine.mul_(scale)
                self.dist.all_gather_into_tensor(body, mine)
            else:
                for r in range(self.world):                            # gloo: one reduce per slice = a reduce-scatter
                    self.dist.reduce(body[r * per:(r + 1) * per], dst=r)
                mine.mul_(scale)
                parts = [body[r * per:(r + 1) * per] for r in range(self.world)]
                self.dist.all_gather(parts, mine.clone())
        if m < n:
            tail = t[m:]
            self.dist.all_reduce(tail)
            tail.mul_(scale)

    def stage_done(self, stage):
        if stage not in self.buckets:
            raise KeyError(stage)
        if stage in self._done:
            raise RuntimeError('stage %s reported twice' % stage)
        expect = self.stages[len(self._done)]
        if stage != expect:
            raise RuntimeError('backward stages out of order: got %s, expected %s' % (stage, expect))
        self._done.add(stage)
        if not self.exchange:
            return
        scale = 1.0 / self.world
        if self.cuda:
            self._reserve(True)               # GEMMs launched from here on leave room for the collectives' kernels
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.flat.device))
            with torch.cuda.stream(self.comm):
                self.comm.wait_event(ev)
                for a, b in self.buckets[stage]:
                    self._exchange(a, b)             # collectives are enqueued on the side stream; the host does not wait
                    self.launched_bytes += (b - a) * self.flat.element_size()
        elif self.algo == 'all_reduce' and self.wire == 'f32':
            for a, b in self.buckets[stage]:
                t = self.flat[a:b]
                self._works.append((self.dist.all_reduce(t, async_op=True), t))
                self.launched_bytes += (b - a) * t.element_size()
        else:
            for a, b in self.buckets[stage]:
                self._exchange(a, b)
                self.launched_bytes += (b - a) * self.flat.element_size()

    def finish(self):
        if len(self._done) != len(self.stages) and self.exchange:
            missing = [s for s in self.stages if s not in self._done]
            raise RuntimeError('finish() before stages %s completed' % missing)
        if not self.exchange:
            return
        if self.cuda:
            torch.cuda.current_stream(self.flat.device).wait_stream(self.comm)
            self._reserve(False)
        else:
            scale = 1.0 / self.world
            for work, t in self._works:
                work.wait()
                t.mul_(scale)
            self._works = []
