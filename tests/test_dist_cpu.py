"""N>1 plumbing of bench.py on CPU: world_size 2, gloo.  (The data path has no collective: replicas only.)"""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port))
    from vitcap_amd import dist_util as D
    from vitcap_amd import weights as W
    dist = D.init('gloo')
    r, w, _ = D.env_rank_world()
    assert (r, w) == (rank, world)
    img = W.gen_image_batch(1, D.shard_seed(1234, rank))
    dist.barrier()
    elapsed = 1.0 + rank                       # rank 1 is the slow one
    mx = D.max_over_ranks(elapsed, dist)
    out.put((rank, mx, float(img.sum())))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_timing_and_sharding():
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res[0][1] == res[1][1] == 2.0        # max over ranks seen by everyone
    assert res[0][2] != res[1][2]                # different images per rank
    from vitcap_amd import dist_util as D
    assert D.whole_job_rate(64, 10, 2, 2.0) == 640.0


# ---------------------------------------------------------------------------------------------------------------
# training: the one exchange step of the path (gradient mean), bucketed behind the backward pass
def test_grad_buckets_cover_exactly_the_tensors_that_get_gradients():
    from vitcap_amd import weights as W
    from vitcap_amd.train import CH, GRAD_STAGES, NO_GRAD_PREFIXES, flat_layout, grad_buckets, grad_stage
    for tied in (True, False):
        order, off, shape, nflat = flat_layout(tied)
        assert len(set(order)) == len(order) and set(order) == set(W.state_dict_spec()) - ({W.TIED_DST} if tied else set())
        buckets = grad_buckets(order, off, shape)
        assert list(buckets) == GRAD_STAGES
        covered = torch.zeros(nflat, dtype=torch.int8)
        for st in GRAD_STAGES:
            assert buckets[st], st
            for a, b in buckets[st]:
                assert 0 <= a < b <= nflat
                covered[a:b] += 1
        assert int(covered.max()) == 1, 'buckets overlap'
        for k in order:
            n = int(torch.Size(shape[k]).numel())
            want = 0 if k.startswith(NO_GRAD_PREFIXES) else 1
            assert int(covered[off[k]:off[k] + n].min()) == want and int(covered[off[k]:off[k] + n].max()) == want, k
            assert off[k] % CH == 0 or '.attention.self.' in k
        # a stage never finishes before one that owns a later layer of the same stack
        assert grad_stage('module.bert.encoder.blocks.11.mlp.fc2.weight') == 'blk5'
        assert grad_stage('module.bert.encoder.blocks.0.norm1.bias') == 'blk0'
        assert grad_stage('module.bert.tag_logit.predictions.decoder.weight') is None
        # optimizer chunks: every element of a tensor sees exactly that tensor's (lr, wd) of the reference's groups
        from vitcap_amd.train import chunk_hparams, param_groups
        pg = param_groups(order, 1e-4, 0.05, 0.1)
        lr, wd = chunk_hparams(order, off, shape, nflat, pg)
        for k in order:
            n = int(torch.Size(shape[k]).numel())
            c0, c1 = off[k] // CH, (off[k] + n + CH - 1) // CH
            hp = (0.0, 0.0) if (pg[k] is None or k.startswith(NO_GRAD_PREFIXES)) else pg[k]
            assert torch.all(lr[c0:c1] == torch.tensor(hp[0])) and torch.all(wd[c0:c1] == torch.tensor(hp[1])), k
        sizes = [sum(b - a for a, b in buckets[st]) * 4 / 2 ** 20 for st in GRAD_STAGES]
        assert max(sizes) < 128 and sum(s > 32 for s in sizes) >= 10      # MB: few large messages


def _reduce_worker(rank, world, port, out, algo='all_reduce', wire='f32'):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port))
    from vitcap_amd import dist_util as D
    dist = D.init('gloo')
    n = 4096
    g = torch.Generator().manual_seed(100 + rank)
    flat = torch.randn(n, generator=g)
    mine = flat.clone()
    buckets = {'last': [(3000, 4000)], 'mid': [(1024, 2048), (2100, 2901)], 'first': [(0, 1000)]}    # 801: odd, exercises the tail
    red = D.BucketedAllReduce(flat, buckets, ['last', 'mid', 'first'], dist, algo=algo, wire=wire)
    red.begin()
    err = None
    try:
        red.stage_done('mid')                   # out of order: must be refused
    except RuntimeError as e:
        err = str(e)
    red.stage_done('last')
    red.stage_done('mid')
    try:
        red.finish()                            # a stage is still missing
        missing_caught = False
    except RuntimeError:
        missing_caught = True
    red.stage_done('first')
    red.finish()
    red.begin()                                 # second step on the same object
    for st in ('last', 'mid', 'first'):
        red.stage_done(st)
    red.finish()
    # plain numpy through the queue: a torch tensor travels as a file descriptor owned by THIS process, and the parent's q.get
    # fails with FileNotFoundError when it runs after the worker has exited
    out.put((rank, mine.numpy().copy(), flat.numpy().copy(), err, missing_caught, red.launched_bytes))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('algo', ['all_reduce', 'rs_ag'])
def test_bucketed_all_reduce_two_ranks_gloo(algo):
    """Both gradient-exchange algorithms (one all-reduce per range / reduce-scatter + scale own slice + all-gather) give every
    rank the mean of the ranks' buckets and leave everything outside the buckets untouched."""
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_reduce_worker, args=(r, world, port, q, algo)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=120) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, m0, f0, e0, c0, nb), (_, m1, f1, e1, c1, _) = res
    m0, f0, m1, f1 = (torch.from_numpy(a) for a in (m0, f0, m1, f1))
    assert e0 and 'out of order' in e0 and e1 and c0 and c1
    mean = (m0 + m1) / 2                        # after step 1 both hold the mean; step 2 averages two equal copies
    inside = torch.zeros(4096, dtype=torch.bool)
    for a, b in ((3000, 4000), (1024, 2048), (2100, 2901), (0, 1000)):
        inside[a:b] = True
    assert torch.allclose(f0[inside], mean[inside]) and torch.equal(f0[inside], f1[inside])
    assert torch.equal(f0[~inside], m0[~inside]) and torch.equal(f1[~inside], m1[~inside])   # untouched outside buckets
    assert nb == int(inside.sum()) * 4


@pytest.mark.parametrize('world', [2, 3])
def test_bucketed_exchange_bf16_wire_vs_fp32(world):
    """wire='bf16' (bf16 on the links, fp32 accumulate: all-to-all of slices, local fp32 sum in rank order, all-gather of the rounded
    means) against the exact mean: within bf16 rounding of the contributions and of the result, IDENTICAL bits on every rank (the
    replicas must not drift apart), untouched outside the buckets, odd range tails exact."""
    port = _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_reduce_worker, args=(r, world, port, q, 'rs_ag', 'bf16')) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=120) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    mine = [torch.from_numpy(r[1]) for r in res]
    flat = [torch.from_numpy(r[2]) for r in res]
    mean = sum(mine) / world
    inside = torch.zeros(4096, dtype=torch.bool)
    for a, b in ((3000, 4000), (1024, 2048), (2100, 2901), (0, 1000)):
        inside[a:b] = True
    for f in flat[1:]:
        assert torch.equal(f, flat[0]) or torch.equal(f[inside], flat[0][inside])
    # two steps ran: step 1 leaves bf16(mean of bf16 contributions), step 2 averages equal copies of a bf16 value (exact)
    bound = (sum(m.abs() for m in mine) / world + mean.abs()) * 2.0 ** -8 + 1e-12
    assert bool(((flat[0] - mean).abs()[inside] <= bound[inside]).all())
    body = inside.clone()
    body[2100 + (801 // world) * world:2901] = False         # the tail of the odd range travels as fp32: exact mean
    tail = inside & ~body
    if bool(tail.any()):
        assert torch.allclose(flat[0][tail], mean[tail], rtol=1e-6, atol=1e-7)
    for m, f in zip(mine, flat):
        assert torch.equal(f[~inside], m[~inside])


def _bcast_worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    from vitcap_amd import dist_util as D
    dist = D.init('gloo')
    g = torch.Generator().manual_seed(7)
    P = torch.randn(5000, generator=g)
    M, V = torch.zeros(5000), torch.ones(5000)
    if rank != 0:                      # a replica built from another file: perturbed parameters, stale moments
        P += 0.1 * (rank + 1)
        M += rank
        V *= 3.0
    n = D.broadcast_from_rank0([P, M, V], dist, chunk_elems=2048)      # 3 pieces per tensor: the chunked path
    out.put((rank, n, P.clone(), M.clone(), V.clone()))
    dist.barrier()
    dist.destroy_process_group()


def test_parameter_broadcast_from_rank0_gloo():
    """DDP's wrap-time broadcast (uni_pipeline.py:497-505): rank 1 starts from perturbed weights / moments and ends equal to rank 0
    (TrainEngine.__init__ and sync_from_rank0 call this helper on the flat parameter / moment buffers)."""
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_bcast_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=120) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    want = torch.randn(5000, generator=torch.Generator().manual_seed(7))
    for rank, n, P, M, V in res:
        assert n == 3 * 5000 * 4
        assert torch.equal(P, want) and torch.equal(M, torch.zeros(5000)) and torch.equal(V, torch.ones(5000)), rank
    from vitcap_amd import dist_util as D
    assert D.broadcast_from_rank0([torch.zeros(3)], None) == 0         # no process group: nothing to do


def test_bench_self_launches_eight_ranks():
    """The driver's widest launch: `python bench.py --gpus 8` starts eight ranks (child torch.distributed.run), every rank takes
    part in the barriers and the max-over-ranks, rank 0 prints the ONE line (CPU stand-in workload, gloo)."""
    import json
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env['OMP_NUM_THREADS'] = '1'
    out = subprocess.run([sys.executable, os.path.join(repo, 'bench.py'), '--gpus', '8', '--steps', '3', '--warmup', '1', '--stub',
                          '--no-cpu-baseline'], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, out.stdout
    rec = json.loads(lines[0])
    assert rec['n_gpus'] == 8 and rec['steps'] == 3 and rec['scaling'] == 'weak' and rec['value'] > 0
    assert abs(rec['value'] - 8 * 3 / (rec['ms_per_step'] * 3e-3)) < 1e-2 * rec['value'] + 1.0


def test_bench_self_launches_n_ranks():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment (how the driver calls --gpus 1) must start the two ranks
    itself -- as a child torch.distributed.run, before the parent touches a GPU -- and print ONE JSON line from rank 0.  Driven on
    the CPU stand-in workload (gloo); the launch path is the one the GPU workloads use."""
    import json
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    out = subprocess.run([sys.executable, os.path.join(repo, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1', '--stub',
                          '--no-cpu-baseline'], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, out.stdout
    rec = json.loads(lines[0])
    assert rec['n_gpus'] == 2 and rec['steps'] == 3 and rec['warmup'] == 1 and rec['scaling'] == 'weak'
    assert rec['value'] > 0 and abs(rec['value'] - 2 * 3 / (rec['ms_per_step'] * 3e-3)) < 1e-2 * rec['value'] + 1.0
    # a world size that contradicts --gpus is refused, not silently benchmarked
    bad = subprocess.run([sys.executable, os.path.join(repo, 'bench.py'), '--gpus', '2', '--stub', '--steps', '1'],
                         env=dict(env, WORLD_SIZE='1', RANK='0', LOCAL_RANK='0'), capture_output=True, text=True, timeout=120)
    assert bad.returncode != 0 and 'WORLD_SIZE' in bad.stderr
