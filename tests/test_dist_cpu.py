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
