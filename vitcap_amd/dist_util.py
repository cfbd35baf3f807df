"""Minimal multi-process plumbing for the captioning path: inference shards by image ("replicas only", SURVEY.md
section 8e) -- every rank runs its own sub-batch, there is NO data-path collective.  torch.distributed (RCCL on the
GPU box, gloo in CPU tests) is used only for barriers and the max-over-ranks of the elapsed time."""
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
