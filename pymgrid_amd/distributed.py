"""Multi-GPU: the grids of a batch are independent (no cross-grid term anywhere in ``Microgrid.run``), so they
shard contiguously over ranks with NO data-path collective.  The only exchange is an optional all-reduce(sum) of a
small metrics vector (RCCL over xGMI via ``torch.distributed`` backend "nccl"; "gloo" on CPU in tests)."""
import os

import torch
import torch.distributed as dist


def shard_bounds(n_total, rank, world):
    """Contiguous block [lo, hi) of rank ``rank``; n_total must divide evenly (weak scaling: fixed per-GPU work)."""
    if n_total % world:
        raise ValueError(f"n_total={n_total} is not divisible by world size {world}")
    per = n_total // world
    return rank * per, (rank + 1) * per


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* (torchrun).  Returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        # MGX_DIST_BACKEND: override for tests (e.g. gloo to exercise the multi-rank code path on a one-GPU box)
        backend = backend or os.environ.get("MGX_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def all_reduce_metrics(local_sums):
    """Sum a small metrics vector over all ranks (in place, returns it).  No-op for a single process."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(local_sums, op=dist.ReduceOp.SUM)
    return local_sums


def max_over_ranks(value, device):
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_over_ranks(value, device):
    """[value of rank 0, ..., value of rank W-1] on every rank (one small all-gather; a one-element list for one process)."""
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
        dist.all_gather(out, t)
        return [float(x.item()) for x in out]
    return [float(value)]


def barrier():
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
