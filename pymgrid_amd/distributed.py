"""Multi-GPU: the grids of a batch are independent (no cross-grid term anywhere in ``Microgrid.run``), so they
shard contiguously over ranks with NO data-path collective.  The only exchange is an optional all-reduce(sum) of a
small metrics vector (RCCL over xGMI via ``torch.distributed`` backend "nccl"; "gloo" on CPU in tests)."""
import datetime
import os

import torch
import torch.distributed as dist

# Control plane.  The data path has no collective, so everything a multi-rank run needs besides the final metrics all-reduce --
# barriers around timed regions, gathering per-rank timings -- goes over a small GLOO group on host tensors: a rank that died
# shows up as a timeout of ``monitored_barrier`` (naming the missing rank) instead of a hang inside RCCL, and RCCL itself is
# touched exactly once, by ``all_reduce_metrics`` (with a gloo fallback that is reported, should RCCL refuse to come up).
_ctrl = None
CTRL_TIMEOUT_S = float(os.environ.get("MGX_CTRL_TIMEOUT_S", "600"))
last_collective = {"backend": None, "error": None, "hung": False}


def shard_bounds(n_total, rank, world):
    """Contiguous block [lo, hi) of rank ``rank``; n_total must divide evenly (weak scaling: fixed per-GPU work)."""
    if n_total % world:
        raise ValueError(f"n_total={n_total} is not divisible by world size {world}")
    per = n_total // world
    return rank * per, (rank + 1) * per


class _stdout_to_stderr:
    """File-descriptor-level redirect of stdout into stderr: gloo announces its connections on stdout ("[Gloo] Rank 0 is connected
    to ..."), and a bench's stdout is a one-line JSON contract."""

    def __enter__(self):
        import sys
        sys.stdout.flush()
        self._saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        import sys
        sys.stdout.flush()
        os.dup2(self._saved, 1)
        os.close(self._saved)


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* (torchrun).  Returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("MGX_FORCE_LOCAL_RANK", os.environ.get("LOCAL_RANK", "0")))   # tests: several ranks on one GPU
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        # MGX_DIST_BACKEND: override for tests (e.g. gloo to exercise the multi-rank code path on a one-GPU box)
        backend = backend or os.environ.get("MGX_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
        global _ctrl
        with _stdout_to_stderr():
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
            if backend != "gloo":          # every rank gets here: new_group is itself a collective over the default group's store
                _ctrl = dist.new_group(backend="gloo", timeout=datetime.timedelta(seconds=CTRL_TIMEOUT_S))
            if backend == "gloo" or _ctrl is not None:     # (gloo connects lazily: bring the control plane up -- and its banner out -- here)
                dist.barrier(group=_ctrl)
    return rank, world, local


def _multi():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def all_reduce_metrics(local_sums, timeout_s=None):
    """Sum a small metrics vector over all ranks and RETURN the sum -- use the return value: it is ``local_sums`` reduced in place
    on the normal paths, but when the backend hangs it is a FRESH host tensor (the stuck helper thread may still own
    ``local_sums``, which is then left as it was -- possibly half reduced -- and must not be read).  THE collective of the engine
    (RCCL over xGMI on a GPU node).  With a gloo control group beside the default backend the attempt is BOUNDED: the all-reduce runs in a helper thread,
    every rank waits ``timeout_s`` (MGX_RCCL_TIMEOUT_S, default 120 s) for it, the ranks then agree over the control group whether
    it came through everywhere, and if it did not -- RCCL raised, or hangs (IPC handles, a missing peer) -- the sum is taken over
    the control group instead.  The failure is kept in ``last_collective`` (bench.py prints it; ``hung`` tells the caller that a
    thread is still stuck inside the backend: leave with ``os._exit`` after flushing).  The per-rank throughputs never depend on
    this call.  No-op for one process."""
    if not _multi():
        return local_sums
    last_collective.update(backend=dist.get_backend(), error=None, hung=False)
    if _ctrl is None:                                        # one backend only (gloo in tests): nothing to fall back to
        dist.all_reduce(local_sums, op=dist.ReduceOp.SUM)
        return local_sums
    import threading
    timeout_s = float(os.environ.get("MGX_RCCL_TIMEOUT_S", "120")) if timeout_s is None else float(timeout_s)
    host = local_sums.detach().cpu().clone()                 # what the fallback sums (the attempt may leave garbage behind)
    res = {}

    def attempt():
        try:
            if local_sums.is_cuda:
                torch.cuda.set_device(local_sums.device)
            dist.all_reduce(local_sums, op=dist.ReduceOp.SUM)
            if local_sums.is_cuda:
                torch.cuda.synchronize(local_sums.device)   # surface an asynchronous RCCL failure here, not later
            res["ok"] = True
        except Exception as e:                               # noqa: BLE001
            res["err"] = f"{type(e).__name__}: {e}"
    th = threading.Thread(target=attempt, daemon=True)
    th.start()
    th.join(timeout_s)
    ok = torch.tensor([1.0 if res.get("ok") else 0.0], dtype=torch.float64)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=_ctrl)   # a collective comes through everywhere or is not trusted anywhere
    if float(ok.item()) == 1.0:
        return local_sums
    hung = th.is_alive()
    dist.all_reduce(host, op=dist.ReduceOp.SUM, group=_ctrl)
    if not hung:
        local_sums.copy_(host)
        out = local_sums
    else:                                                    # the stuck thread may still own local_sums: hand out a fresh tensor
        out = host.to(local_sums.device) if not local_sums.is_cuda else host
    err = res.get("err") or (f"no answer from the {dist.get_backend()} all-reduce within {timeout_s:g} s" if hung
                             else "the all-reduce failed on another rank")
    last_collective.update(backend="gloo (fallback)", error=err, hung=hung)
    return out


def max_over_ranks(value, device=None):
    """max of a host scalar over the ranks (control plane: gloo when the default backend is RCCL)."""
    if not _multi():
        return float(value)
    if _ctrl is not None:
        t = torch.tensor([float(value)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=_ctrl)
    else:
        t = torch.tensor([float(value)], dtype=torch.float64, device=device if dist.get_backend() != "gloo" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_over_ranks(value, device=None):
    """[value of rank 0, ..., value of rank W-1] on every rank (one small all-gather; a one-element list for one process)."""
    if not _multi():
        return [float(value)]
    on_host = _ctrl is not None or dist.get_backend() == "gloo"
    t = torch.tensor([float(value)], dtype=torch.float64, device="cpu" if on_host else device)
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t, group=_ctrl)
    return [float(x.item()) for x in out]


def barrier():
    """All ranks meet (control plane).  With the gloo control group a missing rank is a RuntimeError that names it after
    MGX_CTRL_TIMEOUT_S seconds, not a hang."""
    if not _multi():
        return
    if _ctrl is not None:
        dist.monitored_barrier(group=_ctrl, timeout=datetime.timedelta(seconds=CTRL_TIMEOUT_S))
    else:
        dist.barrier()
