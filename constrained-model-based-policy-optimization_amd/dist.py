"""Collectives for the sharded path: one process per GPU, torch.distributed (backend "nccl" = RCCL over
xGMI on ROCm; "gloo" for the CPU tests).

Replaces the mpi4py wrappers of ``utilities/mpi_tools.py:43-92`` at their call sites:
every payload is tiny (<= 2P+4 floats), so calls are fused (one all-reduce per CG / line-search step,
one per statistics pass) and weighted by sample counts instead of assuming equal shards
(``mpi_avg``, ``mpi_tools.py:67-69``).  Counts travel as float64 / int64, never float32
(``mpi_tools.py:59,83`` loses exactness above 2**24 samples).
"""
import os

import torch
import torch.distributed as td


class Comm:
    def __init__(self, device=None):
        self.enabled = td.is_available() and td.is_initialized()
        self.rank = td.get_rank() if self.enabled else 0
        self.world = td.get_world_size() if self.enabled else 1
        self.device = device
        # gloo moves host memory: device tensors are staged through the host (rehearsals of the multi-rank path on
        # fewer GPUs than ranks; the production backend is nccl = RCCL, which takes device tensors directly)
        self._stage = self.enabled and td.get_backend() == "gloo"

    def _run(self, fn, t):
        if self._stage and t.is_cuda:
            h = t.detach().cpu()
            fn(h)
            t.copy_(h)
        else:
            fn(t)
        return t

    @staticmethod
    def init_from_env(backend=None):
        """Join the job described by RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (torchrun)."""
        world = int(os.environ.get("WORLD_SIZE", "1"))
        if world > 1 and not td.is_initialized():
            backend = os.environ.get("CMBPO_DIST_BACKEND", backend)
            if backend is None:
                # nccl (= RCCL) wants a GPU per rank of this node ("Duplicate GPU detected" otherwise): ranks that share
                # a card -- a rehearsal of N ranks on a one-GPU box -- reduce through gloo and host staging instead
                n_dev = torch.cuda.device_count() if torch.cuda.is_available() else 0
                local_world = int(os.environ.get("LOCAL_WORLD_SIZE", world))
                backend = "nccl" if n_dev >= max(local_world, 1) else "gloo"
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            td.init_process_group(backend=backend)
        return Comm()

    def all_reduce_sum(self, t):
        """In-place SUM of a (device) tensor; a view of a larger tensor is reduced through a copy."""
        if self.world > 1:
            if t.is_contiguous() and t.storage_offset() == 0:
                self._run(lambda x: td.all_reduce(x, op=td.ReduceOp.SUM), t)
            else:
                tmp = t.contiguous().clone()
                self._run(lambda x: td.all_reduce(x, op=td.ReduceOp.SUM), tmp)
                t.copy_(tmp)
        return t

    def broadcast(self, t, root=0):
        """In-place broadcast of a contiguous tensor from `root` (trained ensemble weights: training runs as
        independent replicas, rank 0's result is the one every rank rolls out with)."""
        if self.world > 1:
            self._run(lambda x: td.broadcast(x, src=root), t)
        return t

    def all_reduce_max(self, t):
        if self.world > 1:
            self._run(lambda x: td.all_reduce(x, op=td.ReduceOp.MAX), t)
        return t

    def all_gather_i32(self, row):
        """row: int32[k] on the device -> int32[world, k] (rank-major)."""
        row = row.contiguous().clone()
        out = torch.empty((self.world, row.numel()), dtype=row.dtype, device=row.device)
        if self.world > 1:
            if self._stage and row.is_cuda:
                h_out, h_row = out.cpu(), row.cpu()
                td.all_gather_into_tensor(h_out.view(-1), h_row)
                out.copy_(h_out)
            else:
                td.all_gather_into_tensor(out.view(-1), row)
        else:
            out[0] = row
        return out

    def all_reduce_host(self, values):
        """SUM of a few host scalars (float64)."""
        if self.world == 1:
            return [float(v) for v in values]
        dev = "cpu" if self._stage else (self.device if self.device is not None else "cuda")
        t = torch.tensor([float(v) for v in values], dtype=torch.float64, device=dev)
        td.all_reduce(t, op=td.ReduceOp.SUM)
        return t.cpu().tolist()

    def barrier(self):
        if self.world > 1:
            td.barrier()


def budget_plan(rows, rank, max_samples):
    """Cross-shard form of the budget rule of ``samplers/model_sampler.py:282-287``.

    rows: int array [world, >=3] of per-rank {n_alive, n_too_uncertain, total_samples} in rank order
    (ranks own contiguous blocks of global branch ids, so rank order == global index order).
    Returns (excess, rank_off): `excess` surviving rows must be early-terminated, counted from global
    index 0; `rank_off` of them precede this rank's first surviving row.
    """
    rows = [[int(v) for v in r] for r in rows]
    g_alive = sum(r[0] for r in rows)
    g_unc = sum(r[1] for r in rows)
    g_total = sum(r[2] for r in rows)
    excess = max(g_total + g_alive - g_unc - int(max_samples), 0)
    rank_off = sum(r[0] - r[1] for r in rows[:rank])
    return excess, rank_off
