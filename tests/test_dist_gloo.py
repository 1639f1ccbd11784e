"""CPU, world_size 2, gloo: the sharded path's host logic -- collectives wrapper, the
cross-shard budget rule, count-weighted reductions and the two-pass advantage statistics.

The kernels themselves need a GPU; what is checked here is everything the N > 1 path adds around them.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out_dir):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as td
    td.init_process_group("gloo", rank=rank, world_size=world)
    import cmbpo_amd  # noqa: F401
    from cmbpo_amd.dist import Comm, budget_plan
    comm = Comm(device="cpu")
    assert comm.rank == rank and comm.world == world
    res = {}

    # 1. collectives: sum on a view of a larger tensor, gather of int rows, host scalars
    stats = torch.arange(16, dtype=torch.float64) * (rank + 1)
    comm.all_reduce_sum(stats[8:13])
    res["stats"] = stats.numpy()
    rows = comm.all_gather_i32(torch.tensor([10 + rank, rank, 100 * rank, 0], dtype=torch.int32))
    res["rows"] = rows.numpy()
    res["host"] = np.array(comm.all_reduce_host([rank + 1, 2.5]))
    bc = torch.full((5,), float(rank + 7))
    comm.broadcast(bc, root=1)                     # trained-weights hand-over (PE.sync_weights)
    res["bcast"] = bc.numpy()

    # 2. budget rule across shards == the unsharded rule (model_sampler.py:282-287)
    rng = np.random.default_rng(0)                 # same stream on both ranks: the global problem
    B = 101
    unc = rng.random(B) < 0.3
    total, max_samples = 700, 740
    n = total + B - unc.sum()
    n = max(n - max_samples, 0)
    expect = unc.copy()
    surv = np.flatnonzero(~unc)
    expect[surv[:n]] = True
    lo, hi = (0, 47) if rank == 0 else (47, B)     # unequal contiguous shards
    my_unc = unc[lo:hi]
    # per-rank total_samples: any split of the global total
    row = torch.tensor([hi - lo, int(my_unc.sum()), 300 if rank == 0 else 400, 0], dtype=torch.int32)
    g = comm.all_gather_i32(row).numpy()
    excess, rank_off = budget_plan(g, rank, max_samples)
    local_rank = np.cumsum(~my_unc) - (~my_unc)     # exclusive scan of the survivor flags (the kernel's job)
    mine = my_unc | ((~my_unc) & (rank_off + local_rank < excess))
    res["budget_ok"] = bool(np.array_equal(mine, expect[lo:hi]))
    res["excess"] = excess

    # 3. count-weighted mean of per-shard gradient sums (not mpi_avg of per-rank means)
    x = rng.standard_normal((B, 5))
    part = torch.from_numpy(x[lo:hi].sum(0))
    cnt = comm.all_reduce_host([hi - lo])[0]
    comm.all_reduce_sum(part)
    res["wmean_err"] = float(np.abs(part.numpy() / cnt - x.mean(0)).max())

    # 4. two-pass statistics (utilities/mpi_tools.py:71-87) from per-shard partial sums
    adv = rng.standard_normal(B) * 3 + 1
    s = torch.tensor([adv[lo:hi].sum(), float(hi - lo)], dtype=torch.float64)
    comm.all_reduce_sum(s)
    mean = s[0] / s[1]
    sq = torch.tensor([((adv[lo:hi] - mean.item()) ** 2).sum()], dtype=torch.float64)
    comm.all_reduce_sum(sq)
    res["stat_err"] = float(max(abs(mean.item() - adv.mean()), abs(np.sqrt(sq.item() / s[1].item()) - adv.std())))
    comm.barrier()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), **res)
    td.destroy_process_group()


@pytest.mark.timeout(180)
def test_world2_gloo(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r = [np.load(os.path.join(tmp_path, f"rank{k}.npz")) for k in range(world)]
    for k in range(world):
        want = np.arange(16, dtype=np.float64) * (k + 1)
        want[8:13] = np.arange(8, 13) * 3.0                      # (1 + 2) * value, only the reduced slice
        np.testing.assert_array_equal(r[k]["stats"], want)
        np.testing.assert_array_equal(r[k]["rows"], [[10, 0, 0, 0], [11, 1, 100, 0]])
        np.testing.assert_array_equal(r[k]["host"], [3.0, 5.0])
        np.testing.assert_array_equal(r[k]["bcast"], [8.0] * 5)
        assert bool(r[k]["budget_ok"]) and int(r[k]["excess"]) > 0
        assert float(r[k]["wmean_err"]) < 1e-12 and float(r[k]["stat_err"]) < 1e-12


def test_budget_plan_single_rank_matches_reference_formula():
    import cmbpo_amd  # noqa: F401
    from cmbpo_amd.dist import budget_plan
    assert budget_plan([[50, 7, 300]], 0, 333) == (10, 0)      # 300 + 50 - 7 - 333
    assert budget_plan([[50, 7, 100]], 0, 333) == (0, 0)
    assert budget_plan([[10, 2, 0], [20, 5, 0], [30, 0, 0]], 2, 40) == (13, 23)
