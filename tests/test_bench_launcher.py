"""`python bench.py --gpus N` must start N ranks itself (the reference's analogue is
``mpi_fork``, utilities/mpi_tools.py:7-37).  CPU, gloo, world 2: the launcher half and the rendezvous of the rank half
run end to end (`--dry-run`: no kernels -- those need a GPU; the N-rank rollout itself is covered on the GPU box by
tests/test_world2_gpu.py and `bench.py --gpus 2` under CMBPO_DIST_BACKEND=gloo)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(argv, env_extra=None, drop=("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")):
    env = {k: v for k, v in os.environ.items() if k not in drop}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, env=env, capture_output=True,
                          text=True, timeout=240)


def test_gpus2_spawns_two_ranks_and_reports_them():
    r = _run(["--gpus", "2", "--dry-run", "--steps", "3", "--warmup", "1"], {"CMBPO_DIST_BACKEND": "gloo"})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                      # exactly one JSON line, relayed from rank 0
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["ranks_seen"] == 2 and out["steps"] == 3 and out["warmup"] == 1


def test_a_rank_that_dies_at_start_up_ends_the_job():
    # rank 1 exits before the rendezvous: the launcher reports it and stops rank 0 instead of waiting for torch's
    # rendezvous timeout
    import time
    t0 = time.time()
    r = _run(["--gpus", "2", "--dry-run"], {"CMBPO_DIST_BACKEND": "gloo", "CMBPO_BENCH_TEST_FAIL_RANK": "1"})
    assert r.returncode == 7, (r.returncode, r.stderr[-2000:])
    assert "rank 1 exited with code 7" in r.stderr
    assert time.time() - t0 < 120


def test_world_mismatch_is_an_error_not_a_silent_one_rank_run():
    # a torchrun-style environment with fewer ranks than --gpus asks for must not fall back to that many ranks
    r = _run(["--gpus", "4", "--dry-run"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"}, drop=())
    assert r.returncode != 0
    assert "--gpus 4" in r.stderr


def test_strong_scaling_shards_are_a_partition():
    sys.path.insert(0, ROOT)
    import importlib
    bench = importlib.import_module("bench")
    for total, world in ((100000, 4), (1000000, 8), (1001, 3), (7, 8)):
        blocks = [bench.shard_of(total, r, world) for r in range(world)]
        assert blocks[0][0] == 0 and blocks[-1][1] == total
        assert all(a[1] == b[0] for a, b in zip(blocks, blocks[1:]))
        assert max(h - l for l, h in blocks) - min(h - l for l, h in blocks) <= 1


import pytest  # noqa: E402


@pytest.mark.gpu
def test_gpus2_runs_the_rollout_on_two_ranks():
    """The launcher with the real kernels: two ranks (both on this box's one GPU, gloo instead of RCCL -- only the
    transport differs) run the sharded rollout phase, weak and strong scaling."""
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    for extra, total in ((["--branches", "3000"], 6000), (["--branches", "3000", "--scaling", "strong"], 3000)):
        r = _run(["--gpus", "2", "--steps", "1", "--warmup", "1", "--no-extras", "--maxroll", "6"] + extra,
                 {"CMBPO_DIST_BACKEND": "gloo"})
        assert r.returncode == 0, r.stderr[-2000:]
        out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
        assert out["n_gpus"] == 2 and out["config"]["branches_total"] == total
        assert out["value"] > 0 and out["config"]["samples_per_step"] == total * 5      # every branch lives 5 steps


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_gpus2_strong_config5_shards_with_a_binding_budget():
    """BASELINE config 5's shard at size: 250 000 branches x maxroll 26 as two contiguous shards of 125 000 (the per-rank
    shape of 1 M over 8), 'uncertainty' mode with a max_samples that binds -- the cross-rank budget plan (one all-gather
    per step, dist.budget_plan) decides the early terminations.  Two ranks on this box's one GPU over gloo."""
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "1", "--no-extras", "--scaling", "strong", "--branches", "250000",
              "--maxroll", "26", "--rollout-mode", "uncertainty", "--budget-frac", "0.5"], {"CMBPO_DIST_BACKEND": "gloo"})
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    cfg = out["config"]
    assert out["n_gpus"] == 2 and cfg["branches_total"] == 250000 and cfg["branches_per_gpu"] == 125000
    assert cfg["max_samples"] == int(0.5 * 250000 * 25)
    assert cfg["samples_per_step"] == cfg["max_samples"]              # the job's budget, met exactly across the two shards
    assert cfg["n_budget_terminated_per_phase"] > 0 and cfg["sampler_steps_per_phase_rank0"] >= 12
