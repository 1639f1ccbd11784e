"""bench.py -- imagined env-steps/s of the MI355X-native CMBPO rollout (+ CPO update ms).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--branches B] [--task AntSafe-v2]
                    [--scaling weak|strong] [--maxroll T]

One "step" = one full imagined-rollout phase of the reference trainer on one batch of start states:
``ModelSampler.reset -> sample() x (maxroll-1) -> finish_all_paths -> ModelBuffer.get()``
(algorithms/cmbpo.py:251-269, the span the reference times as `times/epoch_rollout_model`, :293),
with the inputs (start states, weights) already resident in HBM.  metric value = imagined env-steps
(sum of alive branches over the steps, `msampler/samples_added`) per second, whole job over all ranks.

Workload at N = 1: AntSafe-v2 shapes (obs 29, act 8), 7-member 512x512 swish ensemble, 5 elites, 3+3
critic members, 128x128 tanh policy, B = 100 000 branches per GPU, maxroll 35 (34 stored steps),
fixed-horizon mode -- the north-star configuration ("AntSafe 7-ensemble 100k-branch rollouts at
1 MI355X").  N > 1, one process per GPU: ``--scaling weak`` (default) rolls out B branches on every rank,
``--scaling strong`` shards B branches over the ranks as contiguous blocks of global branch ids (BASELINE
config 4: ``--gpus 4 --scaling strong``; config 5: ``--gpus 8 --scaling strong --branches 1000000 --maxroll 26``).
The data path has no collective; per step the ranks exchange three host scalars (global alive ratio, the
reference's `alive_ratio`), per get() the advantage statistics.  Synthetic seeded weights / states.

Launch: under ``torch.distributed.run`` (RANK / WORLD_SIZE set) every process is one rank.  Started plainly as
``python bench.py --gpus N`` with N > 1, this process becomes a launcher (the reference's analogue is
``mpi_fork``, utilities/mpi_tools.py:7-37): it starts N rank processes with RANK / LOCAL_RANK / WORLD_SIZE /
MASTER_ADDR / MASTER_PORT set, relays rank 0's JSON line and exits with the first non-zero exit code.  The
launcher never imports torch or touches the GPU, and nothing is exec'ed from a process that has.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BIND_DKL_SCALE = 4.0            # budget-binding sub-results: DKL limit = 4 x the 5-step calibration: ~40 % of the branches die of uncertainty, the rest live until the budget binds (tools/probe_budget.py)
BIND_BUDGET_FRAC = 0.5          # ... and max_samples = 0.5 * B * T: binds at step ~19


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--branches", type=int, default=100000,
                    help="rollout branches per GPU (--scaling weak) or in total (--scaling strong)")
    ap.add_argument("--task", default="AntSafe-v2")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak")
    ap.add_argument("--maxroll", type=int, default=35, help="max_path_length of the rollout (stored steps = maxroll - 1)")
    ap.add_argument("--rollout-mode", choices=("schedule", "uncertainty"), default="schedule",
                    help="'uncertainty': branches end when their accumulated ensemble DKL passes --dkl-scale x the calibrated limit")
    ap.add_argument("--dkl-scale", type=float, default=BIND_DKL_SCALE)
    ap.add_argument("--budget-frac", type=float, default=0.0,
                    help="> 0: every sample() gets max_samples = frac * (the job's branches) * (maxroll - 1), the early-termination "
                         "rule of samplers/model_sampler.py:282-287 (sharded: the cross-rank budget plan, one all-gather per step)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="headline rollout only (no update / training / sub-configs)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--dry-run", action="store_true",
                    help="rendezvous, one all-reduce and the JSON line only (launcher test on machines without a GPU)")
    return ap.parse_args(argv)


def spawn_ranks(args, argv, deadline_s=900.0):
    """Launcher half of ``python bench.py --gpus N``: N rank processes of this same script, one per GPU.  All children
    are watched from launch time on: the first one to exit non-zero (or the deadline) ends the job -- the others are
    terminated, never left waiting in a rendezvous or a collective for a rank that is gone."""
    import threading
    t_launch = time.time()
    s = socket.socket()
    s.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for rank in range(args.gpus):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(args.gpus),
                   LOCAL_WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if rank == 0 else sys.stderr, text=(rank == 0)))
    result = []

    def relay():                         # rank 0 prints the one JSON line; anything else it prints is passed on
        for out in procs[0].stdout:
            out = out.rstrip("\n")
            if out.startswith("{") and '"metric"' in out:
                result.append(out)
            else:
                print(out, file=sys.stderr, flush=True)

    reader = threading.Thread(target=relay, daemon=True)
    reader.start()
    rc, failed = 0, None
    while True:
        codes = [p.poll() for p in procs]
        bad = [(k, c) for k, c in enumerate(codes) if c not in (None, 0)]
        if bad:
            failed, rc = bad[0]
            print(f"bench.py launcher: rank {failed} exited with code {rc}; stopping the other ranks", file=sys.stderr)
            break
        if all(c == 0 for c in codes):
            break
        if time.time() - t_launch > deadline_s:
            print(f"bench.py launcher: no result after {deadline_s:.0f} s; stopping all ranks", file=sys.stderr)
            rc = 124
            break
        time.sleep(0.05)
    if rc != 0:
        for p in procs:                  # the exact processes we started
            if p.poll() is None:
                p.terminate()
        t_kill = time.time() + 10
        for p in procs:
            try:
                p.wait(timeout=max(0.1, t_kill - time.time()))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
    reader.join(timeout=5)
    line = result[-1] if result else None
    if rc == 0 and line is None:
        print("bench.py launcher: rank 0 printed no result line", file=sys.stderr)
        rc = 1
    if rc == 0:
        n = json.loads(line).get("n_gpus")
        if n != args.gpus:
            print(f"bench.py launcher: result line says n_gpus = {n}, expected {args.gpus}", file=sys.stderr)
            rc = 1
    if rc == 0:
        print(line, flush=True)
    return rc


if __name__ == "__main__" and "WORLD_SIZE" not in os.environ:
    _args = parse_args()
    if _args.gpus > 1:                   # before torch is imported: the launcher never initialises a GPU
        sys.exit(spawn_ranks(_args, sys.argv[1:]))

import numpy as np  # noqa: E402
import torch  # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_BF16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: bf16 dense
SPLIT_TERMS = 6                 # bf16 MFMAs per float32 product on the split path (csrc/ens_split.hip)
MAXROLL = 35                    # default max_path_length (--maxroll)


class _Space:
    def __init__(self, d):
        self.shape = (d,)


def build_world(seed, task, hidden=512, E=7):
    """Seeded synthetic weights of the reference's shapes."""
    from cmbpo_amd import synthetic
    rng = np.random.default_rng(seed)
    D, A = synthetic.ENV_DIMS[task]
    ws, bs = synthetic.ensemble_weights(rng, E, D + A, hidden, 2 * (D + 1), out_scale=0.05)
    sc_in = synthetic.scaler(rng, D + A, hit_clamp=False)
    sc_out = synthetic.scaler(rng, D + 1, hit_clamp=False)
    sc_out = (sc_out[0] * 0.1, (sc_out[1] * 1e-3).astype(np.float32))     # small deltas: branches survive
    pol = synthetic.policy_params(rng, D, A)
    crit = []
    for _ in range(2):
        cw, cb = synthetic.ensemble_weights(rng, 3, D, 128, 1)
        crit.append((cw, cb, synthetic.scaler(rng, D, hit_clamp=False), synthetic.scaler(rng, 1, hit_clamp=False)))
    return dict(obs_dim=D, act_dim=A, ws=ws, bs=bs, sc_in=sc_in, sc_out=sc_out, pol=pol, v=crit[0], vc=crit[1],
                elites=[0, 2, 3, 5, 6])


def build_hip(w, task, B, device, comm=None, maxroll=MAXROLL, mode="schedule"):
    from cmbpo_amd.cpo_policy import CPOPolicy
    from cmbpo_amd.fake_env import FakeEnv
    from cmbpo_amd.model_sampler import ModelSampler
    from cmbpo_amd.modelbuffer import ModelBuffer
    from cmbpo_amd.pens import PE
    D, A = w["obs_dim"], w["act_dim"]
    hidden = w["ws"][1].shape[1]
    model = PE(D + A, D + 1, hidden_dims=(hidden, hidden), num_networks=w["ws"][0].shape[0], num_elites=5,
               loss="MSPE", use_scaler_in=True, use_scaler_out=True, device=device)
    model.set_weights(w["ws"], w["bs"], w["sc_in"], w["sc_out"])
    model.set_elites(w["elites"])
    policy = CPOPolicy(_Space(D), _Space(A), a_hidden_layer_sizes=(128, 128), vf_hidden_layer_sizes=(128, 128),
                       vf_ensemble_size=3, vf_elites=2, vf_activation="swish", vf_loss="MSE", device=device,
                       cost_gamma=0.97, cost_lam=0.5, lam=0.95, max_path_length=maxroll)
    policy.actor.set_params(w["pol"])
    policy.v.set_weights(*w["v"])
    policy.vc.set_weights(*w["vc"])

    class _Env:
        observation_space, action_space = _Space(D), _Space(A)

    env = FakeEnv(_Env(), task, model, True, True, False)
    pool = ModelBuffer(B, D, A, maxroll, device=device, comm=comm)
    pool.initialize(policy.pi_info_shapes, gamma=0.99, lam=0.95, cost_gamma=0.97, cost_lam=0.5)
    sampler = ModelSampler(max_path_length=maxroll, batch_size=B, rollout_mode=mode, comm=comm)
    sampler.initialize(env, policy, pool)
    return sampler, pool, env, policy


def rollout_phase(sampler, pool, start, max_samples=None):
    """reset -> sample until everything is finished -> finish_all_paths -> get(); returns samples."""
    sampler.reset(start)
    while sampler.any_alive() and pool.has_room:
        sampler.sample_many(max_samples=max_samples)     # the loop over sample() in native code (same steps, same results)
    diag = sampler.finish_all_paths()
    res, bdiag = pool.get(as_tensors=True)
    return int(diag["msampler/samples_added"]), res


def ens_flops_per_row(w):
    D, A, E, H = w["obs_dim"], w["act_dim"], w["ws"][0].shape[0], w["ws"][1].shape[1]
    return 2.0 * E * ((D + A) * H + H * H + H * 2 * (D + 1))


def ens_roofline(w, events):
    """Roofline entry of the dominant kernel (the fused ensemble forward) from HIP events on its launch stream.
    `achieved` = ALGORITHMIC float32 flops per second; the peak is that of the matrix path the launches took."""
    from cmbpo_amd import _lib
    H = w["ws"][1].shape[1]
    k_ms = [s.elapsed_time(e) for s, e, _ in events]
    k_rows = [n for _, _, n in events]
    if not k_ms:
        return None
    flop_row = ens_flops_per_row(w)
    achieved = flop_row * float(np.sum(k_rows)) / (float(np.sum(k_ms)) * 1e-3) / 1e12
    path = _lib.lib().cmbpo_get_ens_matrix_path() if H == 512 else 0
    peak, basis, kernel = {
        0: (PEAK_FP32_MFMA_TFLOPS, "fp32 MFMA dense 157.3 TFLOP/s", "ens_mlp_kernel<512,*,swish,prob>"),
        1: (PEAK_BF16_MFMA_TFLOPS / 6, "bf16 dense 2500 TFLOP/s / 6 MFMAs per f32 product",
            "ens_split_kernel (6 x v_mfma_f32_32x32x16_bf16 per f32 product)"),
        2: (PEAK_BF16_MFMA_TFLOPS / 3, "f16 dense 2500 TFLOP/s / 3 MFMAs per f32 product",
            "ens_h3_kernel (3 x v_mfma_f32_32x32x16_f16 per f32 product; a forward is the full rounds of one item size -- 128 rows, "
            "<S0,OTP,4>, at the headline shape -- plus, when the cost model prefers it, the leftovers as shorter items -- 64 rows, "
            "<S0,OTP,2>, there -- in a launch of their own: avg_launch_ms covers both)"),
    }[path]
    return {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
            "kernel": kernel, "avg_launch_ms": float(np.mean(k_ms)), "avg_rows_per_launch": float(np.mean(k_rows)),
            "launches": len(k_ms), "flop_per_branch_step": flop_row, "peak_basis": basis,
            # executed matrix flops over the dense f16 / bf16 peak: how busy the matrix pipe is at nominal clock
            "mfma_pipe_util": achieved * {0: 1.0, 1: 6.0, 2: 3.0}[path] / (PEAK_FP32_MFMA_TFLOPS if path == 0 else PEAK_BF16_MFMA_TFLOPS),
            "vs_fp32_mfma_peak": achieved / PEAK_FP32_MFMA_TFLOPS,
            "vs_r01_basis_bf16_over_6": achieved / (PEAK_BF16_MFMA_TFLOPS / 6)}


def get_roofline(w, get_events):
    """HBM entry of ModelBuffer.get(): the flatten kernels' algorithmic bytes (every stored float read once, written
    once: SURVEY 8d, ~490 B per sample at AntSafe shapes) over their HIP-event time."""
    if not get_events:
        return None
    D, A = w["obs_dim"], w["act_dim"]
    per_sample = 2.0 * 4.0 * (D + 3 * A + 8)
    fl_ms = [b.elapsed_time(c) for _, b, c, _ in get_events]
    all_ms = [a.elapsed_time(c) for a, _, c, _ in get_events]
    n = [k for _, _, _, k in get_events]
    achieved = per_sample * float(np.sum(n)) / (float(np.sum(fl_ms)) * 1e-3) / 1e9
    return {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0,
            "kernel": "flatten_kernel (vector-field tiles + scalar-field tiles in one launch)", "flatten_us": float(np.mean(fl_ms)) * 1e3,
            "get_us_offsets_to_flatten": float(np.mean(all_ms)) * 1e3, "bytes_per_sample": per_sample,
            "samples": float(np.mean(n))}


def run_config(name, task, B, maxroll, mode, device, reps=3, dkl_scale=0.6, budget_frac=0.75, seed=0, budget_binds=False):
    """One sub-result of the bench line: a shipped configuration's rollout phase on this rank (no collectives).
    budget_binds: the entry is only valid if the phase ends BECAUSE `max_samples` bound (the shipped configurations stop
    their rollouts at the model batch, algorithms/cmbpo.py:254-263): >= 12 steps, samples >= 0.99 * max_samples and
    branches ended by the early-termination rule of samplers/model_sampler.py:282-287 -- anything else is a bench error."""
    from cmbpo_amd import synthetic
    w = build_world(seed, task)
    sampler, pool, env, policy = build_hip(w, task, B, device, None, maxroll, mode)
    start = torch.from_numpy(synthetic.start_states(np.random.default_rng(100), B, task)).to(device)
    max_samples, lim = None, None
    if mode == "uncertainty":
        # the DKL limit calibrated like the trainer's (algorithms/cmbpo.py:197-199), tightened so that branches die of
        # uncertainty along the rollout, and a sample budget that the early-termination rule has to enforce
        # (samplers/model_sampler.py:275-287)
        sampler.reset(start)
        lim = float(sampler.compute_dynamics_dkl(start[: min(B, 5000)], depth=5)) * dkl_scale
        sampler.set_rollout_dkl(lim)
        max_samples = int(budget_frac * B * (maxroll - 1))
    rollout_phase(sampler, pool, start, max_samples)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    samples = 0
    for _ in range(reps):
        n, _ = rollout_phase(sampler, pool, start, max_samples)
        samples += n
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    env.kernel_events, pool.get_events = [], []      # one more phase through the instrumented (separate-call) path
    steps_before = sampler._n_episodes if hasattr(sampler, "_n_episodes") else 0
    rollout_phase(sampler, pool, start, max_samples)
    torch.cuda.synchronize()
    n_steps = sampler._n_episodes
    n_budget = int(sampler.n_budget_terminated)
    out = {"name": name, "task": task, "branches": B, "maxroll": maxroll, "rollout_mode": mode,
           "value": samples / dt, "unit": "imagined env-steps/s", "ms_per_phase": dt / reps * 1e3,
           "samples_per_phase": samples / reps, "steps_per_phase": n_steps,
           "us_per_step": dt / reps / max(n_steps, 1) * 1e6,
           "roofline": ens_roofline(w, env.kernel_events), "gae_get": get_roofline(w, pool.get_events)}
    if mode == "uncertainty":
        out.update(dkl_lim=lim, max_samples=max_samples, dkl_scale=dkl_scale, budget_frac=budget_frac,
                   n_budget_terminated=n_budget,
                   ended_by="budget (max_samples bound)" if n_budget > 0 else "uncertainty (every branch passed the DKL limit)")
        if budget_binds:
            ok = n_steps >= 12 and n_budget > 0 and out["samples_per_phase"] >= 0.99 * max_samples
            if not ok:
                raise RuntimeError(f"bench.py: {name}: the sample budget did not bind "
                                   f"(steps {n_steps}, samples {out['samples_per_phase']:.0f} of {max_samples}, "
                                   f"{n_budget} budget-terminated branches)")
    env.kernel_events, pool.get_events = None, None
    del sampler, pool, env, policy
    torch.cuda.empty_cache()
    return out


def constrained_batch(res, seed=11):
    """SURVEY 8d M1's update batch: the rollout's own (obs, act, logp, mu, log_std) with adv, cadv ~ N(0, 1) and
    cost ~ Bernoulli(0.05) -- a non-zero cost gradient, so the update takes the reference's full path (two CG solves +
    two more products, policies/cpo_policy.py:210-224) instead of the TRPO short cut (:216-219)."""
    gen = torch.Generator(device=res[0].device)
    gen.manual_seed(seed)
    n = res[0].shape[0]
    buf = list(res)
    buf[2] = torch.randn(n, generator=gen, device=res[0].device)
    buf[3] = torch.randn(n, generator=gen, device=res[0].device)
    buf[9] = (torch.rand(n, generator=gen, device=res[0].device) < 0.05).float()
    return buf


def time_update(policy, res, n, reps=3, cost_lim=None):
    """CPOPolicy.update_policy on the first n samples of a get() list (device tensors); median ms.  cost_lim: the limit
    of this update (the learned margin starts at 0 every repetition, real-cost history = the limit)."""
    buf = [x[:n].contiguous() for x in res]
    p0 = policy.actor.get_flat_params()
    times, info = [], None
    keep = policy.cost_lim, policy.real_c_buffer
    if cost_lim is not None:
        policy.cost_lim, policy.real_c_buffer = float(cost_lim), [float(cost_lim)] * 300
    for _ in range(reps + 1):
        policy.set_params(p0)
        policy.agent.margin = 0
        policy.ops.n_fvp = 0
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        info = policy.update_policy(buf)
        torch.cuda.synchronize()
        times.append((time.perf_counter() - t0) * 1e3)
    policy.set_params(p0)
    policy.cost_lim, policy.real_c_buffer = keep
    info = dict(info, n_fvp=policy.ops.n_fvp)
    return float(np.median(times[1:])), info


def time_fvp(policy, reps=5):
    """One Fisher-vector product (direction pack + pi_kernel<FVP> on the saved activations + the fixed-order sum of the
    workgroups' partials) on the batch the last update_policy left bound; ms per product by events on the launch stream."""
    ops = policy.ops
    with torch.cuda.device(ops.device):
        ops.loss_grad(0)                       # saves the hidden activations, as the update's first gradient does
        ops.dirv.normal_()
        ops.fvp_raw(ops.dirv)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            ops.fvp_raw(ops.dirv)
        e1.record()
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def time_train_steps(pe, x, t, batch, steps):
    """Ensemble training: average microseconds of one train_op (forward, loss, backward, weight
    gradients, Adam) with per-member bootstrap rows of device-resident data."""
    tr = pe._ensure_trainer(batch)
    n = int(x.shape[0])
    E = pe.num_nets
    idx = torch.randint(0, n, (E, batch * 8), dtype=torch.int32, device=x.device)
    ws, bs = pe.get_weights()
    for k in range(3):
        tr.step(x, t, idx.data_ptr() + 4 * batch * (k % 8), batch * 8, batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        tr.step(x, t, idx.data_ptr() + 4 * batch * (k % 8), batch * 8, batch)
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / steps * 1e6
    pe.set_weights(ws, bs)        # the timing steps must not leak into the benchmark's ensembles
    tr.reset_optimizer()
    return us


def cpu_train_baseline(E, I, H, O, loss, batch):
    """One train_op of the oracle (torch-CPU autograd restatement of the TF graph); microseconds."""
    from oracle import reftrain
    rng = np.random.default_rng(3)
    ws = [(rng.standard_normal(s) * 0.05).astype(np.float32) for s in ((E, I, H), (E, H, H), (E, H, O))]
    bs = [np.zeros((E, 1, s), np.float32) for s in (H, H, O)]
    ref = reftrain.EnsembleTrainer(ws, bs, loss_type=loss, decays=(2.5e-7, 5e-7, 1e-6))
    D = O // 2 if loss == "MSPE" else O
    x = rng.standard_normal((E, batch, I)).astype(np.float32)
    t = rng.standard_normal((E, batch, D)).astype(np.float32)
    ref.step(x, t)
    t0 = time.perf_counter()
    for _ in range(2):
        ref.step(x, t)
    return (time.perf_counter() - t0) / 2 * 1e6


def cpu_update_baseline(w, res, n, cost_lim=10.0, maxroll=MAXROLL):
    """One CPO update of the oracle (torch-CPU autograd graph + update_pi) on n samples; ms."""
    from oracle import refupdate
    host = [x[:n].cpu().numpy() if hasattr(x, "cpu") else np.asarray(x[:n]) for x in res]
    obs, act, adv, cadv, _, _, logp, _, _, cost, ls, mu = host
    D, A = w["obs_dim"], w["act_dim"]
    graph = refupdate.PolicyGraph(D, A, dict(obs=obs, act=act, adv=adv, cadv=cadv, logp_old=logp, cost=cost,
                                             mu_old=mu, log_std_old=ls), max_path_length=maxroll)
    params = np.concatenate([p.reshape(-1) for p in w["pol"]]).astype(np.float32)
    agent = refupdate.AgentState(maxroll, constrained=True)

    def grads():
        g, b, lo, sc = graph.grads(params)
        return g, b, lo, sc, float(graph.cur_cret_avg())

    t0 = time.perf_counter()
    with np.errstate(all="ignore"):
        refupdate.update_pi(agent, dict(grads=grads, Hx=lambda v: graph.hvp(params, v, 0.1),
                                        set_and_eval=lambda p: graph.evals(np.asarray(p, np.float32))),
                            params, 0.01, cost_lim, [cost_lim] * 300)
    return (time.perf_counter() - t0) * 1e3


def host_threads():
    """Threads the CPU baselines run on: the cores this process may use (a GPU box hands a job a share of its cores;
    BLAS pools sized for the whole machine oversubscribe that share several times over)."""
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    try:        # cgroup v2 CPU quota, if any
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, -(-int(quota) // int(period))))
    except Exception:
        pass
    n = max(1, min(n, int(os.environ.get("CMBPO_CPU_THREADS", 16))))     # a one-GPU box's share of the host is 16 cores
    try:
        from threadpoolctl import threadpool_limits
        threadpool_limits(limits=n)
    except Exception:
        pass
    torch.set_num_threads(n)
    return n


def cpu_baseline(w, task, seconds=20.0, runs=5):
    """The oracle (NumPy restatement of the reference semantics, NOT TF 1.14) on the host cores: median of `runs`
    bounded rollout phases (B = 2000 branches x 6 steps + finish / get each), plus one step of the full 100 000-branch
    workload."""
    from oracle import refcpu
    from cmbpo_amd import synthetic
    threads = host_threads()
    model = lambda x: refcpu.ens_forward(x, w["ws"], w["bs"], w["sc_in"], w["sc_out"])
    policy = lambda obs, eps: refcpu.policy_forward(obs, w["pol"], eps)
    v = lambda obs: refcpu.ens_predict_mean(obs, *w["v"])[:, 0]
    vc = lambda obs: refcpu.ens_predict_mean(obs, *w["vc"])[:, 0]
    elites = np.asarray(w["elites"], np.int32)

    def phase(B, horizon, seed):
        rng = np.random.default_rng(seed)
        orc = refcpu.RolloutOracle(model, policy, v, vc, task, w["obs_dim"], w["act_dim"], horizon + 1, "schedule",
                                   float("inf"))
        start = synthetic.start_states(rng, B, task)
        t0 = time.perf_counter()
        orc.reset(start)
        steps = 0
        with np.errstate(all="ignore"):
            while orc.alive.any():
                n = int(orc.alive.sum())
                orc.sample(rng.standard_normal((n, w["act_dim"])).astype(np.float32),
                           elites[rng.integers(0, len(elites), n)])
                steps += 1
            orc.finish_all()
            orc.get()
        return orc.tot["samples"] / (time.perf_counter() - t0), steps

    B, H = 2000, 6
    phase(B, 2, 0)                                   # warm-up (BLAS threads, page faults)
    rates, t_begin = [], time.perf_counter()
    for r in range(runs):
        rates.append(phase(B, H, 1 + r)[0])
    dt = time.perf_counter() - t_begin
    t1 = time.perf_counter()
    rate_full, _ = phase(100000, 1, 99)              # one step + finish / get of the headline batch
    dt_full = time.perf_counter() - t1
    return dict(value=float(np.median(rates)), unit="imagined env-steps/s", cores=int(threads), kind="port",
                runs=[float(x) for x in rates], value_b100k_one_step=float(rate_full),
                sample=f"NumPy oracle (restatement of reference semantics, not TF 1.14), {int(threads)} threads: median of "
                       f"{runs} phases of B={B} branches x {H} steps of the same workload + finish/get ({dt:.1f} s in all); "
                       f"value_b100k_one_step = one step + finish/get at B=100000 ({dt_full:.1f} s)")


def dry_run(args):
    """Launcher / rendezvous check without a GPU: every rank joins, one all-reduce, rank 0 prints the line."""
    import cmbpo_amd  # noqa: F401
    from cmbpo_amd.dist import Comm
    if os.environ.get("CMBPO_BENCH_TEST_FAIL_RANK") == os.environ.get("RANK", "0"):
        sys.exit(7)                      # launcher test: a rank that dies before the rendezvous
    comm = Comm.init_from_env(None if torch.cuda.is_available() else "gloo")
    seen = torch.zeros(max(comm.world, 1), dtype=torch.float64)
    seen[comm.rank] = 1.0
    comm.all_reduce_sum(seen)
    comm.barrier()
    if comm.world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but the job has {comm.world} ranks", file=sys.stderr)
        sys.exit(2)
    if comm.rank == 0:
        print(json.dumps({"metric": "imagined env-steps/sec (ensemble rollout) + CPO update ms", "value": None,
                          "unit": "imagined env-steps/s", "n_gpus": comm.world, "steps": args.steps,
                          "warmup": args.warmup, "scaling": args.scaling, "dry_run": True,
                          "ranks_seen": int(seen.sum().item())}), flush=True)
    comm.barrier()


def shard_of(total, rank, world):
    """Contiguous block of global branch ids of one rank: [lo, hi)."""
    return (total * rank) // world, (total * (rank + 1)) // world


def main():
    args = parse_args()
    if args.dry_run:
        return dry_run(args)

    import cmbpo_amd  # noqa: F401
    from cmbpo_amd import _lib, synthetic
    from cmbpo_amd.dist import Comm
    _lib.lib()   # fail loudly if the HIP library is missing

    local_rank = int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    comm = Comm.init_from_env(None)      # nccl (= RCCL) with a GPU per rank, gloo when ranks share a card (dist.py)
    comm.device = device
    rank, world = comm.rank, comm.world
    if world != args.gpus:      # never fall back to fewer ranks than asked for
        print(f"bench.py: --gpus {args.gpus} but the job has {world} rank(s)", file=sys.stderr)
        sys.exit(2)

    task, maxroll = args.task, args.maxroll
    if args.scaling == "strong":   # the job's B branches as contiguous shards of global branch ids
        lo, hi = shard_of(args.branches, rank, world)
        B = hi - lo
        all_states = synthetic.start_states(np.random.default_rng(100), args.branches, task)
        start = torch.from_numpy(all_states[lo:hi]).to(device)
    else:
        B = args.branches
        start = torch.from_numpy(synthetic.start_states(np.random.default_rng(100 + rank), B, task)).to(device)
    w = build_world(0, task)
    sampler, pool, env, policy = build_hip(w, task, B, device, comm if world > 1 else None, maxroll, args.rollout_mode)
    max_samples, dkl_lim = None, None
    branches_total = args.branches if args.scaling == "strong" else B * world
    if args.rollout_mode == "uncertainty":
        # one limit for the whole job: the mean of the ranks' calibrations (algorithms/cmbpo.py:197-199)
        sampler.reset(start)
        lim = float(sampler.compute_dynamics_dkl(start[: min(B, 5000)], depth=5))
        dkl_lim = float(comm.all_reduce_host([lim])[0]) / world * args.dkl_scale
        sampler.set_rollout_dkl(dkl_lim)
    if args.budget_frac > 0:
        max_samples = int(args.budget_frac * branches_total * (maxroll - 1))

    for _ in range(args.warmup):
        rollout_phase(sampler, pool, start, max_samples)
    comm.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    samples, n_budget, n_steps = 0, 0, 0
    for _ in range(args.steps):
        n, _ = rollout_phase(sampler, pool, start, max_samples)
        samples += n
        n_budget += int(sampler.n_budget_terminated)
        n_steps += int(sampler._n_episodes)
    torch.cuda.synchronize()
    comm.barrier()
    dt = time.perf_counter() - t0
    # one more phase with HIP events around the dominant kernel and get()'s kernels (events need the step as separate
    # calls: the timed phases above run it as the single call the sampler makes by default)
    env.kernel_events, pool.get_events = [], []
    rollout_phase(sampler, pool, start, max_samples)
    torch.cuda.synchronize()
    events, env.kernel_events = env.kernel_events, None
    get_events, pool.get_events = pool.get_events, None

    t = torch.tensor([dt], dtype=torch.float64, device=device)
    comm.all_reduce_max(t)
    dt_max = float(t.item())
    tot, n_budget = comm.all_reduce_host([samples, n_budget])

    extras = not args.no_extras
    upd, train, subs = None, None, []
    D, A, E, H = w["obs_dim"], w["act_dim"], w["ws"][0].shape[0], w["ws"][1].shape[1]
    if extras:
        # Metric B: CPO trust-region update (update_policy, algorithms/cmbpo.py:357) on the rollout's samples
        _, res = rollout_phase(sampler, pool, start)
        n_full = int(res[0].shape[0])
        n50 = min(50000, n_full)
        from cmbpo_amd import _lib as _l
        # (a) the constrained update the north star names: 22 Hessian-vector products (cg(Hx, g), Hx(v), cg(Hx, b), Hx(w),
        #     policies/cpo_policy.py:210-224) on SURVEY M1's batch; cost_lim just above the batch's mean episode cost
        #     (c < 0: optim case 2 or 3); (b) the same update on the rollout's own samples, whose costs are all zero:
        #     the reference's TRPO short cut (:216-219), 11 products
        cbuf = constrained_batch(res)
        c_lim = float(cbuf[9].mean()) * policy.max_path_length + 0.5
        upd = {"unit": "ms", "n_50k": n50, "n_full": n_full, "per_rank": True,
               "matrix_path": "3 x v_mfma_f32_32x32x16_f16 per f32 product" if _l.lib().cmbpo_get_pi_matrix_path() else "fp32 MFMA"}
        for tag, b_, lim_, want in (("constrained", cbuf, c_lim, 22), ("unconstrained", res, None, 11)):
            ms50, info50 = time_update(policy, b_, n50, cost_lim=lim_)
            msfull, infofull = time_update(policy, b_, n_full, reps=2, cost_lim=lim_)
            ent = {"ms_50k": ms50, "ms_full": msfull, "hvps": int(infofull["n_fvp"]), "hvps_50k": int(info50["n_fvp"]),
                   "optim_case": int(infofull["OptimCase"]), "optim_case_50k": int(info50["OptimCase"]),
                   "backtrack_iters": int(infofull["BacktrackIters"]), "accepted": bool(infofull["accepted"]),
                   "fvp_ms_full": time_fvp(policy)}
            if tag == "constrained":
                ent["cost_lim"] = lim_
                ent["batch"] = "rollout (obs, act, logp, mu, log_std); adv, cadv ~ N(0,1); cost ~ Bernoulli(0.05) (SURVEY 8d M1)"
                if ent["hvps"] != want or ent["hvps_50k"] != want:
                    raise RuntimeError(f"bench.py: the constrained update ran {ent['hvps']} / {ent['hvps_50k']} Hessian-vector "
                                       f"products, not {want} (optim case {ent['optim_case']})")
            upd[tag] = ent
        # Metric B = the constrained update
        upd.update(ms_50k=upd["constrained"]["ms_50k"], ms_full=upd["constrained"]["ms_full"], hvps=upd["constrained"]["hvps"],
                   optim_case=upd["constrained"]["optim_case"], fvp_ms_full=upd["constrained"]["fvp_ms_full"])
        cbuf_host = [x[:n50].cpu() for x in cbuf]     # for the CPU update baseline (the same constrained batch)
        del cbuf
        # Metric C: ensemble training steps (dynamics model on (obs, act) -> (d_obs, rew); critic on obs -> ret)
        obs_t, act_t, ret_t = res[0], res[1], res[4]
        n_tr = min(n_full, 200000)
        x_dyn = torch.cat([obs_t[:n_tr], act_t[:n_tr]], dim=1).contiguous()
        t_dyn = torch.cat([0.01 * torch.randn_like(obs_t[:n_tr]), ret_t[:n_tr, None]], dim=1).contiguous()
        model_us = time_train_steps(env._model, x_dyn, t_dyn, 2048, 100)
        critic_us = time_train_steps(policy.v, obs_t[:n_tr].contiguous(), ret_t[:n_tr, None].contiguous(), 2048, 300)
        fl_model = 6.0 * E * 2048 * ((D + A) * H + H * H + H * 2 * (D + 1))
        train = {"unit": "us/step", "batch": 2048, "model_step_us": model_us, "model_tflops": fl_model / model_us / 1e6,
                 "critic_step_us": critic_us, "model": f"E={E} {D + A}->{H}->{H}->{2 * (D + 1)} MSPE + Adam",
                 "critic": "E=3 obs->128->128->1 MSE + Adam", "per_rank": True}
        del res, obs_t, act_t, ret_t, x_dyn, t_dyn

    roof = ens_roofline(w, events)
    # HBM traffic of the dominant kernel from PMC counters collected OFFLINE (separate rocprofv3 --pmc passes, see the
    # json file), scaled to this run's rows per launch -- not a measurement of this run
    pmc_file = {"ens_h3": ("r03", "pmc_traffic_h3.json"), "ens_split": ("r01", "pmc_traffic_split.json"),
                "ens_mlp": ("r01", "pmc_traffic.json")}[roof["kernel"].split("_kernel")[0]] if roof else None
    if roof is not None:
        roof["traffic"] = None
        roof["traffic_offline_pmc"] = None
        try:
            with open(os.path.join(ROOT, "profiles", pmc_file[0], pmc_file[1])) as f:
                pmc = json.load(f)
            if task == "AntSafe-v2":
                roof["traffic_offline_pmc"] = pmc["hbm_bytes_per_launch"] * roof["avg_rows_per_launch"] / pmc["rows_per_launch"]
                roof["traffic"] = roof["traffic_offline_pmc"]
                roof["traffic_source"] = f"profiles/{pmc_file[0]}/{pmc_file[1]} (offline rocprofv3 --pmc passes, scaled by rows)"
        except Exception:
            pass

    if extras and world == 1 and rank == 0:
        # what ships: the rollout phase at the shipped configurations' shapes and modes (configs/cmbpo_hcs.py:17-31,
        # configs/cmbpo_hs.py:5,28, configs/cmbpo_antsafe.py:33; budget rule samplers/model_sampler.py:275-287)
        del sampler, pool, env
        torch.cuda.empty_cache()
        for spec in (("hcs_b10k", "HalfCheetahSafe-v2", 10000, 35, "schedule"),
                     ("humanoid_b10k_h14", "HumanoidSafe-v2", 10000, 15, "schedule"),
                     ("hopper_b10k_h14", "HopperSafe-v2", 10000, 15, "schedule"),
                     ("antsafe_b1000", "AntSafe-v2", 1000, 35, "schedule"),
                     ("antsafe_b1000_uncertainty_dies_of_dkl", "AntSafe-v2", 1000, 35, "uncertainty"),
                     ("antsafe_b100k_uncertainty_dies_of_dkl", "AntSafe-v2", 100000, 35, "uncertainty")):
            subs.append(run_config(*spec, device))
        # the shipped stop rule: 'uncertainty' rollouts that end because the sample budget binds (a DKL limit most
        # branches stay under for the phase, max_samples = BUDGET_FRAC * B * T)
        for spec in (("antsafe_b1000_uncertainty_budget_binds", "AntSafe-v2", 1000, 35, "uncertainty"),
                     ("antsafe_b100k_uncertainty_budget_binds", "AntSafe-v2", 100000, 35, "uncertainty")):
            subs.append(run_config(*spec, device, dkl_scale=BIND_DKL_SCALE, budget_frac=BIND_BUDGET_FRAC, budget_binds=True))

    if rank == 0:
        arith = {
            "ens_h3": ("float32 inputs, outputs and accumulation; each float32 product of the ensemble forward runs as three exact "
                       "f16 partial products (operands lifted by a power of two and split exactly into 2 f16 pieces, 11 + 1 + 11 "
                       "significant bits; measured error 3.2e-7 of sum|a_k b_k| at K = 512 against 7.6e-7 for the fp32 MFMA chain, "
                       "tools/split_f16_probe.hip); cmbpo_set_ens_matrix_path(1) selects six bf16 terms, (0) fp32 MFMAs"),
            "ens_split": ("float32 inputs, outputs and accumulation; the ensemble forward's float32 products run as six exact bf16 "
                          "partial products each (operands split exactly into 3 bf16 pieces)"),
            "ens_mlp": "float32 throughout (fp32 MFMAs)"}[roof["kernel"].split("_kernel")[0]] if roof else None
        out = {
            "metric": "imagined env-steps/sec (ensemble rollout) + CPO update ms",
            "value": tot / dt_max,
            "unit": "imagined env-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt_max / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"{task} imagined rollout: E={E} x ({D + A}->{H}->{H}->{2 * (D + 1)}) swish ensemble, 5 elites, "
                                   f"3+3 critics 128x128, tanh policy 128x128, "
                                   + (f"B={args.branches} branches sharded over {world} GPU(s)" if args.scaling == "strong"
                                      else f"B={B} branches/GPU")
                                   + f", maxroll {maxroll} ({maxroll - 1} steps), "
                                   + ("fixed-horizon mode" if args.rollout_mode == "schedule" else
                                      f"'uncertainty' mode (DKL limit {args.dkl_scale} x the 5-step calibration)")
                                   + (f", max_samples = {max_samples} (budget rule)" if max_samples else "")
                                   + ", reset->sample*->finish_all_paths->get()",
                       "rollout_mode": args.rollout_mode, "max_samples": max_samples, "dkl_lim": dkl_lim,
                       "n_budget_terminated_per_phase": n_budget / args.steps, "sampler_steps_per_phase_rank0": n_steps / args.steps,
                       "branches_per_gpu": B, "branches_total": args.branches if args.scaling == "strong" else B * world,
                       "horizon": maxroll - 1, "task": task, "samples_per_step": tot / args.steps, "arithmetic": arith},
            "roofline": roof,
            "gae_get": get_roofline(w, get_events),
        }
        if upd is not None:
            out["cpo_update"] = upd
            out["ensemble_train"] = train
        if subs:
            out["configs"] = subs
        if extras and not args.no_cpu_baseline and world == 1:     # the CPU baseline is timed at N = 1 only
            out["cpu_baseline"] = cpu_baseline(w, task, seconds=args.cpu_seconds)
            out["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
            out["cpu_baseline"]["cpo_update_ms_50k"] = cpu_update_baseline(w, cbuf_host, n50, c_lim, maxroll)
            out["cpu_baseline"]["model_train_step_us"] = cpu_train_baseline(E, D + A, H, 2 * (D + 1), "MSPE", 2048)
            out["cpu_baseline"]["critic_train_step_us"] = cpu_train_baseline(3, D, 128, 1, "MSE", 2048)
        print(json.dumps(out), flush=True)
    comm.barrier()


if __name__ == "__main__":
    main()
