"""Generate golden vectors by running the REFERENCE's own NumPy code in the build container.

Usage (build container only; /root/reference does not exist on the GPU box):
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Writes tests/golden/*.npz (data only: inputs and the reference's outputs).  The TF-graph half of the
reference cannot run (no TensorFlow), so the model / policy objects handed to the reference's
FakeEnv / ModelSampler are duck-typed stand-ins computing the restated forward passes of
oracle/refcpu.py; everything downstream of them (FakeEnv.step, average_dkl, statics, ModelSampler,
ModelBuffer, discount_cumsum, mpi_statistics_scalar, cg, CPOAgent.update_pi, CPOBuffer) is the
reference's code, imported from /root/reference behind inert stubs for the modules it imports at the
top of its files but never touches on these paths (tensorflow, gym, ray, mpi4py, ...).
"""
import importlib.abc
import importlib.machinery
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("CMBPO_REFERENCE", "/root/reference")
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
from worlds import build_world, make_update_batch  # noqa: E402,F401

sys.path.insert(0, REF)      # the reference tree: only this generator imports from it

if not hasattr(np, "int"):
    np.int = int       # buffers/cpobuffer.py:135, utilities/utils.py:300
if not hasattr(np, "bool"):
    np.bool = bool


class _Inert(types.ModuleType):
    """Module-like object: any attribute is another inert object, callable, iterable, subclassable."""

    def __init__(self, name="stub"):
        super().__init__(name)
        self.__path__ = []

    def __getattr__(self, item):
        if item.startswith("__") and item.endswith("__"):
            raise AttributeError(item)
        child = _InertClass
        return child

    def __call__(self, *a, **k):
        return _Inert()


class _InertMeta(type):
    def __getattr__(cls, item):
        if item.startswith("__") and item.endswith("__"):
            raise AttributeError(item)
        return _InertClass

    def __iter__(cls):
        return iter(())

    def __contains__(cls, item):
        return False

    def __getitem__(cls, item):
        return _InertClass


class _InertClass(metaclass=_InertMeta):
    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _InertClass()

    def __getattr__(self, item):
        if item.startswith("__") and item.endswith("__"):
            raise AttributeError(item)
        return _InertClass()

    def __iter__(self):
        return iter(())


_STUB_ROOTS = {"tensorflow", "gtimer", "dotmap", "gym", "ray", "mujoco_py", "wrappers", "cv2", "safety_gym",
               "joblib", "psutil_stub"}


class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        if fullname.split(".")[0] in _STUB_ROOTS:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        return _Inert(spec.name)

    def exec_module(self, module):
        pass


def install_stubs():
    sys.meta_path.insert(0, _StubFinder())
    mpi4py = types.ModuleType("mpi4py")

    class _Comm:
        def Get_rank(self):
            return 0

        def Get_size(self):
            return 1

        def Allreduce(self, x, buf, op=None):
            buf[...] = x

        def Bcast(self, x, root=0):
            pass

    class _MPI:
        COMM_WORLD = _Comm()
        SUM, MIN, MAX = "sum", "min", "max"

    mpi4py.MPI = _MPI
    sys.modules["mpi4py"] = mpi4py


# ------------------------------------------------------------------------------------------------
class _Space:
    def __init__(self, d):
        self.shape = (d,)


class _TrueEnv:
    def __init__(self, obs_dim, act_dim):
        self.observation_space, self.action_space = _Space(obs_dim), _Space(act_dim)


class OracleModel:
    """Duck type of EnsembleModel (models/base_model.py) computing the restated forward."""

    def __init__(self, ws, bs, sc_in, sc_out, elites):
        self.ws, self.bs, self.sc_in, self.sc_out = ws, bs, sc_in, sc_out
        self.is_ensemble, self.is_probabilistic = True, True
        self.in_dim, self.out_dim = ws[0].shape[1], ws[2].shape[2] // 2
        self.elite_inds = elites

    def predict_ensemble(self, x):
        from oracle import refcpu
        return refcpu.ens_forward(x, self.ws, self.bs, self.sc_in, self.sc_out)


class OraclePolicy:
    """Duck type of CPOPolicy for the sampler (samplers/model_sampler.py:249-256,401-407)."""

    class agent:
        reward_penalized = False

    def __init__(self, params, v, vc, rng):
        self.params, self.v, self.vc, self.rng = params, v, vc, rng
        self.eps_log = []

    def reset(self):
        pass

    def get_v(self, obs):
        from oracle import refcpu
        return refcpu.ens_predict_mean(obs, *self.v)[:, 0]

    def get_vc(self, obs):
        from oracle import refcpu
        return refcpu.ens_predict_mean(obs, *self.vc)[:, 0]

    def get_action_outs(self, obs):
        from oracle import refcpu
        eps = self.rng.standard_normal((obs.shape[0], self.params[-1].shape[0])).astype(np.float32)
        self.eps_log.append(eps)
        out = refcpu.policy_forward(obs, self.params, eps)
        return dict(pi=out["pi"], logp_pi=out["logp_pi"], pi_info=dict(mu=out["mu"], log_std=out["log_std"]),
                    v=self.get_v(obs), vc=self.get_vc(obs))


def gen_statics(out):
    from models import statics
    rng = np.random.default_rng(101)
    n = 400
    data = {}
    for task, (od, ad) in (("AntSafe-v2", (29, 8)), ("HalfCheetahSafe-v2", (20, 6))):
        obs = rng.standard_normal((n, od)).astype(np.float32)
        act = rng.standard_normal((n, ad)).astype(np.float32)
        nxt = (rng.standard_normal((n, od)) * 1.5).astype(np.float32)
        nxt[:, 0] = rng.uniform(-0.2, 1.4, n).astype(np.float32)
        # edges: z boundaries, z_rot at the threshold, non-finite entries, cost thresholds
        nxt[0, 0], nxt[1, 0], nxt[2, 0], nxt[3, 0] = 0.2, 1.0, np.float32(0.2) - 1e-7, np.float32(1.0) + 1e-6
        nxt[4, 0] = 0.5; nxt[4, 2] = np.sqrt(0.85); nxt[4, 3] = 0.0       # z_rot = -0.7 (approximately)
        nxt[5, 0] = 0.5; nxt[5, 2] = 2.0
        nxt[6, 5] = np.nan; nxt[7, 2] = np.inf; nxt[8, 0] = np.nan; nxt[9, 3] = -np.inf
        nxt[10, -1], nxt[11, -1], nxt[12, -1] = 3.2, np.float32(3.2) + 1e-6, -4.0
        nxt[13, -1], nxt[14, -1] = 0.2, np.float32(0.2) - 1e-8
        with np.errstate(all="ignore"):
            term = statics.TERMS_BY_TASK[task](obs, act, nxt)
            cost = statics.COST_BY_TASK[task](obs, act, nxt)
        key = task.split("-")[0]
        data.update({f"{key}_obs": obs, f"{key}_act": act, f"{key}_next": nxt,
                     f"{key}_term": term, f"{key}_cost": np.asarray(cost)})
    nd = statics.no_done(obs, act, nxt)
    data["no_done"] = nd
    np.savez_compressed(os.path.join(out, "g1_statics.npz"), **data)


def gen_dkl(out):
    from models.pens.utils import average_dkl, gaussian_kl_np
    rng = np.random.default_rng(102)
    mu = rng.standard_normal((7, 300, 29)).astype(np.float32)
    std = np.exp(rng.uniform(-8, 1, (7, 300, 29))).astype(np.float32)
    std[:, 0, 0] = 0.0
    std[2, 1, 1] = 1e-30
    with np.errstate(all="ignore"):
        d = average_dkl(mu, std)
        k = gaussian_kl_np(mu[0], np.log(std[0] + 1e-3), mu[1], np.log(std[1] + 1e-3))
    np.savez_compressed(os.path.join(out, "g2_dkl.npz"), mu=mu, std=std, average_dkl=d, pair_kl=k)


def gen_gae_and_stats(out):
    from utilities.utils import discount_cumsum
    from utilities.mpi_tools import mpi_statistics_scalar
    from utilities.trust_region import cg
    rng = np.random.default_rng(103)
    x32 = rng.standard_normal((50, 34)).astype(np.float32)
    y = discount_cumsum(x32, 0.99, 0.95, axis=-1)
    yc = discount_cumsum(x32, 0.97, 0.5, axis=-1)
    x1 = rng.standard_normal(1000).astype(np.float32)
    y1 = discount_cumsum(x1, 0.99, 0.95, axis=-1)
    s = rng.standard_normal(5000).astype(np.float32) * 3 + 1
    mean, std = mpi_statistics_scalar(s)
    # cg on SPD operators, fixed 10 iterations (trust_region.py:32-45)
    n = 40
    M = rng.standard_normal((n, n)).astype(np.float32)
    Aop = (M @ M.T / n + 0.1 * np.eye(n)).astype(np.float32)
    b = rng.standard_normal(n).astype(np.float32)
    x = cg(lambda p: (Aop @ p).astype(np.float32), b)
    np.savez_compressed(os.path.join(out, "g3_gae_stats_cg.npz"), x32=x32, gae_r=y, gae_c=yc, x1=x1, gae_1d=y1,
                        stat_in=s, stat_mean=mean, stat_std=std, cg_A=Aop, cg_b=b, cg_x=x)


def run_sampler_trace(seed, task, B, T, hidden, dkl_lim, budget, mode, out_scale=1.0, q_boost=0.0, max_steps=100):
    from buffers.modelbuffer import ModelBuffer
    from models.fake_env import FakeEnv
    from samplers.model_sampler import ModelSampler
    from cmbpo_amd import synthetic
    w = build_world(seed, task, hidden, out_scale=out_scale, q_boost=q_boost)
    rng = np.random.default_rng(seed + 1)
    model = OracleModel(w["ws"], w["bs"], w["sc_in"], w["sc_out"], w["elites"])
    policy = OraclePolicy(w["pol"], w["v"], w["vc"], rng)
    env = FakeEnv(_TrueEnv(w["obs_dim"], w["act_dim"]), task, model, True, True, False)
    inds_log = []
    orig = env.random_inds

    def rec(size):
        r = orig(size)
        inds_log.append(np.asarray(r, dtype=np.int32))
        return r

    env.random_inds = rec
    buf = ModelBuffer(batch_size=B, obs_dim=w["obs_dim"], act_dim=w["act_dim"], max_path_length=T)
    buf.initialize({"mu": [w["act_dim"]], "log_std": [w["act_dim"]]}, gamma=0.99, lam=0.95,
                   cost_gamma=0.97, cost_lam=0.5)
    sampler = ModelSampler(max_path_length=T, batch_size=B, rollout_mode=mode, logger=object())
    sampler.initialize(env, policy, buf)
    sampler.set_rollout_dkl(dkl_lim)
    start = synthetic.start_states(rng, B, task)
    np.random.seed(seed)
    sampler.reset(start)
    alive_log, tot_log, ratio_log = [], [], []
    with np.errstate(all="ignore"):
        for _ in range(max_steps):
            _, _, _, info = sampler.sample(max_samples=budget)
            alive_log.append(buf.alive_paths.copy())
            tot_log.append(sampler._total_samples)
            ratio_log.append(info["alive_ratio"])
            if budget and sampler._total_samples >= .99 * budget:
                break
            if info["alive_ratio"] <= 0.1:
                break
        dkl_acc = np.array(sampler._dyn_dkl_path, dtype=np.float64)
        diag = sampler.finish_all_paths()
        res, bdiag = buf.get()
    nsteps = len(alive_log)
    eps_pad = np.zeros((nsteps, B, w["act_dim"]), np.float32)
    inds_pad = np.zeros((nsteps, B), np.int32)
    n_rows = np.zeros(nsteps, np.int32)
    for s in range(nsteps):
        k = policy.eps_log[s].shape[0]
        n_rows[s] = k
        eps_pad[s, :k] = policy.eps_log[s]
        inds_pad[s, :k] = inds_log[s]
    names = ["obs", "act", "adv", "cadv", "ret", "cret", "logp", "val", "cval", "cost", "log_std", "mu"]
    data = dict(seed=seed, task=task, B=B, T=T, hidden=hidden, dkl_lim=dkl_lim, budget=budget or 0, mode=mode,
                out_scale=out_scale, q_boost=q_boost, start=start, eps=eps_pad, inds=inds_pad, n_rows=n_rows,
                alive=np.array(alive_log), total_samples=np.array(tot_log, dtype=np.float64),
                alive_ratio=np.array(ratio_log, dtype=np.float64),
                dkl_acc=dkl_acc, poolm_batch_size=bdiag["poolm_batch_size"], poolm_ret_mean=bdiag["poolm_ret_mean"],
                poolm_cret_mean=bdiag["poolm_cret_mean"])
    for k, v in zip(names, res):
        data["get_" + k] = v
    for k, v in diag.items():
        data["diag_" + k.replace("/", "__")] = np.float64(v)
    return data


def gen_sampler_traces(out):
    traces = {
        # uncertainty-limited, AntSafe terminations, budget hit mid-rollout
        "g5_trace_ant_unc": dict(seed=7, task="AntSafe-v2", B=96, T=12, hidden=128, dkl_lim=None, budget=330,
                                 mode="uncertainty", out_scale=1.0, q_boost=1.2),
        # env terminations (AntSafe rule) spread over several steps, then the horizon finish
        "g5_trace_ant_term": dict(seed=7, task="AntSafe-v2", B=96, T=8, hidden=128, dkl_lim=float("inf"),
                                  budget=None, mode="uncertainty", out_scale=1.0, q_boost=1.2),
        # fixed horizon ('schedule' style), no budget, HalfCheetah cost rule
        "g5_trace_hcs_sched": dict(seed=8, task="HalfCheetahSafe-v2", B=64, T=9, hidden=128, dkl_lim=float("inf"),
                                   budget=None, mode="schedule"),
        # no statics entry (Hopper): never terminates, bool zero cost; tight budget
        "g5_trace_hopper_budget": dict(seed=9, task="HopperSafe-v2", B=50, T=15, hidden=128, dkl_lim=float("inf"),
                                       budget=333, mode="uncertainty"),
        # widest shapes (obs 47 / act 17: 64 inputs, 96 raw outputs) at the production width, uncertainty deaths
        "g5_trace_humanoid_512": dict(seed=11, task="HumanoidSafe-v2", B=70, T=7, hidden=512, dkl_lim=None, budget=None,
                                      mode="uncertainty"),
    }
    only = os.environ.get("CMBPO_GOLDEN_TRACES")
    if only:
        traces = {k: v for k, v in traces.items() if k in only.split(",")}
    for name, cfg in traces.items():
        if cfg["dkl_lim"] is None:
            # calibrate a limit that kills a fraction of the branches over the rollout
            # limit = median accumulated DKL after 4 steps: about half the branches die at step 4,
            # the rest one step later (partial uncertainty deaths + terminations + budget interplay)
            probe = run_sampler_trace(**{**cfg, "dkl_lim": float("inf"), "budget": None, "max_steps": 4})
            acc = np.sort(probe["dkl_acc"])
            lo, hi = int(.35 * len(acc)), int(.65 * len(acc))
            k = lo + int(np.argmax(acc[lo + 1:hi + 1] - acc[lo:hi]))      # widest gap near the median:
            cfg["dkl_lim"] = float(0.5 * (acc[k] + acc[k + 1]))           # no branch sits on the limit
        data = run_sampler_trace(**cfg)
        np.savez_compressed(os.path.join(out, name + ".npz"), **data)
        print(name, "steps", len(data["n_rows"]), "rows/step", data["n_rows"].tolist(),
              "samples", int(data["poolm_batch_size"]))


def main():
    install_stubs()
    gen_statics(HERE)
    gen_dkl(HERE)
    gen_gae_and_stats(HERE)
    gen_sampler_traces(HERE)
    gen_update_pi(HERE)
    gen_cpobuffer(HERE)
    gen_pe_train(HERE)
    gen_loop_helpers(HERE)
    gen_cpobuffer_archive(HERE)
    gen_cpo_sampler(HERE)
    print("golden vectors written to", HERE)




# ------------------------------------------------------------------------------------------------
# G7: the reference's CPOAgent.update_pi driven by a fake session evaluating the restated graph
# ------------------------------------------------------------------------------------------------
def gen_update_pi(out):
    from policies.cpo_policy import CPOAgent
    from oracle import refupdate
    obs_dim, act_dim, hidden, n, T = 6, 3, 16, 256, 20
    K = {k: object() for k in ("flat_g", "flat_b", "v_ph", "hvp", "get", "set", "pi_loss", "surr_cost", "d_kl",
                               "cur_cret_avg")}

    class Logger:
        def __init__(self):
            self.stored = {}

        def log(self, *a, **k):
            pass

        def store(self, **kw):
            self.stored.update(kw)

    class FakeSession:
        def __init__(self, graph, params):
            self.g, self.p = graph, np.asarray(params, np.float32).copy()
            self.calls = dict(hvp=0, evals=0)

        def run(self, fetches, feed_dict=None):
            if fetches is K["hvp"]:
                self.calls["hvp"] += 1
                return self.g.hvp(self.p, feed_dict[K["v_ph"]], 0.1)
            if fetches is K["get"]:
                return self.p.copy()
            if fetches is K["set"]:
                self.p = np.asarray(feed_dict[K["v_ph"]], np.float32).copy()
                return None
            if isinstance(fetches, list) and len(fetches) == 5:
                g, b, lo, sc = self.g.grads(self.p)
                return [g, b, np.float32(lo), np.float32(sc), np.float32(float(self.g.cur_cret_avg()))]
            if isinstance(fetches, list) and len(fetches) == 3:
                self.calls["evals"] += 1
                return [np.float32(t) for t in self.g.evals(self.p)]
            raise KeyError(fetches)

    scenarios = [   # (name, cost_p, cadv_scale, cost_lim, constrained, real_cost, seed)
        ("feasible", 0.05, 1.0, 10.0, True, 3.0, 700),
        ("violating", 0.9, 1.0, 10.0, True, 25.0, 701),
        ("violating_hard", 1.0, 0.02, 2.0, True, 30.0, 702),
        ("zero_costgrad", 0.0, 0.0, 10.0, True, 1.0, 703),
        ("unconstrained_ls_fail", 0.5, 1.0, 10.0, False, 10.0, 704),
        ("feasible_tight", 0.45, 3.0, 10.0, True, 9.0, 705),
        ("slightly_violating", 0.56, 10.0, 10.0, True, 10.5, 706),
        ("unconstrained", 0.5, 1.0, 10.0, False, 10.0, 707),
    ]
    data = dict(obs_dim=obs_dim, act_dim=act_dim, hidden=hidden, T=T, names=np.array([s[0] for s in scenarios]))
    for si, (name, cost_p, cadv_scale, cost_lim, constrained, real_cost, seed) in enumerate(scenarios):
        rng = np.random.default_rng(seed)
        params, batch = make_update_batch(rng, n, obs_dim, act_dim, hidden, cost_p, cadv_scale, T)
        graph = refupdate.PolicyGraph(obs_dim, act_dim, batch, max_path_length=T, hidden=hidden)
        sess = FakeSession(graph, params)
        agent = CPOAgent(constrained=constrained, reward_penalized=False, objective_penalized=False,
                         learn_penalty=False, penalty_param_loss=False, learn_margin=True, c_gamma=0.97,
                         max_path_length=T)
        logger = Logger()
        agent.set_logger(logger)
        agent.prepare_session(sess)
        real_buf = [real_cost] * 300
        agent.prepare_update(dict(flat_g=K["flat_g"], flat_b=K["flat_b"], v_ph=K["v_ph"], hvp=K["hvp"],
                                  get_pi_params=K["get"], set_pi_params=K["set"], pi_loss=K["pi_loss"],
                                  surr_cost=K["surr_cost"], d_kl=K["d_kl"], target_kl=0.01, cost_lim=cost_lim,
                                  cur_cret_avg=K["cur_cret_avg"], real_cost_buf=real_buf))
        agent.margin = 0.002 * si
        with np.errstate(all="ignore"):
            agent.update_pi({})
        st = logger.stored
        pre = f"s{si}_"
        data[pre + "params"] = params
        for k, v in batch.items():
            data[pre + "b_" + k] = v
        data[pre + "cost_lim"], data[pre + "constrained"], data[pre + "real_cost"] = cost_lim, constrained, real_cost
        data[pre + "margin_in"] = 0.002 * si
        data[pre + "new_params"] = sess.p
        data[pre + "hvp_calls"], data[pre + "eval_calls"] = sess.calls["hvp"], sess.calls["evals"]
        for k in ("Optim_A", "Optim_B", "Optim_c", "Optim_q", "Optim_r", "Optim_s", "Optim_Lam", "Optim_Nu",
                  "Margin", "OptimCase", "BacktrackIters"):
            data[pre + k] = np.float64(st[k])
        print("update_pi", name, "case", int(st["OptimCase"]), "backtrack", int(st["BacktrackIters"]),
              "hvp calls", sess.calls["hvp"], "evals", sess.calls["evals"],
              "moved", bool(np.any(sess.p != params)))
    np.savez_compressed(os.path.join(out, "g7_update_pi.npz"), **data)


def gen_cpobuffer(out):
    """G8: the reference CPOBuffer on three real-env style paths (float32 and float64-zero bootstraps)."""
    from buffers.cpobuffer import CPOBuffer
    rng = np.random.default_rng(808)
    D, A = 6, 2
    buf = CPOBuffer(size=64, archive_size=256, observation_space=_Space(D), action_space=_Space(A))
    buf.initialize({"mu": [A], "log_std": [A]}, gamma=0.99, lam=0.95, cost_gamma=0.97, cost_lam=0.5)
    lengths = [7, 1, 12, 20]
    boots = [("f32", "f32"), ("zero64", "f32"), ("f32", "f32"), ("zero64", "f32")]
    n = sum(lengths)
    data = dict(D=D, A=A, lengths=np.array(lengths), obs=rng.standard_normal((n, D)).astype(np.float32),
                act=rng.standard_normal((n, A)).astype(np.float32),
                rew=rng.standard_normal(n).astype(np.float32), val=rng.standard_normal(n).astype(np.float32),
                cost=(rng.random(n) < 0.3).astype(np.float32), cval=rng.standard_normal(n).astype(np.float32),
                logp=rng.standard_normal(n).astype(np.float32), mu=rng.standard_normal((n, A)).astype(np.float32),
                log_std=np.full((n, A), -0.5, np.float32),
                last_val=rng.standard_normal(len(lengths)).astype(np.float32),
                last_cval=rng.standard_normal(len(lengths)).astype(np.float32),
                zero_val=np.array([b[0] == "zero64" for b in boots]))
    i = 0
    for p, L in enumerate(lengths):
        for _ in range(L):
            buf.store(data["obs"][i], data["act"][i], data["obs"][i], data["rew"][i], data["val"][i], data["cost"][i],
                      data["cval"][i], data["logp"][i], {"mu": data["mu"][i], "log_std": data["log_std"][i]}, False, 3)
            i += 1
        lv = np.zeros((1,)) if data["zero_val"][p] else data["last_val"][p:p + 1]
        buf.finish_path(lv, data["last_cval"][p:p + 1])
    res, diag = buf.get()
    names = ["obs", "act", "adv", "cadv", "ret", "cret", "logp", "val", "cval", "cost", "log_std", "mu"]
    for k, v in zip(names, res):
        data["get_" + k] = v
    data["poolr_ret_mean"], data["poolr_cret_mean"] = diag["poolr_ret_mean"], diag["poolr_cret_mean"]
    data["arch_size"] = buf.arch_size
    np.savez_compressed(os.path.join(out, "g8_cpobuffer.npz"), **data)
    print("cpobuffer: samples", n, "archive", buf.arch_size)


# ------------------------------------------------------------------------------------------------
# G9: the reference's PE.train control flow (and TensorStandardScaler.fit) driven by a fake session
# ------------------------------------------------------------------------------------------------
def gen_pe_train(out):
    """PE.train (models/pens/pe.py:457-646), _save_best / _end_train (:367-400) and TensorStandardScaler.fit
    (models/pens/utils.py:119-138,220-231) are the reference's code; the session is a stand-in that records
    which rows every train_op is fed (column 0 of the inputs carries the row id) and answers `self.loss` with a
    scripted sequence of holdout losses, so the NumPy half of training is pinned without TensorFlow."""
    import contextlib
    from models.pens.pe import PE
    from models.pens.utils import TensorStandardScaler

    class Var:
        def __init__(self, v):
            self.v = np.asarray(v, np.float32)

        def load(self, v):
            self.v = np.asarray(v, np.float32)      # the TF variables are float32

        def eval(self):
            return self.v

    def make_scaler(dim):
        sc = object.__new__(TensorStandardScaler)
        sc.fitted = False
        sc.count, sc.mu, sc.var = Var(0.0), Var(np.zeros([1, dim])), Var(np.ones([1, dim]))
        sc.cached_count, sc.cached_mu, sc.cached_var = 0, np.zeros([1, dim]), np.ones([1, dim])
        return sc

    class Sess:
        def __init__(self, script):
            self.script, self.k = script, 0
            self.steps, self.holdout_rows = [], None

        @contextlib.contextmanager
        def as_default(self):
            yield self

        def run(self, op, feed_dict=None):
            x = feed_dict["in"]
            if op == "train_op":
                self.steps.append(x[..., 0].astype(np.int32))
                return None
            if op == "loss":
                rows = x[0, :, 0].astype(np.int32)
                if self.holdout_rows is not None and rows.shape == self.holdout_rows.shape and \
                        np.array_equal(rows, self.holdout_rows):
                    losses = self.script(self.k)
                    self.k += 1
                    return losses
                return np.zeros(x.shape[0])            # the progress-bar evaluation on training rows
            return [np.zeros(1), np.zeros(1)]          # [tensor_loss, debug_mean]: display only

    cases = {}
    scen = [  # name, n, E, elites, batch, kwargs, script
        ("early_stop", 230, 5, 3, 32, dict(max_epochs=40, holdout_ratio=0.2, max_epochs_since_update=3, min_epoch_before_break=4),
         lambda k: np.array([1.0, 2.0, 0.5, 3.0, 1.5]) * (0.8 ** min(k, 6)) + 0.01 * np.array([3, 1, 4, 1, 5]) * (k % 3)),
        ("max_epochs", 100, 3, 2, 64, dict(max_epochs=5, holdout_ratio=0.1, min_epoch_before_break=5),
         lambda k: np.array([0.3, 0.1, 0.2]) / (1.0 + k)),
        ("grad_updates", 157, 4, 2, 16, dict(max_epochs=50, holdout_ratio=0.25, max_grad_updates=30),
         lambda k: np.array([4.0, 3.0, 2.0, 1.0]) * (0.99 ** k)),
        ("max_logging", 400, 2, 1, 128, dict(max_epochs=2, holdout_ratio=0.5, max_logging=50),
         lambda k: np.array([1.0, 1.0 - 0.2 * k])),
    ]
    for name, n, E, n_el, bs, kw, script in scen:
        seed = int(len(name) * 17 + n)
        rs = np.random.RandomState(seed + 1)
        in_dim, out_dim = 4, 2
        inputs = rs.standard_normal((n, in_dim)).astype(np.float32)
        inputs[:, 0] = np.arange(n)
        targets = rs.standard_normal((n, out_dim)).astype(np.float32)
        fake = types.SimpleNamespace()
        fake.num_nets, fake.num_elites, fake.name = E, n_el, "G9"
        fake.clip_loss, fake.loss_type, fake.weights = False, "MSPE", None
        fake.use_scaler_in = fake.use_scaler_out = True
        fake.scaler_in, fake.scaler_out = make_scaler(in_dim), make_scaler(out_dim)
        fake.sy_train_in, fake.sy_train_targ = "in", "targ"
        fake.train_op, fake.loss, fake.tensor_loss, fake.debug_mean = "train_op", "loss", "tensor_loss", "debug_mean"
        sess = Sess(script)
        fake.sess = sess
        fake.layers = []
        for meth in ("_start_train", "_save_best", "_save_state", "_end_train"):
            setattr(fake, meth, types.MethodType(getattr(PE, meth), fake))
        # the holdout rows are what the permutation draws first: replay the generator to know them
        probe = np.random.RandomState(seed)
        perm = probe.permutation(n)
        num_holdout = min(int(n * kw.get("holdout_ratio", 0.0)), kw.get("max_logging", 5000))
        sess.holdout_rows = perm[:num_holdout].astype(np.int32)
        np.random.seed(seed)
        metrics = PE.train(fake, inputs, targets, batch_size=bs, **kw)
        steps = sess.steps
        widths = np.array([s.shape[1] for s in steps], np.int32)
        flat = np.concatenate([s.reshape(-1) for s in steps])
        pre = name + "/"
        cases[pre + "seed"], cases[pre + "n"], cases[pre + "E"] = seed, n, E
        cases[pre + "num_elites"], cases[pre + "batch_size"] = n_el, bs
        for k, v in kw.items():
            cases[pre + "kw_" + k] = v
        cases[pre + "inputs"], cases[pre + "targets"] = inputs, targets
        cases[pre + "step_widths"], cases[pre + "step_rows"] = widths, flat
        cases[pre + "holdout_rows"] = sess.holdout_rows
        cases[pre + "script"] = np.stack([script(k) for k in range(sess.k)])
        cases[pre + "elites"] = np.asarray(fake._model_inds, np.int32)
        cases[pre + "val_loss"] = float(list(metrics.values())[0])
        cases[pre + "in_mu"], cases[pre + "in_var"] = fake.scaler_in.cached_mu, fake.scaler_in.cached_var
        cases[pre + "out_mu"], cases[pre + "out_var"] = fake.scaler_out.cached_mu, fake.scaler_out.cached_var
        # a second fit on other rows exercises the running-moment branch (count > 0)
        more = rs.standard_normal((57, in_dim)).astype(np.float32) * 2 + 1
        with sess.as_default():
            fake.scaler_in.fit(more)
        cases[pre + "more"], cases[pre + "in_mu2"], cases[pre + "in_var2"] = more, fake.scaler_in.cached_mu, fake.scaler_in.cached_var
        cases[pre + "in_count2"] = float(fake.scaler_in.cached_count)
        print("pe_train", name, "steps", len(steps), "holdout evals", sess.k, "elites", fake._model_inds)
    np.savez_compressed(os.path.join(out, "g9_pe_train.npz"), **cases)



# ------------------------------------------------------------------------------------------------
# G10: small pure functions of the trainer loop
# ------------------------------------------------------------------------------------------------
def gen_loop_helpers(out):
    """format_samples_for_dyn (models/pens/pe_factory.py:74-107), update_dict (models/pens/logger.py:7-17) and the
    rollout-length schedule of CMBPO._set_rollout_length (algorithms/cmbpo.py:493-509), run from the reference."""
    from models.pens.pe_factory import format_samples_for_dyn
    from models.pens.logger import update_dict
    from algorithms.cmbpo import CMBPO
    rng = np.random.default_rng(77)
    n, D, A = 50, 6, 2
    samples = dict(observations=rng.standard_normal((n, D)).astype(np.float32),
                   actions=rng.standard_normal((n, A)).astype(np.float32),
                   next_observations=rng.standard_normal((n, D)).astype(np.float32),
                   rewards=rng.standard_normal((n, 1)).astype(np.float32),
                   costs=(rng.random((n,)) < 0.3).astype(np.float32),
                   terminals=np.zeros((n, 1), np.float32))
    data = {"s_" + k: v for k, v in samples.items()}
    for tag, kw in (("r", dict(append_r=True, append_c=False)), ("rc", dict(append_r=True, append_c=True)),
                    ("none", dict(append_r=False, append_c=False))):
        x, y = format_samples_for_dyn(samples, **kw)
        data["dyn_in_" + tag], data["dyn_out_" + tag] = x, y
    a = {"k1": 1.0, "k2": 4.0, "only_a": 7.0}
    b = {"k1": 3.0, "k2": -2.0, "only_b": 5.0}
    for i, (wa, wb) in enumerate(((0.5, 0.5), (0.25, 0.75), (1.0, 1.0))):
        d = update_dict(a, b, weight_a=wa, weight_b=wb)
        data[f"ud{i}_keys"] = np.array(sorted(d))
        data[f"ud{i}_vals"] = np.array([d[k] for k in sorted(d)])
        data[f"ud{i}_w"] = np.array([wa, wb])
    sched = []
    for schedule in ([10, 500, 5, 30], [20, 100, 1, 15], [0, 1, 4, 4]):
        for epoch in (0, 5, 10, 11, 60, 255, 500, 900):
            fake = types.SimpleNamespace(_rollout_schedule=schedule, _epoch=epoch,
                                         model_sampler=types.SimpleNamespace(set_max_path_length=lambda v: None))
            CMBPO._set_rollout_length(fake)
            sched.append(schedule + [epoch, fake._rollout_length])
    data["schedule"] = np.array(sched)
    np.savez_compressed(os.path.join(out, "g10_loop_helpers.npz"), **data)
    print("loop helpers:", len(sched), "schedule points")



# ------------------------------------------------------------------------------------------------
# G11: CPOBuffer archive accessors over several epochs (start-state sampling of the trainer loop)
# ------------------------------------------------------------------------------------------------
def gen_cpobuffer_archive(out):
    """buffers/cpobuffer.py:292-530 after three epochs have been moved to the archive: boltz_dist,
    distributed_batch_from_archive, epoch_batch, rand_batch_from_archive, get_archive under np.random.seed."""
    from buffers.cpobuffer import CPOBuffer
    rng = np.random.default_rng(909)
    D, A = 5, 2
    buf = CPOBuffer(size=40, archive_size=100, observation_space=_Space(D), action_space=_Space(A))
    buf.initialize({"mu": [A], "log_std": [A]}, gamma=0.99, lam=0.95, cost_gamma=0.97, cost_lam=0.5)
    plan = [(3, [5, 9]), (4, [12]), (6, [3, 3, 10])]          # (epoch tag, path lengths)
    n = sum(sum(ls) for _, ls in plan)
    data = dict(D=D, A=A, obs=rng.standard_normal((n, D)).astype(np.float32), act=rng.standard_normal((n, A)).astype(np.float32),
                rew=rng.standard_normal(n).astype(np.float32), val=rng.standard_normal(n).astype(np.float32),
                cost=(rng.random(n) < 0.3).astype(np.float32), cval=rng.standard_normal(n).astype(np.float32),
                logp=rng.standard_normal(n).astype(np.float32), mu=rng.standard_normal((n, A)).astype(np.float32),
                log_std=np.full((n, A), -0.5, np.float32), last=rng.standard_normal((8, 2)).astype(np.float32),
                plan_epochs=np.array([e for e, _ in plan]), plan_lengths=np.array([len(ls) for _, ls in plan]),
                path_lengths=np.array([l for _, ls in plan for l in ls]))
    i = p = 0
    for epoch, ls in plan:
        for L in ls:
            for _ in range(L):
                buf.store(data["obs"][i], data["act"][i], data["obs"][i] + 1, data["rew"][i], data["val"][i], data["cost"][i],
                          data["cval"][i], data["logp"][i], {"mu": data["mu"][i], "log_std": data["log_std"][i]}, False, epoch)
                i += 1
            buf.finish_path(data["last"][p, 0:1], data["last"][p, 1:2])
            p += 1
        buf.get()
    data["arch_size"], data["epochs_list"] = buf.arch_size, np.array(buf.epochs_list)
    data["max_ep"], data["min_ep"] = buf.max_ep, buf.min_ep
    kls = np.array([0.02, 0.5, 0.1])
    data["kls"] = kls
    dist = buf.boltz_dist(kls, alpha=2)
    data["boltz"] = dist
    np.random.seed(5)
    b = buf.distributed_batch_from_archive(23, dist, fields=["observations", "pi_infos"])
    data["dist_obs"], data["dist_mu"] = b["observations"], b["mu"]
    e = buf.epoch_batch(7, buf.epochs_list, fields=["observations", "pi_infos"])
    data["ep_obs"], data["ep_ls"] = e["observations"], e["log_std"]
    r = buf.rand_batch_from_archive(11, fields=["observations", "rewards"])
    data["rand_obs"], data["rand_rew"] = r["observations"], r["rewards"]
    arch = buf.get_archive(["observations", "actions", "next_observations", "rewards", "costs", "terminals", "epochs"])
    for k, v in arch.items():
        data["arch_" + k] = v
    np.savez_compressed(os.path.join(out, "g11_cpobuffer_archive.npz"), **data)
    print("cpobuffer archive:", buf.arch_size, "samples, epochs", list(buf.epochs_list))


# ------------------------------------------------------------------------------------------------
# G12: the reference's real-environment sampler (samplers/cpo_sampler.py:125-235) + CPOBuffer, driven by a scripted toy
# environment and a stub policy (tests/toyworld.py): what it hands to its buffer, what it logs, and the buffer's get()
# ------------------------------------------------------------------------------------------------
def gen_cpo_sampler(out):
    sys.path.insert(0, os.path.dirname(HERE))
    import toyworld
    from samplers.cpo_sampler import CpoSampler
    from buffers.cpobuffer import CPOBuffer
    data = {}
    # (a) against a recording pool: the exact store() / finish_path() call sequence
    pool = toyworld.RecordingPool()
    res = toyworld.drive(CpoSampler, pool)
    data["stores"] = np.array(pool.stores)
    data["finishes"] = np.array(pool.finishes)
    for k, v in res.items():
        data["rec_" + k] = v
    # (b) into the reference's own CPOBuffer: the 12-array list of get()
    D, A = toyworld.ToyEnv.D, toyworld.ToyEnv.A
    buf = CPOBuffer(size=128, archive_size=512, observation_space=_Space(D), action_space=_Space(A))
    buf.initialize({"mu": [A], "log_std": [A]}, gamma=0.99, lam=0.95, cost_gamma=0.97, cost_lam=0.5)
    res2 = toyworld.drive(CpoSampler, buf)
    got, diag = buf.get()
    names = ["obs", "act", "adv", "cadv", "ret", "cret", "logp", "val", "cval", "cost", "log_std", "mu"]
    for k, v in zip(names, got):
        data["get_" + k] = v
    data["poolr_ret_mean"], data["poolr_cret_mean"] = diag["poolr_ret_mean"], diag["poolr_cret_mean"]
    assert np.array_equal(res2["rets"], res["rets"])
    np.savez_compressed(os.path.join(out, "g12_cpo_sampler.npz"), **data)
    print("cpo sampler:", len(pool.stores), "stores,", len(pool.finishes), "finished paths, episodes", res["n_episodes"])


if __name__ == "__main__":
    if "--cpo-sampler-only" in sys.argv:
        install_stubs()
        gen_cpo_sampler(HERE)
    elif "--traces-only" in sys.argv:
        install_stubs()
        gen_sampler_traces(HERE)
    elif "--archive-only" in sys.argv:
        install_stubs()
        gen_cpobuffer_archive(HERE)
    elif "--loop-helpers-only" in sys.argv:
        install_stubs()
        gen_loop_helpers(HERE)
    elif "--pe-train-only" in sys.argv:
        install_stubs()
        gen_pe_train(HERE)
    elif "--cpobuffer-only" in sys.argv:
        install_stubs()
        gen_cpobuffer(HERE)
    elif "--update-only" in sys.argv:
        install_stubs()
        gen_update_pi(HERE)
    else:
        main()
