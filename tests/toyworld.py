"""A scripted real-environment stand-in and a stub policy, shared by the golden generator (which drives the REFERENCE's
CpoSampler + CPOBuffer with them: tests/golden/make_golden.py, G12) and by the tests that drive this repo's mirror.
Nothing here comes from the reference: the environment and the policy are arbitrary deterministic functions."""
import numpy as np


class Space:
    def __init__(self, d):
        self.shape = (d,)


class ToyEnv:
    """gym-like: reset() -> obs [1, D] (the sampler squeezes it), step(a) -> (obs [1, D], reward [1], done [1], info).
    Episodes end after the scripted lengths (a length above the sampler's max_path_length is a time-out)."""
    D, A = 5, 2

    def __init__(self, lengths=(5, 40, 3, 1, 9, 40), seed=7):
        self.observation_space, self.action_space = Space(self.D), Space(self.A)
        self.lengths = list(lengths)
        self.rng = np.random.default_rng(seed)
        self.M = (self.rng.standard_normal((self.D, self.D)) * 0.3).astype(np.float32)
        self.Bm = (self.rng.standard_normal((self.A, self.D)) * 0.5).astype(np.float32)
        self.episode = -1
        self.t = 0
        self.x = None
        self.closed = False

    def reset(self):
        self.episode += 1
        self.t = 0
        self.x = (0.1 * (1 + self.episode) * np.cos(np.arange(self.D, dtype=np.float32) + self.episode)).astype(np.float32)
        return self.x[None].copy()

    def step(self, a):
        a = np.asarray(a, np.float32).reshape(-1)
        self.x = (self.x @ self.M + a @ self.Bm + np.float32(0.01)).astype(np.float32)
        self.t += 1
        rew = np.array([-float(np.abs(self.x).sum())], np.float32)
        done = np.array([self.t >= self.lengths[self.episode % len(self.lengths)]])
        info = {"cost": float(self.x[0] > 0.05)} if self.episode % 3 != 2 else {}    # some steps carry no cost entry
        return self.x[None].copy(), rew, done, info

    def close(self):
        self.closed = True


class StubPolicy:
    """Duck type of CPOPolicy for the real-env sampler (samplers/cpo_sampler.py:131-139,203-211)."""

    def __init__(self, seed=11):
        rng = np.random.default_rng(seed)
        D, A = ToyEnv.D, ToyEnv.A
        self.W = (rng.standard_normal((D, A)) * 0.4).astype(np.float32)
        self.wv = rng.standard_normal(D).astype(np.float32)
        self.wc = rng.standard_normal(D).astype(np.float32)
        self.eps = rng.standard_normal((4096, A)).astype(np.float32)
        self.k = 0
        self.resets = 0

    def reset(self):
        self.resets += 1

    def get_v(self, obs):
        return np.asarray(np.dot(obs, self.wv), np.float32).reshape(1)

    def get_vc(self, obs):
        return np.asarray(np.dot(obs, self.wc), np.float32).reshape(1)

    def get_action_outs(self, obs):
        mu = np.tanh(obs @ self.W).astype(np.float32)
        log_std = np.full_like(mu, -0.5)
        e = self.eps[self.k]
        self.k += 1
        pi = (mu + np.exp(log_std) * e).astype(np.float32)
        logp = np.float32(-0.5 * float((e ** 2).sum()))
        return dict(pi=pi, logp_pi=logp, pi_info=dict(mu=mu, log_std=log_std), v=self.get_v(obs)[0], vc=self.get_vc(obs)[0])


class RecordingLogger:
    """Stands in for utilities/logx.EpochLogger: keeps every store() call in order."""

    def __init__(self):
        self.calls = []

    def store(self, **kw):
        self.calls.append({k: np.asarray(v, np.float64).reshape(-1) for k, v in kw.items()})

    def series(self, key):
        return np.concatenate([c[key] for c in self.calls if key in c]) if any(key in c for c in self.calls) else np.zeros(0)

    def log_tabular(self, *a, **k):
        pass


class RecordingPool:
    """Records what a sampler hands to its buffer: the arguments of store() and finish_path()."""
    max_size = 10 ** 9

    def __init__(self):
        self.stores, self.finishes = [], []

    @property
    def size(self):
        return len(self.stores)

    def store(self, obs, act, next_obs, rew, val, cost, cval, logp, pi_info, term, time_step):
        self.stores.append(np.concatenate([np.ravel(obs), np.ravel(act), np.ravel(next_obs), np.ravel(rew), np.ravel(val),
                                           np.ravel(np.float64(cost)), np.ravel(cval), np.ravel(logp),
                                           np.ravel(pi_info["mu"]), np.ravel(pi_info["log_std"]),
                                           np.ravel(np.float64(term)), [time_step]]).astype(np.float64))

    def finish_path(self, last_val=0, last_cval=0):
        lv, lc = np.asarray(last_val), np.asarray(last_cval)
        # the dtype matters downstream: float64 zeros promote the reward deltas
        self.finishes.append(np.array([len(self.stores), float(lv.reshape(-1)[0]), float(lc.reshape(-1)[0]),
                                       float(lv.dtype == np.float64), float(lc.dtype == np.float64)]))


def drive(sampler_cls, pool, steps=75, max_path_length=10, logger=None):
    """Run `steps` sampler steps (+ a final finish_all_paths with bootstraps, like the trainer at an epoch's end,
    algorithms/cmbpo.py:317-320); returns everything observable from outside."""
    env, policy = ToyEnv(), StubPolicy()
    logger = logger or RecordingLogger()
    sampler = sampler_cls(max_path_length=max_path_length, logger=logger)
    sampler.initialize(env, policy, pool)
    rets = []
    for t in range(steps):
        nxt, rew, done, info = sampler.sample(timestep=t)
        rets.append(np.concatenate([np.ravel(nxt), np.ravel(rew), np.ravel(np.float64(done)), [float(info.get("cost", -1.0))]]))
    sampler.finish_all_paths(append_val=True, append_cval=True, reset_path=False)
    diag = sampler.get_diagnostics()
    out = dict(rets=np.array(rets), cum_cost=float(sampler.cum_cost), total_samples=float(sampler._total_samples),
               n_episodes=float(sampler._n_episodes), max_path_return=float(diag["max-path-return"]),
               last_path_return=float(diag["last-path-return"]), pool_size=float(diag["pool-size"]),
               path_length_open=float(sampler._path_length), policy_resets=float(policy.resets),
               last_path_obs=np.asarray(sampler.last_path["observations"], np.float64),
               last_path_rew=np.asarray(sampler.last_path["rewards"], np.float64).reshape(-1),
               last_path_cost=np.asarray(sampler.last_path["cost"], np.float64).reshape(-1),
               last_path_term=np.asarray(sampler.last_path["terminals"], np.float64).reshape(-1))
    for key in ("VVals", "CostVVals", "RetEp", "EpLen", "CostEp", "CostFullEp"):
        out["log_" + key] = logger.series(key)
    return out
