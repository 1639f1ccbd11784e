"""Object construction from the reference's experiment variant (config schema of ``configs/*.py`` merged over
``configs/baseconfig/base.py``): the registries of ``buffers/utils.py``, ``samplers/utils.py``, ``policies/utils.py``
and ``algorithms/utils.py`` and the build order of ``scripts/run.py:44-76``, with this package's classes behind them.
The variant is the RESOLVED dict (ray-tune ``sample_from`` entries already evaluated); the environment is passed in
(``get_env_from_params`` builds gym / safety-gym environments in the reference, which are out of scope here)."""
from copy import deepcopy


def get_cpobuffer(env, *args, **kwargs):
    from .cpobuffer import CPOBuffer
    return CPOBuffer(*args, observation_space=env.observation_space, action_space=env.action_space, **kwargs)


def get_cposampler(*args, **kwargs):
    from .cpo_sampler import CpoSampler
    return CpoSampler(*args, **kwargs)


def get_cpo_policy(env, session=None, *args, **kwargs):
    from .cpo_policy import CPOPolicy
    return CPOPolicy(obs_space=env.observation_space, act_space=env.action_space, session=session, *args, **kwargs)


def create_CMBPO_algorithm(variant, *args, **kwargs):
    from .cmbpo import CMBPO
    return CMBPO(*args, **kwargs)


BUFFER_FUNCTIONS = {'CPOBuffer': get_cpobuffer}
SAMPLERS_FUNCTIONS = {'CPOSampler': get_cposampler}
POLICY_FUNCTIONS = {'cpopolicy': get_cpo_policy}
ALGORITHM_CLASSES = {'CMBPO': create_CMBPO_algorithm}


def get_buffer_from_params(params, env, *args, **kwargs):
    p = params['buffer_params']
    return BUFFER_FUNCTIONS[p.get('type', 'CPOBuffer')](env, *args, **deepcopy(p.get('kwargs', {})), **kwargs)


def get_sampler_from_params(params, *args, **kwargs):
    p = params['sampler_params']
    kw = deepcopy(p.get('kwargs', {}))
    return SAMPLERS_FUNCTIONS[p.get('type', 'CPOSampler')](*deepcopy(p.get('args', ())), *args, **kw, **kwargs)


def get_policy_from_params(params, env, *args, **kwargs):
    p = params['policy_params']
    return POLICY_FUNCTIONS[p['type']](env, *args, **deepcopy(p['kwargs']), **kwargs)


def get_algorithm_from_params(variant, *args, **kwargs):
    p = variant['algorithm_params']
    kw = deepcopy(p['kwargs'])
    if hasattr(kw, 'toDict'):
        kw = kw.toDict()
    return ALGORITHM_CLASSES[p['type']](variant, *args, **kw, **kwargs)


def build_experiment(params, env, device=None):
    """scripts/run.py:44-76: buffer, sampler, policy, algorithm -- returns the algorithm (``.train()`` is the generator
    the reference's tune trainable steps).  Entries the reference fills in through ray-tune lambdas may be left out:
    the buffer size defaults to ``epoch_length``, the archive to 3e5 samples, the sampler's / policy's
    ``max_path_length`` to the environment's ``max_episode_steps`` attribute (or 1000), the task to ``params['task']``."""
    params = deepcopy(params)
    akw = params['algorithm_params']['kwargs']
    mpl = int(getattr(env, 'max_episode_steps', getattr(env, '_max_episode_steps', 1000)))
    params.setdefault('buffer_params', {})
    bkw = params['buffer_params'].setdefault('kwargs', {})
    bkw.setdefault('size', int(akw.get('epoch_length', 50000)))
    bkw.setdefault('archive_size', int(3e5))
    params.setdefault('sampler_params', {})
    skw = params['sampler_params'].setdefault('kwargs', {})
    skw.setdefault('max_path_length', mpl)
    params['policy_params']['kwargs'].setdefault('max_path_length', skw['max_path_length'])
    akw.setdefault('task', params.get('task', 'default'))
    dev = {} if device is None else {'device': device}
    buffer = get_buffer_from_params(params, env)
    sampler = get_sampler_from_params(params)
    policy = get_policy_from_params(params, env, None, **dev)
    for k in ('eval_render_mode', 'eval_n_episodes', 'eval_deterministic'):
        akw.pop(k, None)
    return get_algorithm_from_params(variant=params, env=env, policy=policy, buffer=buffer, sampler=sampler, **dev)
