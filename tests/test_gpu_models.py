"""GPU parity tests (-m gpu) of the reward / disturbance variants: every disturbance model of
quadjax/dynamics/free.py:9-72 and both rewards Quad3D binds (dynamics/utils.py:285-313) in the rollout, the Hessian, the env
step, covo-offline's nominal kernels and the fused step -- through the C ABI, against the CPU oracle (oracle/ref_np.py,
covo_oracle.c) on the same inputs and the same explicit draws.  Tolerances: rollout cost <= 1e-5 relative (north_star);
Hessian <= 1e-9 (no force table) / 2e-8 relative to max|R| (force table: its entries are fp32 like the reference's f_disturb).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
if not torch.cuda.is_available():
    pytest.skip("needs the MI355X", allow_module_level=True)

from covo_mpc_amd import _lib  # noqa: E402
from covo_mpc_amd import random as cr  # noqa: E402
from covo_mpc_amd.controllers._core import SamplingCore  # noqa: E402
from covo_mpc_amd.dynamics.dataclass import EnvParams3D  # noqa: E402
from oracle import c_oracle as CO  # noqa: E402
from oracle import ref_np as R  # noqa: E402
from tests.conftest import make_problem, table_ref  # noqa: E402
from tests.test_gpu_parity import DEV, dev_state, rel_err, sample_actions, to_stripes  # noqa: E402

KINDS = ["periodic", "sin", "drag", "mixed"]
# disturb_params: every component live (dataclass.py:88; DR / reset draw them); fp32 numbers, like every parameter of the reference
DP = tuple(float(np.float32(x)) for x in (0.7, -0.4, 0.9, 0.15, -0.35, 0.6))


def params_c(p: R.Params, kind="none", reward="penyaw", rollover=False):
    e = EnvParams3D(disturb_params=np.asarray(p.disturb_params, dtype=np.float32), disturb_period=p.disturb_period,
                    disturb_scale=p.disturb_scale, dyn_noise_scale=p.dyn_noise_scale)
    return e.to_c(rollover_terminate=rollover, reward=reward, disturb_type=kind)


def disturb_key(k):
    """disturb_key of step_env(k): quadrotor.py:262, free.py:136,144"""
    return cr.split(cr.split(cr.split(k)[1])[0])[0]


def step_keys(key, mode, H=32):
    """the key step_env receives at every rollout step (include/covo_hip.h: COVO_DISTURB_KEYS_*)"""
    out = []
    for _ in range(H):
        if mode == _lib.DISTURB_KEYS_SHARED:
            out.append(key)
        elif mode == _lib.DISTURB_KEYS_HESSIAN:  # covo.py:151
            rk, key = cr.split(key)
            out.append(rk)
        else:  # covo.py:60,66
            _, key = cr.split(key)
            rs, key = cr.split(key)
            out.append(rs)
    return out


def uniform_draws(p, key, mode, H=32):
    """(H,3): uniform(disturb_key, (3,), -scale, scale) of every step (free.py:16-21)"""
    return np.stack([cr.uniform(disturb_key(k), (3,), -p.disturb_scale, p.disturb_scale) for k in step_keys(key, mode, H)])


# ------------------------------------------------------------------------------------------ the table
@pytest.mark.parametrize("kind", KINDS + ["gaussian"])
@pytest.mark.parametrize("mode", [_lib.DISTURB_KEYS_SHARED, _lib.DISTURB_KEYS_HESSIAN, _lib.DISTURB_KEYS_NOMINAL])
def test_disturb_table_vs_model_functions(kind, mode):
    s, p, rng = make_problem(seed=5, time=93)  # steps 7 (time 100) and none other redraw; time 93 + 31 < 150
    p = p.replace(disturb_params=DP, disturb_period=10 if mode == _lib.DISTURB_KEYS_HESSIAN else 50)  # period 10: three redraws
    core = SamplingCore(64, 32, 0.01, 1.0, device=DEV)
    key = cr.PRNGKey(77)
    ds = dev_state(s)
    tab = core.disturb_table(params_c(p, kind), ds.packed, key=key, key_mode=mode, deterministic=False)[0].cpu().numpy()
    if kind == "gaussian":
        z = np.stack([cr.normal(disturb_key(k), (3,)) for k in step_keys(key, mode)])
        assert np.abs(tab[1:, :3] - np.float32(p.dyn_noise_scale) * z[:-1]).max() < 1e-7 and np.all(tab[:, 3] == 0)
        det = core.disturb_table(params_c(p, kind), ds.packed, key=key, key_mode=mode, deterministic=True)[0].cpu().numpy()
        assert np.all(det == 0)
        return
    draws = uniform_draws(p, key, mode)
    ref = table_ref(p, s, kind, draws)
    # fp32 sin of an argument of ~15 rad (the host env's own arithmetic, free.py:27-38 in fp32): ~2e-7 on a 0.1 N force
    assert np.abs(tab - ref).max() < 1e-6, np.abs(tab - ref).max()
    hits = [(s.time + k) % p.disturb_period == 0 for k in range(31)]
    assert sum(hits) == (3 if mode == _lib.DISTURB_KEYS_HESSIAN else 1)
    # batch: per-entry keys from a device array, per-entry states
    keys = np.stack([cr.PRNGKey(77), cr.PRNGKey(78)]).astype(np.uint32)
    s2 = s.replace(time=s.time + 3, f_disturb=s.f_disturb * 0.5)
    packed = torch.stack([ds.packed, dev_state(s2).packed])
    kd = torch.from_numpy(keys.view(np.int32)).to(DEV)
    tb = core.disturb_table(params_c(p, kind), packed, keys_dev=kd, key_mode=mode, deterministic=True, batch=2).cpu().numpy()
    assert np.array_equal(tb[0], tab)
    assert np.abs(tb[1] - table_ref(p, s2, kind, uniform_draws(p, keys[1], mode))).max() < 1e-6


@pytest.mark.parametrize("kind", KINDS + ["gaussian"])
@pytest.mark.parametrize("mode", [_lib.DISTURB_KEYS_SHARED, _lib.DISTURB_KEYS_HESSIAN, _lib.DISTURB_KEYS_NOMINAL])
def test_host_table_builder_equals_the_device_table(kind, mode):
    """Quad3D.rollout_disturbance_table (the host restatement the jax-stream path uses with random_jax) fed the library's own
    key module must reproduce the device kernel's table: same key threading, same draws (bit-equal), sin to fp32 rounding."""
    import covo_mpc_amd as cm
    s, p, rng = make_problem(seed=5, time=93)
    p = p.replace(disturb_params=DP, disturb_period=10)
    env = cm.envs.Quad3D(task="tracking_zigzag", enable_randomizer=False, disturb_type=kind, disable_rollover_terminate=True,
                         generate_noisy_state=True, device=DEV)
    ep = env.default_params.replace(disturb_params=np.asarray(DP, dtype=np.float32), disturb_period=10)
    core = SamplingCore(64, 32, 0.01, 1.0, device=DEV)
    key = cr.PRNGKey(77)
    for det in (False, True):
        dev = core.disturb_table(params_c(p, kind), dev_state(s).packed, key=key, key_mode=mode, deterministic=det)[0].cpu().numpy()
        host = env.rollout_disturbance_table(key, ep, s.time, s.f_disturb, mode, det, rng=cr)
        assert host.shape == dev.shape == (32, 4)
        if kind in ("sin", "mixed"):
            assert np.abs(host - dev).max() < 1e-6
        else:
            assert np.array_equal(host, dev), (kind, mode, det)
    assert np.any(dev != 0) or kind in ("gaussian", "drag")


@pytest.mark.parametrize("name,kind", [("covo-online", "periodic"), ("mppi", "mixed")])
def test_jax_stream_with_table_models(name, kind):
    """noise_stream = "jax" with a table-driven disturbance model: step key and the models' uniform draws come from jax.random's
    bitstream (host-built table, random_jax) -- the rollout costs equal the oracle's with exactly those draws, and differ from the
    run on the library's own stream."""
    import covo_mpc_amd as cm
    from oracle import jax_rng_np as J  # the checker's restatement of jax's stream (not the product's host twin)

    class rj:
        PRNGKey = staticmethod(J.prng_key)
        split = staticmethod(J.split)
        uniform = staticmethod(lambda key, shape, lo, hi: J.uniform(key, int(np.prod(shape)), lo, hi).reshape(shape))
    N = 512
    env = cm.envs.Quad3D(task="tracking_zigzag", enable_randomizer=False, disturb_type=kind, disable_rollover_terminate=True,
                         generate_noisy_state=True, device=DEV)
    params = env.default_params.replace(disturb_params=np.asarray(DP, dtype=np.float32), disturb_period=4)
    c, cp = cm.envs.get_controller(env, name, f"N{N}_H32_lam0.01", device=DEV)
    c.noise_stream = "jax"
    obs, info, state = env.reset(cr.PRNGKey(3), params)
    rng_act = rj.PRNGKey(11)
    u, cp2, _ = c(obs, state, params, rng_act, cp, info)
    assert np.all(np.isfinite(cp2.a_mean.cpu().numpy()))
    cost_jax = c.core.cost.cpu().numpy().copy()
    a_dev = c.core.a.permute(1, 0, 2).contiguous().cpu().numpy()
    ns = info["noisy_state"]
    so = R.State(pos=ns.pos, vel=ns.vel, quat=ns.quat, omega=ns.omega, f_disturb=ns.f_disturb, pos_tar=ns.pos_tar,
                 vel_tar=ns.vel_tar, acc_tar=ns.acc_tar, time=ns.time, pos_traj=ns.pos_traj, vel_traj=ns.vel_traj,
                 acc_traj=ns.acc_traj).astype(np.float64)
    p = R.Params().fp32().replace(disturb_params=DP, disturb_period=4)
    step_key = rj.split(rj.split(rng_act)[0])[1]  # rng, act_key = split(rng_act); rng, step_key = split(rng) on jax's stream
    dk = rj.split(rj.split(rj.split(step_key)[1])[0])[0]  # quadrotor.py:262, free.py:136,144
    draw = np.asarray(rj.uniform(dk, (3,), -p.disturb_scale, p.disturb_scale), dtype=np.float64)
    ref = CO.rollout(so, p, a_dev.astype(np.float64), 1.0, dtype=np.float64, disturb=R.Disturb(kind, draw, name != "mppi"))
    assert rel_err(cost_jax, ref).max() < 1e-5, rel_err(cost_jax, ref).max()
    # the library's own stream on the same key bits draws another force: other costs
    draw_philox = np.asarray(cr.uniform(disturb_key(cr.split(cr.split(rng_act)[0])[1]), (3,), -p.disturb_scale, p.disturb_scale))
    assert np.abs(draw - draw_philox).max() > 1e-3


# ------------------------------------------------------------------------------------------ rollout
def _rollout_dev(core, s, pc, a, f_shared=(0.0, 0.0, 0.0), tab=None, want_stats=False):
    core.a.copy_(to_stripes(a))
    return core.rollout(dev_state(s), pc, f_shared, want_stats, f_steps=tab).cpu().numpy()


@pytest.mark.parametrize("reward", ["penyaw", "realworld"])
@pytest.mark.parametrize("kind", ["none", "gaussian"] + KINDS)
def test_rollout_reward_and_disturbance_variants_vs_fp64_oracle(kind, reward):
    if kind in ("none", "gaussian") and reward == "penyaw":
        pytest.skip("the default family: tests/test_gpu_parity.py")
    s, p, rng = make_problem(seed=17, time=37)  # time 37: step 13 (time 50) redraws the periodic part
    p = p.replace(disturb_params=DP)
    N = 3000
    a = sample_actions(p, rng, N)
    key = cr.PRNGKey(5)
    core = SamplingCore(N, 32, 0.01, 0.97, device=DEV)
    pc = params_c(p, kind, reward)
    ds = dev_state(s)
    if kind in KINDS:
        tab = core.disturb_table(pc, ds.packed, key=key, key_mode=_lib.DISTURB_KEYS_SHARED, deterministic=True)
        draw = cr.uniform(disturb_key(key), (3,), -p.disturb_scale, p.disturb_scale).astype(np.float64)
        cost = _rollout_dev(core, s, pc, a, tab=tab, want_stats=True)
        ref, rew, poses = CO.rollout(s, p, a.astype(np.float64), 0.97, dtype=np.float64, want_rewards=True, want_poses=True,
                                     reward=reward, disturb=R.Disturb(kind, draw, True))
    else:
        fs = np.array([0.02, -0.03, 0.01], dtype=np.float32)
        cost = _rollout_dev(core, s, pc, a, f_shared=fs, want_stats=True)
        ref, rew, poses = CO.rollout(s, p, a.astype(np.float64), 0.97, fs.astype(np.float64), dtype=np.float64, want_rewards=True,
                                     want_poses=True, reward=reward)
    assert rel_err(cost, ref).max() < 1e-5, rel_err(cost, ref).max()
    # the variant matters (guards against a silently ignored switch)
    base = CO.rollout(s, p, a.astype(np.float64), 0.97, dtype=np.float64)
    assert np.abs(base - ref).max() > 1e-3
    # position statistics of the same launch (covo.py:281) and the per-wave minima
    info = core.info(ds)
    pm, ps = R.pos_stats(poses)
    assert np.abs(info["pos_mean"].cpu().numpy() - pm).max() < 2e-5 and np.abs(info["pos_std"].cpu().numpy() - ps).max() < 2e-5
    assert np.array_equal(core.blockmin.cpu().numpy(), np.array([cost[i:i + 64].min() for i in range(0, N, 64)], dtype=np.float32))
    # no statistics: the same arithmetic (the rollout TUs are compiled with -ffp-contract=off so that template variants agree)
    cost2 = _rollout_dev(core, s, pc, a, f_shared=(0.02, -0.03, 0.01), tab=tab if kind in KINDS else None)
    assert np.array_equal(cost, cost2)


@pytest.mark.parametrize("kind,reward,N", [("mixed", "penyaw", 70016), ("periodic", "realworld", 20000), ("drag", "realworld", 257),
                                           ("sin", "penyaw", 1)])
def test_rollout_variants_shapes_rollover_and_freeze(kind, reward, N):
    """1 / 2 / 4 groups per workgroup, ragged tails, rollover termination on, samples leaving the box (frozen rewards)."""
    s, p, rng = make_problem(seed=23, time=290)  # the horizon passes the episode end: time >= 300 terminates
    p = p.replace(disturb_params=DP, disturb_period=7)  # several redraws inside the horizon
    s = s.replace(pos=s.pos + np.array([2.7, 0, 0]), vel=s.vel + np.array([2.0, 0, 0]))
    a = np.clip(R.hover_action(p, 32, np.float64)[None] + np.array([0.3, 1.2, 1.2, 0.5]) * rng.normal(size=(N, 32, 4)), -1, 1)
    a = a.astype(np.float32)
    key = cr.PRNGKey(9)
    core = SamplingCore(N, 32, 0.01, 1.0, device=DEV)
    pc = params_c(p, kind, reward, rollover=True)
    tab = core.disturb_table(pc, dev_state(s).packed, key=key, key_mode=_lib.DISTURB_KEYS_SHARED, deterministic=True)
    cost = _rollout_dev(core, s, pc, a, tab=tab)
    idx = rng.choice(N, min(N, 3000), replace=False)
    draw = cr.uniform(disturb_key(key), (3,), -p.disturb_scale, p.disturb_scale).astype(np.float64)
    d = R.Disturb(kind, draw, True)
    ref = CO.rollout(s, p, a[idx].astype(np.float64), 1.0, dtype=np.float64, rollover=True, reward=reward, disturb=d)
    ref32 = CO.rollout(s.astype(np.float32), p, a[idx], 1.0, dtype=np.float32, rollover=True, reward=reward, disturb=d)
    err = np.minimum(rel_err(cost[idx], ref), rel_err(cost[idx], ref32.astype(np.float64)))  # rollover ties: see test_gpu_parity
    assert (err < 1e-5).mean() > 0.995 and np.median(err) < 3e-6, (err.max(), np.median(err))


# ------------------------------------------------------------------------------------------ Hessian
@pytest.mark.parametrize("kind,reward,method", [("none", "realworld", "adjoint"), ("none", "realworld", "pairs"),
                                                ("periodic", "penyaw", "adjoint"), ("periodic", "realworld", "pairs"),
                                                ("sin", "realworld", "adjoint"), ("sin", "penyaw", "pairs"),
                                                ("drag", "penyaw", "adjoint"), ("mixed", "realworld", "adjoint"),
                                                ("mixed", "penyaw", "pairs"), ("drag", "realworld", "pairs")])
def test_hessian_reward_and_disturbance_variants_vs_ad_oracle(kind, reward, method):
    """covo_hessian (second-order adjoint) and covo_hessian_pairs for every model against the fp64 hyper-dual C oracle (itself
    equal to torch forward-over-forward AD: tests/test_oracle.py).  drag / mixed: the force is part of the differentiated state
    -- the adjoint kernels' 16-component instantiation (hessian_adj.hip: adj16)."""
    s, p, rng = make_problem(seed=3, time=41)  # step 9 (time 50) redraws
    p = p.replace(disturb_params=DP)
    a = (R.hover_action(p, 32, np.float64) + 0.1 * rng.normal(size=(32, 4))).astype(np.float32)
    a[3, 1] = 1.0  # clip tie (two clips on the path)
    key = cr.PRNGKey(21)
    core = SamplingCore(256, 32, 0.01, 1.0, device=DEV)
    ds = dev_state(s)
    pc = params_c(p, kind, reward)
    tab = None
    draws = None
    if kind in KINDS:
        tab = core.disturb_table(pc, ds.packed, key=key, key_mode=_lib.DISTURB_KEYS_HESSIAN, deterministic=True)
        draws = uniform_draws(p, key, _lib.DISTURB_KEYS_HESSIAN).astype(np.float64)
    Rm = core.hessian(ds.packed, ds, pc, torch.from_numpy(a.reshape(-1)).to(DEV), method=method, f_steps=tab)[0].cpu().numpy()
    ref = CO.hessian(s, p, a.reshape(-1).astype(np.float64), 32, reward=reward, kind=kind, draws=draws)
    assert np.abs(Rm - Rm.T).max() == 0.0 and np.abs(Rm[124:]).max() == 0.0
    if kind == "none":
        assert np.abs(Rm - ref).max() < 1e-9 * max(1.0, np.abs(ref).max()), np.abs(Rm - ref).max()
    else:
        # (i) the kernel given ITS table (fp32 entries, like the reference's fp32 f_disturb): the oracle fed the same rows, 1e-9
        ref_t = CO.hessian(s, p, a.reshape(-1).astype(np.float64), 32, reward=reward, kind=kind, table=tab[0].cpu().numpy())
        assert np.abs(Rm - ref_t).max() < 1e-9 * max(1.0, np.abs(ref_t).max()), np.abs(Rm - ref_t).max()
        # (ii) against the model functions evaluated in fp64: what the table's fp32 rounding (sin of a ~15 rad argument: 2e-7 N)
        # moves the curvature of the log-shaped reward by
        assert np.abs(Rm - ref).max() < 1e-6 * max(1.0, np.abs(ref).max()), np.abs(Rm - ref).max()
    other = CO.hessian(s, p, a.reshape(-1).astype(np.float64), 32)  # penyaw, no force: a different matrix
    assert np.abs(other - ref).max() > 1e-6


def test_hessian_batched_with_force_tables():
    """batch > 1 (covo-offline's table): per-entry states, means, keys and force tables."""
    s, p, rng = make_problem(seed=8, time=45)
    p = p.replace(disturb_params=DP)
    core = SamplingCore(256, 32, 0.01, 1.0, device=DEV)
    B = 5
    states, means, keys, refs = [], [], [], []
    pc = params_c(p, "periodic", "realworld")
    for b in range(B):
        sb = s.replace(time=s.time + b, pos=s.pos + 0.01 * b, f_disturb=s.f_disturb * (1 + 0.1 * b))
        ab = (R.hover_action(p, 32, np.float64) + 0.1 * rng.normal(size=(32, 4))).astype(np.float32)
        kb = cr.PRNGKey(100 + b)
        states.append(dev_state(sb).packed)
        means.append(torch.from_numpy(ab.reshape(-1)).to(DEV))
        keys.append(kb)
        refs.append(CO.hessian(sb, p, ab.reshape(-1).astype(np.float64), 32, reward="realworld", kind="periodic",
                               draws=uniform_draws(p, kb, _lib.DISTURB_KEYS_HESSIAN).astype(np.float64)))
    packed, am = torch.stack(states), torch.stack(means)
    kd = torch.from_numpy(np.stack(keys).astype(np.uint32).view(np.int32)).to(DEV)
    tab = core.disturb_table(pc, packed, keys_dev=kd, key_mode=_lib.DISTURB_KEYS_HESSIAN, deterministic=True, batch=B)
    Rm = core.hessian(packed, dev_state(s), pc, am, batch=B, f_steps=tab).cpu().numpy()
    for b in range(B):
        assert np.abs(Rm[b] - refs[b]).max() < 1e-6 * max(1.0, np.abs(refs[b]).max()), b  # fp32 table entries, see above


# ------------------------------------------------------------------------------------------ env step / nominal / fused step
@pytest.mark.parametrize("task,kind", [("tracking_zigzag", "periodic"), ("tracking_slow", "mixed"), ("tracking_slow", "sin"),
                                       ("tracking", "drag"), ("tracking_slow", "none")])
def test_env_step_kernel_models_vs_host_env(task, kind):
    """covo_env_step with every disturbance model / both rewards against the Python env on the same keys and actions."""
    import covo_mpc_amd as cm
    env = cm.envs.Quad3D(task=task, enable_randomizer=False, disturb_type=kind, disable_rollover_terminate=True,
                         generate_noisy_state=True, device=DEV)
    params = env.default_params.replace(disturb_params=np.asarray(DP, dtype=np.float32), disturb_period=9)
    core = SamplingCore(256, 32, 0.01, 1.0, device=DEV)
    ep = cm.envs.DeviceEpisode(env, cr.PRNGKey(11), params, (core.lib, core.h), DEV)
    obs, info, state = env.reset(cr.PRNGKey(11), params)
    rng = np.random.default_rng(5)
    key = cr.PRNGKey(12)
    rewards, forces = [], []
    for t in range(40):
        key, k_step = cr.split(key)
        u = np.clip(np.array([-0.3378, 0, 0, 0]) + 0.3 * rng.normal(size=4), -1.2, 1.2).astype(np.float32)
        ep.step(k_step, torch.from_numpy(u).to(DEV))
        obs, state, reward, done, info = env.step(k_step, state, u, params)
        rewards.append(reward)
        forces.append(state.f_disturb.copy())
        t_dev = ep.true.cpu().numpy()
        assert np.abs(t_dev - state.pack()).max() < 2e-5, (t, np.abs(t_dev - state.pack()).max())
        assert np.abs(ep.noisy.cpu().numpy() - info["noisy_state"].pack()).max() < 2e-5, t
    log = ep.read_log()
    assert np.abs(log[:, 0] - np.asarray(rewards)).max() < 2e-5
    forces = np.asarray(forces)
    if kind != "none":
        assert np.abs(forces).max() > 1e-3 and len(np.unique(forces[:, 0])) > (2 if kind == "periodic" else 10)


@pytest.mark.parametrize("kind", ["gaussian", "periodic", "mixed", "drag"])
def test_offline_nominal_and_table_with_disturbance_models(kind):
    """covo-offline reset under the state / time dependent models: device chain + nominal rollouts (covo_pid_nominal) against
    the Python loop, and rows of the Sigma table against the oracle's Hessian -> optimize_sigma of those nominal means."""
    import covo_mpc_amd as cm
    env = cm.envs.Quad3D(task="tracking_zigzag", enable_randomizer=False, disturb_type=kind, disable_rollover_terminate=True,
                         generate_noisy_state=True, device=DEV)
    controller, cp = cm.envs.get_controller(env, "covo-offline", "N1024_H32_lam0.01", device=DEV)
    params = env.default_params.replace(disturb_params=np.asarray(DP, dtype=np.float32))
    obs, info, state = env.reset(cr.PRNGKey(31), params)
    ph, ah = controller._nominal_host(state, params, cr.PRNGKey(32))
    pd, ad, kd = controller._nominal_device(state, params, cr.PRNGKey(32))
    pd, ad = pd.cpu().numpy(), ad.cpu().numpy()
    assert np.abs(pd[:, :25] - ph[:, :25]).max() < 5e-5, np.abs(pd[:, :25] - ph[:, :25]).max()
    assert np.abs(ad - ah).max() < 3e-4, np.abs(ad - ah).max()
    assert np.abs(ph[:, 13:16]).max() > 1e-3
    # the scan's carry keys
    key = cr.PRNGKey(32)
    kd = kd.cpu().numpy().view(np.uint32)
    for t in range(5):
        assert np.array_equal(kd[t], key)
        _, key = cr.split(key)
        _, key = cr.split(key)
    # the table rows (batch-300 Hessian + Sigma) against the oracle on the DEVICE's nominal states / means
    cp2 = controller.reset(state, params, cp, cr.PRNGKey(32))
    p = R.Params().fp32().replace(disturb_params=DP)
    for t in (1, 49, 50, 137, 299):  # (not row 0: err_pos is exactly 0 at reset, where norm'(0) is NaN in JAX's and the oracle's AD)
        st = pd[t]
        so = R.State(pos=st[0:3], vel=st[3:6], quat=st[6:10], omega=st[10:13], f_disturb=st[13:16], pos_tar=st[16:19],
                     vel_tar=st[19:22], acc_tar=st[22:25], time=int(st[25:26].view(np.int32)[0]), pos_traj=state.pos_traj,
                     vel_traj=state.vel_traj, acc_traj=state.acc_traj).astype(np.float64)
        draws = uniform_draws(p, kd[t], _lib.DISTURB_KEYS_HESSIAN).astype(np.float64) if kind in KINDS else None
        Rm = CO.hessian(so, p, ad[t].astype(np.float64), 32, kind=kind, draws=draws)
        Sref = R.optimize_sigma(Rm, 0.5, 32, 4)
        S = cp2.a_cov_offline[t].cpu().numpy()
        assert np.linalg.norm(S - Sref) / np.linalg.norm(Sref) < 2e-5, (t, np.linalg.norm(S - Sref) / np.linalg.norm(Sref))
        L = cp2.a_chol_offline[t].cpu().numpy().astype(np.float64)
        assert np.linalg.norm(L @ L.T - S) / np.linalg.norm(S) < 1e-6


@pytest.mark.parametrize("graph", ["graph", "eager"])
@pytest.mark.parametrize("name,task,kind", [("covo-online", "tracking_zigzag", "periodic"), ("covo-online", "tracking_slow", "mixed"),
                                            ("mppi", "tracking_slow", "drag"), ("covo-offline", "tracking_slow", "sin"),
                                            ("mppi", "hovering", "periodic"), ("covo-online", "tracking_slow", "gaussian")])
def test_fused_step_with_models_equals_kernel_by_kernel_and_oracle(name, task, kind, graph, monkeypatch):
    """The fused step derives the step's force tables on the device from the raw controller key (disturb.hip); the
    kernel-by-kernel path builds them through covo_disturb_table from host-split keys: same bits.  The costs of the last step
    are then checked against the oracle with the draws the reference's key threading gives."""
    import covo_mpc_amd as cm
    monkeypatch.setenv("COVO_GRAPH" if graph == "graph" else "COVO_NO_GRAPH", "1")
    env = cm.envs.Quad3D(task=task, enable_randomizer=False, disturb_type=kind, disable_rollover_terminate=True,
                         generate_noisy_state=True, device=DEV)
    params = env.default_params.replace(disturb_params=np.asarray(DP, dtype=np.float32), disturb_period=4)
    N = 2048
    outs = []
    for fused in (True, False):
        controller, cp = cm.envs.get_controller(env, name, f"N{N}_H32_lam0.01", device=DEV)
        controller.materialize_eps = not fused
        obs, info, state = env.reset(cr.PRNGKey(3), params)
        cp = controller.reset(state, params, cp, cr.PRNGKey(4))
        key = cr.PRNGKey(6)
        for i in range(5):
            key, k_act, k_step = cr.split(key, 3)
            u, cp_new, cinfo = controller(obs, state, params, k_act, cp, info)
            ns, cp_prev, cp = info["noisy_state"], cp, cp_new
            obs, state, reward, done, info = env.step(k_step, state, u.cpu().numpy(), params)
        outs.append((cp.a_mean.cpu().numpy().copy(), controller.core.cost.cpu().numpy().copy(),
                     controller.core.a.permute(1, 0, 2).contiguous().cpu().numpy().copy(),
                     cinfo["pos_mean"].cpu().numpy().copy(), ns, k_act))
    if task != "tracking_slow":
        assert np.array_equal(outs[0][2], outs[1][2]) and np.array_equal(outs[0][1], outs[1][1])
        assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][3], outs[1][3])
    else:
        # The quadratic reward keeps the costs within ~0.5 of each other: at lambda = 0.01 hundreds of samples carry weight
        # (penyaw: one or two, the rest underflow to exactly 0), and the two paths sum them in different orders -- per-workgroup
        # online-softmax records merged (fused) vs one global-minimum pass (kernel by kernel): equal to fp32 rounding, and the
        # closed loop carries the last-ulp difference of the mean through the five steps
        assert np.abs(outs[0][2] - outs[1][2]).max() < 5e-6 and np.abs(outs[0][0] - outs[1][0]).max() < 5e-6
        assert rel_err(outs[0][1], outs[1][1]).max() < 1e-5 and np.abs(outs[0][3] - outs[1][3]).max() < 2e-5
    # oracle: the last step's costs on the device's own actions
    a_dev, cost_dev, ns, k_act = outs[0][2], outs[0][1], outs[0][4], outs[0][5]
    so = R.State(pos=ns.pos, vel=ns.vel, quat=ns.quat, omega=ns.omega, f_disturb=ns.f_disturb, pos_tar=ns.pos_tar,
                 vel_tar=ns.vel_tar, acc_tar=ns.acc_tar, time=ns.time, pos_traj=ns.pos_traj, vel_traj=ns.vel_traj,
                 acc_traj=ns.acc_traj).astype(np.float64)
    p = R.Params().fp32().replace(disturb_params=DP, disturb_period=4)
    step_key = cr.split(cr.split(k_act)[0])[1]  # rng, act_key = split(rng_act); rng, step_key = split(rng)  (covo.py:212,225)
    det = name != "mppi"
    reward = "realworld" if task == "tracking_slow" else "penyaw"
    if kind == "gaussian":
        d = R.Disturb("gaussian", cr.normal(disturb_key(step_key), (3,)).astype(np.float64), det)
    else:
        d = R.Disturb(kind, cr.uniform(disturb_key(step_key), (3,), -p.disturb_scale, p.disturb_scale).astype(np.float64), det)
    ref = CO.rollout(so, p, a_dev.astype(np.float64), 1.0, dtype=np.float64, reward=reward, disturb=d)
    assert rel_err(cost_dev, ref).max() < 1e-5, rel_err(cost_dev, ref).max()
    if name == "covo-online":  # the Sigma the last step sampled from: oracle Hessian with get_hessian's per-step keys
        am = R.shift_mean(cp_prev.a_mean.cpu().numpy().astype(np.float64))
        draws = uniform_draws(p, k_act, _lib.DISTURB_KEYS_HESSIAN).astype(np.float64) if kind in KINDS else None
        Rm = CO.hessian(so, p, am.reshape(-1), 32, reward=reward, kind=kind, draws=draws)
        Sref = R.optimize_sigma(Rm, 0.5, 32, 4)
        S = cp.a_cov.cpu().numpy()
        assert np.linalg.norm(S - Sref) / np.linalg.norm(Sref) < 2e-5


@pytest.mark.parametrize("task,kind", [("tracking", "periodic"), ("tracking", "sin"), ("tracking_slow", "drag"), ("tracking", "mixed")])
def test_batched_step_with_disturbance_models_equals_replicas(task, kind):
    """covo_mpc_step_batched with the table-driven disturbance models: every instance's rollout and Hessian tables are built
    inside the batched graph from that instance's state, raw controller key and (domain-randomised, quadrotor.py:152)
    disturb_params; drag / mixed take the per-pair Hessian with per-instance constants.  Against one covo-online controller per
    instance on the same states and keys: plans and Sigmas bit-identical (eager call, capture, replays)."""
    import covo_mpc_amd as cm
    N, E = 1024, 3
    env = cm.envs.Quad3D(task=task, obs_type="quad_params", enable_randomizer=True, disturb_type=kind,
                         disable_rollover_terminate=True, generate_noisy_state=True, device=DEV)
    inst = []
    for e in range(E):
        params = env.sample_params(cr.PRNGKey(100 + e)).replace(disturb_period=5 + 2 * e)
        controller, cp = cm.envs.get_controller(env, "covo-online", f"N{N}_H32_lam0.01", device=DEV, compute_info=False)
        obs, info, state = env.reset(cr.PRNGKey(200 + e), params)
        cp = controller.reset(state, params, controller.init_control_params, cr.PRNGKey(2))
        inst.append(dict(params=params, controller=controller, cp=cp, obs=obs, info=info, state=state, key=cr.PRNGKey(300 + e)))
    assert np.abs(np.asarray(inst[0]["params"].disturb_params) - np.asarray(inst[1]["params"].disturb_params)).max() > 1e-3
    cp0 = inst[0]["cp"]
    batched = cm.controllers.BatchedCoVOController(env, E, N, 32, 0.01, discount=cp0.discount, gamma_mean=cp0.gamma_mean,
                                                   sample_sigma=cp0.sample_sigma, a_mean_init=cp0.a_mean, device=DEV)
    batched.set_instances([i["state"] for i in inst], [i["params"] for i in inst])
    for step in range(4):
        k_acts = []
        for i in inst:
            i["key"], k_act, i["k_step"] = cr.split(i["key"], 3)
            k_acts.append(np.asarray(k_act))
        u_b = batched([i["info"]["noisy_state"] for i in inst], np.stack(k_acts)).clone()
        for e, i in enumerate(inst):
            u, i["cp"], _ = i["controller"](i["obs"], i["state"], i["params"], k_acts[e], i["cp"], i["info"])
            assert torch.equal(batched.a_cov[e], i["cp"].a_cov), (step, e)
            assert torch.equal(batched.a_mean[e].view(32, 4), i["cp"].a_mean), (step, e)
            assert torch.equal(u_b[e], u), (step, e)
            i["obs"], i["state"], _, _, i["info"] = env.step(i["k_step"], i["state"], u.cpu().numpy(), i["params"])
    assert torch.isfinite(batched.a_mean).all() and (batched.a_mean[0] - batched.a_mean[1]).abs().max() > 1e-4


@pytest.mark.parametrize("name,task,kind", [("covo-online", "tracking_slow", "periodic"), ("mppi", "tracking_zigzag", "mixed")])
def test_run_episode_with_models_equals_python_loop(name, task, kind):
    import covo_mpc_amd as cm
    env = cm.envs.Quad3D(task=task, enable_randomizer=False, disturb_type=kind, disable_rollover_terminate=True,
                         generate_noisy_state=True, device=DEV)
    params = env.default_params.replace(disturb_params=np.asarray(DP, dtype=np.float32), disturb_period=5)
    n = 16
    outs = []
    for fused in (False, True):
        controller, _ = cm.envs.get_controller(env, name, "N2048_H32_lam0.01", device=DEV, compute_info=False)
        controller.alias_outputs = True
        core = controller.core
        ep = cm.envs.DeviceEpisode(env, cr.PRNGKey(41), params, (core.lib, core.h), DEV)
        cp = controller.reset(ep.state0, params, controller.init_control_params, cr.PRNGKey(42))
        rng = cr.PRNGKey(43)
        if fused:
            cp, rng = controller.run_episode(ep, params, cp, rng, n)
        else:
            for _ in range(n):
                rng, rng_act, rng_step, rng_control = cr.split(rng, 4)
                u, cp, _ = controller(None, None, params, rng_act, cp, {"noisy_state": ep.noisy_state})
                ep.step(rng_step, u)
                rng, rng_control = cr.split(rng)
        outs.append((ep.read_log().copy(), cp.a_mean.cpu().numpy().copy(), ep.true.cpu().numpy().copy()))
    assert all(np.array_equal(x, y) for x, y in zip(outs[0], outs[1]))
    assert np.abs(outs[0][2][13:16]).max() > 1e-4  # a force is acting


def test_tracking_slow_closed_loop_sanity():
    """tracking_slow (quadratic reward, slow lissajous) closed loop on the device: the controller tracks."""
    import covo_mpc_amd as cm
    env = cm.envs.Quad3D(task="tracking_slow", enable_randomizer=False, disturb_type="periodic",
                         disable_rollover_terminate=True, generate_noisy_state=True, device=DEV)
    controller, _ = cm.envs.get_controller(env, "covo-online", "N4096_H32_lam0.01", device=DEV, compute_info=False)
    err = cm.envs.eval_env_device(env, controller, total_steps=300, num_trajs=1, verbose=False)
    assert err.shape == (1,) and err[0] < 0.25, err


# ------------------------------------------------------------------------------------------ f1 / f3 against oracle/ directly
def _oracle_state(packed, traj):
    """ref_np.State (fp64) from a packed float[32] state and the episode's (pos, vel, acc) trajectories"""
    st = np.asarray(packed, dtype=np.float32)
    return R.State(pos=st[0:3], vel=st[3:6], quat=st[6:10], omega=st[10:13], f_disturb=st[13:16], pos_tar=st[16:19],
                   vel_tar=st[19:22], acc_tar=st[22:25], time=int(st[25:26].view(np.int32)[0]), pos_traj=traj[0],
                   vel_traj=traj[1], acc_traj=traj[2]).astype(np.float64)


def _pack_oracle(s):
    x = np.zeros(32)
    x[0:3], x[3:6], x[6:10], x[10:13], x[13:16] = s.pos, s.vel, s.quat, s.omega, s.f_disturb
    x[16:19], x[19:22], x[22:25] = s.pos_tar, s.vel_tar, s.acc_tar
    return x


@pytest.mark.parametrize("task,kind,rollover", [("tracking_zigzag", "gaussian", False), ("tracking_slow", "periodic", True),
                                                ("tracking", "mixed", False)])
def test_env_step_kernel_vs_oracle_step_env(task, kind, rollover):
    """covo_env_step against oracle/ref_np.py directly (Weak 8): every step the oracle's step_env + noisy_state
    (quadrotor.py:215-263, 314-361 in fp64) advances the DEVICE's previous true state with the draws the step key gives
    (covo_mpc_amd.random: Philox, pinned by tests/test_oracle.py::test_philox_known_answers) and must land on the device's
    new true / noisy state, reward, errors and termination flag."""
    import covo_mpc_amd as cm
    env = cm.envs.Quad3D(task=task, enable_randomizer=False, disturb_type=kind, disable_rollover_terminate=not rollover,
                         generate_noisy_state=True, device=DEV)
    params = env.default_params.replace(disturb_params=np.asarray(DP, dtype=np.float32), disturb_period=6)
    p = R.Params().fp32().replace(disturb_params=DP, disturb_period=6)
    reward_fn = R.REWARD_FNS["realworld" if task == "tracking_slow" else "penyaw"]
    core = SamplingCore(256, 32, 0.01, 1.0, device=DEV)
    # auto_reset=False: the oracle's step_env has no reset (base.py's select is host plumbing, tests/test_gpu_reset.py), and the
    # rollover case terminates on purpose
    ep = cm.envs.DeviceEpisode(env, cr.PRNGKey(11), params, (core.lib, core.h), DEV, auto_reset=False)
    traj = (ep.state0.pos_traj, ep.state0.vel_traj, ep.state0.acc_traj)
    rng = np.random.default_rng(5)
    key = cr.PRNGKey(12)
    exp = []
    for t in range(40):
        before = ep.true.cpu().numpy()
        key, k_step = cr.split(key)
        u = np.clip(np.array([-0.3378, 0, 0, 0]) + 0.3 * rng.normal(size=4), -1.2, 1.2).astype(np.float32)
        if rollover and t > 22:  # late in the run: a saturated roll command tips the vehicle past 90 degrees (quat[3] < cos(pi/4))
            u[1] = 1.1
        ep.step(k_step, torch.from_numpy(u).to(DEV))
        kd, kp, kv, kq, ko = cm.envs.DeviceEpisode.leaf_keys(k_step)
        draw = cr.normal(kd, (3,)) if kind == "gaussian" else cr.uniform(kd, (3,), -p.disturb_scale, p.disturb_scale)
        s = _oracle_state(before, traj)
        nxt, r, done = R.step_env(s, u.astype(np.float64), p, R.Disturb(kind, draw.astype(np.float64)), rollover, reward_fn)
        noisy = R.noisy_state(nxt, p, cr.normal(kp, (3,)).astype(np.float64), cr.normal(kv, (3,)).astype(np.float64),
                              cr.normal(kq, (4,)).astype(np.float64), cr.normal(ko, (3,)).astype(np.float64))
        t_dev, n_dev = ep.true.cpu().numpy(), ep.noisy.cpu().numpy()
        assert t_dev[25:26].view(np.int32)[0] == nxt.time
        assert np.abs(t_dev[:25] - _pack_oracle(nxt)[:25]).max() < 2e-6, (t, np.abs(t_dev[:25] - _pack_oracle(nxt)[:25]).max())
        assert np.abs(n_dev[:25] - _pack_oracle(noisy)[:25]).max() < 2e-6, t
        exp.append((float(r), float(R.norm(s.pos_tar - s.pos)), float(R.norm(s.vel_tar - s.vel)), float(done)))
    log, exp = ep.read_log(), np.asarray(exp)
    assert np.abs(log[:, :3] - exp[:, :3]).max() < 3e-6 and np.array_equal(log[:, 3], exp[:, 3])
    if rollover:
        assert exp[:, 3].max() == 1.0  # the rollover termination fired on both sides


def test_pid_nominal_vs_oracle_pid_action():
    """covo_pid_nominal against oracle/ref_np.py::pid_action + step_env (pid.py:38-84, covo.py:58-99) directly (Weak
    8): from every device start state the oracle takes one PID-tracked, non-deterministic env step (the chain) and H
    deterministic ones (the nominal mean)."""
    import covo_mpc_amd as cm
    env = cm.envs.Quad3D(task="tracking_zigzag", enable_randomizer=False, disturb_type="gaussian",
                         disable_rollover_terminate=True, generate_noisy_state=True, device=DEV)
    controller, cp = cm.envs.get_controller(env, "covo-offline", "N1024_H32_lam0.01", device=DEV)
    params = env.default_params
    obs, info, state = env.reset(cr.PRNGKey(31), params)
    pd, ad, kd = controller._nominal_device(state, params, cr.PRNGKey(32))
    pd, ad, kd = pd.cpu().numpy(), ad.cpu().numpy(), kd.cpu().numpy().view(np.uint32)
    p = R.Params().fp32()
    traj = (state.pos_traj, state.vel_traj, state.acc_traj)
    worst_chain = worst_mean = 0.0
    for t in list(range(0, 60)) + [137, 250, 298]:
        s = _oracle_state(pd[t], traj)
        # the chain step (covo.py:80-90): rng_step, key = split(key) [PID]; rng_step, key = split(key) [env step, deterministic=False]
        _, k1 = cr.split(kd[t])
        rs, _ = cr.split(k1)
        z = cr.normal(disturb_key(rs), (3,)).astype(np.float64)
        act, _ = R.pid_action(s, p)
        nxt, _, _ = R.step_env(s, act, p, R.Disturb("gaussian", z))
        worst_chain = max(worst_chain, np.abs(pd[t + 1][:25] - _pack_oracle(nxt)[:25]).max())
        # the nominal mean (covo.py:58-76): H deterministic PID steps
        sr, mean = s, []
        for k in range(32):
            a, _ = R.pid_action(sr, p)
            mean.append(a)
            sr, _, _ = R.step_env(sr, a, p, np.zeros(3))
        worst_mean = max(worst_mean, np.abs(ad[t] - np.concatenate(mean)).max())
    assert worst_chain < 5e-6, worst_chain
    assert worst_mean < 2e-4, worst_mean  # the attitude gain amplifies fp32 rounding of the 32-step roll-out (cf. the host-env test)


# ------------------------------------------------------------------------------------------ MPPI covariance adaptation
@pytest.mark.parametrize("lam,N,gs,gm", [(1.0, 3000, 0.4, 1.0), (0.05, 8192, 1.0, 0.7), (0.01, 257, 0.25, 1.0)])
def test_mppi_covariance_adaptation_vs_oracle(lam, N, gs, gm):
    """covo_softmax_update_cov (mppi.py:109-125 with gamma_sigma != 0) against oracle/ref_np.py::softmax_update +
    mppi_cov_update (fp64) on the same costs and samples: new mean and the H 4x4 covariances about the NEW mean."""
    s, p, rng = make_problem(seed=N, time=60)
    mu = R.hover_action(p, 32, np.float64) + 0.05 * rng.normal(size=(32, 4))
    a = np.clip(mu[None] + 0.3 * rng.normal(size=(N, 32, 4)), -1, 1).astype(np.float32)
    cov_old = np.stack([(lambda B: B @ B.T * 0.05 + 0.02 * np.eye(4))(rng.normal(size=(4, 4))) for _ in range(32)]).astype(np.float32)
    core = SamplingCore(N, 32, lam, 1.0, device=DEV)
    core.a.copy_(to_stripes(a))
    cost = core.rollout(dev_state(s), params_c(p), (0.0, 0.0, 0.0), False).cpu().numpy().astype(np.float64)
    mean_d, cov_d = core.update_cov(torch.from_numpy(mu.reshape(-1).astype(np.float32)).to(DEV), gm,
                                    torch.from_numpy(cov_old).to(DEV), gs)
    mu32 = mu.astype(np.float32).astype(np.float64)
    a_new, w = R.softmax_update(cost, a.astype(np.float64), lam, gm, mu32)
    cov_ref = R.mppi_cov_update(w, a.astype(np.float64), a_new, cov_old.astype(np.float64), gs)
    assert np.abs(mean_d.cpu().numpy().reshape(32, 4) - a_new).max() < 1e-5
    assert np.abs(cov_d.cpu().numpy() - cov_ref).max() < 1e-5, np.abs(cov_d.cpu().numpy() - cov_ref).max()
    assert np.abs(cov_ref - cov_old).max() > 1e-3  # the adaptation moved it


@pytest.mark.parametrize("graph", ["graph", "eager"])
def test_mppi_step_with_covariance_adaptation(graph, monkeypatch):
    """The fused MPPI step with gamma_sigma != 0 (covariances shifted, sampled from, adapted in place) equals the
    kernel-by-kernel path, and a few closed-loop steps keep the covariances symmetric positive definite."""
    import covo_mpc_amd as cm
    monkeypatch.setenv("COVO_GRAPH" if graph == "graph" else "COVO_NO_GRAPH", "1")
    env = cm.envs.Quad3D(task="tracking_zigzag", enable_randomizer=False, disturb_type="gaussian", disable_rollover_terminate=True,
                         generate_noisy_state=True, device=DEV)
    params = env.default_params
    outs = []
    for fused in (True, False):
        controller, cp = cm.envs.get_controller(env, "mppi", "N4096_H32_lam0.5", device=DEV)
        cp = cp.replace(gamma_sigma=0.3)
        controller.materialize_eps = not fused
        obs, info, state = env.reset(cr.PRNGKey(3), params)
        key = cr.PRNGKey(6)
        for i in range(6):
            key, k_act, k_step = cr.split(key, 3)
            u, cp, _ = controller(obs, state, params, k_act, cp, info)
            obs, state, reward, done, info = env.step(k_step, state, u.cpu().numpy(), params)
        outs.append((cp.a_mean.cpu().numpy().copy(), cp.a_cov.cpu().numpy().copy()))
    assert np.abs(outs[0][0] - outs[1][0]).max() < 1e-6 and np.abs(outs[0][1] - outs[1][1]).max() < 1e-6
    cov = outs[0][1].astype(np.float64)
    assert np.abs(cov - np.transpose(cov, (0, 2, 1))).max() < 1e-7
    assert all(np.linalg.eigvalsh(c).min() > 0 for c in cov) and np.abs(cov - 0.25 * np.eye(4)).max() > 1e-3


# ------------------------------------------------------------------------------------------ BASELINE configs[4] end to end (round 4)
def _dr_env():
    import covo_mpc_amd as cm
    return cm.envs.Quad3D(task="tracking", obs_type="quad_params", enable_randomizer=True, disturb_type="gaussian",
                          disable_rollover_terminate=True, generate_noisy_state=True, device=DEV)


def _oracle_params(params):
    """ref_np.Params of one domain-randomised instance (quadrotor.py:135-160 varies m, I, action_scale, alpha_bodyrate,
    disturb_params; I does not enter the body-rate model)."""
    return R.Params(m=float(params.m), action_scale=float(params.action_scale), alpha_bodyrate=float(params.alpha_bodyrate),
                    disturb_params=tuple(float(x) for x in params.disturb_params)).fp32()


def test_env_instances_32_batched_step_vs_fp64_oracle():
    """BASELINE configs[4] at its per-GPU shape: E = 32 domain-randomised `tracking` instances x N = 4096 in
    ONE covo_mpc_step_batched call; three sampled instances are checked against the fp64 oracle run with THEIR OWN parameters:
    per-sample rollout costs (1e-5), the Sigma each sampled from (oracle hyper-dual Hessian -> eigh, 2e-5) and the new mean
    (oracle softmax of the oracle costs)."""
    import covo_mpc_amd as cm
    N, E = 4096, 32
    env = _dr_env()
    params = [env.sample_params(cr.PRNGKey(500 + e)) for e in range(E)]
    ep = None
    c0, cp0 = cm.envs.get_controller(env, "covo-online", f"N{N}_H32_lam0.01", device=DEV, compute_info=False)
    cp0 = c0.init_control_params
    b = cm.controllers.BatchedCoVOController(env, E, N, 32, 0.01, discount=cp0.discount, gamma_mean=cp0.gamma_mean,
                                             sample_sigma=cp0.sample_sigma, a_mean_init=cp0.a_mean, device=DEV)
    ep = cm.envs.BatchedDeviceEpisode(env, [cr.PRNGKey(600 + e) for e in range(E)], params, (b.core.lib, b.core.h), DEV)
    b.bind_episode(ep)
    assert np.ptp([float(p.m) for p in params]) > 1e-3  # different plants
    # a few closed-loop steps first (states off the reset point), all on the device
    rngs = np.stack([np.asarray(cr.PRNGKey(700 + e)) for e in range(E)])
    rngs = b.run_episode(ep, rngs, 4)
    torch.cuda.synchronize()
    noisy = ep.noisy.cpu().numpy().copy()
    a_mean_before = b.a_mean.cpu().numpy().copy()
    k_acts = np.stack([np.asarray(cr.split(cr.PRNGKey(800 + e))[1]) for e in range(E)])
    b(None, k_acts)  # ONE batched control step on the episode's current noisy states
    torch.cuda.synchronize()
    for e in (0, 13, 31):
        traj = (ep.states0[e].pos_traj, ep.states0[e].vel_traj, ep.states0[e].acc_traj)
        so = _oracle_state(noisy[e], traj)
        po = _oracle_params(params[e])
        a_dev = b._a[e].permute(1, 0, 2).contiguous().cpu().numpy()
        cost_ref = CO.rollout(so, po, a_dev.astype(np.float64), 1.0, np.zeros(3), dtype=np.float64)
        assert rel_err(b._cost[e].cpu().numpy(), cost_ref).max() < 1e-5, e
        am = R.shift_mean(a_mean_before[e].reshape(32, 4).astype(np.float64))
        Sref = R.optimize_sigma(CO.hessian(so, po, am.reshape(-1), 32), 0.5, 32, 4)
        S = b.a_cov[e].cpu().numpy()
        assert np.linalg.norm(S - Sref) / np.linalg.norm(Sref) < 2e-5, e
        a_ref, _ = R.softmax_update(cost_ref, a_dev.astype(np.float64), 0.01, 1.0, am)
        top2 = np.sort(cost_ref)[:2]
        err = np.abs(b.a_mean[e].view(32, 4).cpu().numpy() - a_ref).max()
        assert err < 1e-4 or (top2[1] - top2[0]) < 1e-3, (e, err)
    assert np.abs(b.a_mean[0].cpu().numpy() - b.a_mean[13].cpu().numpy()).max() > 1e-4  # different plans


def test_batched_closed_loop_sigma_tracks_lapack_on_its_own_hessians():
    """Every step of a batched closed-loop episode (E = 32 domain-randomised instances, control + env step on the device): the
    Sigma an instance sampled from against LAPACK eigh on THAT step's Hessian (covo_debug_batched_hessians: the Sigma chain's
    input), <= 1e-6 -- the early Ritz evaluations inside the batched squaring launch take lambda_min from whichever filter
    iterate passes first, step after step, on the matrices a closed loop really produces."""
    import covo_mpc_amd as cm
    N, E = 4096, 32
    env = _dr_env()
    params = [env.sample_params(cr.PRNGKey(1000 + e)) for e in range(E)]
    c0, _ = cm.envs.get_controller(env, "covo-online", f"N{N}_H32_lam0.01", device=DEV, compute_info=False)
    cp0 = c0.init_control_params
    b = cm.controllers.BatchedCoVOController(env, E, N, 32, 0.01, discount=cp0.discount, gamma_mean=cp0.gamma_mean,
                                             sample_sigma=cp0.sample_sigma, a_mean_init=cp0.a_mean, device=DEV)
    ep = cm.envs.BatchedDeviceEpisode(env, [cr.PRNGKey(5000 + e) for e in range(E)], params, (b.core.lib, b.core.h), DEV)
    rngs = np.stack([np.asarray(cr.PRNGKey(6000 + e)) for e in range(E)])
    Rh = torch.zeros((2, 128 * 128), dtype=torch.float64).pin_memory()
    sl = torch.zeros(16, dtype=torch.float64, device=DEV)
    kwins = []
    for t in range(40):
        rngs = b.run_episode(ep, rngs, 1)
        for i, e in enumerate((4, 19)):
            _lib.check(b.core.lib.covo_debug_batched_hessians(b.core.h, _lib.ptr(Rh[i]), e * 128 * 128, 128 * 128, b.core.stream()))
        _lib.check(b.core.lib.covo_debug_sigma_workspace(b.core.h, _lib.ptr(sl), 11 * E * 128 * 128 + 4 * 3712, 16, b.core.stream()))
        torch.cuda.synchronize()
        kwins.append((int(sl[7]), int(sl[8])))  # SC_KWIN, SC_SQ of instance 4
        for i, e in enumerate((4, 19)):
            ref = R.optimize_sigma(Rh[i].numpy().reshape(128, 128), 0.5, 32, 4)
            S = b.a_cov[e].cpu().numpy()
            assert np.linalg.norm(S - ref) / np.linalg.norm(ref) < 1e-6, (t, e)
    assert b.core.device_status() == 0
    assert all(2 <= k <= s <= 16 for k, s in kwins), kwins


def test_batched_env_step_vs_oracle_and_single_instance_kernel():
    """covo_env_step_batched (E = 32, one launch): every instance's step equals covo_env_step on that instance alone bit for bit,
    and three sampled instances land where oracle/ref_np.py::step_env + noisy_state (fp64, THEIR parameters, the draws their
    step keys give) put them."""
    import covo_mpc_amd as cm
    E = 32
    env = _dr_env()
    params = [env.sample_params(cr.PRNGKey(900 + e)) for e in range(E)]
    core = SamplingCore(256, 32, 0.01, 1.0, device=DEV)
    keys0 = [cr.PRNGKey(1000 + e) for e in range(E)]
    ep = cm.envs.BatchedDeviceEpisode(env, keys0, params, (core.lib, core.h), DEV)
    singles = {e: cm.envs.DeviceEpisode(env, keys0[e], params[e], (core.lib, core.h), DEV) for e in (0, 13, 31)}
    rng = np.random.default_rng(7)
    key = cr.PRNGKey(77)
    a_mean = torch.zeros((E, 128), dtype=torch.float32, device=DEV)
    for t in range(12):
        before = ep.true.cpu().numpy().copy()
        key, sub = cr.split(key)
        step_keys = np.asarray(cr.split(sub, E))
        u = np.clip(np.array([-0.3378, 0, 0, 0]) + 0.3 * rng.normal(size=(E, 4)), -1.2, 1.2).astype(np.float32)
        a_mean[:, :4] = torch.from_numpy(u).to(DEV)
        ep.step(step_keys, a_mean)
        t_dev, n_dev = ep.true.cpu().numpy(), ep.noisy.cpu().numpy()
        for e, se in singles.items():
            se.step(step_keys[e], torch.from_numpy(u[e]).to(DEV))
            assert torch.equal(se.true, ep.true[e]) and torch.equal(se.noisy, ep.noisy[e]), (t, e)
            kd, kp, kv, kq, ko = cm.envs.DeviceEpisode.leaf_keys(step_keys[e])
            p = _oracle_params(params[e])
            traj = (ep.states0[e].pos_traj, ep.states0[e].vel_traj, ep.states0[e].acc_traj)
            s = _oracle_state(before[e], traj)
            nxt, r, done = R.step_env(s, u[e].astype(np.float64), p, R.Disturb("gaussian", cr.normal(kd, (3,)).astype(np.float64)),
                                      False, R.REWARD_FNS["penyaw"])
            noisy = R.noisy_state(nxt, p, cr.normal(kp, (3,)).astype(np.float64), cr.normal(kv, (3,)).astype(np.float64),
                                  cr.normal(kq, (4,)).astype(np.float64), cr.normal(ko, (3,)).astype(np.float64))
            assert np.abs(t_dev[e, :25] - _pack_oracle(nxt)[:25]).max() < 2e-6, (t, e)
            assert np.abs(n_dev[e, :25] - _pack_oracle(noisy)[:25]).max() < 2e-6, (t, e)
    log = ep.read_log()
    assert log.shape == (E, 12, 4)
    for e, se in singles.items():
        assert np.array_equal(se.read_log(), log[e])


def test_run_episode_batched_equals_per_instance_episodes():
    """covo_run_episode_batched (E instances: control step + env step, every key chain threaded on the host side of ONE C call)
    against E separate covo_run_episode runs of plain covo-online controllers with the same keys: logs, means and final states
    bit-identical per instance (the batched step is bit-identical to the single step, the batched env step to the single one)."""
    import covo_mpc_amd as cm
    N, E, n = 1024, 4, 20
    env = _dr_env()
    params = [env.sample_params(cr.PRNGKey(40 + e)) for e in range(E)]
    c0, _ = cm.envs.get_controller(env, "covo-online", f"N{N}_H32_lam0.01", device=DEV, compute_info=False)
    cp0 = c0.init_control_params
    b = cm.controllers.BatchedCoVOController(env, E, N, 32, 0.01, discount=cp0.discount, gamma_mean=cp0.gamma_mean,
                                             sample_sigma=cp0.sample_sigma, a_mean_init=cp0.a_mean, device=DEV)
    reset_keys = [cr.PRNGKey(50 + e) for e in range(E)]
    ep = cm.envs.BatchedDeviceEpisode(env, reset_keys, params, (b.core.lib, b.core.h), DEV)
    rngs0 = np.stack([np.asarray(cr.PRNGKey(60 + e)) for e in range(E)])
    rngs = b.run_episode(ep, rngs0, n // 2)
    rngs = b.run_episode(ep, rngs, n - n // 2)  # two segments: the chains and the log rows continue
    log = ep.read_log()
    assert log.shape == (E, n, 4) and np.all(log[:, 1:, 1] > 0)  # (row 0: the reset state sits on its target)
    for e in range(E):
        c, _ = cm.envs.get_controller(env, "covo-online", f"N{N}_H32_lam0.01", device=DEV, compute_info=False)
        c.alias_outputs = True
        se = cm.envs.DeviceEpisode(env, reset_keys[e], params[e], (c.core.lib, c.core.h), DEV)
        cp = c.reset(se.state0, params[e], c.init_control_params, cr.PRNGKey(2))
        cp, rng = c.run_episode(se, params[e], cp, rngs0[e], n)
        assert np.array_equal(se.read_log(), log[e]), e
        assert torch.equal(cp.a_mean.reshape(-1), b.a_mean[e]) and torch.equal(se.true, ep.true[e]), e
        assert np.array_equal(np.asarray(rng, dtype=np.uint32), rngs[e]), e
    # the driver: parameters, reset keys and key chains drawn from one seed
    err = cm.envs.eval_env_batched(env, 3, f"N{N}_H32_lam0.01", n_steps=30, seed=5, device=DEV, verbose=False)
    assert err.shape == (3,) and np.all(np.isfinite(err)) and np.all(err < 0.5)
