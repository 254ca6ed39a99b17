"""GPU tests (-m gpu) of the device env's AUTO-RESET (VERDICT r05 "Missing 2"): BaseEnvironment.step selects reset_env(key_reset)
whenever the pre-step state is terminal (quadjax/envs/base.py:22-40) and eval_env steps through it (quadjax/envs/quadrotor.py:
531-538).  env_step.hip / env_reset.hpp do the same on the device -- new trajectory (all four generators of
quadjax/dynamics/utils.py:49-251), zero state, f_disturb ~ U(-scale, scale), noisy copy from reset_env's own info key, the
controller's state carried on -- through covo_env_step, covo_env_step_batched, covo_run_episode and covo_run_episode_batched.
The checker is the Python env (covo_mpc_amd/envs: the host mirror of the reference's env, itself checked against oracle/ in
tests/test_gpu_models.py), whose `step` performs the reference's select on the host.
Tolerances: trajectories <= 1 fp32 ulp (fp64 sin / cos / acos / atan2 of ocml against libm, rounded to fp32; the count of
elements that differ is asserted to be tiny), reset state / noise / log bit-exact on equal trajectories, stepped states 2e-5
(as tests/test_gpu_parity.py::test_env_step_kernel_vs_host_env)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
if not torch.cuda.is_available():
    pytest.skip("needs the MI355X", allow_module_level=True)

from covo_mpc_amd import random as cr  # noqa: E402
from covo_mpc_amd.controllers._core import SamplingCore  # noqa: E402
from tests.test_gpu_parity import DEV  # noqa: E402

f32 = np.float32


def _ulps(a, b):
    """max distance in fp32 ulps between two float32 arrays (0 where both are 0)"""
    a, b = np.asarray(a, dtype=f32), np.asarray(b, dtype=f32)
    scale = np.spacing(np.maximum(np.abs(a), np.abs(b)).astype(f32))
    d = np.abs(a.astype(np.float64) - b.astype(np.float64)) / scale
    return float(np.max(np.where(a == b, 0.0, d)))


def _push_out(ep, state, pos, vel):
    """put the true state (device and host copy) at `pos` with velocity `vel`"""
    import dataclasses
    st = dataclasses.replace(state, pos=np.asarray(pos, dtype=f32), vel=np.asarray(vel, dtype=f32))
    ep.true.copy_(torch.from_numpy(st.pack()).to(ep.true.device))
    return st


@pytest.mark.parametrize("task,kind", [("tracking_zigzag", "gaussian"), ("tracking", "periodic"), ("tracking_slow", "none"),
                                       ("hovering", "gaussian")])
def test_device_reset_vs_host_reset_env(task, kind):
    """One covo_env_step on a state outside the 3 m box: the device stores what the Python env's step returns for it -- the
    reset_env(key_reset) state, its noisy copy, its trajectory -- and logs {reward of the terminal state, the reset state's
    errors, 1}.  With auto_reset=False (COVO_TRAJ_NONE) the old behaviour stays: done is logged and the state flies on."""
    import covo_mpc_amd as cm
    env = cm.envs.Quad3D(task=task, enable_randomizer=False, disturb_type=kind, disable_rollover_terminate=True,
                         generate_noisy_state=True, device=DEV)
    params = env.default_params.replace(disturb_scale=0.3)
    core = SamplingCore(256, 32, 0.01, 1.0, device=DEV)
    differing = 0
    for trial in range(6):
        ep = cm.envs.DeviceEpisode(env, cr.PRNGKey(11 + trial), params, (core.lib, core.h), DEV)
        obs, info, state = env.reset(cr.PRNGKey(11 + trial), params)
        state = _push_out(ep, state, [0.2, -3.2 - 0.1 * trial, 0.4], [0.1, -1.0, 0.0])
        k_step = cr.split(cr.PRNGKey(500 + trial))[1]
        u = np.array([-0.3, 0.1, -0.1, 0.05], dtype=f32)
        ep.step(k_step, torch.from_numpy(u).to(DEV))
        obs, st_h, reward, done, info = env.step(k_step, state, u, params)
        assert done and st_h.time == 0
        traj_d = [t.cpu().numpy() for t in (ep.pos_traj, ep.vel_traj, ep.acc_traj)]
        traj_h = [st_h.pos_traj, st_h.vel_traj, st_h.acc_traj]
        for d, h in zip(traj_d, traj_h):
            assert d.shape == h.shape and _ulps(d, h) <= 1.0, (trial, _ulps(d, h))
            differing += int(np.sum(d != h))
        t_dev, n_dev = ep.true.cpu().numpy(), ep.noisy.cpu().numpy()
        want = st_h.pack()
        # zero state, unit quaternion, time 0 and the uniform disturbance: exact; the targets = row 0 of the device's trajectory
        assert np.array_equal(t_dev[:16], want[:16]) and np.array_equal(t_dev[25:], want[25:]), trial
        assert np.abs(t_dev[13:16]).max() > 0 and np.abs(t_dev[13:16]).max() <= 0.3
        assert np.array_equal(t_dev[16:25], np.concatenate([traj_d[0][0], traj_d[1][0], traj_d[2][0]]))
        # the noisy copy: reset_env's own draws (quadrotor.py:365-366) on the device's reset state, bit for bit
        n_host = info["noisy_state"].pack()
        assert np.array_equal(n_dev[:16], n_host[:16]) and np.array_equal(n_dev[16:], t_dev[16:]), trial
        assert np.abs(n_dev[:13] - t_dev[:13]).max() > 1e-4
        log = ep.read_log()
        assert log.shape == (1, 4) and log[0, 3] == 1.0
        assert abs(log[0, 0] - reward) < 2e-5 * max(1.0, abs(reward))
        assert abs(log[0, 1] - info["err_pos"]) < 1e-6 and abs(log[0, 2] - info["err_vel"]) < 1e-6  # errors of the RESET state
        if task != "hovering" and trial == 0:
            assert np.abs(traj_d[0]).max() > 0.1 and not np.array_equal(traj_d[0], ep.state0.pos_traj)  # a NEW trajectory
    assert differing <= 2, differing  # fp32 elements that differ by one ulp, of 6 trials x 3 x T x 3
    # auto_reset=False: pre-round-6 behaviour
    ep = cm.envs.DeviceEpisode(env, cr.PRNGKey(11), params, (core.lib, core.h), DEV, auto_reset=False)
    obs, info, state = env.reset(cr.PRNGKey(11), params)
    state = _push_out(ep, state, [0.2, -3.2, 0.4], [0.1, -1.0, 0.0])
    ep.step(cr.PRNGKey(9), torch.zeros(4, device=DEV))
    t_dev = ep.true.cpu().numpy()
    assert ep.read_log()[0, 3] == 1.0 and t_dev[1] < -3.2 and t_dev[25:26].view(np.int32)[0] == 1
    assert np.array_equal(ep.pos_traj.cpu().numpy(), ep.state0.pos_traj)


def test_env_step_through_reset_vs_host_env():
    """A scripted action sequence from 5 cm inside the box with outward velocity: device env step against the Python env's
    auto-resetting step, before, at and after the reset (same step on both sides; the time field restarts at 0)."""
    import covo_mpc_amd as cm
    env = cm.envs.Quad3D(task="tracking_zigzag", enable_randomizer=False, disturb_type="gaussian",
                         disable_rollover_terminate=True, generate_noisy_state=True, device=DEV)
    params = env.default_params
    core = SamplingCore(256, 32, 0.01, 1.0, device=DEV)
    ep = cm.envs.DeviceEpisode(env, cr.PRNGKey(21), params, (core.lib, core.h), DEV)
    obs, info, state = env.reset(cr.PRNGKey(21), params)
    state = _push_out(ep, state, [2.95, 0.0, 0.0], [2.0, 0.0, 0.0])
    rng = np.random.default_rng(3)
    key = cr.PRNGKey(22)
    dones, errs = [], []
    for t in range(30):
        key, k_step = cr.split(key)
        u = np.clip(np.array([-0.3378, 0, 0, 0]) + 0.2 * rng.normal(size=4), -1, 1).astype(f32)
        ep.step(k_step, torch.from_numpy(u).to(DEV))
        obs, state, reward, done, info = env.step(k_step, state, u, params)
        dones.append(done)
        errs.append(info["err_pos"])
        t_dev, n_dev = ep.true.cpu().numpy(), ep.noisy.cpu().numpy()
        assert t_dev[25:26].view(np.int32)[0] == state.time, t
        assert np.abs(t_dev - state.pack()).max() < 2e-5, (t, np.abs(t_dev - state.pack()).max())
        assert np.abs(n_dev - info["noisy_state"].pack()).max() < 2e-5, t
    log = ep.read_log()
    assert sum(dones) == 1 and dones.index(True) in (1, 2, 3)
    assert np.array_equal(log[:, 3] == 1.0, np.asarray(dones))
    assert np.abs(log[:, 1] - np.asarray(errs)).max() < 2e-5
    assert _ulps(ep.pos_traj.cpu().numpy(), state.pos_traj) <= 1.0 and _ulps(ep.vel_traj.cpu().numpy(), state.vel_traj) <= 1.0
    assert state.time == 30 - dones.index(True) - 1


@pytest.mark.parametrize("name", ["covo-online", "mppi", "covo-offline"])
def test_run_episode_through_reset_equals_python_loop(name):
    """covo_run_episode (control + env step enqueued from C) through an auto-reset against the Python loop over
    controller.__call__ + DeviceEpisode.step: same log (one done row), same trajectories after the reset, same mean, same rng
    -- and the state after the reset is the Python env's reset_env(key_reset) state (checked on the step it happens)."""
    import covo_mpc_amd as cm
    env = cm.envs.Quad3D(task="tracking_zigzag", enable_randomizer=False, disturb_type="gaussian",
                         disable_rollover_terminate=True, generate_noisy_state=True, device=DEV)
    params = env.default_params
    n = 14
    outs = []
    for fused in (False, True):
        controller, _ = cm.envs.get_controller(env, name, "N2048_H32_lam0.01", device=DEV, compute_info=False)
        controller.alias_outputs = True
        core = controller.core
        ep = cm.envs.DeviceEpisode(env, cr.PRNGKey(41), params, (core.lib, core.h), DEV)
        cp = controller.reset(ep.state0, params, controller.init_control_params, cr.PRNGKey(42))
        st = _push_out(ep, ep.state0, [0.0, 0.0, 2.95], [0.0, 0.0, 2.5])
        ep.noisy.copy_(ep.true)
        rng = cr.PRNGKey(43)
        if fused:
            cp, rng = controller.run_episode(ep, params, cp, rng, n)
        else:
            for t in range(n):
                rng, rng_act, rng_step, rng_control = cr.split(rng, 4)
                u, cp, _ = controller(None, None, params, rng_act, cp, {"noisy_state": ep.noisy_state})
                before = ep.true.cpu().numpy().copy()
                ep.step(rng_step, u)
                if np.abs(before[:3]).max() > 3.0:  # this step resets: the Python env's select on the same pre-step state
                    import dataclasses
                    pre = dataclasses.replace(st, pos=before[0:3], vel=before[3:6], quat=before[6:10], omega=before[10:13],
                                              f_disturb=before[13:16], time=int(before[25:26].view(np.int32)[0]))
                    _, st_h, _, done, info = env.step(rng_step, pre, u.cpu().numpy(), params)
                    assert done
                    t_dev = ep.true.cpu().numpy()
                    assert np.array_equal(t_dev[:16], st_h.pack()[:16]) and np.abs(t_dev - st_h.pack()).max() < 1e-6
                    assert np.abs(ep.noisy.cpu().numpy() - info["noisy_state"].pack()).max() < 1e-6
                rng, rng_control = cr.split(rng)
        outs.append((ep.read_log().copy(), cp.a_mean.cpu().numpy().copy(), ep.true.cpu().numpy().copy(),
                     ep.pos_traj.cpu().numpy().copy(), ep.vel_traj.cpu().numpy().copy(), np.asarray(rng).copy()))
    assert all(np.array_equal(x, y) for x, y in zip(outs[0], outs[1]))
    log = outs[0][0]
    assert log[:, 3].sum() == 1.0 and log[int(np.argmax(log[:, 3])), 1] == 0.0  # one reset; its row carries the reset state's error
    assert not np.array_equal(outs[0][3], ep.state0.pos_traj)  # flying on a new trajectory
    assert np.abs(outs[0][2][:3]).max() < 1.0  # and near it
    assert np.all(np.isfinite(outs[0][1]))


def test_run_episode_batched_through_reset_equals_per_instance_episodes():
    """covo_run_episode_batched with two of four domain-randomised instances pushed out of the box: per instance bit-identical
    to covo_run_episode on that instance alone (log with the done row, state, trajectories after the reset, mean, key chain);
    the instances that stayed inside never reset."""
    import covo_mpc_amd as cm
    N, E, n = 1024, 4, 12
    env = cm.envs.Quad3D(task="tracking", obs_type="quad_params", enable_randomizer=True, disturb_type="gaussian",
                         disable_rollover_terminate=True, generate_noisy_state=True, device=DEV)
    params = [env.sample_params(cr.PRNGKey(40 + e)) for e in range(E)]
    c0, _ = cm.envs.get_controller(env, "covo-online", f"N{N}_H32_lam0.01", device=DEV, compute_info=False)
    cp0 = c0.init_control_params
    b = cm.controllers.BatchedCoVOController(env, E, N, 32, 0.01, discount=cp0.discount, gamma_mean=cp0.gamma_mean,
                                             sample_sigma=cp0.sample_sigma, a_mean_init=cp0.a_mean, device=DEV)
    reset_keys = [cr.PRNGKey(50 + e) for e in range(E)]
    ep = cm.envs.BatchedDeviceEpisode(env, reset_keys, params, (b.core.lib, b.core.h), DEV)
    out = {1: ([2.96, 0.0, 0.0], [2.5, 0.0, 0.0]), 3: ([0.0, -2.97, 0.1], [0.0, -3.0, 0.0])}
    for e, (p, v) in out.items():
        ep.true[e, 0:3] = torch.tensor(p, device=DEV)
        ep.true[e, 3:6] = torch.tensor(v, device=DEV)
        ep.noisy[e].copy_(ep.true[e])
    traj0 = ep.pos_traj.cpu().numpy().copy()
    rngs0 = np.stack([np.asarray(cr.PRNGKey(60 + e)) for e in range(E)])
    rngs = b.run_episode(ep, rngs0, n)
    log = ep.read_log()
    assert [int(log[e, :, 3].sum()) for e in range(E)] == [0, 1, 0, 1]
    traj1 = ep.pos_traj.cpu().numpy()
    assert np.array_equal(traj1[0], traj0[0]) and np.array_equal(traj1[2], traj0[2])
    assert not np.array_equal(traj1[1], traj0[1]) and not np.array_equal(traj1[3], traj0[3])
    for e in range(E):
        c, _ = cm.envs.get_controller(env, "covo-online", f"N{N}_H32_lam0.01", device=DEV, compute_info=False)
        c.alias_outputs = True
        se = cm.envs.DeviceEpisode(env, reset_keys[e], params[e], (c.core.lib, c.core.h), DEV)
        cp = c.reset(se.state0, params[e], c.init_control_params, cr.PRNGKey(2))
        if e in out:
            se.true[0:3] = torch.tensor(out[e][0], device=DEV)
            se.true[3:6] = torch.tensor(out[e][1], device=DEV)
            se.noisy.copy_(se.true)
        cp, rng = c.run_episode(se, params[e], cp, rngs0[e], n)
        assert np.array_equal(se.read_log(), log[e]), e
        assert torch.equal(cp.a_mean.reshape(-1), b.a_mean[e]) and torch.equal(se.true, ep.true[e]), e
        assert torch.equal(se.pos_traj, ep.pos_traj[e]) and torch.equal(se.vel_traj, ep.vel_traj[e]), e
        assert np.array_equal(np.asarray(rng, dtype=np.uint32), rngs[e]), e
    assert np.abs(ep.true[:, :3].cpu().numpy()).max() < 1.5  # everybody back on a track
