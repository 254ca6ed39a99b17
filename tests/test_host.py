"""CPU: host-side logic -- sharding, PRNG keys, env plumbing, PID, and the world_size-2 exchange."""
import os
import subprocess
import sys

import numpy as np
import pytest

from covo_mpc_amd import random as cr
from covo_mpc_amd.controllers._core import shard_range
from covo_mpc_amd.envs import Quad3D

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_range():
    assert shard_range(65536, 3, 8) == (3 * 8192, 8192)
    assert [shard_range(64, r, 4)[0] for r in range(4)] == [0, 16, 32, 48]
    with pytest.raises(ValueError):
        shard_range(100, 0, 8)


def test_prng_split_is_deterministic_and_distinct():
    k = cr.PRNGKey(1)
    a, b = cr.split(k)
    assert not np.array_equal(a, b) and np.array_equal(cr.split(k)[0], a)
    z = cr.normal(a, (20000,))
    assert abs(z.mean()) < 0.03 and abs(z.std() - 1) < 0.03
    u = cr.uniform(b, (20000,), -0.2, 0.2)
    assert u.min() >= -0.2 and u.max() <= 0.2 and abs(u.mean()) < 0.01
    # the host philox equals the oracle's philox (and hence the device kernel's)
    from oracle import rng_np
    r = cr._philox([1, 2], [3, 4], [5, 6], [7, 8], 9, 10)
    r2 = rng_np.philox4x32_10([1, 2], [3, 4], [5, 6], [7, 8], 9, 10)
    assert all(np.array_equal(x, y) for x, y in zip(r, r2))


def test_env_plumbing_matches_oracle_step():
    """Quad3D.step_env (host plumbing) against the oracle's literal restatement on the same inputs."""
    from oracle import ref_np as R
    env = Quad3D(task="tracking_zigzag", enable_randomizer=False, disturb_type="none",
                 disable_rollover_terminate=True, generate_noisy_state=True)
    p = env.default_params
    obs, info, st = env.reset(cr.PRNGKey(5), p)
    assert obs.shape == (env.obs_dim,) == (49,) and st.pos_traj.shape == (320, 3) and st.time == 0
    assert np.all(np.abs(st.f_disturb) <= 0.2) and info["noisy_state"] is not None
    so = R.State(pos=st.pos.astype(np.float64), vel=st.vel.astype(np.float64), quat=st.quat.astype(np.float64),
                 omega=st.omega.astype(np.float64), f_disturb=st.f_disturb.astype(np.float64),
                 pos_tar=st.pos_tar.astype(np.float64), vel_tar=st.vel_tar.astype(np.float64), acc_tar=np.zeros(3),
                 time=0, pos_traj=st.pos_traj.astype(np.float64), vel_traj=st.vel_traj.astype(np.float64),
                 acc_traj=st.acc_traj.astype(np.float64))
    key = cr.PRNGKey(6)
    for i in range(5):
        a = np.array([0.1 * i - 0.3, 0.2, -0.1, 0.05 * i], dtype=np.float32)
        key, k = cr.split(key)
        _, st, r, d, info = env.step_env(k, st, a, p)
        so, ro, do = R.step_env(so, a.astype(np.float64), R.Params(), np.zeros(3))
        assert abs(r - ro) < 1e-5 and d == bool(do)
        assert np.abs(st.pos - so.pos).max() < 1e-6 and np.abs(st.quat - so.quat).max() < 1e-6
        assert np.abs(st.vel - so.vel).max() < 1e-5 and np.abs(st.omega - so.omega).max() < 1e-5
        assert np.array_equal(st.pos_tar, st.pos_traj[st.time])
    packed = info["noisy_state"].pack()
    assert packed.shape == (32,) and packed[25:26].view(np.int32)[0] == st.time


def test_pid_tracks_hover_and_get_controller_errors():
    import covo_mpc_amd as cm
    env = Quad3D(task="hovering", enable_randomizer=False, disturb_type="none", disable_rollover_terminate=True)
    pid = cm.controllers.PIDController(env, cm.controllers.PIDParams(Kp=10.0, Kd=5.0, Ki=0.0, Kp_att=10.0))
    obs, info, st = env.reset(cr.PRNGKey(2), env.default_params)
    cp, key = pid.init_control_params, cr.PRNGKey(3)
    for _ in range(150):
        a, cp, _ = pid(obs, st, env.default_params, key, cp)
        key, k = cr.split(key)
        obs, st, r, d, info = env.step(k, st, a, env.default_params)
    assert info["err_pos"] < 0.2 and not d
    with pytest.raises(NotImplementedError):
        Quad3D(task="hover")  # README's name is not a valid task (quadrotor.py:83-84)
    with pytest.raises(NotImplementedError):
        cm.envs.get_controller(env, "lqr")


def test_zigzag_generator_shape_and_speed():
    from covo_mpc_amd.dynamics import utils
    pos, vel, acc = utils.generate_zigzag_traj(300, 0.02, cr.PRNGKey(11))
    assert pos.shape == vel.shape == acc.shape == (320, 3) and np.all(pos[0] == 0) and np.all(acc == 0)
    seg_len = np.linalg.norm(pos[40] - pos[0])
    assert 0.99 <= seg_len <= 1.51
    assert np.allclose(np.linalg.norm(vel[0]), seg_len / 41 / 0.02, rtol=1e-5)  # divides by point_per_seg + 1


@pytest.mark.parametrize("world,kind", [(2, "plain"), (8, "plain"), (8, "cov")])
def test_world_size_N_gloo_exchange(world, kind):
    """`world` CPU processes (gloo; 8 = the ranks of one MI355X node): each reduces its shard (oracle arithmetic) to ONE rank
    record -- softmax partial + position sums, "cov": + MPPI's second moments (mppi.py:119-125) --, the product's
    exchange_records makes all records known to all ranks, every rank merges device-free -> identical to the unsharded update
    on every rank (tests/_dist_worker.py)."""
    script = os.path.join(ROOT, "tests", "_dist_worker.py")
    port = str(29533 + world + (1 if kind == "cov" else 0))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port, PYTHONPATH=ROOT, OMP_NUM_THREADS="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
                        "--master-addr", "127.0.0.1", "--master-port", port, script, kind],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert f"DIST_OK {world} {kind}" in r.stdout


def test_env_reward_and_disturbance_model_reach_the_c_params():
    """The kernels evaluate env.reward_fn and the env's disturbance model themselves: which ones must travel in struct
    covo_env_params (quadrotor.py:35,49-89), and an env bound to a reward the kernels do not know must be refused instead of
    letting the controller plan against penyaw while env.step pays something else (round-2 VERDICT, Weak 2)."""
    import types
    import covo_mpc_amd as cm
    from covo_mpc_amd import _lib
    from covo_mpc_amd.controllers.base import BaseController, env_reward_kind
    p = cm.dynamics.EnvParams3D(disturb_params=np.arange(6, dtype=np.float32) / 10)
    for task, kind in (("tracking", 0), ("tracking_zigzag", 0), ("hovering", 0), ("tracking_slow", 1)):
        for dt in ("none", "gaussian", "periodic", "sin", "drag", "mixed"):
            env = cm.envs.Quad3D(task=task, enable_randomizer=False, disturb_type=dt, disable_rollover_terminate=True)
            c = BaseController(env, None)._params_c(p)
            assert c.reward_kind == kind and c.disturb_kind == _lib.DISTURB_KINDS[dt]
            assert c.disturb_period == 50 and abs(c.disturb_scale - 0.2) < 1e-7 and abs(c.dyn_noise_scale - 0.05) < 1e-7
            assert np.allclose(list(c.disturb_params), np.arange(6) / 10)
    env = cm.envs.Quad3D(task="tracking_slow", enable_randomizer=False, disturb_type="none")
    assert env_reward_kind(env) == "realworld"
    env.reward_fn = lambda state, params=None: 0.0  # a user-supplied reward: no kernel evaluates it
    cp = types.SimpleNamespace(discount=1.0)
    with pytest.raises(NotImplementedError, match="reward_fn"):
        cm.controllers.CoVOController(env, cp, 1024, 32, 0.01, mode="online")  # refused before any device work
    with pytest.raises(NotImplementedError, match="reward_fn"):
        cm.controllers.MPPIController(env, cp, 1024, 32, 0.01)
    with pytest.raises(NotImplementedError, match="reward_fn"):
        BaseController(env, None)._params_c(p)


def test_rollover_flag_reaches_the_c_params():
    import covo_mpc_amd as cm
    p = cm.dynamics.EnvParams3D()
    assert p.to_c().rollover_terminate == 0 and p.to_c(rollover_terminate=True).rollover_terminate == 1
    from covo_mpc_amd.controllers.base import BaseController
    for off in (True, False):
        env = cm.envs.Quad3D(task="hovering", enable_randomizer=False, disturb_type="none", disable_rollover_terminate=off)
        assert BaseController(env, None)._params_c(p).rollover_terminate == (0 if off else 1)


def test_jax_bitstream_known_answers():
    """covo_mpc_amd/random_jax.py restates jax.random's threefry stream.  jax is not installable here; what
    IS published pins it: the three Random123 threefry2x32-20 vectors (the ones jax's own tests use) and the values jax's
    documentation prints for PRNGKey(0) / split / uniform / normal."""
    from covo_mpc_amd import random_jax as rj
    for key, ctr, exp in [((0, 0), (0, 0), (0x6B200159, 0x99BA4EFE)),
                          ((0xFFFFFFFF, 0xFFFFFFFF), (0xFFFFFFFF, 0xFFFFFFFF), (0x1CB996FC, 0xBB002BE7)),
                          ((0x13198A2E, 0x03707344), (0x243F6A88, 0x85A308D3), (0xC4923A9C, 0x483DF7A0))]:
        y0, y1 = rj.threefry2x32(key[0], key[1], [ctr[0]], [ctr[1]])
        assert (int(y0[0]), int(y1[0])) == exp
    k = rj.PRNGKey(0)
    assert k.tolist() == [0, 0]
    assert rj.split(k).tolist() == [[4146024105, 967050713], [2718843009, 1272950319]]   # jax docs, "JAX PRNG design"
    assert abs(float(rj.uniform(k)) - 0.41845703) < 1e-8
    assert abs(float(rj.normal(k)) - (-0.20584226)) < 2e-8                               # "Sharp bits": random.normal(key)
    quick = [-0.3721109, 0.26423115, -0.18252768, -0.7368197, -0.44030377, -0.1521442, -0.67135346, -0.5908641, 0.73168886,
             0.5673026]                                                                  # jax quickstart: normal(PRNGKey(0), (10,))
    assert np.abs(rj.normal(k, (10,)) - np.array(quick, dtype=np.float32)).max() < 1e-7
    assert abs(float(rj.normal(rj.PRNGKey(42))) - (-0.18471177)) < 2e-8
    # layout: an odd count is padded with one zero counter and cut again
    assert np.array_equal(rj.random_bits(k, 3), rj._threefry_counts(k, [0, 1, 2]))
    e = rj.controller_epsilon(rj.PRNGKey(7), 6, n=128, sample_offset=2, n_samples=3)
    assert e.shape == (3, 128) and np.array_equal(e[0], rj.normal(rj.split(rj.PRNGKey(7), 6)[2], (128,)))


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` WITHOUT a torchrun environment must run two ranks (the parent starts
    torch.distributed.run as a child before any GPU call and relays rank 0's line).  No GPU here: --rendezvous-only runs
    everything of the N-rank bench except the device work (process group, barrier bracket, max over ranks, one line)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["COVO_BENCH_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--rendezvous-only"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["rendezvous_only"] is True
    # a failing rank's exit code comes back through the launcher
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=300)  # no GPU here: the ranks' assert fires
    assert r.returncode != 0


def test_bench_spread_indices_cover_the_episode():
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    assert b.spread_indices(20, 300) == [7 + 15 * i for i in range(20)]
    assert b.spread_indices(300, 300) == list(range(300)) and b.spread_indices(5, 300) == [30, 90, 150, 210, 270]
    idx = b.spread_indices(200, 300)
    assert len(idx) == 200 and idx[0] == 0 and idx[-1] == 299 and all(b2 - a2 in (1, 2) for a2, b2 in zip(idx, idx[1:]))
    assert b.spread_indices(650, 300)[:301] == list(range(300)) + [0] and b.spread_indices(0, 300) == []
