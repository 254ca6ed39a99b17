"""world_size-2 worker for tests/test_gpu_parity.py::test_two_ranks_one_gpu: two processes share cuda:0 and run the
sample-sharded controller (covo_mpc_step writing this shard's rank record -> ONE exchange of the 516-float records -> covo_merge_ranks);
every rank checks the sharded result (new mean AND covo.py:281's pos_mean / pos_std, which ride in the same record) against an
unsharded controller fed the same keys.  argv: controller name, exchange ("collective": all-gather over gloo, staged through
the host; "peer": direct writes into hipIpc-mapped buffers, csrc/exchange.hip -- then also a whole sharded episode segment
enqueued from C by covo_run_episode).  Controller "mppi-cov": MPPI with gamma_sigma = 0.3 (mppi.py:119-125) -- the 836-float
rank records with the weighted second moments; the adapted covariances are compared too.  "mppi-cov0": the same core (836-float
records) stepping with gamma_sigma = 0 -- covo_merge_ranks_wide must read the records at THEIR stride."""
import os
import sys

NCCL = len(sys.argv) > 3 and sys.argv[3] == "nccl"  # one GPU per rank, RCCL (only on boxes with >= 2 GPUs)
if not NCCL:
    os.environ["COVO_SHARED_DEVICE"] = "1"  # two ranks share cuda:0: no launch may rely on co-resident workgroups (_core.py)

import numpy as np
import torch
import torch.distributed as dist


def main():
    name, exchange = sys.argv[1], sys.argv[2]
    inject = exchange == "auto_fail"  # rank 1 cannot export its exchange buffer: every rank must fall back, nobody may hang
    coarse = exchange == "auto_coarse"  # coarse-grained exchange buffers on (pretended) different devices: connect must refuse
    if inject:
        exchange = "auto"
        os.environ["COVO_DEBUG_FAIL_EXCHANGE_RANK"] = "1"
    if coarse:
        exchange = "auto"
        os.environ["COVO_DEBUG_EXCHANGE_COARSE"] = "1"
        os.environ["COVO_DEBUG_EXCHANGE_BUS_ID"] = "fake:%s" % os.environ.get("RANK", "0")
    if exchange == "collective":
        os.environ.pop("COVO_EXCHANGE", None)  # the DEFAULT must be the collective (the peer path is opt-in)
    else:
        os.environ["COVO_EXCHANGE"] = exchange
    DEV = "cuda:0"
    if NCCL:
        DEV = "cuda:%d" % int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(DEV)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=torch.device(DEV))
    else:
        dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    import covo_mpc_amd as cm
    from covo_mpc_amd import random as cr
    task = "hovering" if name.startswith("mppi") else "tracking_zigzag"
    env = cm.envs.Quad3D(task=task, enable_randomizer=False, disturb_type="gaussian", disable_rollover_terminate=True,
                         generate_noisy_state=True, device=DEV)
    N = 4096
    cov = name == "mppi-cov"
    cov0 = name == "mppi-cov0"  # a core BUILT for the covariance records (836 floats) whose steps run with gamma_sigma == 0
    if cov or cov0:  # quadjax's factory fixes gamma_sigma = 0 (quadrotor.py:715): built directly, lam = 0.5 so that many samples carry weight
        _, cp0 = cm.envs.get_controller(env, "mppi", f"N{N}_H32_lam0.5", device=DEV)
        cp0 = cp0.replace(gamma_sigma=0.3)
        cs = cm.controllers.MPPIController(env=env, control_params=cp0, N=N, H=32, lam=0.5, device=DEV, process_group=dist.group.WORLD)
        c1 = cm.controllers.MPPIController(env=env, control_params=cp0, N=N, H=32, lam=0.5, device=DEV)
        cps = cp1 = cp0
        assert cs.core.cov_records and cs.core.rec_floats == 836
        if cov0:
            cs.init_control_params = c1.init_control_params = cp0.replace(gamma_sigma=0.0)
    else:
        cs, cps = cm.envs.get_controller(env, name, f"N{N}_H32_lam0.01", device=DEV, process_group=dist.group.WORLD)
        c1, cp1 = cm.envs.get_controller(env, name, f"N{N}_H32_lam0.01", device=DEV)
    # "auto": the construction-time self-test picks the peer path when it works on every rank (here: two ranks, one GPU)
    want_exchange = ("collective" if (inject or coarse) else "peer") if exchange == "auto" else exchange
    assert cs.core.shared_device == (not NCCL)  # physical identity (PCI bus id), not the per-process device index
    assert cs.core.n_local == N // world and cs.core.offset == rank * (N // world) and cs.core.exchange == want_exchange
    assert cs.core.device_status() == 0
    params = env.default_params
    obs, info, state = env.reset(cr.PRNGKey(4), params)
    cps = cs.reset(state, params, cs.init_control_params, cr.PRNGKey(5))  # (built with gamma_sigma = 0.3 for "mppi-cov")
    cp1 = c1.reset(state, params, c1.init_control_params, cr.PRNGKey(5))
    key = cr.PRNGKey(6)
    for step in range(5):  # eager, capture, replays
        key, k_act, k_step = cr.split(key, 3)
        us, cps, infs = cs(obs, state, params, k_act, cps, info)
        u1, cp1, inf1 = c1(obs, state, params, k_act, cp1, info)
        err = (cps.a_mean - cp1.a_mean).abs().max().item()
        # online-softmax merge: fp32 reassociation only; "mppi-cov" (lam = 0.5: thousands of samples carry weight, and the adapted
        # covariances feed the next step's draws) accumulates it over the steps
        assert err < (1e-5 if (cov or cov0) else 2e-6), (name, rank, step, err)  # (cov0: lam = 0.5 as well)
        if cov0:  # the merge read {m, s, v} and the position sums at the 836-float stride; a_cov only shifts
            assert float(cps.gamma_sigma) == 0.0 and (cps.a_cov - cp1.a_cov).abs().max().item() == 0.0
        if cov:
            ec = (cps.a_cov - cp1.a_cov).abs().max().item()
            assert ec < 1e-5 and (cps.a_cov - 0.25 * torch.eye(4, device=DEV)).abs().max().item() > 1e-3, (rank, step, ec)
        for k in ("pos_mean", "pos_std"):  # the shards' position sums travelled in the rank records (no second collective)
            e = (infs[k] - inf1[k]).abs().max().item()
            assert e < 2e-6, (name, rank, step, k, e)
        obs, state, reward, done, info = env.step(k_step, state, u1.cpu().numpy(), params)
    out = [None] * world
    dist.all_gather_object(out, cps.a_mean.cpu().numpy().tobytes())
    assert out[0] == out[1]  # every rank merges identically
    if want_exchange == "peer":
        # a sharded episode segment enqueued by ONE C call per rank (covo_run_episode: step -> peer-write exchange -> merge ->
        # env step, no host in between) against the unsharded controller's
        n = 12
        logs = []
        for ctrl in (cs, c1):
            ctrl.alias_outputs = True
            ep = cm.envs.DeviceEpisode(env, cr.PRNGKey(41), params, (ctrl.core.lib, ctrl.core.h), DEV)
            cp = ctrl.reset(ep.state0, params, ctrl.init_control_params, cr.PRNGKey(42))
            cp, rng = ctrl.run_episode(ep, params, cp, cr.PRNGKey(43), n)
            logs.append((ep.read_log().copy(), cp.a_mean.cpu().numpy().copy(), ep.true.cpu().numpy().copy()))
        assert logs[0][0].shape == (n, 4)
        tol_mean = 1e-3 if cov else 1e-4  # 12 closed-loop steps; "mppi-cov": dense weights + adapted covariances feed back
        assert np.abs(logs[0][0] - logs[1][0]).max() < 1e-4 and np.abs(logs[0][1] - logs[1][1]).max() < tol_mean, \
            (np.abs(logs[0][0] - logs[1][0]).max(), np.abs(logs[0][1] - logs[1][1]).max())
        out = [None] * world
        dist.all_gather_object(out, logs[0][2].tobytes())
        assert out[0] == out[1]  # both ranks drove their replicated env to the same state
    dist.barrier()
    if rank == 0:
        print("DIST_GPU_OK")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
