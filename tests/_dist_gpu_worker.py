"""world_size-2 worker for tests/test_gpu_parity.py::test_two_ranks_one_gpu: two processes share cuda:0 and run the
sample-sharded controller (covo_mpc_step with partial_out -> all-gather of the 132-float records -> covo_merge) over
gloo; every rank checks the sharded result against an unsharded controller fed the same keys."""
import os
import sys

os.environ["COVO_SHARED_DEVICE"] = "1"  # two ranks share cuda:0: no launch may rely on co-resident workgroups (_core.py)

import numpy as np
import torch
import torch.distributed as dist


def main():
    name = sys.argv[1]
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    import covo_mpc_amd as cm
    from covo_mpc_amd import random as cr
    task = "hovering" if name == "mppi" else "tracking_zigzag"
    env = cm.envs.Quad3D(task=task, enable_randomizer=False, disturb_type="gaussian", disable_rollover_terminate=True,
                         generate_noisy_state=True, device="cuda:0")
    N = 4096
    cs, cps = cm.envs.get_controller(env, name, f"N{N}_H32_lam0.01", device="cuda:0", process_group=dist.group.WORLD)
    c1, cp1 = cm.envs.get_controller(env, name, f"N{N}_H32_lam0.01", device="cuda:0")
    assert cs.core.n_local == N // world and cs.core.offset == rank * (N // world)
    params = env.default_params
    obs, info, state = env.reset(cr.PRNGKey(4), params)
    cps = cs.reset(state, params, cs.init_control_params, cr.PRNGKey(5))
    cp1 = c1.reset(state, params, c1.init_control_params, cr.PRNGKey(5))
    key = cr.PRNGKey(6)
    for step in range(5):  # eager, capture, replays
        key, k_act, k_step = cr.split(key, 3)
        us, cps, _ = cs(obs, state, params, k_act, cps, info)
        u1, cp1, _ = c1(obs, state, params, k_act, cp1, info)
        err = (cps.a_mean - cp1.a_mean).abs().max().item()
        assert err < 2e-6, (name, rank, step, err)  # online-softmax merge: fp32 reassociation only
        obs, state, reward, done, info = env.step(k_step, state, u1.cpu().numpy(), params)
    out = [None] * world
    dist.all_gather_object(out, cps.a_mean.cpu().numpy().tobytes())
    assert out[0] == out[1]  # every rank merges identically
    dist.barrier()
    if rank == 0:
        print("DIST_GPU_OK")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
