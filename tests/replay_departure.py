"""Replay of a DEPARTING domain-randomised instance of BASELINE configs[4] through the oracle (Next 1).

Config [4]'s `tracking` episodes under domain randomisation leave their track now and then (24 of 3 072 episodes above 0.3 m
mean error; the instance on profiles/r05_bench_envs_line.json ended 23 m away).  Build or algorithm?  This runs the bench's own
closed loop (32 instances x N = 4096, seeds as bench.py --config envs; auto-reset off so that the departure can be followed),
finds the instance that leaves, and replays ITS control steps one by one through oracle/ (test infrastructure: the C fp64
rollout and hyper-dual Hessian, numpy eigh-based optimize_sigma and softmax update -- covo.py:116-132, 134-185, 227-278),
teacher-forced on the device's inputs of that step: the noisy state, the mean that went in and the N sampled action sequences.
Per step: rollout costs (<= 1e-5 of max(|cost|, 1) against fp64 -- or 1.5 x what the fp32 C oracle itself loses against fp64
on the same samples where that is more: |pos| ~ 3 m; samples whose trajectory passes within 1e-5 m of the 3 m box, where the
cost is discontinuous, set aside and counted), the Sigma it sampled from (<= 2e-5), the new mean (<= 2e-5 against the oracle's
softmax of the device's costs; against the oracle's own costs <= 1e-4, or a near-tie, or within the 1 / lambda amplification of
the cost difference).  Verdict: every step up to the departure within those bars => the ALGORITHM leaves the track on that
plant, not the build.

    python -m tests.replay_departure [--out gpurun_out/departure.txt] [--window 60]     (on the MI355X)
"""
import argparse
import json
import sys

import numpy as np


def find_and_replay(window=60, out=None, seeds=(1000, 5000, 6000), E=32, N=4096, verbose=True, max_instances=1):
    import torch
    import covo_mpc_amd as cm
    from covo_mpc_amd import random as cr
    from oracle import c_oracle as CO
    from oracle import ref_np as R
    from tests.test_gpu_models import _oracle_params, _oracle_state
    dev = "cuda:0"
    lam = 0.01
    env = cm.envs.Quad3D(task="tracking", obs_type="quad_params", enable_randomizer=True, disturb_type="gaussian",
                         disable_rollover_terminate=True, generate_noisy_state=True, device=dev)
    c0, _ = cm.envs.get_controller(env, "covo-online", f"N{N}_H32_lam{lam}", device=dev, compute_info=False)
    cp0 = c0.init_control_params
    c0.core.close()
    params = [env.sample_params(cr.PRNGKey(seeds[0] + g)) for g in range(E)]

    def fresh():
        b = cm.controllers.BatchedCoVOController(env, E, N, 32, lam, discount=cp0.discount, gamma_mean=cp0.gamma_mean,
                                                 sample_sigma=cp0.sample_sigma, a_mean_init=cp0.a_mean, device=dev)
        ep = cm.envs.BatchedDeviceEpisode(env, [cr.PRNGKey(seeds[1] + g) for g in range(E)], params, (b.core.lib, b.core.h), dev,
                                          auto_reset=False)
        rngs = np.stack([np.asarray(cr.PRNGKey(seeds[2] + g)) for g in range(E)])
        return b, ep, rngs

    T = params[0].max_steps_in_episode
    # pass 1: the whole episode on the device; who leaves, and when
    b, ep, rngs = fresh()
    b.run_episode(ep, rngs, T)
    log = ep.read_log()
    b.core.close()
    err = log[:, :, 1]
    leavers = [int(e) for e in np.argsort(-err.mean(axis=1)) if err[e].mean() > 0.3][:max_instances]
    lines = [f"closed loop, {E} instances x N={N}, seeds {seeds}, auto-reset off: mean err_pos per instance (m) "
             f"median {np.median(err.mean(axis=1)):.4f} max {err.mean(axis=1).max():.3f}; instances above 0.3 m: "
             f"{[int(e) for e in np.nonzero(err.mean(axis=1) > 0.3)[0]]}"]
    report = {"seeds": list(seeds), "E": E, "N": N, "leavers": leavers, "instances": []}
    if not leavers:
        lines.append("no instance leaves its track with these seeds")
    for e in leavers:
        t_dep = int(np.argmax(err[e] > 0.3))          # first step above 0.3 m
        t_box = int(np.argmax(log[e, :, 3] > 0)) if log[e, :, 3].any() else -1
        t_end = min(T, max(t_dep + 10, t_box + 3))  # through the departure, up to the exit from the box
        t_beg = max(0, t_end - window)
        p = params[e]
        lines.append(f"instance {e}: m={float(p.m):.5f} action_scale={float(p.action_scale):.4f} alpha_bodyrate="
                     f"{float(p.alpha_bodyrate):.4f}; err_pos first above 0.3 m at step {t_dep}, leaves the 3 m box at step {t_box}; "
                     f"replaying steps {t_beg}..{t_end - 1} through the oracle")
        lines.append(" step  err_pos  |pos|max  cost_rel   Sigma_rel  a_mean_err  (on dev costs)  top2_gap  ESS    u_dev[0]  u_ref[0]   lam_min(R)  verdict")
        # pass 2: the same episode step by step (bit-identical: same kernels, same keys), oracle on instance e in the window
        b, ep, rngs = fresh()
        traj = (ep.states0[e].pos_traj, ep.states0[e].vel_traj, ep.states0[e].acc_traj)
        po = _oracle_params(p)
        rows, ok_all = [], True
        for t in range(t_end):
            noisy = ep.noisy[e].cpu().numpy().copy()
            am_in = b.a_mean[e].cpu().numpy().copy()
            rngs = b.run_episode(ep, rngs, 1)
            if t < t_beg:
                continue
            torch.cuda.synchronize()
            so = _oracle_state(noisy, traj)
            a_dev = b._a[e].permute(1, 0, 2).contiguous().cpu().numpy().astype(np.float64)
            cost_dev = b._cost[e].cpu().numpy()
            cost_ref = CO.rollout(so, po, a_dev, 1.0, np.zeros(3), dtype=np.float64)
            rel = np.abs(cost_dev - cost_ref) / np.maximum(np.abs(cost_ref), 1.0)
            note = ""
            bar = 1e-5
            if rel.max() >= 1e-5:
                # The cost is DISCONTINUOUS where a rollout touches the 3 m box (is_terminal, quadrotor.py:484: the rewards freeze
                # from that step on): a sample whose fp64 trajectory passes within 1e-5 m of the boundary may terminate a step
                # earlier or later in fp32.  Such samples are set aside (counted), and the fp32 C oracle -- the reference's own
                # arithmetic type -- is shown beside the fp64 one for the rest.
                _, poses = CO.rollout(so, po, a_dev, 1.0, np.zeros(3), dtype=np.float64, want_poses=True)
                dist = np.abs(np.abs(poses).max(axis=2) - 3.0).min(axis=0)           # (N,)
                dist = np.minimum(dist, abs(np.abs(so.pos).max() - 3.0))
                edge = dist < 1e-5
                c32 = CO.rollout(so.astype(np.float32) if hasattr(so, "astype") else so, po, a_dev.astype(np.float32), 1.0,
                                 np.zeros(3), dtype=np.float32)
                rel32 = np.abs(c32 - cost_ref) / np.maximum(np.abs(cost_ref), 1.0)
                relo = np.abs(cost_dev - c32) / np.maximum(np.abs(cost_ref), 1.0)
                i = int(np.argmax(np.where(edge, 0.0, rel)))
                note = (f"  [{int((rel >= 1e-5).sum())} samples >= 1e-5, {int((edge & (rel >= 1e-5)).sum())} of them within 1e-5 m of "
                        f"the box boundary; rest: max {np.where(edge, 0.0, rel).max():.2e} (sample {i}: dev {cost_dev[i]:.6f} "
                        f"fp64 {cost_ref[i]:.6f} fp32-oracle {c32[i]:.6f}); fp32 oracle vs fp64 max {np.where(edge, 0.0, rel32).max():.2e}, "
                        f"dev vs fp32 oracle max {np.where(edge, 0.0, relo).max():.2e}]")
                rel = np.where(edge, 0.0, rel)
                # far from the origin fp32 itself is the limit (|pos| ~ 3 m: half an ulp of a position is 1.2e-7 m, the reward's
                # slope in the position error is 8 / m, 32 steps): the bar is the larger of 1e-5 and 1.5 x what the reference's
                # own arithmetic type loses on the same samples
                bar = max(1e-5, 1.5 * float(np.where(edge, 0.0, rel32).max()))
            cost_rel = float(rel.max())
            am = R.shift_mean(am_in.reshape(32, 4).astype(np.float64))
            Rm = CO.hessian(so, po, am.reshape(-1), 32)
            Sref = R.optimize_sigma(Rm, 0.5, 32, 4)
            S = b.a_cov[e].cpu().numpy()
            s_rel = float(np.linalg.norm(S - Sref) / np.linalg.norm(Sref))
            a_ref, w = R.softmax_update(cost_ref, a_dev, lam, 1.0, am)
            a_err = float(np.abs(b.a_mean[e].view(32, 4).cpu().numpy() - a_ref).max())
            # the update alone: the oracle's softmax on the DEVICE's costs (the 1 / lambda = 100 amplification of the cost
            # differences taken out)
            a_ref_dc, _ = R.softmax_update(cost_dev.astype(np.float64), a_dev, lam, 1.0, am)
            a_err_dc = float(np.abs(b.a_mean[e].view(32, 4).cpu().numpy() - a_ref_dc).max())
            top2 = np.sort(cost_ref)[:2]
            gap = float(top2[1] - top2[0])
            ess = float(1.0 / np.sum(np.asarray(w) ** 2))
            ok = cost_rel < bar and s_rel < 2e-5 and a_err_dc < 2e-5 and (a_err < 1e-4 or gap < 1e-3 or a_err < 400 * cost_rel)
            ok_all &= ok
            true = ep.true[e].cpu().numpy()
            row = dict(step=t, err_pos=float(err[e, t]), pos_max=float(np.abs(true[:3]).max()), cost_rel=cost_rel, sigma_rel=s_rel,
                       a_mean_err=a_err, a_mean_err_on_device_costs=a_err_dc, cost_bar=bar, top2_gap=gap, ess=ess, u_dev=float(b.a_mean[e, 0]), u_ref=float(a_ref[0, 0]),
                       lam_min=float(np.linalg.eigvalsh(Rm)[0]), ok=bool(ok))
            rows.append(row)
            lines.append(f"{t:5d}  {row['err_pos']:7.4f}  {row['pos_max']:7.3f}  {cost_rel:9.2e}  {s_rel:9.2e}  {a_err:9.2e}  {a_err_dc:9.2e}       {gap:8.2e}  "
                         f"{ess:5.2f}  {row['u_dev']:+.5f}  {row['u_ref']:+.5f}  {row['lam_min']:+.4e}  {'ok' if ok else 'MISMATCH'}{note}")
        b.core.close()
        lines.append(f"instance {e}: {'every replayed step within the bars: the algorithm leaves the track, not the build' if ok_all else 'MISMATCH: see rows'}")
        report["instances"].append(dict(instance=e, m=float(p.m), action_scale=float(p.action_scale),
                                        alpha_bodyrate=float(p.alpha_bodyrate), t_departure=t_dep, t_box=t_box, all_ok=bool(ok_all),
                                        rows=rows))
    text = "\n".join(lines)
    if verbose:
        print(text)
    if out:
        with open(out, "w") as f:
            f.write(text + "\n")
        with open(out.rsplit(".", 1)[0] + ".json", "w") as f:
            json.dump(report, f)
    return report


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--window", type=int, default=60)
    ap.add_argument("--seeds", type=int, nargs=3, default=[1000, 5000, 6000])
    ap.add_argument("--max-instances", type=int, default=1)
    a = ap.parse_args()
    rep = find_and_replay(window=a.window, out=a.out, seeds=tuple(a.seeds), max_instances=a.max_instances)
    sys.exit(0 if all(i["all_ok"] for i in rep["instances"]) else 1)
