import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from covo_mpc_amd.envs import quadrotor as Q
name = sys.argv[1]
env = Q.Quad3D(task="tracking_zigzag", obs_type="quad", lower_controller="base", enable_randomizer=False,
               disturb_type="gaussian", disable_rollover_terminate=True, generate_noisy_state=True, device="cuda")
ctrl, cp = Q.get_controller(env, name, "N8192_H32_lam0.01")
np.set_printoptions(precision=3, linewidth=200)
allerr = []
for seed in range(1, 13):
    errs = Q.eval_env_device(env, controller=ctrl, total_steps=300*4*10, seed=seed, verbose=False)
    allerr.append(errs)
    print(name, "seed", seed, "mean %.3f med %.3f max %.3f  n>0.1: %d" % (errs.mean(), np.median(errs), errs.max(), (errs > 0.1).sum()), flush=True)
allerr = np.concatenate(allerr)
print(name, "TOTAL n", len(allerr), "crashes(>0.3)", (allerr > 0.3).sum(), "outliers(>0.06)", (allerr > 0.06).sum(), "median %.4f" % np.median(allerr), "mean-noncrash %.4f" % allerr[allerr<0.3].mean())
