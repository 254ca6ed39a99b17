// rollout_var_r0.hip -- rollout_pipe3_kernel for tracking_penyaw_reward_fn with the per-step (periodic, sin) and per-sample
// (drag, mixed) disturbance models of quadjax/dynamics/free.py:10-58.  A translation unit of its own only so that the
// variants compile in parallel with rollout.hip; the kernel is rollout_pipe.hpp's.
#include "rollout_launch.hpp"

void launch_rollout_variant_r0(const RolloutArgs &A, const RolloutArgs *batch, int nb, bool batched, int groups, bool stats, hipStream_t s)
{
    if (batched) {  // env-batched step: instance y's table rides in its argument block
        if (A.fdist == 1) launch_pipe3_family<false, true, 0, 1>(A, batch, nb, groups, false, s);
        else launch_pipe3_family<false, true, 0, 2>(A, batch, nb, groups, false, s);
    } else if (A.fdist == 1) {
        launch_pipe3_family<false, false, 0, 1>(A, batch, nb, groups, stats, s);
    } else {
        launch_pipe3_family<false, false, 0, 2>(A, batch, nb, groups, stats, s);
    }
}
