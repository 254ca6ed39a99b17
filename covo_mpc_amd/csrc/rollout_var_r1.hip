// rollout_var_r1.hip -- rollout_pipe3_kernel for tracking_realworld_reward_fn (quadjax/dynamics/utils.py:297-313, task
// "tracking_slow") with every disturbance variant.  A translation unit of its own only so that the variants compile in
// parallel with rollout.hip; the kernel is rollout_pipe.hpp's.
#include "rollout_launch.hpp"

void launch_rollout_variant_r1(const RolloutArgs &A, const RolloutArgs *batch, int nb, bool batched, int groups, bool stats, hipStream_t s)
{
    if (A.fdist == 0) {
        if (batched) launch_pipe3_family<false, true, 1, 0>(A, batch, nb, groups, false, s);
        else launch_pipe3_family<false, false, 1, 0>(A, batch, nb, groups, stats, s);
    } else if (A.fdist == 1) {
        if (batched) launch_pipe3_family<false, true, 1, 1>(A, batch, nb, groups, false, s);
        else launch_pipe3_family<false, false, 1, 1>(A, batch, nb, groups, stats, s);
    } else {
        if (batched) launch_pipe3_family<false, true, 1, 2>(A, batch, nb, groups, false, s);
        else launch_pipe3_family<false, false, 1, 2>(A, batch, nb, groups, stats, s);
    }
}
