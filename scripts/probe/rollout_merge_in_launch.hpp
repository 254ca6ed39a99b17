// rollout_merge_in_launch.hpp -- NOT part of the library (round 6: removed from the product).
// Round 5's experiment: the softmax update's merge finished INSIDE the record-leaving rollout launch by the workgroup that takes
// the last ticket, instead of by merge_kernel as a launch of its own.  Bit-identical (shared merge_body, softmax_merge.hpp) and
// SLOWER at every size on the MI355X (covo-online N = 65 536: 5 005 against 5 070 steps/s; covo-offline N = 8 192 staged: 32.2k
// against 38.6k): every workgroup's tail gains a write-through acknowledgement and an agent-scope atomic (two fabric round trips)
// and the last one pulls the records through sc1 loads -- more than the ~2 us launch boundary + 2.4-4.3 us merge_kernel it
// replaces.  The one-launch small step (step_small.hip) keeps its own last-ticket merge: there it removes a launch from a
// host-bound path.  What the product carried for it (tree at commit 3b7ab14 and before): RolloutArgs::{merge_ticket, merge_out,
// merge_mean_old, merge_gamma, merge_final}, struct RolloutMerge + launch_rollout(..., merge), step.hip's g_merge_in_rollout /
// COVO_MERGE_IN_ROLLOUT / covo_debug_set_merge_in_rollout, the call at the end of rollout_pipe3_kernel, and
// tests/test_gpu_parity.py::test_merge_inside_the_rollout_launch_equals_the_merge_launch.  The device function, for the record:

// The update's second stage inside the launch that left the records: every workgroup takes a ticket once its (coherently stored)
// record is acknowledged; the one that takes the last merges all gridDim.x records.  atomicInc wraps to 0 at the last arrival:
// the counter re-arms itself for the next launch.  Called by every thread of every workgroup (barriers); THREADS = blockDim.x.
template <int THREADS>
__device__ __forceinline__ void rollout_merge_last(const RolloutArgs &A, MergeLds &M, int &last_flag)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) last_flag = (atomicInc(A.merge_ticket, gridDim.x - 1) == gridDim.x - 1) ? 1 : 0;
    __syncthreads();
    if (!last_flag) return;
    if (A.merge_final) merge_body<THREADS, true, true>(A.records, (int)gridDim.x, A.inv_lam, A.merge_mean_old, A.merge_gamma, A.merge_out, COVO_PARTIAL_FLOATS, M);
    else merge_body<THREADS, false, true>(A.records, (int)gridDim.x, A.inv_lam, nullptr, 1.0f, A.merge_out, COVO_PARTIAL_FLOATS, M);
}

