// step_small.hpp -- interface of the fused small step (step_small.hip) towards step.hip.
#pragma once
#include "rollout_common.hpp"
#include "step_begin.hpp"

struct SmallStepArgs {
    RolloutArgs R;              // the rollout's argument block (state, trajectories, work buffers, records workspace, 1 / lambda,
                                // and where the last workgroup's merge goes: the new mean, or a sharded rank's merged record)
    const float *a_mean_in;     // control_params.a_mean of this call [128]
    float *a_mean_shift_out;    // receives the shifted old mean (the rank merge of a sharded step blends with it); with dyn_mem set it
                                // already HOLDS it (the begin launch of this graph replay) and is read instead of a_mean_in
    int n_table;
    const float *L_table;       // covo-offline: [n_table][128][128] lower factors
    float *mppi_cov;            // MPPI: a_cov [H][4][4], shifted in place by the launch's last workgroup
    int64_t sample_offset;
    DynBlock blk;               // eager launches: the per-step scalars as kernel arguments (step_begin.hpp) ...
    const uint32_t *dyn_mem;    // ... captured graphs: in device memory, left there by the begin launch (else null)
    int derive_keys;
    float shared_noise_scale;
    int nanp;
};

// can the step (args, params) run as the one fused launch?
bool step_small_eligible(const covo_ctx *h, const covo_env_params &p, const covo_step_args &a);
// state: the noisy state this launch reads (args.state, or the graph's fixed-address copy); blk / dyn_mem: exactly one non-null
int launch_step_small(covo_ctx *h, const covo_env_params &p, const covo_step_args &a, const float *state, float *a_mean_shift,
                      const DynBlock *blk, const uint32_t *dyn_mem, float shared_noise_scale, unsigned *ticket, hipStream_t s);
