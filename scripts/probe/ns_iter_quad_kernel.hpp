// ns_iter_quad_kernel.hpp -- NOT part of the library: round 6's experiment on VERDICT r05 Next 3 (batched Newton-Schulz), removed
// after it measured SLOWER.  One Newton-Schulz iteration of a batch as ONE launch: ns_T_quad_kernel's 16 blocks per matrix produce T
// (write-through stores + one flag per block), ns_YZ_quad_kernel's 32 blocks wait for the four flags of the panel of T they read
// inside the same launch (all producers of the launch ahead of all consumers in dispatch order).  Bit-identical to the two
// launches (batch 3 / 32 / 300, every matrix <= 1e-6 from LAPACK) -- and, same box, bench.py --config envs, 3 x alternating:
//   two launches per iteration: 87 287 / 87 609 / 87 378 control-steps/s, Sigma 238.3 / 239.7 / 239.0 us per batched step
//   one launch per iteration:   83 841 / 83 923 / 84 384,                 Sigma 254.6 / 254.2 / 254.9
// rocprofv3: the fused launch 18.56 us against 7.32 + 10.2 for the pair.  Why: at 32 matrices neither launch is latency-bound --
// 512 + 1 024 workgroups, each pulling 64 KB of operands through its XCD's L2 and issuing 32 fp64 MFMAs of 64 cycles per wave: the
// pair's 17.5 us IS its L2 + matrix-pipe time (plain T accesses instead of sc1, or no wait at all, change nothing: 232.5 / 233.7
// against 234.2 us on one box), the 1 536 workgroups of the fused launch exceed what is resident (LDS: 4 per CU) and run in two
// rounds, and the one launch boundary it saves (~2 us) is less than what the spinning consumers cost the producers.  With the
// matrix-major id order of ns_block the last matrices' producers queued behind the first ones' spinning consumers: 289 us.
// What would pay instead: fewer operand bytes per product (bigger blocks) and a software-pipelined tile loop -- a different kernel.
// The code as it was (it needs sigma_ns.hip's QuadOps, quad_load, quad_mma_reduce, ns_converged, store_both, gld / gst):

// "the 2 x 2 tile block q of T_iter is in memory" (ns_iter_quad_kernel): 16 flags per iteration, kept in the unused tail of the
// squaring norms' slot rows (SC_SQN row r holds 36 partials and, in slot 63, t_r); cleared by the chain's first squaring
__host__ __device__ constexpr int ns_tflag_slot(int iter, int quad) { return SC_SQN + iter * 64 + 40 + quad; }
static_assert(NS_ITERS <= NS_SQUARINGS + 1 && 40 + 16 <= 63, "T flags fit the norm rows");

// ---- one Newton-Schulz iteration of a BATCH in ONE launch (round 6): ns_T_quad_kernel's 16 blocks and ns_YZ_quad_kernel's 32 of
// every matrix, the consumers waiting for the blocks of T they read instead of for a launch boundary.  Per batched iteration the
// pair cost 7.3 + 10.2 us + a boundary at 32 matrices, at 26 % / 38 % of the fp64 MFMA rate (profiles/r05_bench_envs_pmc_summary.json):
// launch ramp, first-touch operand latency and the end-of-kernel write-back, twice.  Here workgroups w < 16 of a matrix are the
// producers -- ns_T_quad_kernel's body, T and T^T stored write-through (agent-scope relaxed atomics: sc1) and, once acknowledged, one
// flag per block --, workgroups 16 .. 47 the consumers -- ns_YZ_quad_kernel's body: the operand that is NOT T (Y^T rows / Z columns) is
// requested first, then the four flags of the column panel (Y' = Y.T) / row panel (Z' = T.Z) of T are polled by four lanes, then T
// is read with sc1 loads.  Same tiles, same K-split, same sums: bit-identical to the two launches.  Forward progress: the
// consumers wait only for workgroups with SMALLER linear ids of the same launch (every producer's id is below every consumer's),
// which the dispatcher has started before them; the wait is bounded anyway (0.2 s -> SC_BARFAIL -> NaN Sigma).
template <int COH = COH_NONE, class F>
__device__ __forceinline__ void quad_load_a(QuadOps &o, const double *A, int mi, int lane, int kq, F f)
{
    const int lo = lane & 15, hi = lane >> 4;
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
        const int k = 32 * kq + 4 * kk + hi;
#pragma unroll
        for (int h = 0; h < 2; ++h) o.a[h][kk] = f(gld<COH>(A + (size_t)k * SN + 32 * mi + 16 * h + lo), k, 32 * mi + 16 * h + lo);
    }
}
template <int COH = COH_NONE, class F>
__device__ __forceinline__ void quad_load_b(QuadOps &o, const double *B, int mj, int lane, int kq, F f)
{
    const int lo = lane & 15, hi = lane >> 4;
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
        const int k = 32 * kq + 4 * kk + hi;
#pragma unroll
        for (int h = 0; h < 2; ++h) o.b[h][kk] = f(gld<COH>(B + (size_t)k * SN + 32 * mj + 16 * h + lo), k, 32 * mj + 16 * h + lo);
    }
}
#ifndef NS_POLL_SLEEP
#define NS_POLL_SLEEP 1
#endif
constexpr int NS_ITERQ_WG = 48;  // per matrix: 16 producers (T) + 16 (Y') + 16 (Z')
__global__ __launch_bounds__(256) void ns_iter_quad_kernel(const double *__restrict__ Yall, const double *__restrict__ Ytall,
                                                           const double *__restrict__ Zall, const double *__restrict__ Ztall,
                                                           double *Tall, double *Ttall, double *__restrict__ Yout,
                                                           double *__restrict__ Ytout, double *__restrict__ Zout,
                                                           double *__restrict__ Ztout, double *scall, int iter, int zbuf_out, int batch, int dbg)
{
    __shared__ double redq[4][4][4][64];
    __shared__ double partq[4][4];
    __shared__ int wait_ok;
    // Linear ids: ALL producers of the launch first, then all consumers (the dispatcher starts workgroups in id order and 1 536 of
    // them at 32 matrices are more than fit: with ns_block's matrix-major order the last matrices' producers queued behind the first
    // ones' spinning consumers and the launch took two rounds -- 289 against 240 us of Sigma per batched step).  Inside either
    // region ns_block's rule: the matrix is 8 (slot / per-matrix count) + (id & 7), so that all workgroups of a matrix share an XCD.
    int b, w;
    {
        const unsigned id = blockIdx.y * gridDim.x + blockIdx.x, nprod = 16u * gridDim.y;  // gridDim.y = matrices rounded up to 8
        if (id < nprod) {
            const unsigned slot = id >> 3;
            b = (int)(slot / 16u) * 8 + (int)(id & 7u);
            w = (int)(slot % 16u);
        } else {
            const unsigned idc = id - nprod, slot = idc >> 3;
            b = (int)(slot / 32u) * 8 + (int)(idc & 7u);
            w = 16 + (int)(slot % 32u);
        }
        if (b >= batch) return;
    }
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    double *s = scall + (size_t)b * SC_COUNT;
    const size_t off = (size_t)b * SN * SN;
    if (s[SC_NS_DONE] != 0.0) return;  // (decided a launch ago: every workgroup of the matrix reads the same value)
    QuadOps ops;
    if (w < 16) {
        // ================================================================ producer: block (mi, mj) of T = a I + b Z.Y (ns_T_quad_kernel)
        const int mi = w >> 2, mj = w & 3;
        quad_load(ops, Ztall + off, Yall + off, mi, mj, lane, wv, LoadPlain{});
        const double a = s[SC_COEF + 2 * iter], bq = s[SC_COEF + 2 * iter + 1];
        if (ns_converged<COH_NONE>(s, iter, lane, w == 0 && tid == 0)) return;  // (the consumers take the same decision from the same slots)
        double p[4];
        quad_mma_reduce(ops, redq, wv, lane, p);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int ti = 2 * mi + (t >> 1), tj = 2 * mj + (t & 1);
            const int row = 16 * ti + (lane >> 4) + 4 * wv, col = 16 * tj + (lane & 15);
            if (dbg & 2) store_both<COH_NONE>(Tall + off, Ttall + off, row, col, fma(bq, p[t], (row == col) ? a : 0.0)); else
            store_both<COH_AGENT>(Tall + off, Ttall + off, row, col, fma(bq, p[t], (row == col) ? a : 0.0));
            const double d = p[t] - ((row == col) ? 1.0 : 0.0);
            const double ws = wr::wave64_allsum(d * d);
            if (lane == 0) partq[t][wv] = ws;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this thread's write-through stores have been acknowledged
        __syncthreads();
        if (tid == 0) gst<COH_AGENT>(s + ns_tflag_slot(iter, w), 1.0);
        if (tid < 4) {
            const int ti = 2 * mi + (tid >> 1), tj = 2 * mj + (tid & 1);
            s[SC_ERR + iter * 64 + ti * 8 + tj] = (partq[tid][0] + partq[tid][1]) + (partq[tid][2] + partq[tid][3]);
        }
        return;
    }
    // ==================================================================== consumer: block (mi, mj) of Y' = Y.T or Z' = T.Z (ns_YZ_quad_kernel)
    const int wx = w - 16;
    const bool isZ = wx >= 16;
    const int q = wx & 15, mi = q >> 2, mj = q & 3;
    if (isZ) quad_load_b(ops, Zall + off, mj, lane, wv, LoadPlain{});
    else quad_load_a(ops, Ytall + off, mi, lane, wv, LoadPlain{});
    if (ns_converged<COH_NONE>(s, iter, lane, false)) return;
    if (wx == 0 && tid == 0) {
        s[SC_ZBUF] = (double)zbuf_out;
        s[SC_ITERS] = (double)(iter + 1);
    }
    if (dbg & 1) { if (tid == 0) wait_ok = 1; } else
    if (tid < 64) {
        // Y' block: the column panel of T = its blocks (r, mj); Z' block: the row panel = blocks (mi, r); lane r < 4 polls one flag
        const int need = isZ ? 4 * mi + (lane & 3) : 4 * (lane & 3) + mj;
        const double *flag = s + ns_tflag_slot(iter, need);
        const long long t0 = wall_clock64();
        int good = 1;
        for (;;) {
            const double v = (lane < 4) ? gld<COH_AGENT>(flag) : 1.0;
            if (__builtin_amdgcn_ballot_w64(v == 0.0) == 0) break;
            __builtin_amdgcn_s_sleep(NS_POLL_SLEEP);
            if (wall_clock64() - t0 > 20000000LL) {  // 0.2 s: never silently -- the finalize launch turns this into NaN outputs
                good = 0;
                if (lane == 0) gst<COH_AGENT>(s + SC_BARFAIL, 1.0);
                break;
            }
        }
        if (lane == 0) wait_ok = good;
    }
    __syncthreads();
    if (!wait_ok) return;
    if (dbg & 2) {
        if (isZ) quad_load_a<COH_NONE>(ops, Ttall + off, mi, lane, wv, LoadPlain{});
        else quad_load_b<COH_NONE>(ops, Tall + off, mj, lane, wv, LoadPlain{});
    } else
    if (isZ) quad_load_a<COH_AGENT>(ops, Ttall + off, mi, lane, wv, LoadPlain{});
    else quad_load_b<COH_AGENT>(ops, Tall + off, mj, lane, wv, LoadPlain{});
    double v[4];
    quad_mma_reduce(ops, redq, wv, lane, v);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int ti = 2 * mi + (t >> 1), tj = 2 * mj + (t & 1);
        const int row = 16 * ti + (lane >> 4) + 4 * wv, col = 16 * tj + (lane & 15);
        store_both((isZ ? Zout : Yout) + off, (isZ ? Ztout : Ytout) + off, row, col, v[t]);
    }
}

