// xcd_chain_probe.hip -- dependent 128^3 fp64 GEMM phases inside ONE persistent launch whose workgroups all sit on one XCD.
// Inside an XCD every CU shares the L2, so a phase boundary needs no cross-XCD coherence: plain (write-through-to-L2)
// stores, s_waitcnt, a counter that lives in that L2, and loads that skip the per-CU L1.  Question: what does a phase cost
// that way, against 2.6-3.2 us for a separate launch and 2.9-3.7 us for the coherent cross-XCD barrier (barrier_probe)?
// 256 workgroups are launched (one per CU); each reads its XCC id and takes a ticket on that XCD; the first XCD to hand out
// `nw` tickets becomes the leader (by pigeonhole some XCD receives >= 32 of 256), everybody else leaves.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/probe/xcd_chain_probe.hip -o scripts/probe/xcd_chain_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double f64x4 __attribute__((ext_vector_type(4)));
constexpr int SN = 128;

// LD: 0 plain, 1 agent-scope relaxed atomic (sc1), 2 nontemporal, 3 sc0 only (workgroup scope: skips the CU's L1? -- the check tells)
// Round 4 adds the L2-LOCAL variants: plain loads from a buffer no wave of this launch has touched before (FRESH: every phase
// writes a new 128 KB slab, so the L1s cannot hold a stale line of it), sc0 loads, and a counter whose atomics carry no sc1.
template <int LD>
__device__ __forceinline__ double ld(const double *p)
{
    if (LD == 1 || LD == 4) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (LD == 2) return __builtin_nontemporal_load(p);
    if (LD == 3) {
        double v;
        asm volatile("global_load_dwordx2 %0, %1, off sc0" : "=v"(v) : "v"(p) : "memory");
        return v;
    }
    return *p;
}

template <int LD>
__device__ __forceinline__ void tile(const double *__restrict__ A, double *__restrict__ C, int t, double (*red)[4][64])
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, lo = lane & 15, hi = lane >> 4;
    const int ti = t >> 3, tj = t & 7;
    if (LD == 5) asm volatile("buffer_inv sc1" ::: "memory");  // the CU's L1 (and the L2's non-coherent lines) forget; the XCD's L2 serves
    if (LD == 6) asm volatile("buffer_inv sc0" ::: "memory");
    double a[8], b[8];
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
        const int k = 32 * wv + 4 * kk + hi;
        a[kk] = ld<LD>(A + (size_t)k * SN + 16 * ti + lo);
        b[kk] = ld<LD>(A + (size_t)k * SN + 16 * tj + lo);
    }
    if (LD == 3) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) asm volatile("" : "+v"(a[kk]), "+v"(b[kk]));
    }
    f64x4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[kk], b[kk], acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wv][r][lane] = acc[r];
    __syncthreads();
    const double v = (red[0][wv][lane] + red[1][wv][lane]) + (red[2][wv][lane] + red[3][wv][lane]);
    if (LD == 4) __hip_atomic_store(C + (size_t)(16 * ti + hi + 4 * wv) * SN + 16 * tj + lo, v * 1e-2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else C[(size_t)(16 * ti + hi + 4 * wv) * SN + 16 * tj + lo] = v * 1e-2;
    __syncthreads();
}

// SCOPE: 1 agent-scope HIP atomics on one counter (what sigma_ns.hip's tail launches use);
//        0 the same counter through inline-asm read-modify-writes WITHOUT sc1 (hipcc turns fetch_add(0) into a load and gives
//          agent and workgroup RMWs the same encoding, so this is spelled out): do they execute in the XCD's own L2?
//        2 no atomics: every workgroup stores the phase number into its own flag word, wave 0 polls all nw words with ONE
//          sc0 load;  3 the same with an sc1 load
template <int SCOPE, int SLEEP>
__device__ __forceinline__ bool xcd_barrier(unsigned *ctr, unsigned *flags, unsigned phase, int rank, int nw)
{
    __shared__ int ok;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this thread's stores have reached the L2
    __syncthreads();
    const unsigned target = (unsigned)nw * phase;
    if (SCOPE >= 2) {
        if (threadIdx.x < 64) {
            const int lane = threadIdx.x;
            if (lane == 0) {
                if (SCOPE == 3) asm volatile("global_store_dword %0, %1, off sc1\n s_waitcnt vmcnt(0)" ::"v"(flags + rank), "v"(phase) : "memory");
                else asm volatile("global_store_dword %0, %1, off\n s_waitcnt vmcnt(0)" ::"v"(flags + rank), "v"(phase) : "memory");
            }
            const long long t0 = wall_clock64();
            int good = 1;
            for (;;) {
                unsigned v = phase;
                if (lane < nw) {
                    if (SCOPE == 2) asm volatile("global_load_dword %0, %1, off sc0\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(flags + lane) : "memory");
                    else asm volatile("global_load_dword %0, %1, off sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(flags + lane) : "memory");
                }
                if (__builtin_amdgcn_ballot_w64(v < phase) == 0) break;
                __builtin_amdgcn_s_sleep(SLEEP);
                if (wall_clock64() - t0 > 20000000LL) { good = 0; break; }
            }
            if (lane == 0) ok = good;
        }
        __syncthreads();
        return ok != 0;
    }
    if (threadIdx.x == 0) {
        if (SCOPE == 0) asm volatile("global_atomic_add %0, %1, off\n s_waitcnt vmcnt(0)" ::"v"(ctr), "v"(1u) : "memory");
        else __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const long long t0 = wall_clock64();
        int good = 1;
        for (;;) {
            unsigned v;
            if (SCOPE == 0) asm volatile("global_atomic_add %0, %1, %2, off sc0\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(ctr), "v"(0u) : "memory");
            else v = __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (v >= target) break;
            __builtin_amdgcn_s_sleep(SLEEP);
            if (wall_clock64() - t0 > 20000000LL) { good = 0; break; }
        }
        ok = good;
    }
    __syncthreads();
    return ok != 0;
}

struct Ctl {
    unsigned tickets[8];
    int leader;       // -1 until decided
    unsigned bar;     // phase barrier of the leader's workgroups
    unsigned failed;
    unsigned pad[5];
    unsigned flags[64];
};

template <int LD, int SCOPE, int SLEEP, bool FRESH = false>
__global__ __launch_bounds__(256) void persist_k(double *b0, double *b1, Ctl *ctl, int phases, int nw, int fixed)
{
    __shared__ double red[4][4][64];
    __shared__ int s_rank;
    if (fixed) {  // no election: the linear workgroup id round-robins over the XCDs, ids = 0 (mod 8) stay
        if (blockIdx.x & 7) return;
        if (threadIdx.x == 0) {
            unsigned xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            __hip_atomic_fetch_add(&ctl->tickets[xcc & 7u], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_rank = (int)(blockIdx.x >> 3);
        }
    } else if (threadIdx.x == 0) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        xcc &= 7u;
        const unsigned tk = __hip_atomic_fetch_add(&ctl->tickets[xcc], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int rank = -1;
        if ((int)tk < nw) {
            if ((int)tk == nw - 1) {  // this XCD has filled up: claim leadership (first one wins)
                int expect = -1;
                __hip_atomic_compare_exchange_strong(&ctl->leader, &expect, (int)xcc, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            const long long t0 = wall_clock64();
            int ld_ = -1;
            while ((ld_ = __hip_atomic_load(&ctl->leader, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < 0) {
                __builtin_amdgcn_s_sleep(2);
                if (wall_clock64() - t0 > 20000000LL) break;
            }
            if (ld_ == (int)xcc) rank = (int)tk;
        }
        s_rank = rank;
    }
    __syncthreads();
    const int rank = s_rank;
    if (rank < 0) return;
    double *in = b0, *out = b1;
    for (int p = 0; p < phases; ++p) {
        for (int t = rank; t < 64; t += nw) tile<LD>(in, out, t, red);
        if (!xcd_barrier<SCOPE, SLEEP>(&ctl->bar, ctl->flags, (unsigned)(p + 1), rank, nw)) { ctl->failed = 1; return; }
        if (FRESH) {  // b0 is a ring of phases + 1 slabs
            in = out;
            out = (p == 0 ? b0 : in) + (size_t)SN * SN;
        } else {
            double *x = in; in = out; out = x;
        }
    }
}
__global__ __launch_bounds__(256) void one_k(const double *in, double *out)
{
    __shared__ double red[4][4][64];
    tile<0>(in, out, blockIdx.x, red);
}

static int g_fixed = 0;
template <int LD, int SCOPE, int SLEEP, bool FRESH = false>
static void run(const char *name, int nw, double *dA, double *dB, Ctl *ctl, const std::vector<double> &h, const std::vector<double> &ref, int phases)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f, t1 = 0;
    Ctl hc;
    for (int ph : {1, phases}) {
        best = 1e9f;
        for (int rep = 0; rep < 4; ++rep) {
            hipMemcpy(dA, h.data(), SN * SN * 8, hipMemcpyHostToDevice);
            Ctl z; for (int i = 0; i < 8; ++i) z.tickets[i] = 0; z.leader = -1; z.bar = 0; z.failed = 0;
            for (int i = 0; i < 64; ++i) z.flags[i] = 0;
            hipMemcpy(ctl, &z, sizeof(z), hipMemcpyHostToDevice);
            hipEventRecord(e0);
            hipLaunchKernelGGL((persist_k<LD, SCOPE, SLEEP, FRESH>), dim3(nw * 8), dim3(256), 0, 0, dA, FRESH ? dA + (size_t)SN * SN : dB, ctl, ph, nw, g_fixed);
            hipEventRecord(e1);
            hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        if (ph == 1) t1 = best;
    }
    hipMemcpy(&hc, ctl, sizeof(hc), hipMemcpyDeviceToHost);
    std::vector<double> out(SN * SN);
    hipMemcpy(out.data(), FRESH ? dA + (size_t)phases * SN * SN : ((phases & 1) ? dB : dA), SN * SN * 8, hipMemcpyDeviceToHost);
    double err = 0;
    for (int i = 0; i < SN * SN; ++i) err = fmax(err, fabs(out[i] - ref[i]));
    printf("%-44s nw=%2d  %6.2f us/phase (1 phase %6.2f us, %d phases %7.1f us)  leader XCD %d tickets %u %u %u %u %u %u %u %u  failed %u  max|diff| %.1e\n",
           name, nw, (best - t1) * 1e3f / (phases - 1), t1 * 1e3f, phases, best * 1e3f, hc.leader, hc.tickets[0], hc.tickets[1], hc.tickets[2],
           hc.tickets[3], hc.tickets[4], hc.tickets[5], hc.tickets[6], hc.tickets[7], hc.failed, err);
}

int main()
{
    double *dA, *dB;
    Ctl *ctl;
    hipMalloc(&dA, (size_t)64 * SN * SN * 8); hipMalloc(&dB, SN * SN * 8); hipMalloc(&ctl, sizeof(Ctl));  // dA: slab 0, or a ring of 64
    std::vector<double> h(SN * SN);
    for (int i = 0; i < SN * SN; ++i) h[i] = (i % 129 == 0) ? 9.0 : 0.3 * ((i * 37) % 11 - 5);
    const int phases = 41;
    hipMemcpy(dA, h.data(), SN * SN * 8, hipMemcpyHostToDevice);
    for (int p = 0; p < phases; ++p) hipLaunchKernelGGL(one_k, dim3(64), dim3(256), 0, 0, (p & 1) ? dB : dA, (p & 1) ? dA : dB);
    hipDeviceSynchronize();
    std::vector<double> ref(SN * SN);
    hipMemcpy(ref.data(), (phases & 1) ? dB : dA, SN * SN * 8, hipMemcpyDeviceToHost);
    for (int nw : {32, 36, 64}) {
        run<1, 1, 1>("sc1 loads, agent counter, sleep 1", nw, dA, dB, ctl, h, ref, phases);
        run<2, 1, 1>("nt loads, agent counter, sleep 1", nw, dA, dB, ctl, h, ref, phases);
        run<1, 1, 0>("sc1 loads, agent counter, sleep 0", nw, dA, dB, ctl, h, ref, phases);
        run<1, 0, 0>("sc1 loads, asm RMW counter", nw, dA, dB, ctl, h, ref, phases);
        run<1, 3, 0>("sc1 loads, flag words polled sc1", nw, dA, dB, ctl, h, ref, phases);
        run<5, 3, 0>("buffer_inv sc1 + plain loads, plain st, flags", nw, dA, dB, ctl, h, ref, phases);
        run<6, 3, 0>("buffer_inv sc0 + plain loads, plain st, flags", nw, dA, dB, ctl, h, ref, phases);
        run<4, 3, 0>("sc1 loads + sc1 stores, flags sc1", nw, dA, dB, ctl, h, ref, phases);
        run<4, 3, 1>("sc1 loads + sc1 stores, flags sc1, sleep 1", nw, dA, dB, ctl, h, ref, phases);
        run<4, 1, 0>("sc1 loads + sc1 stores, agent counter", nw, dA, dB, ctl, h, ref, phases);
        g_fixed = 1;
        run<4, 3, 0>("FIXED placement: sc1 ld + st, flags sc1", nw, dA, dB, ctl, h, ref, phases);
        run<4, 1, 0>("FIXED placement: sc1 ld + st, agent counter", nw, dA, dB, ctl, h, ref, phases);
        g_fixed = 0;
        run<3, 0, 0>("sc0 loads, asm RMW counter", nw, dA, dB, ctl, h, ref, phases);
        run<3, 2, 0>("sc0 loads, flag words polled sc0", nw, dA, dB, ctl, h, ref, phases);
        run<3, 3, 0>("sc0 loads, flag words polled sc1", nw, dA, dB, ctl, h, ref, phases);
        run<0, 1, 0, true>("plain loads of fresh slabs, agent counter", nw, dA, dB, ctl, h, ref, phases);
        run<0, 0, 0, true>("plain loads of fresh slabs, asm RMW counter", nw, dA, dB, ctl, h, ref, phases);
        run<0, 2, 0, true>("plain loads of fresh slabs, flags sc0", nw, dA, dB, ctl, h, ref, phases);
        run<0, 3, 0, true>("plain loads of fresh slabs, flags sc1", nw, dA, dB, ctl, h, ref, phases);
    }
    printf("final: %s\n", hipGetErrorString(hipDeviceSynchronize()));
    return 0;
}
