// exchange.hip -- the ONE exchange of a sample-sharded control step as direct peer writes.
//
// A sharded step (quadjax has none: its N samples live in one XLA program) leaves one RANK RECORD per GPU --
// the online-softmax partial {m, s, v[128]} of controllers/covo.py:266-272 and, when the caller wants covo.py:281's
// pos_mean / pos_std, the 192 fp64 position sums -- COVO_RANK_RECORD_FLOATS * 4 = 2 064 bytes.  Every rank needs all G records
// before it can merge.  Through RCCL that is an all-gather whose cost is pure latency (10-20 us for 2 KB); the xGMI mesh
// is fully connected and the payload fits one store burst, so here every rank WRITES its record straight into a slot of
// every peer's exchange buffer (hipIpc-mapped device memory) and raises a per-(step parity, source rank) sequence flag;
// the receiving side spins on its OWN memory until all G flags carry this step's sequence number, then copies the slots
// out.  No collective library, no host round trip: the whole control step, exchange included, can be enqueued from C
// (covo_run_episode on sharded ranks).
//   buffer (per rank):  float  slot[2][G][COVO_RANK_RECORD_COV_FLOATS] two parities: a rank can run at most one step ahead
//                       uint64 flag[2][G]                            of the slowest (it needs everybody's record to finish)
// Visibility: payload stores, then a system-scope fence, then the flag as a system-scope release store; the reader
// acquires the flag and reads the payload with system-scope (sc0 sc1) loads -- nothing depends on a cache being flushed at
// a kernel boundary.  The spin is bounded (covo_exchange_set_timeout; default 60 s -- a collective would simply wait, and host-side
// rank skew of seconds is ordinary: a first-step graph capture, a garbage collection, a rank that enqueued a whole episode
// segment ahead): a missing peer surfaces as COVO_DEVSTAT_EXCHANGE + NaN records, never a hang.
// Validated functionally with two processes sharing one GPU (tests/test_gpu_parity.py); NOT measured over xGMI (the pool
// has single-GPU boxes only), which is why the host side keeps torch.distributed (RCCL) unless the caller opts in
// (controllers/_core.py: exchange="peer" / "auto", COVO_EXCHANGE).  The handle blob the ranks all-gather also carries whether the
// buffer is fine-grained and the PCI bus id of its device: a coarse-grained buffer (the allocation fallback) is only accepted
// between ranks that share ONE device (they share its L2); across devices remote xGMI writes into it are not guaranteed to be
// seen by the owner's L2, so covo_exchange_connect refuses.
#include <cstdlib>
#include <cstring>
#include "covo_common.hpp"

constexpr int EX_MAX_WORLD = 16;

struct ExPeers {
    float *base[EX_MAX_WORLD];
};

// what a rank publishes: the hipIpc handle + the facts the peers need to decide whether mapping it is safe
struct ExBlob {
    hipIpcMemHandle_t ipc;
    unsigned int finegrained;  // 1: hipDeviceMallocFinegrained succeeded
    char bus_id[32];           // hipDeviceGetPCIBusId of the owning device ("0000:c1:00.0")
};
static_assert(sizeof(ExBlob) <= COVO_EXCHANGE_HANDLE_BYTES, "handle blob size");

struct Exchange {
    int world, rank;
    float *local;                      // this rank's buffer (slots + flags)
    float *peer[EX_MAX_WORLD];         // everybody's buffer as mapped here (peer[rank] = local)
    float *gathered;                   // [world][COVO_RANK_RECORD_FLOATS] local staging the merge reads
    unsigned long long seq;            // exchanges done
    bool connected;
    long long timeout_ticks;           // bound of the wait kernel's spin, 100 MHz wall-clock ticks
    ExBlob blob;
};

// a slot holds the LARGER record kind (MPPI's covariance adaptation carries 320 second moments more); an exchange moves the
// first `nfloats` of it
constexpr int EX_SLOT = COVO_RANK_RECORD_COV_FLOATS;
__host__ __device__ inline size_t ex_slot_floats(int world) { return (size_t)2 * world * EX_SLOT; }
static size_t ex_bytes(int world) { return ex_slot_floats(world) * sizeof(float) + (size_t)2 * world * sizeof(unsigned long long); }

// workgroup p: this rank's record -> slot [parity][rank] of rank p's buffer, then the flag
__global__ __launch_bounds__(256) void exchange_push_kernel(const float *__restrict__ record, const ExPeers peers, int world, int rank,
                                                            int parity, unsigned long long seq, int nfloats)
{
    const int p = blockIdx.x;
    float *base = peers.base[p];
    float *dst = base + ((size_t)parity * world + rank) * EX_SLOT;
    for (int i = threadIdx.x; i < nfloats; i += 256)
        __hip_atomic_store(dst + i, record[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long *flags = reinterpret_cast<unsigned long long *>(base + ex_slot_floats(world));
        __hip_atomic_store(flags + (size_t)parity * world + rank, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// one workgroup: wait until all `world` flags of this parity carry `seq`, then copy the slots to `gathered`
__global__ __launch_bounds__(512) void exchange_wait_kernel(float *__restrict__ local, int world, int parity, unsigned long long seq,
                                                            float *__restrict__ gathered, int *status, long long timeout_ticks,
                                                            int nfloats)
{
    __shared__ int ok;
    const int tid = threadIdx.x;
    // fail fast: once an exchange of this handle has timed out (sticky COVO_DEVSTAT_EXCHANGE, cleared by the host) every wait
    // still queued behind it -- covo_run_episode enqueues whole episode segments -- leaves its NaN records without spinning its own
    // full time-out in stream order (300 queued steps x 60 s otherwise)
    if (tid == 0)
        ok = (status != nullptr && (__hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) & COVO_DEVSTAT_EXCHANGE)) ? 0 : 1;
    __syncthreads();
    if (ok != 0 && tid < world) {
        const unsigned long long *flag = reinterpret_cast<const unsigned long long *>(local + ex_slot_floats(world)) +
                                         (size_t)parity * world + tid;
        const long long t0 = wall_clock64();
        while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < seq) {
            __builtin_amdgcn_s_sleep(4);
            if (wall_clock64() - t0 > timeout_ticks) {  // 100 MHz wall clock: a peer is gone
                ok = 0;
                break;
            }
        }
    }
    __syncthreads();
    const bool good = ok != 0;
    if (!good && tid == 0 && status != nullptr)
        __hip_atomic_fetch_or(status, COVO_DEVSTAT_EXCHANGE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const float *src = local + (size_t)parity * world * EX_SLOT;
    for (int i = tid; i < world * nfloats; i += 512) {  // gathered: [world][nfloats], dense
        const int g = i / nfloats, j = i - g * nfloats;
        gathered[i] = good ? __hip_atomic_load(src + (size_t)g * EX_SLOT + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : __builtin_nanf("");
    }
}

// sum_g of the fp64 position sums riding in the rank records (covo.py:281 over all shards)
// (stride: floats per record; offset: where its 192 doubles start -- after the partial, or after the second moments too)
__global__ __launch_bounds__(256) void rank_stats_sum_kernel(const float *__restrict__ records, int G, double *__restrict__ out,
                                                             int stride, int offset)
{
    const int i = threadIdx.x;
    if (i >= COVO_POS_STATS_DOUBLES) return;
    double acc = 0.0;
    for (int g = 0; g < G; ++g)
        acc += reinterpret_cast<const double *>(records + (size_t)g * stride + offset)[i];
    out[i] = acc;
}

int launch_rank_stats_sum(const float *records, int G, double *out, hipStream_t s, bool cov)
{
    hipLaunchKernelGGL(rank_stats_sum_kernel, dim3(1), dim3(256), 0, s, records, G, out,
                       cov ? COVO_RANK_RECORD_COV_FLOATS : COVO_RANK_RECORD_FLOATS,
                       cov ? COVO_PARTIAL_FLOATS + COVO_COV_FLOATS : COVO_PARTIAL_FLOATS);
    COVO_CHECK_HIP(hipGetLastError());
    return 0;
}

void exchange_destroy(covo_ctx *h)
{
    Exchange *x = reinterpret_cast<Exchange *>(h->exchange);
    if (!x) return;
    (void)hipDeviceSynchronize();  // nothing of this rank still reads or writes the buffers (peers: the caller tears ranks down together)
    for (int p = 0; p < x->world; ++p)  // every mapping made, also those of a connect that failed partway
        if (p != x->rank && x->peer[p]) (void)hipIpcCloseMemHandle(x->peer[p]);
    (void)hipFree(x->local);
    (void)hipFree(x->gathered);
    delete x;
    h->exchange = nullptr;
}

int exchange_create(covo_ctx *h, int world, int rank, void *handle_out)
{
    if (world < 2 || world > EX_MAX_WORLD || rank < 0 || rank >= world) {
        covo_set_error("covo_exchange_create: world=%d rank=%d (2 <= world <= %d)", world, rank, EX_MAX_WORLD);
        return COVO_E_BADARG;
    }
    exchange_destroy(h);
    Exchange *x = new Exchange();
    std::memset(x, 0, sizeof(*x));
    x->world = world;
    x->rank = rank;
    x->timeout_ticks = 60LL * 100000000LL;
    if (const char *e = std::getenv("COVO_EXCHANGE_TIMEOUT_S")) {
        const double sec = std::atof(e);
        if (sec > 0.0) x->timeout_ticks = (long long)(sec * 1e8);
    }
    h->exchange = x;  // owned by the handle from here on: a failure below is cleaned up by exchange_destroy / covo_destroy
    // fine-grained device memory (uncached in this GPU's L2): the peers' writes arrive over xGMI behind the L2's back, and the
    // wait kernel must see them; plain hipMalloc is the fallback (ranks sharing one GPU share its L2 anyway)
    hipError_t e = hipExtMallocWithFlags(reinterpret_cast<void **>(&x->local), ex_bytes(world), hipDeviceMallocFinegrained);
    x->blob.finegrained = (e == hipSuccess) ? 1u : 0u;
    if (std::getenv("COVO_DEBUG_EXCHANGE_COARSE")) {  // tests: behave as if the fine-grained allocation had failed
        if (e == hipSuccess) (void)hipFree(x->local);
        x->blob.finegrained = 0u;
        e = hipErrorOutOfMemory;
    }
    if (e != hipSuccess) {
        (void)hipGetLastError();
        x->local = nullptr;
        e = hipMalloc(&x->local, ex_bytes(world));
    }
    if (e == hipSuccess) e = hipMemset(x->local, 0, ex_bytes(world));  // flags = 0 < every sequence number (they start at 1)
    if (e == hipSuccess) e = hipMalloc(&x->gathered, (size_t)world * EX_SLOT * sizeof(float));
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipIpcGetMemHandle(&x->blob.ipc, x->local);
    if (e == hipSuccess) {
        int dev = 0;
        e = hipGetDevice(&dev);
        if (e == hipSuccess) e = hipDeviceGetPCIBusId(x->blob.bus_id, (int)sizeof(x->blob.bus_id), dev);
        if (const char *o = std::getenv("COVO_DEBUG_EXCHANGE_BUS_ID")) {  // tests: pretend this rank sits on another device
            std::memset(x->blob.bus_id, 0, sizeof(x->blob.bus_id));
            std::strncpy(x->blob.bus_id, o, sizeof(x->blob.bus_id) - 1);
        }
    }
    if (e != hipSuccess) {
        covo_set_error("covo_exchange_create: %s", hipGetErrorString(e));
        exchange_destroy(h);
        return (int)e;
    }
    std::memset(handle_out, 0, COVO_EXCHANGE_HANDLE_BYTES);
    std::memcpy(handle_out, &x->blob, sizeof(ExBlob));
    return 0;
}

int exchange_connect(covo_ctx *h, const void *handles)
{
    Exchange *x = reinterpret_cast<Exchange *>(h->exchange);
    if (!x) {
        covo_set_error("covo_exchange_connect: covo_exchange_create first");
        return COVO_E_BADARG;
    }
    // first the check that needs no mapping: a coarse-grained buffer on either side of a pair of DIFFERENT devices is refused
    for (int p = 0; p < x->world; ++p) {
        if (p == x->rank) continue;
        ExBlob pb;
        std::memcpy(&pb, reinterpret_cast<const char *>(handles) + (size_t)p * COVO_EXCHANGE_HANDLE_BYTES, sizeof(pb));
        pb.bus_id[sizeof(pb.bus_id) - 1] = 0;
        const bool same_device = std::strcmp(pb.bus_id, x->blob.bus_id) == 0;
        if (!same_device && !(pb.finegrained && x->blob.finegrained)) {
            covo_set_error("covo_exchange_connect: rank %d (device %s) and rank %d (device %s) sit on different devices and the "
                           "exchange buffer of rank %d is coarse-grained (hipDeviceMallocFinegrained failed): remote writes would "
                           "not be guaranteed visible -- use the collective exchange",
                           x->rank, x->blob.bus_id, p, pb.bus_id, pb.finegrained ? x->rank : p);
            return COVO_E_UNSUPPORTED;
        }
    }
    for (int p = 0; p < x->world; ++p) {
        if (p == x->rank) {
            x->peer[p] = x->local;
            continue;
        }
        ExBlob pb;
        std::memcpy(&pb, reinterpret_cast<const char *>(handles) + (size_t)p * COVO_EXCHANGE_HANDLE_BYTES, sizeof(pb));
        void *ptr = nullptr;
        COVO_CHECK_HIP(hipIpcOpenMemHandle(&ptr, pb.ipc, hipIpcMemLazyEnablePeerAccess));
        x->peer[p] = reinterpret_cast<float *>(ptr);
    }
    x->connected = true;
    return 0;
}

// bound of the wait kernel's spin for the exchanges enqueued from now on (seconds; <= 0 leaves it unchanged)
int exchange_set_timeout(covo_ctx *h, double seconds)
{
    Exchange *x = reinterpret_cast<Exchange *>(h->exchange);
    if (!x) {
        covo_set_error("covo_exchange_set_timeout: covo_exchange_create first");
        return COVO_E_BADARG;
    }
    if (seconds > 0.0) x->timeout_ticks = (long long)(seconds * 1e8);
    return 0;
}

bool exchange_ready(const covo_ctx *h)
{
    const Exchange *x = reinterpret_cast<const Exchange *>(h->exchange);
    return x != nullptr && x->connected;
}
int exchange_world(const covo_ctx *h)
{
    const Exchange *x = reinterpret_cast<const Exchange *>(h->exchange);
    return x ? x->world : 1;
}

// enqueue: push this rank's record to every peer, wait for everybody's, leave them in *gathered_out (the handle's staging
// buffer when gathered_dst == null)
int exchange_records(covo_ctx *h, const float *record, float *gathered_dst, const float **gathered_out, hipStream_t s, int nfloats)
{
    if (nfloats != COVO_RANK_RECORD_FLOATS && nfloats != COVO_RANK_RECORD_COV_FLOATS) {
        covo_set_error("covo_exchange_records: nfloats=%d (COVO_RANK_RECORD_FLOATS or COVO_RANK_RECORD_COV_FLOATS)", nfloats);
        return COVO_E_BADARG;
    }
    Exchange *x = reinterpret_cast<Exchange *>(h->exchange);
    if (!x || !x->connected) {
        covo_set_error("covo_exchange_records: the exchange is not connected (covo_exchange_create / covo_exchange_connect)");
        return COVO_E_BADARG;
    }
    const unsigned long long seq = ++x->seq;
    const int parity = (int)(seq & 1ull);
    ExPeers peers;
    std::memset(&peers, 0, sizeof(peers));
    for (int p = 0; p < x->world; ++p) peers.base[p] = x->peer[p];
    float *dst = gathered_dst ? gathered_dst : x->gathered;
    hipLaunchKernelGGL(exchange_push_kernel, dim3(x->world), dim3(256), 0, s, record, peers, x->world, x->rank, parity, seq, nfloats);
    hipLaunchKernelGGL(exchange_wait_kernel, dim3(1), dim3(512), 0, s, x->local, x->world, parity, seq, dst, h->status_dev,
                       x->timeout_ticks, nfloats);
    COVO_CHECK_HIP(hipGetLastError());
    if (gathered_out) *gathered_out = dst;
    return 0;
}
