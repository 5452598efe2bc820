// allreduce.hip - one-shot peer-to-peer SUM all-reduce of the training step's gradient bucket over xGMI (gfx950, one node).
//
// What it serves: the ONE collective of the data-parallel step (SURVEY.md par. 8e; reference loop src/runner/runner_udf.py:166-168
// has none - EMAP is single-GPU): 462 985 fp32 gradients + the step statistics in the bucket's tail = 1.85 MB.  That message is
// latency-bound: a ring all-reduce over R ranks is 2 (R - 1) dependent steps, while MI355X's xGMI is a full mesh of point-to-point
// links (7 x ~153 GB/s per GPU) - every rank can read every other rank's bucket directly, all links at once:
//     1.85 MB per link = ~12 us of wire time, ONE synchronisation, and the sum is formed in rank order on every rank,
// so all ranks hold bit-identical results (a ring's result depends on where a rank sits in it).  Above a few MB the (R - 1) x size
// reads per rank stop paying and RCCL's reduce-scatter + all-gather wins: this is for the small bucket only (emap_ar_local_bytes
// refuses more than 64 MiB).
//
// Memory: every rank owns one REGION = [256 B control][2 staging buffers of n floats], allocated uncached / fine-grained
// (hipDeviceMallocUncached: peer accesses and the owner's accesses bypass the non-coherent L2 paths) and exported with
// hipIpcGetMemHandle; the ranks exchange the 64-byte handles once (host side: torch.distributed all_gather_object, any backend)
// and map each other's regions with hipIpcOpenMemHandle.  No host involvement after that: the kernel is graph-capturable.
//
// One launch (every rank, same n, in lock step t = 1, 2, ...):
//   1  copy its gradient bucket into its staging buffer t & 1;
//   2  the last workgroup to finish that (device-scope counter behind a system-scope fence) publishes flag = t with a
//      system-scope RELEASE store;
//   3  every workgroup ACQUIRE-polls the flags of all peers until they show >= t (bounded by s_memrealtime: emap_ar_set_timeout_ms,
//      default 10 s, 6x that for the first two launches whose peers may still be loading code objects / allocating workspaces;
//      a dead peer must not hang the GPU).  A time-out is LOUD: the sticky error word is set AND the workgroup writes NaN instead
//      of a partial sum, so every consumer of the bucket (Adam, the loss) turns NaN on this rank - ranks cannot silently apply
//      different gradients;
//   4  out[i] = sum over ranks r = 0 .. R-1, in that order, of staging_r[t & 1][i]  (16-byte loads straight from the peers);
//   5  the last workgroup to finish stores step = t for the next launch.
// Double buffering makes one flag wait per launch enough: a rank overwrites buffer b again in launch t + 2, which it enters only
// after it has seen every peer's flag t + 1 - and a peer publishes t + 1 after its launch t (its reads of buffer b) has completed.
#include "emap_common.h"
#include <stddef.h>
#include <string.h>

namespace emap {

typedef float ar_f4 __attribute__((ext_vector_type(4)));
constexpr int AR_MAX_RANKS = 16;
constexpr int AR_CTRL_BYTES = 256;
constexpr int AR_BLOCKS = 64;            // workgroups per launch: 1.85 MB / 64 = 29 KB each; far fewer than the 256 CUs, so a peer process
constexpr int AR_THREADS = 256;          // sharing the GPU (the one-GPU tests) always finds CUs for its own launch

struct ArCtrl {            // first 256 bytes of a region
    uint32_t flag;         // last step whose staging buffer is complete (written by the owner, polled by the peers)
    uint32_t step;         // last completed launch of the owner (read by the owner's next launch)
    uint32_t arrive0;      // workgroups that have finished phase 1 of the current launch
    uint32_t arrive1;      // workgroups that have finished phase 4
    uint32_t error;        // 1 = a peer's flag did not arrive in time
    uint32_t pad[59];
};
static_assert(sizeof(ArCtrl) == AR_CTRL_BYTES, "control block");

struct ArArgs {
    float* data;                       // in: this rank's values, out: the sum over the ranks
    long long n;                       // floats (multiple of 4 is not required)
    char* region[AR_MAX_RANKS];        // region of rank r as mapped into this process (region[rank] = own)
    int rank, world;
    long long stage_floats;            // capacity of one staging buffer
    long long timeout_ticks;           // bound of one flag wait in s_memrealtime ticks (100 MHz)
};

__device__ __forceinline__ float* ar_staging(char* region, long long stage_floats, uint32_t b) {
    return reinterpret_cast<float*>(region + AR_CTRL_BYTES) + (size_t)b * (size_t)stage_floats;
}

__global__ __launch_bounds__(AR_THREADS) void allreduce_oneshot_kernel(const ArArgs a) {
    ArCtrl* const me = reinterpret_cast<ArCtrl*>(a.region[a.rank]);
    const uint32_t t = __hip_atomic_load(&me->step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;   // written by the previous launch
    const uint32_t b = t & 1u;
    const long long n4 = a.n >> 2;
    const long long tid = (long long)blockIdx.x * AR_THREADS + threadIdx.x, nth = (long long)gridDim.x * AR_THREADS;

    // ---- 1: own bucket -> own staging buffer b ----
    {
        float* st = ar_staging(a.region[a.rank], a.stage_floats, b);
        const ar_f4* src4 = reinterpret_cast<const ar_f4*>(a.data);
        ar_f4* dst4 = reinterpret_cast<ar_f4*>(st);
        for (long long i = tid; i < n4; i += nth) dst4[i] = src4[i];
        for (long long i = (n4 << 2) + tid; i < a.n; i += nth) st[i] = a.data[i];
    }
    // ---- 2: the last workgroup publishes the buffer ----
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t old = __hip_atomic_fetch_add(&me->arrive0, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (old == gridDim.x - 1) {
            __hip_atomic_store(&me->arrive0, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&me->flag, t, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    // ---- 3: wait for every rank's buffer b of step t (own included: its last workgroup may still be copying) ----
    __shared__ int timed_out;
    if (threadIdx.x == 0) timed_out = 0;
    __syncthreads();
    if ((int)threadIdx.x < a.world) {
        ArCtrl* const pc = reinterpret_cast<ArCtrl*>(a.region[threadIdx.x]);
        const long long t0 = (long long)__builtin_amdgcn_s_memrealtime();      // 100 MHz
        bool ok = false;
        while (true) {
            const uint32_t f = __hip_atomic_load(&pc->flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
            if ((int32_t)(f - t) >= 0) { ok = true; break; }
            if ((long long)__builtin_amdgcn_s_memrealtime() - t0 > (t <= 2u ? 6 * a.timeout_ticks : a.timeout_ticks)) break;
            __builtin_amdgcn_s_sleep(8);
        }
        if (!ok) {
            __hip_atomic_store(&me->error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            timed_out = 1;
        }
    }
    __syncthreads();
    __atomic_thread_fence(__ATOMIC_ACQUIRE);     // (every thread's later loads are ordered behind the polls of its workgroup)
    // ---- 4: sum in rank order (a peer that never arrived: poison this workgroup's share instead of summing stale buffers) ----
    if (timed_out) {
        const float qnan = __builtin_nanf("");
        for (long long i = tid; i < a.n; i += nth) a.data[i] = qnan;
    } else {
        for (long long i = tid; i < n4; i += nth) {
            ar_f4 acc = {0.f, 0.f, 0.f, 0.f};
            for (int r = 0; r < a.world; ++r)
                acc += __builtin_nontemporal_load(reinterpret_cast<const ar_f4*>(ar_staging(a.region[r], a.stage_floats, b)) + i);
            reinterpret_cast<ar_f4*>(a.data)[i] = acc;
        }
        for (long long i = (n4 << 2) + tid; i < a.n; i += nth) {
            float acc = 0.f;
            for (int r = 0; r < a.world; ++r) acc += __builtin_nontemporal_load(ar_staging(a.region[r], a.stage_floats, b) + i);
            a.data[i] = acc;
        }
    }
    // ---- 5: the last workgroup closes the step ----
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t old = __hip_atomic_fetch_add(&me->arrive1, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (old == gridDim.x - 1) {
            __hip_atomic_store(&me->arrive1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&me->step, t, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

}  // namespace emap

using namespace emap;

extern "C" {

static long long g_ar_timeout_ticks = 1000000000ll;     // 10 s of the 100 MHz s_memrealtime counter

int emap_ar_set_timeout_ms(int64_t ms) {
    if (ms < 1 || ms > 600000) { set_error("ar_set_timeout_ms: %lld ms is outside [1, 600000]", (long long)ms); return EMAP_E_INVALID; }
    g_ar_timeout_ticks = (long long)ms * 100000ll;
    return EMAP_OK;
}

int emap_ar_local_bytes(int64_t n_floats, size_t* bytes) {
    if (!bytes || n_floats <= 0 || n_floats > (64ll << 20) / 4) { set_error("ar_local_bytes: n_floats must be in (0, 16 Mi] (a latency-bound bucket; use RCCL above that)"); return EMAP_E_INVALID; }
    const size_t stage = ((size_t)n_floats * 4 + 255) & ~(size_t)255;
    *bytes = AR_CTRL_BYTES + 2 * stage;
    return EMAP_OK;
}

int emap_ar_alloc(size_t bytes, void** region, void* ipc_handle64) {
    if (!region || !ipc_handle64 || bytes < AR_CTRL_BYTES) { set_error("ar_alloc: bad arguments"); return EMAP_E_INVALID; }
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "hipIpcMemHandle_t is 64 bytes");
    void* p = nullptr;
    // uncached / fine-grained: flags and staging are read by peers WHILE kernels run on both sides.  There is no fall-back to a plain
    // hipMalloc: coarse-grained memory gives no cross-device visibility guarantee during a running kernel, whatever scope the
    // atomics name - the failure would surface as time-outs.  Callers fall back to RCCL instead (Trainer(allreduce="rccl")).
    if (hipExtMallocWithFlags(&p, bytes, hipDeviceMallocUncached) != hipSuccess) {
        (void)hipGetLastError();
        set_error("ar_alloc: fine-grained (uncached) allocation of %zu bytes failed; use the RCCL transport", bytes);
        return EMAP_E_LAUNCH;
    }
    if (hipMemset(p, 0, bytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess) { set_error("ar_alloc: memset failed"); (void)hipFree(p); return EMAP_E_LAUNCH; }
    hipIpcMemHandle_t h;
    const hipError_t e = hipIpcGetMemHandle(&h, p);
    if (e != hipSuccess) {
        set_error("ar_alloc: hipIpcGetMemHandle failed (%s) - is HSA_ENABLE_IPC_MODE_LEGACY=0 exported?", hipGetErrorString(e));
        (void)hipGetLastError(); (void)hipFree(p);
        return EMAP_E_LAUNCH;
    }
    memcpy(ipc_handle64, &h, 64);
    *region = p;
    return EMAP_OK;
}

int emap_ar_open(const void* ipc_handle64, void** peer_region) {
    if (!ipc_handle64 || !peer_region) { set_error("ar_open: null pointer"); return EMAP_E_INVALID; }
    hipIpcMemHandle_t h;
    memcpy(&h, ipc_handle64, 64);
    void* p = nullptr;
    const hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) { set_error("ar_open: hipIpcOpenMemHandle failed (%s)", hipGetErrorString(e)); (void)hipGetLastError(); return EMAP_E_LAUNCH; }
    *peer_region = p;
    return EMAP_OK;
}

int emap_ar_close(void* peer_region) {
    if (peer_region && hipIpcCloseMemHandle(peer_region) != hipSuccess) { (void)hipGetLastError(); set_error("ar_close failed"); return EMAP_E_LAUNCH; }
    return EMAP_OK;
}

int emap_ar_free(void* region) {
    if (region && hipFree(region) != hipSuccess) { (void)hipGetLastError(); set_error("ar_free failed"); return EMAP_E_LAUNCH; }
    return EMAP_OK;
}

int emap_ar_allreduce_sum(float* data, int64_t n, int rank, int world, void* const* regions_host, size_t region_bytes, void* stream) {
    if (!data || n <= 0 || world < 1 || world > AR_MAX_RANKS || rank < 0 || rank >= world || !regions_host) { set_error("ar_allreduce_sum: bad arguments"); return EMAP_E_INVALID; }
    if (((uintptr_t)data & 15) != 0) { set_error("ar_allreduce_sum: data must be 16-byte aligned"); return EMAP_E_INVALID; }
    if (region_bytes < AR_CTRL_BYTES + 2 * (size_t)n * 4) { set_error("ar_allreduce_sum: region of %zu bytes is too small for %lld floats", region_bytes, (long long)n); return EMAP_E_WORKSPACE; }
    ArArgs a;
    a.data = data; a.n = n; a.rank = rank; a.world = world;
    a.stage_floats = (long long)((region_bytes - AR_CTRL_BYTES) / 2 / 4);
    a.timeout_ticks = g_ar_timeout_ticks;
    for (int r = 0; r < AR_MAX_RANKS; ++r) a.region[r] = (r < world) ? static_cast<char*>(regions_host[r]) : nullptr;
    for (int r = 0; r < world; ++r) if (!a.region[r]) { set_error("ar_allreduce_sum: region of rank %d is null", r); return EMAP_E_INVALID; }
    hipLaunchKernelGGL(allreduce_oneshot_kernel, dim3(AR_BLOCKS), dim3(AR_THREADS), 0, static_cast<hipStream_t>(stream), a);
    return check_launch("ar_allreduce_sum");
}

/* error word of the own region (host read: synchronises the device) - 1 if a peer's flag did not arrive within the time-out in some launch */
int emap_ar_error(void* region, int* error_host) {
    if (!region || !error_host) { set_error("ar_error: null pointer"); return EMAP_E_INVALID; }
    uint32_t e = 0;
    if (hipMemcpy(&e, static_cast<char*>(region) + offsetof(ArCtrl, error), 4, hipMemcpyDeviceToHost) != hipSuccess) { (void)hipGetLastError(); set_error("ar_error: copy failed"); return EMAP_E_LAUNCH; }
    *error_host = (int)e;
    return EMAP_OK;
}

}  // extern "C"
