// grid_barrier.hip — what does a grid-wide barrier cost on MI355X (256 CUs in 8 XCDs, one L2 per XCD)?  The query-time
// encoder forward is 144 dependent launches of ~5 us; a persistent forward would replace the launch
// boundaries by grid barriers, so the barrier has to be well under a launch to pay.  Variants:
//   0  agent-scope RELEASE add + ACQUIRE spin (the compiler's L2 write-back / invalidate on every barrier)
//   1  RELAXED add + RELAXED spin, data exchanged through sc1 sc0 (write-through / L2-bypassing) accesses only
//   2  variant 1, and every workgroup also streams 32 KiB of "weights" between barriers (nt loads)
// Each iteration: every workgroup writes one 256-B record, barrier, reads the record of workgroup (b + it) % G and checks it.
//   hipcc -O3 --offload-arch=gfx950 grid_barrier.hip -o grid_barrier.bin && ./grid_barrier.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int V>
__global__ __launch_bounds__(256) void barrier_kernel(unsigned* counter, unsigned* rec, const u32x4* weights, int iters, unsigned* bad,
                                                      unsigned long long spin_limit) {
    const int G = gridDim.x, b = blockIdx.x;
    unsigned acc = 0;
    for (int it = 0; it < iters; ++it) {
        // publish
        const unsigned val = (unsigned)(it * 1315423911u) ^ (unsigned)b;
        if (threadIdx.x < 64) {
            unsigned* dst = rec + ((it & 1) * G + b) * 64 + threadIdx.x;
            if (V == 0) *dst = val;
            else __hip_atomic_store(dst, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (V == 2) {
            const u32x4* w = weights + ((size_t)b * 2048 + (size_t)(it & 7) * 2048 * G);
            u32x4 v0 = __builtin_nontemporal_load(w + threadIdx.x), v1 = __builtin_nontemporal_load(w + 256 + threadIdx.x);
            u32x4 v2 = __builtin_nontemporal_load(w + 512 + threadIdx.x), v3 = __builtin_nontemporal_load(w + 768 + threadIdx.x);
            u32x4 v4 = __builtin_nontemporal_load(w + 1024 + threadIdx.x), v5 = __builtin_nontemporal_load(w + 1280 + threadIdx.x);
            u32x4 v6 = __builtin_nontemporal_load(w + 1536 + threadIdx.x), v7 = __builtin_nontemporal_load(w + 1792 + threadIdx.x);
            acc += v0.x ^ v1.y ^ v2.z ^ v3.w ^ v4.x ^ v5.y ^ v6.z ^ v7.w;
        }
        // barrier
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned target = (unsigned)(it + 1) * (unsigned)G;
            if (V == 0) {
                __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                unsigned long long n = 0;
                while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target && ++n < spin_limit) __builtin_amdgcn_s_sleep(1);
                if (n >= spin_limit) atomicAdd(bad, 1000000u);
            } else {
                __builtin_amdgcn_s_waitcnt(0);   // this wave's write-through stores have been issued and acknowledged (vmcnt 0)
                __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                unsigned long long n = 0;
                while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target && ++n < spin_limit) __builtin_amdgcn_s_sleep(1);
                if (n >= spin_limit) atomicAdd(bad, 1000000u);
            }
        }
        __syncthreads();
        // consume
        if (threadIdx.x < 64) {
            const int src_b = (b + it + 1) % G;
            const unsigned* src = rec + ((it & 1) * G + src_b) * 64 + threadIdx.x;
            const unsigned got = V == 0 ? *src : __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned want = (unsigned)(it * 1315423911u) ^ (unsigned)src_b;
            if (got != want) atomicAdd(bad, 1u);
        }
    }
    if (acc == 0x12345678u) bad[1] = acc;
}

template <int V>
void run(int G, int iters, unsigned* counter, unsigned* rec, u32x4* weights, unsigned* bad) {
    CK(hipMemset(counter, 0, 4));
    CK(hipMemset(bad, 0, 8));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    unsigned long long limit = 20000000ull;
    void* args[] = {&counter, &rec, &weights, &iters, &bad, &limit};
    CK(hipEventRecord(e0));
    CK(hipLaunchCooperativeKernel(reinterpret_cast<void*>(&barrier_kernel<V>), dim3(G), dim3(256), args, 0, nullptr));
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned hb[2];
    CK(hipMemcpy(hb, bad, 8, hipMemcpyDeviceToHost));
    printf("variant %d  workgroups %3d  iterations %d: %.3f us per (publish + barrier + consume), mismatches %u\n", V, G, iters,
           ms * 1e3 / iters, hb[0]);
}

int main() {
    unsigned *counter, *rec, *bad;
    u32x4* weights;
    CK(hipMalloc(&counter, 256));
    CK(hipMalloc(&bad, 256));
    CK(hipMalloc(&rec, 2 * 256 * 64 * 4));
    CK(hipMalloc(&weights, (size_t)8 * 256 * 2048 * 16));
    CK(hipMemset(weights, 1, (size_t)8 * 256 * 2048 * 16));
    for (int G : {256, 128, 32, 8}) {
        run<0>(G, 2000, counter, rec, weights, bad);
        run<1>(G, 2000, counter, rec, weights, bad);
        run<2>(G, 2000, counter, rec, weights, bad);
    }
    return 0;
}
