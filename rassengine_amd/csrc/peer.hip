// peer.hip — SURVEY §8f-4 (first half): the cross-shard exchange of per-shard top-k WITHOUT a collective.
//
// The exchange is 3.8 KB per rank (32 queries x top-10: nq*k f32 scores | nq*k i64 ids): pure latency.  Instead of
// one RCCL all-gather, every rank STORES its packed record straight into a slot of a buffer that lives in rank 0's
// HBM (mapped into the peer processes through a HIP IPC handle; over xGMI between GPUs) and then releases a
// per-rank sequence flag at system scope; rank 0 runs a one-workgroup kernel that acquires the G flags, and its
// ordinary merge kernel follows in stream order.  No rank but 0 needs the merged result on the serving path
// (the reference's coordinator, app/main.py:89, 357).
//
//   layout of the buffer:  [G flags, 64 B apart][2 parities][G slots of slot_bytes]
//
// Slots are double-buffered by step parity: a rank can only post step s+2 after rank 0's broadcast of step s+2,
// which rank 0's stream orders behind its merge of step s (dist.PeerMergeSearch).
// The wait is BOUNDED: a wave that never sees its flag gives up after max_spins polls and reports it, so a lost
// peer cannot hang the GPU.

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"

namespace rass {

__global__ __launch_bounds__(256) void peer_post_kernel(const uint4* __restrict__ local, uint4* __restrict__ remote,
                                                        int n16, unsigned long long* remote_flag,
                                                        unsigned long long seq) {
    for (int i = threadIdx.x; i < n16; i += blockDim.x) remote[i] = local[i];
    __threadfence_system();  // the record is visible system-wide before the flag
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(remote_flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ __launch_bounds__(64) void peer_wait_kernel(const unsigned long long* __restrict__ flags, int n,
                                                       int flag_stride_u64, unsigned long long seq,
                                                       int* __restrict__ status, long long max_spins) {
    const int r = threadIdx.x;
    bool ok = true;
    if (r < n) {
        const unsigned long long* f = flags + (int64_t)r * flag_stride_u64;
        long long spins = 0;
        while (__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < seq) {
            if (++spins >= max_spins) {
                ok = false;
                break;
            }
            __builtin_amdgcn_s_sleep(8);
        }
    }
    if (!ok) atomicExch(status, 1 + r);  // which rank never arrived (1-based)
    __threadfence_system();
}

hipError_t launch_peer_post(const void* local, size_t bytes, void* remote_slot, void* remote_flag, uint64_t seq,
                            hipStream_t stream) {
    if (bytes == 0 || bytes % 16 != 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(peer_post_kernel, dim3(1), dim3(256), 0, stream, static_cast<const uint4*>(local),
                       static_cast<uint4*>(remote_slot), (int)(bytes / 16), static_cast<unsigned long long*>(remote_flag),
                       (unsigned long long)seq);
    return hipGetLastError();
}

hipError_t launch_peer_wait(const void* flags, int n, int flag_stride_bytes, uint64_t seq, int* status,
                            int64_t max_spins, hipStream_t stream) {
    if (n < 1 || n > 64 || flag_stride_bytes % 8 != 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(peer_wait_kernel, dim3(1), dim3(64), 0, stream, static_cast<const unsigned long long*>(flags), n,
                       flag_stride_bytes / 8, (unsigned long long)seq, status, (long long)max_spins);
    return hipGetLastError();
}

}  // namespace rass
