// sq_hostflag.h -- publishing results to pinned host memory from inside a kernel.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// Completion words in pinned host memory.  The results a kernel leaves in host memory and the word the host polls are
// all posted writes on their way across the link; a system-scope fence orders them inside the GPU, but a later posted
// write was SEEN to overtake earlier ones on the way (per-job flags of the blossom kernel: one result block in 10^5 was
// still arriving when its flag was visible).  A read from host memory cannot pass the posted writes ahead of it: call
// this between the last result store and the store of the polled word.
__device__ __forceinline__ void sq_host_write_flush(const volatile void *any_host_word)
{
    __threadfence_system();
#ifdef SQ_NO_HOST_FLUSH                                  // (measurement only)
    return;
#endif
    const uint32_t back = __hip_atomic_load(reinterpret_cast<const uint32_t *>(const_cast<const void *>(any_host_word)), __ATOMIC_RELAXED,
                                            __HIP_MEMORY_SCOPE_SYSTEM);
    asm volatile("s_waitcnt vmcnt(0)" :: "v"(back) : "memory");
    __threadfence_system();
}

