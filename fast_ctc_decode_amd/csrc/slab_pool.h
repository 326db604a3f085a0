// slab_pool.h -- tree-arena slabs handed out ON THE DEVICE (wide-beam kernel, beam_lane.hip).
//
// Why: a wide-beam job's tree arena used to be one slab per READ of the launch -- 50 GB for BASELINE config 3's 8192
// reads -- although no more wavefronts than the chip holds (256 CUs x 4 SIMDs x 4 = 4096 at the kernel's 128 VGPRs) ever
// touch theirs at the same time.  With the slabs in a pool a wavefront takes one when it STARTS and gives it back after
// its traceback, so the arena is sized by the chip's residency, not by the batch: launches of any size -- and any number
// of launches in flight on different streams (fcd_set_overlap: the stragglers of one call run under the next calls) --
// share the same 4096 wave slabs.
//
// The pool is a ring of slab ids with two 64-bit ticket counters (FIFO: pop ticket t reads entry t mod P, push ticket
// t writes it; an entry carries the low bits of its ticket's generation t / P, so a pop whose entry has not been
// pushed yet simply waits for it -- which also makes a pool SMALLER than the residency correct: the wavefronts that
// hold slabs are resident and finish, the others sleep).  A push releases the slab's contents at agent scope and a pop
// acquires them: the next owner may run on another XCD, whose L2 is not coherent with this one's.
#pragma once

#include <stdint.h>

namespace fcd {
namespace slab_pool {

constexpr int kHeaderWords = 8;  // u64: [0] pop tickets, [1] push tickets, [2] P; entries (u32) follow
constexpr int kMaxSlabs = 65535; // an entry is generation << 16 | id

inline size_t bytes(int slabs) { return (size_t)kHeaderWords * 8 + (((size_t)slabs * 4 + 63) & ~(size_t)63); }

#if defined(__HIPCC__) || defined(FCD_HIPEMU)
// a free slab's id; every lane gets it (called in uniform control flow)
__device__ __forceinline__ int pop(unsigned long long *pool, int lane) {
    int id = 0;
    if (lane == 0) {
        const unsigned long long P = pool[2];
        const unsigned long long t = atomicAdd(&pool[0], 1ull);
        const uint32_t want = (uint32_t)((t / P) & 0xFFFFull);
        uint32_t *const e = reinterpret_cast<uint32_t *>(pool + kHeaderWords) + (t % P);
        uint32_t v = __atomic_load_n(e, __ATOMIC_RELAXED);
        while ((v >> 16) != want) {
            __builtin_amdgcn_s_sleep(32);
            v = __atomic_load_n(e, __ATOMIC_RELAXED);
        }
        id = (int)(v & 0xFFFFu);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    return __builtin_amdgcn_readfirstlane(id);
}

// gives the slab back (called in uniform control flow, after the wavefront's last access to the slab)
__device__ __forceinline__ void push(unsigned long long *pool, int id, int lane) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    if (lane == 0) {
        const unsigned long long P = pool[2];
        const unsigned long long t = atomicAdd(&pool[1], 1ull);
        uint32_t *const e = reinterpret_cast<uint32_t *>(pool + kHeaderWords) + (t % P);
        __atomic_store_n(e, (uint32_t)(((t / P) & 0xFFFFull) << 16) | (uint32_t)id, __ATOMIC_RELAXED);
    }
}
#endif

}  // namespace slab_pool
}  // namespace fcd
