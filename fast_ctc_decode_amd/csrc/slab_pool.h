// slab_pool.h -- tree-arena slabs handed out ON THE DEVICE (wide-beam kernel, beam_lane.hip).
//
// Why: a wide-beam job's tree arena used to be one slab per READ of the launch -- 50 GB for BASELINE config 3's 8192
// reads -- although no more wavefronts than the chip holds (256 CUs x 4 SIMDs x 4 = 4096 at the kernel's 128 VGPRs) ever
// touch theirs at the same time.  With the slabs in a pool a wavefront takes one when it STARTS and gives it back after
// its traceback, so the arena is sized by the chip's residency, not by the batch: launches of any size -- and any number
// of launches in flight on different streams (fcd_set_overlap: the stragglers of one call run under the next calls) --
// share the same 4096 wave slabs.
//
// The pool is a bounded queue of slab ids (D. Vyukov's array queue with the positions taken by fetch-and-add, so that both
// ends WAIT instead of failing): two 64-bit ticket counters and one 64-bit cell per slab, sequence number << 32 | id.  Pop
// ticket t waits until cell t mod P carries sequence t + 1, takes the id and leaves sequence t + P; push ticket t waits
// until the cell carries sequence t and stores (t + 1, id) -- a cell is never overwritten before its reader has been, and
// a pop whose slab has not been pushed yet simply waits for it, which also makes a pool SMALLER than the residency
// correct: the wavefronts that hold slabs are resident and finish, the others sleep.  A push releases the slab's contents
// at agent scope and a pop acquires them: the next owner may run on another XCD, whose L2 is not coherent with this one's.
#pragma once

#include <stdint.h>

namespace fcd {
namespace slab_pool {

constexpr int kHeaderWords = 8;  // u64: [0] pop tickets, [1] push tickets, [2] P; the cells (u64) follow
constexpr int kMaxSlabs = 65535;

inline size_t bytes(int slabs) { return ((size_t)kHeaderWords + (size_t)slabs) * 8; }

#if defined(__HIPCC__) || defined(FCD_HIPEMU)
// a free slab's id; every lane gets it (called in uniform control flow)
__device__ __forceinline__ int pop(unsigned long long *pool, int lane) {
    int id = 0;
    if (lane == 0) {
        const unsigned long long P = pool[2];
        const unsigned long long t = atomicAdd(&pool[0], 1ull);
        unsigned long long *const cell = pool + kHeaderWords + (t % P);
        const uint32_t want = (uint32_t)(t + 1);
        unsigned long long v = __atomic_load_n(cell, __ATOMIC_RELAXED);
        while ((uint32_t)(v >> 32) != want) {
            __builtin_amdgcn_s_sleep(8);
            v = __atomic_load_n(cell, __ATOMIC_RELAXED);
        }
        id = (int)(uint32_t)v;
        __atomic_store_n(cell, (unsigned long long)(uint32_t)(t + P) << 32, __ATOMIC_RELAXED);  // free for push ticket t + P
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
        unsigned long long *const cell = pool + kHeaderWords + (t % P);
        const uint32_t want = (uint32_t)t;
        while ((uint32_t)(__atomic_load_n(cell, __ATOMIC_RELAXED) >> 32) != want) __builtin_amdgcn_s_sleep(8);  // (its reader is about to)
        __atomic_store_n(cell, ((unsigned long long)(uint32_t)(t + 1) << 32) | (uint32_t)id, __ATOMIC_RELAXED);
    }
}
#endif

}  // namespace slab_pool
}  // namespace fcd
