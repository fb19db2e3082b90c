// Device side of the one-shot peer-to-peer gradient exchange over xGMI.
//
// Every rank owns an UNCACHED device buffer of 8-byte granules {tag = epoch, value = float bits},
// [2 banks][world][NG]; all ranks map all buffers through HIP IPC.  To sum a value across ranks a lane
//   1. stores {epoch, its value} into slot [epoch&1][my rank][idx] of EVERY rank's buffer (one aligned
//      8-byte system-scope store each: the data IS the flag -- no separate flag, fence or ordering),
//   2. polls its OWN buffer's W slots for idx until each tag equals the epoch,
//   3. adds the W values in rank order (every rank adds the same bits in the same order: replicas stay
//      bit-identical, its own contribution included since it is read back from its own slot).
// A rank can run at most one epoch ahead of a peer (it needs that peer's granule to finish an epoch),
// so two banks by epoch parity are enough.  Spins are bounded; on give-up the status word is set and
// the host sees it (vaek_comm_status) -- nothing can hang the GPU for good.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vaek {

constexpr int kMaxWorld = 8;
constexpr unsigned kSpinLimit = 1u << 23;      // x (uncached load + s_sleep 8, ~0.4 us measured): a lane gives up after ~3 s in total.
                                               // Every exchange follows a host-side barrier or the previous exchange, so
                                               // ranks are milliseconds apart; once any exchange on this rank has given up
                                               // (status word), later ones do not wait at all -- a dead link costs seconds,
                                               // not world x steps x the limit, before the host falls back to RCCL.

struct CommDev {
    unsigned long long* peer[kMaxWorld];      // peer[r] = rank r's granule region (peer[rank] = local)
    unsigned int* status;                     // local: set to 1 on spin give-up
    int world, rank, ng;                      // ng = granules per (bank, source rank)
};

__device__ __forceinline__ float comm_exchange_sum(const CommDev& c, unsigned epoch, int idx, float v) {
    const unsigned long long mine = ((unsigned long long)epoch << 32) | (unsigned long long)__float_as_uint(v);
    const long long slot = ((long long)(epoch & 1u) * c.world) * c.ng + idx;
    for (int p = 0; p < c.world; ++p)
        __hip_atomic_store(c.peer[p] + slot + (long long)c.rank * c.ng, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const unsigned long long* own = c.peer[c.rank] + slot;
    float sum = 0.f;
    unsigned spins = 0;                       // shared by the polls of all ranks: the bound is per exchange
    bool dead = false;
    for (int r = 0; r < c.world; ++r) {
        unsigned long long g = 0;
        for (;;) {
            g = __hip_atomic_load(own + (long long)r * c.ng, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if ((unsigned)(g >> 32) == epoch || dead) break;
            if (spins == 0) dead = __hip_atomic_load(c.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
            if (++spins > kSpinLimit) { __hip_atomic_store(c.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); dead = true; }
            __builtin_amdgcn_s_sleep(8);
        }
        sum += __uint_as_float((unsigned)g);
    }
    return sum;
}

}  // namespace vaek
