// N consecutive train steps of a LINEAR VAE (encoder = one Dense D -> L, decoder = one Dense L -> D: every "" layer-size
// experiment of seed_linpadding_expts.sh, the configuration the headline metric is quoted on), software-pipelined over
// launches: vaek_train_steps (include/vaek.h).  Same function as N calls of VAE.train_step (networks.py:87-101: forward,
// ELBO, value_and_grad, Adam), evaluated through its sufficient statistic.
//
// Why this is exact.  With no non-linearity between input and loss, every per-sample quantity is LINEAR in
//     u_b = [z1_b (L) | x_b (D) | z2_b (D) | 1]                                   (NF = L + 2 D + 1 features):
//     mu_b = E u_b,        E = [0 | We^T | 0 | be]                                  (networks.py:67-68)
//     samples_b = S u_b,   S = E + [diag(e^{lv/2}) | 0 | 0 | 0]                     (networks.py:73-74)
//     r_b = x_hat_b - x_b = R u_b,   R = [Wd^T diag(e^{lv/2}) | Wd^T We^T - I | e^{eps/2} I | Wd^T be + bd]   (:80-83)
// and the loss and every gradient of SURVEY.md 8(a) row a5 are quadratic in u_b summed over the batch, i.e. functions of
//     M = sum_b u_b u_b^T                                                           (NF x NF, parameter-INDEPENDENT)
// and of the parameters alone:  sum_b |r_b|^2 = tr(R M R^T),  sum_b |mu_b|^2 = tr(E M E^T),  dWd = c0 S M R^T,
// [dWe | dbe] = columns of c0 Wd R M + E M / B,  d lv_l = 1/2 e^{lv_l/2} c0 (Wd R M)[l, z1_l] - 1/2 (1 - e^{lv_l}),
// d eps = (-1/2 tr(RMR^T) e^{-eps} + 1/2 B D + 1/2 e^{-eps/2} sum_d (R M)[d, z2_d]) / B,  c0 = e^{-eps} / B.
//
// What it buys.  The streaming pass over the batch (11.5 MB at the metric's size, the whole algorithmic traffic) no longer
// depends on the parameters, so nothing couples the 256 workgroups of a step to the previous step's all-reduce -> Adam ->
// broadcast: the two-kernel step (fused_mfma.hip + fused_finalize) spends most of its 13 us waiting on exactly that chain.
// Here launch n carries three roles at once (the persistent form below: wave-specialised streamers, the updater on the float64
// matrix cores -- DESIGN.md 3.0):
//     streamers   (one per 256-sample tile)  batch n:   x, z1, z2 tiles land in LDS by global_load_lds, then
//                 M_tile = U^T U on v_mfma_f32_16x16x4_f32 (exact f32 fmaf chains; upper block triangle only) -> one
//                 partial image per workgroup.  No weights, no elementwise pass, no transposition (the k axis of the MFMA
//                 is the sample axis and any sample order will do: operands are read straight from the row-major images).
//     reducers    batch n-1: fixed-order float64 sum of the partial images -> M.
//     updater     (one workgroup) batch n-2: the small dense algebra above in float64, the closed-form KL terms, the three
//                 loss means, Adam (flax.optim.Adam.apply_gradient, networks.py:100), the step counter, the loss ring.
// In that launch-per-step form stream order is the only synchronisation.  The PERSISTENT form (default where it applies) runs
// up to 64 steps in ONE launch with the same three roles as resident workgroups: streamers walk the batches with the tile of
// batch n + 1 / n + 2 in flight while batch n is multiplied, reducers and the updater follow behind through per-batch arrival
// counters (write-through partial images, one counter add per workgroup, relaxed polls: cdna guide G16 R1).  Every batch of
// the launch has its OWN partial / M slot, so a streamer never waits for anybody: nothing can dead-lock, whatever the
// dispatcher does with residency, and every poll is bounded (status word, vaek_train_steps_status).  What it buys on top:
// no launch boundary and no cold start per step (the updater's instruction stream and parameters stay on one CU).
//
// Numerics: M accumulates exact-f32 products in chains of 72 samples (one multiplying wave's 18 MFMA k-steps at the metric's
// 288-row tile; 36 in the launch-per-step form), summed further in
// float64; everything downstream is float64 until the final rounding of each gradient to float32.  Against the float64
// oracle the loss sits at ~1e-7 relative (tests/test_gpu_steps.py), like the sample-by-sample kernels -- but it is a
// different summation order, so this path is NOT bitwise comparable with vaek_train_step.
#include <stdlib.h>

#include <type_traits>

#include "comm_dev.h"
#include "rng_dev.h"
#include "vaek_internal.h"

namespace vaek {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using d2 = __attribute__((ext_vector_type(2))) double;
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;

constexpr int LNT = 512, LNW = LNT / 64;          // threads / waves per workgroup, every role
constexpr int kLinMaxPersist = 64;                // steps per persistent launch (each owns a partial / M slot)
constexpr int kLinReduceSets = 1;                 // persistent form: reducer sets taking alternate batches (one set of 24 beat two of 12)
constexpr int kLinReduceWgs = 24;                 // workgroups per set, each summing NO / 32 / 24 slices of 32 outputs (more resident
                                                  // workgroups measurably slow the streamers: 48 per set cost 2 us per step)

// Data parallel (world > 1): the moment matrix is additive over ranks.  Each reducer publishes its slice of the rank's M to every
// rank's exchange buffer (comm_dev.h's scheme: 8-byte granules {tag = the batch's Adam step, 32 bits of payload} in uncached,
// IPC-mapped memory, one aligned system-scope store each -- the data is the flag), waits for the same slice from all ranks in
// its own buffer, and adds them in rank order: every rank ends with the same bits, and the updater never learns that other
// ranks exist.  A double travels as two granules.  A reducer cannot get more than one batch ahead of a peer's (it needs that
// peer's granules to finish a batch), so banks by tag & 3 are never overwritten before they are read.
constexpr int kLinCommBanks = 4;
struct LinComm { unsigned long long* peer[kMaxWorld]; int world, rank, ng2; };     // ng2 = granules per (bank, source rank) = 2 NO

constexpr int kLinShards = 8, kLinShardStride = 32;   // words
struct LinPtrs { const float* x[64]; const float* z1[64]; const float* z2[64]; };    // batch pointers of a persistent launch (kernarg)
struct LinArgs {
    // roles by blockIdx.x: [0, has_update) the updater, then n_reduce reducers, then n_stream streamers
    int has_update, n_reduce, n_stream;
    int B, D, L, ntiles, T;                       // T: samples per tile (a multiple of 32, <= 512)
    // ---- launch-per-step form: streamers take THE batch of this launch, reducers the one before, the updater the one before that
    const float* x; const float* z1; const float* z2; float* partial_out;        // [ntiles][NBLK * 256]
    const float* partial_in; double* M_out;                                       // [NBLK * 256]
    const double* M_in;
    // ---- persistent form: n_steps batches, pointer tables in device memory, one slot per batch, arrival counters
    int persistent, n_steps, sets;                // sets: reducer sets taking alternate batches
    float* partial_base; double* M_base;                                          // slot n at + n * ntiles * NO resp. + n * NO
    // arrival counters.  cnt_stream: kLinShards shards per batch, each on a 128-byte line of its own (a streamer adds to shard
    // blockIdx & 7: 228 adders on ONE word serialise at the memory side -- the reducers woke 8 us after the last streamer had
    // signalled); cnt_reduce: one word per batch.  All zero when a launch starts: the updater re-zeroes what the launch used as
    // its last act (everybody else is provably done with them by then), lin_init_kernel zeroes them once per workspace.
    unsigned* cnt_stream; unsigned* cnt_reduce; unsigned* status;                 // status: sticky, outside the zeroed range
    // ---- updater
    float* params; float* grads; float* m; float* v; int32_t* step_dev; float lr;
    float inv_bt, eps_cli, rows, rows_over_bt; int off_eps, P;
    float* loss_hist; long long loss_hist_cap;
    int stagger;                                  // streamers: waves 4 .. 7 enter a tile's products this many 64-cycle sleeps late (see there)
    LinComm comm;
};

#ifndef VAEK_LIN_ABL         // diagnostic builds (tools/lin_ablate.sh): 1 pieces issued in front of the products, 2 no MFMAs, 4 no image store,
#define VAEK_LIN_ABL 0       //   8 no LDS-DMA at all (the slots keep whatever they hold).  Results are garbage unless 0.
#endif
#ifdef VAEK_LIN_STAMPS      // diagnostic build (tools/lin_stamps.sh): s_memtime at the updater's phase boundaries, into a buffer nothing reads
__device__ unsigned long long* g_lin_stamp_buf = nullptr;
#define LIN_STAMP(i)                                                                                         \
    do {                                                                                                     \
        if (VAEK_LIN_STAMPS == 2) break;      /* light build: only the drain-free time stamps (LIN_NOWQ) */    \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
        unsigned long long _t;                                                                               \
        unsigned long long _r;                                                                               \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t), "=s"(_r)::"memory"); \
        if (g_lin_stamp_buf && threadIdx.x == 0) { g_lin_stamp_buf[i] = _t; g_lin_stamp_buf[16 + (i)] = _r; } \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
    } while (0)
#define LIN_NOW(v)                                                                                           \
    do {                                                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
        if (VAEK_LIN_STAMPS == 2) asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v)::"memory");             \
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v)::"memory"); \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
    } while (0)
#define LIN_NOWQ(v)  /* no vmcnt wait: does not drain the wave's stores */                                   \
    do {                                                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v)::"memory");                        \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
    } while (0)
#define LIN_PUT(i, v) do { if (g_lin_stamp_buf && threadIdx.x == 0) g_lin_stamp_buf[i] = (v); } while (0)
#define LIN_PUTMAX(i, v) do { if (g_lin_stamp_buf && threadIdx.x == 0) atomicMax(&g_lin_stamp_buf[i], (v)); } while (0)
#else
#define LIN_STAMP(i) do {} while (0)
#define LIN_NOW(v) do {} while (0)
#define LIN_NOWQ(v) do {} while (0)
#define LIN_PUT(i, v) do {} while (0)
#define LIN_PUTMAX(i, v) do {} while (0)
#endif

__device__ __forceinline__ void lin_glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((glb_void_t*)gsrc, (lds_void_t*)lds_wave_base, 16, 0, 0);
}
template <int N> __device__ __forceinline__ void lin_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// Workgroup barrier that leaves LDS-DMA loads in flight: __syncthreads() would drain them (its fence waits vmcnt(0) while a
// global_load_lds is pending: cdna guide, LDS-DMA rules); LDS traffic of this wave is waited for explicitly.
__device__ __forceinline__ void lin_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// wait until at most n (wave-uniform, <= 20) of this wave's memory operations are outstanding
__device__ __forceinline__ void lin_wait_vmcnt_upto(int n) {
    switch (n) {
        case 1: lin_wait_vmcnt<1>(); break;   case 2: lin_wait_vmcnt<2>(); break;   case 3: lin_wait_vmcnt<3>(); break;
        case 4: lin_wait_vmcnt<4>(); break;   case 5: lin_wait_vmcnt<5>(); break;   case 6: lin_wait_vmcnt<6>(); break;
        case 7: lin_wait_vmcnt<7>(); break;   case 8: lin_wait_vmcnt<8>(); break;   case 9: lin_wait_vmcnt<9>(); break;
        case 10: lin_wait_vmcnt<10>(); break; case 11: lin_wait_vmcnt<11>(); break; case 12: lin_wait_vmcnt<12>(); break;
        case 13: lin_wait_vmcnt<13>(); break; case 14: lin_wait_vmcnt<14>(); break; case 15: lin_wait_vmcnt<15>(); break;
        case 16: lin_wait_vmcnt<16>(); break; case 17: lin_wait_vmcnt<17>(); break; case 18: lin_wait_vmcnt<18>(); break;
        case 19: lin_wait_vmcnt<19>(); break; case 20: lin_wait_vmcnt<20>(); break;
        default: lin_wait_vmcnt<0>(); break;
    }
}

// block (b1, b2), b1 <= b2, of the upper block triangle -> its index in the packed image
__device__ __host__ constexpr int lin_blk(int NB, int b1, int b2) { return b1 * NB - b1 * (b1 - 1) / 2 + (b2 - b1); }

// write-through (sc1) accesses of the in-launch hand-offs: relaxed agent-scope atomics lower to global_store / global_load ... sc1
__device__ __forceinline__ void st_sc1(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_sc1(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ld_sc1(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double ld_sc1(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// All threads call; thread 0 polls `*cnt >= target` (relaxed, with s_sleep; bounded: ~2 s, then the status word is set and
// every later wait of the launch returns at once so the grid drains), the workgroup barrier publishes the outcome.  The first
// wait to expire records who it was: 0x80000000 | role << 28 (1 updater, 2 reducer) | batch << 16 | the count it last saw.
__device__ __forceinline__ void lin_wait_count(const unsigned* cnt, unsigned target, unsigned* status, unsigned tag) {
    if (threadIdx.x == 0) {
        unsigned spins = 0, seen;
        while ((seen = __hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < target) {
            __builtin_amdgcn_s_sleep(16);
            if ((++spins & 1023u) == 0) {
                if (__hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
                if (spins > (1u << 21)) {
                    unsigned expect = 0;
                    __hip_atomic_compare_exchange_strong(status, &expect, 0x80000000u | tag | (seen & 0xffffu), __ATOMIC_RELAXED,
                                                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
            }
        }
    }
    __syncthreads();
}

// the same on a batch's SHARDED streamer counter: thread 0 keeps all shards' loads in flight together and compares their sum
__device__ __forceinline__ void lin_wait_shards(const unsigned* cnt, unsigned target, unsigned* status, unsigned tag) {
    if (threadIdx.x == 0) {
        unsigned spins = 0, seen;
        for (;;) {
            unsigned v[kLinShards];
#pragma unroll
            for (int k = 0; k < kLinShards; ++k) v[k] = __hip_atomic_load(cnt + k * kLinShardStride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            seen = 0;
#pragma unroll
            for (int k = 0; k < kLinShards; ++k) seen += v[k];
            if (seen >= target) break;
            __builtin_amdgcn_s_sleep(4);
            if ((++spins & 1023u) == 0) {
                if (__hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
                if (spins > (1u << 21)) {
                    unsigned expect = 0;
                    __hip_atomic_compare_exchange_strong(status, &expect, 0x80000000u | tag | (seen & 0xffffu), __ATOMIC_RELAXED,
                                                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
            }
        }
    }
    __syncthreads();
}

// one output of M across the ranks: publish this rank's value, collect everybody's, add in rank order (bounded spins)
__device__ __forceinline__ double lin_sum_over_ranks(const LinComm& c, unsigned epoch, int o, double v, unsigned* status) {
    const unsigned long long bits = (unsigned long long)__double_as_longlong(v), tag = (unsigned long long)epoch << 32;
    const long long bank = (long long)(epoch & (kLinCommBanks - 1)) * c.world * c.ng2;
    for (int p = 0; p < c.world; ++p) {
        unsigned long long* q = c.peer[p] + bank + (long long)c.rank * c.ng2 + 2 * o;
        __hip_atomic_store(q, tag | (bits & 0xffffffffull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(q + 1, tag | (bits >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    const unsigned long long* own = c.peer[c.rank] + bank + 2 * o;
    double sum = 0.0;
    unsigned spins = 0;
    bool dead = false;
    for (int r = 0; r < c.world; ++r) {
        unsigned long long lo = 0, hi = 0;
        for (;;) {
            lo = __hip_atomic_load(own + (long long)r * c.ng2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            hi = __hip_atomic_load(own + (long long)r * c.ng2 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if (((unsigned)(lo >> 32) == epoch && (unsigned)(hi >> 32) == epoch) || dead) break;
            __builtin_amdgcn_s_sleep(8);
            if ((++spins & 1023u) == 0) {
                if (__hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) dead = true;
                else if (spins > (1u << 21)) {
                    unsigned expect = 0;
                    __hip_atomic_compare_exchange_strong(status, &expect, 0x80000000u | (3u << 28) | ((unsigned)r << 16) | (epoch & 0xffffu),
                                                         __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    dead = true;
                }
            }
        }
        sum += __longlong_as_double((long long)((hi << 32) | (lo & 0xffffffffull)));
    }
    return sum;
}

// ---- streamer pieces ------------------------------------------------------------------------------------------------------------
// LDS image of one T-sample tile: the three tensors' tiles exactly as they lie in HBM (row-major [T][L] / [T][D]), each region a
// whole number of 1 KB PIECES (one LDS-DMA wave-instruction: 64 lanes x 16 bytes).  Piece p of a tile goes to wave p % 8.  The
// validity column V[T] (the "1" feature; 0 for rows past the batch end) and a zero word live outside the slots: a full tile's
// column is all ones and is written once per launch.
struct LinTile {
    int nz1, nx, np, oX, oZ2, bytes;
    __device__ __host__ LinTile(int D, int L, int T) {
        nz1 = (L * 4 * T + 1023) >> 10; nx = (D * 4 * T + 1023) >> 10; np = nz1 + 2 * nx;
        oX = (nz1 << 10) + 80;      // + 20 banks: the feature block that mixes z1 and x columns then reads conflict-free in 3 k-steps of 4 (was 2 of 4)
        oZ2 = oX + (nx << 10); bytes = (oZ2 + (nx << 10) + 255) & ~255;
    }
};

// piece p (wave-uniform) of tile `tile`: 1 KB of z1 / x / z2 from HBM straight into the slot
__device__ __forceinline__ void lin_issue_piece(const LinArgs& a, const LinTile& tl, const float* x, const float* z1, const float* z2,
                                                int tile, int p, char* slot, int lane) {
    const float* src; int cols, idx, lds_off;
    if (p < tl.nz1) { src = z1; cols = a.L; idx = p; lds_off = 0; }
    else if (p < tl.nz1 + tl.nx) { src = x; cols = a.D; idx = p - tl.nz1; lds_off = tl.oX; }
    else { src = z2; cols = a.D; idx = p - tl.nz1 - tl.nx; lds_off = tl.oZ2; }
    const long long tot = (long long)a.B * cols * 4, base = (long long)tile * a.T * cols * 4;
    // the last 16-byte piece this tile may fetch: the tile's own (a region's padding up to whole pieces re-reads it -- a line this
    // CU has just fetched -- instead of pulling the NEXT tile's rows through another XCD's L2), and never past the tensor's last
    // whole 16 bytes; what a clamped piece brings lands in LDS nobody reads, in rows that are zeroed afterwards, or in the
    // tensor's last <= 3 floats, which lin_fix_ragged rewrites
    const long long last = min(tot & ~15ll, base + (long long)a.T * cols * 4) - 16;
    long long off = base + idx * 1024 + lane * 16;
    off = off <= last ? off : last;
    lin_glds16(reinterpret_cast<const char*>(src) + off, slot + lds_off + idx * 1024);
}

// The same with everything that does not depend on the piece worked out once per tile (the persistent streamers issue the pieces
// of a tile one by one between the k-steps of an earlier one: what is left per piece is a handful of scalar selects, one 64-bit
// add, one clamp).  All members are wave-uniform.
struct LinTileSrc {
    const char *z1, *x, *z2; long long baseL, lastL, baseD, lastD; char* slot; bool on;
    __device__ __forceinline__ void prepare(const LinArgs& a, const float* px, const float* pz1, const float* pz2, int tile, char* slot_) {
        auto uni = [](const float* q) {
            const unsigned long long u = (unsigned long long)q;
            // (readfirstlane returns int: through unsigned, or a low word >= 2^31 sign-extends into the high one)
            return reinterpret_cast<const char*>((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)u) |
                                                 ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(u >> 32)) << 32));
        };
        z1 = uni(pz1); x = uni(px); z2 = uni(pz2); slot = slot_; on = true;
        const long long rows = (long long)tile * a.T;
        const long long totL = (long long)a.B * a.L * 4, totD = (long long)a.B * a.D * 4;
        baseL = rows * a.L * 4; baseD = rows * a.D * 4;
        lastL = min(totL & ~15ll, baseL + (long long)a.T * a.L * 4) - 16;
        lastD = min(totD & ~15ll, baseD + (long long)a.T * a.D * 4) - 16;
    }
    __device__ __forceinline__ void issue(const LinTile& tl, int p, int lane) const {      // piece p (wave-uniform)
        const char* src; long long base, last; int idx, lds_off;
        if (p < tl.nz1) { src = z1; base = baseL; last = lastL; idx = p; lds_off = 0; }
        else if (p < tl.nz1 + tl.nx) { src = x; base = baseD; last = lastD; idx = p - tl.nz1; lds_off = tl.oX; }
        else { src = z2; base = baseD; last = lastD; idx = p - tl.nz1 - tl.nx; lds_off = tl.oZ2; }
        long long off = base + idx * 1024 + lane * 16;
        off = off <= last ? off : last;
        lin_glds16(src + off, slot + lds_off + idx * 1024);
    }
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"      // (M0 on the clobber list: it is what the LDS-DMA takes its LDS address from)
    __device__ __forceinline__ void issue_tile(const LinTile& tl, int lw, int NLW, int lane) const {
        auto region = [&](const char* src, long long base, long long last, int n, int lds_off) __attribute__((always_inline)) {
            const long long b2 = base <= last ? base : last;           // (a last tile shorter than 16 bytes: every lane reads the tensor's last 16)
            const char* sbase = src + b2;
            const unsigned lastrel = (unsigned)(last - b2), degenerate = base <= last ? 0xffffffffu : 0u;
            const unsigned m0_0 = (unsigned)(size_t)(lds_void_t*)(slot + lds_off);
            for (int idx = lw; idx < n; idx += NLW) {
                unsigned off = ((unsigned)idx * 1024u + (unsigned)lane * 16u) & degenerate;
                off = off <= lastrel ? off : lastrel;
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(off), "s"(sbase), "s"(m0_0 + (unsigned)idx * 1024u) : "memory", "m0");
            }
        };
        region(z1, baseL, lastL, tl.nz1, 0);
        region(x, baseD, lastD, tl.nx, tl.oX);
        region(z2, baseD, lastD, tl.np - tl.nz1 - tl.nx, tl.oZ2);
    }
#pragma clang diagnostic pop
};

// the last tile of a ragged batch, after it has landed: the tensors' last floats behind their last whole 16 bytes, zeros in the
// rows past the batch end
__device__ __forceinline__ void lin_fix_ragged(const LinArgs& a, const LinTile& tl, const float* x, const float* z1, const float* z2,
                                               int tile, char* slot, int t) {
    const int T = a.T;
    const long long row0 = (long long)tile * T;
    const int valid = (int)min((long long)T, (long long)a.B - row0), D = a.D, L = a.L;
    if (valid < T) {
        auto patch_tail = [&](const float* src, int cols, int lds_off) {     // floats behind the tensor's last whole 16 bytes
            const long long tot = (long long)a.B * cols * 4, full = tot & ~15ll, base = row0 * cols * 4;
            if (t < (int)((tot - full) / 4) && full >= base) reinterpret_cast<float*>(slot + lds_off)[(full - base) / 4 + t] = src[full / 4 + t];
        };
        patch_tail(z1, L, 0); patch_tail(x, D, tl.oX); patch_tail(z2, D, tl.oZ2);
        for (int e = valid * L + t; e < T * L; e += LNT) reinterpret_cast<float*>(slot)[e] = 0.f;
        for (int e = valid * D + t; e < T * D; e += LNT) { reinterpret_cast<float*>(slot + tl.oX)[e] = 0.f; reinterpret_cast<float*>(slot + tl.oZ2)[e] = 0.f; }
    }
    lin_barrier();
}
// validity column of a tile with `valid` rows
__device__ __forceinline__ void lin_write_vcol(float* v, int T, int valid, int t) {
    for (int r = t; r < T; r += LNT) v[r] = r < valid ? 1.f : 0.f;
}

// 16-byte write-through store (the compiler does not count it: every wait on it is ours)
__device__ __forceinline__ void st_sc1_x4(float* p, f32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}

// M_tile = U^T U.  The SAMPLES are dealt to the waves: wave w takes samples (T / 8) w .. -- T / 32 k-steps of 4 samples -- for
// every block of the upper block triangle, so each operand register read from LDS feeds NB (+1) MFMAs, the matrix pipes of
// the four SIMDs carry equal loads and the NBLK accumulator chains are independent.  hook(j) runs between the operand reads
// and the products of k-step j (the persistent streamers issue the LDS-DMA pieces of a later tile there: in the shadow of the
// matrix pipe).  JT > 0: the k-step count at compile time -- the loop unrolls, the reads carry immediate offsets and the waits
// are counted (with a run-time count hipcc waits lgkmcnt(0) before every group of products, prefetched operands included).
constexpr int lin_scratch_bytes(int NB) { return LNW * (NB * (NB + 1) / 2) * 1024; }       // (at 8 images; the persistent form's 4 need half)
constexpr int kLinCW = 4;      // persistent form: waves 0 .. 3 multiply (one per SIMD), waves 4 .. 7 load
template <int NB, int JT, int CW, typename Hook>
__device__ __forceinline__ void lin_tile_products(const LinArgs& a, const LinTile& tl, const char* smem, int slot_off, int v_off, int c_off,
                                                  f32x4 (&acc)[NB * (NB + 1) / 2], int lane, int wave, Hook&& hook) {
    constexpr int NBLK = NB * (NB + 1) / 2;
    const int g = lane >> 4, D = a.D, L = a.L;
    // where this lane's feature 16 u + (lane & 15) lives (byte offset of sample 0, byte stride per sample)
    int fb[NB], fs[NB];
#pragma unroll
    for (int u = 0; u < NB; ++u) {
        const int f = 16 * u + (lane & 15);
        if (f < L) { fb[u] = slot_off + f * 4; fs[u] = L * 4; }
        else if (f < L + D) { fb[u] = slot_off + tl.oX + (f - L) * 4; fs[u] = D * 4; }
        else if (f < L + 2 * D) { fb[u] = slot_off + tl.oZ2 + (f - L - D) * 4; fs[u] = D * 4; }
        else if (f == L + 2 * D) { fb[u] = v_off; fs[u] = 4; }
        else { fb[u] = c_off; fs[u] = 0; }
    }
#pragma unroll
    for (int k = 0; k < NBLK; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    // Wave w takes samples (T / CW) w .. + T / CW - 1 in J = T / (4 CW) k-steps of 4.  Within a run of 16 samples lane group g takes sample
    // s + 4 g in step s: four rows apart, i.e. 16 banks with 80- and 48-byte rows, so the lane groups one ds_read_b32 services
    // together never collide; a run shorter than 16 (T / 8 not a multiple of 16: its last r < 4 steps) takes s + r g.
    // (CW: the waves that share a tile's samples -- all 8 of the launch-per-step form, the 4 compute waves of the persistent one)
    const int J = JT ? JT : a.T / (4 * CW), wbase = (a.T / CW) * wave, jfull = J & ~3;
    auto products = [&](const float (&op)[NB]) __attribute__((always_inline)) {
        int k = 0;
#pragma unroll
        for (int b1 = 0; b1 < NB; ++b1)
#pragma unroll
            for (int b2 = b1; b2 < NB; ++b2, ++k) {
                if (VAEK_LIN_ABL & 2) acc[k][0] += op[b1] * op[b2];
                else acc[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(op[b1], op[b2], acc[k], 0, 0, 0);
            }
    };
    if constexpr (JT > 0) {
        // per-lane address of every feature's operand, advanced from k-step to k-step by adds only: + one sample inside a run of
        // four steps, + 13 samples from one run to the next, the short last run from its own base
        constexpr int JF = JT & ~3;
        int cur[NB], fs13[NB], tail[NB];
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            cur[u] = fb[u] + (wbase + 4 * g) * fs[u]; fs13[u] = 13 * fs[u];
            tail[u] = fb[u] + (wbase + 4 * JF + (JT - JF) * g) * fs[u];
        }
        float op[JT][NB];
        auto fetch = [&](int j) __attribute__((always_inline)) {                              // called for j = 0, 1, 2, ... in order
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                if (j == JF) cur[u] = tail[u];
                op[j][u] = *reinterpret_cast<const float*>(smem + cur[u]);
                cur[u] += (j < JF && (j & 3) == 3) ? fs13[u] : fs[u];
            }
        };
        fetch(0);
        if (JT > 1) fetch(1);
#pragma unroll
        for (int j = 0; j < JT; ++j) {
            if (j + 2 < JT) fetch(j + 2);                      // operands two k-steps ahead
            hook(j);
            __builtin_amdgcn_sched_barrier(0);
            products(op[j]);
            __builtin_amdgcn_sched_barrier(0);
        }
    } else {
        // Operands run one step ahead in a second register set, and the scheduler is told to keep it that way: left alone hipcc
        // reuses one register set and waits out the full LDS latency before every MFMA.
        float op[2][NB];
        auto fetch = [&](int set, int j) __attribute__((always_inline)) {
            const int sample = wbase + (j < jfull ? 16 * (j >> 2) + (j & 3) + 4 * g : 4 * jfull + (j - jfull) + (J - jfull) * g);
#pragma unroll
            for (int u = 0; u < NB; ++u) op[set][u] = *reinterpret_cast<const float*>(smem + fb[u] + sample * fs[u]);
        };
        fetch(0, 0);
        for (int j = 0; j < J; j += 2) {
            if (j + 1 < J) fetch(1, j + 1);
            hook(j);
            __builtin_amdgcn_sched_barrier(0);                 // the reads of step j + 1 are issued before the MFMAs of step j
            products(op[0]);
            __builtin_amdgcn_sched_barrier(0);
            if (j + 1 < J) {
                if (j + 2 < J) fetch(0, j + 2);
                hook(j + 1);
                __builtin_amdgcn_sched_barrier(0);
                products(op[1]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
}

// The eight per-wave images are summed through LDS (in wave order: deterministic) in `scratch` -- the tile's own slot, dead once
// every wave has passed the barrier behind its products -- and leave as ONE image in the accumulators' own layout
// [block][lane][4] (image element (row, col) of a block sits at ((row >> 2) * 16 + col) * 4 + (row & 3): lin_img_index), so the
// sum reads and the store are 16 bytes per lane.
template <int NB, bool SC1, int CW>
__device__ __forceinline__ void lin_tile_combine(const f32x4 (&acc)[NB * (NB + 1) / 2], char* scratch, float* out, int t, int wave, int lane) {
    constexpr int NBLK = NB * (NB + 1) / 2;
    f32x4* scr = reinterpret_cast<f32x4*>(scratch);       // [wave][block][lane]
    if (wave < CW) {
#pragma unroll
        for (int k = 0; k < NBLK; ++k) scr[(wave * NBLK + k) * 64 + lane] = acc[k];
    }
    lin_barrier();
    for (int q = t; q < NBLK * 64; q += LNT) {
        f32x4 sum = scr[q];
#pragma unroll
        for (int w = 1; w < CW; ++w) sum += scr[w * NBLK * 64 + q];
        if ((VAEK_LIN_ABL & 4) && sum[0] != 12345.f) continue;
        if (SC1) st_sc1_x4(out + 4 * q, sum); else *reinterpret_cast<f32x4*>(out + 4 * q) = sum;
    }
}
// (block, row, col) of image element e in that layout
__device__ __forceinline__ void lin_img_coords(int e, int& blk, int& i, int& j) {
    blk = e >> 8; const int ln = (e >> 2) & 63; i = ((ln >> 4) << 2) + (e & 3); j = ln & 15;
}

// ---- reducer: 32 outputs x 16 row groups per workgroup, float64, fixed order ------------------------------------------------
template <bool SC1>
__device__ __forceinline__ void lin_reduce(const float* partial_in, double* M_out, int ntiles, char* smem, int rb, int no,
                                           const LinComm* cm = nullptr, unsigned epoch = 0, unsigned* status = nullptr) {
    double* sums = reinterpret_cast<double*>(smem);       // [16][32]
    const int t = threadIdx.x, o = rb * 32 + (t & 31), rg = t >> 5;
    const int rpg = (ntiles + 15) / 16, r_lo = rg * rpg, r_hi = min(ntiles, r_lo + rpg);
    double s = 0.0;
    if (o < no)
        for (int r0 = r_lo; r0 < r_hi; r0 += 16) {
            float tv[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {                 // unconditional, clamped
                const float* q = partial_in + (long long)min(r0 + u, r_hi - 1) * no + o;
                tv[u] = SC1 ? ld_sc1(q) : *q;
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) s += r0 + u < r_hi ? (double)tv[u] : 0.0;
        }
    sums[rg * 32 + (t & 31)] = s;
    __syncthreads();
    if (t < 32 && o < no) {
        double tot = 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k) tot += sums[k * 32 + t];
        if (cm && cm->world > 1) tot = lin_sum_over_ranks(*cm, epoch, o, tot, status);
        if (SC1) st_sc1(M_out + o, tot); else M_out[o] = tot;
    }
}

// two 32-output slices at once (the persistent form's reducers own two each): both slices' loads are in flight together, one
// memory round trip per batch instead of two.  Same row groups, same order of additions as lin_reduce: bitwise its sums.
__device__ __forceinline__ void lin_reduce_pair(const float* partial_in, double* M_out, int ntiles, char* smem, int rb0, int rb1, int no,
                                                const LinComm& cm, unsigned epoch, unsigned* status) {
    double* sums = reinterpret_cast<double*>(smem);       // [2][16][32]
    const int t = threadIdx.x, q = t & 31, rg = t >> 5, o0 = rb0 * 32 + q, o1 = rb1 * 32 + q;
    const int rpg = (ntiles + 15) / 16, r_lo = rg * rpg, r_hi = min(ntiles, r_lo + rpg);
    double s0 = 0.0, s1 = 0.0;
    for (int r0 = r_lo; r0 < r_hi; r0 += 16) {
        float ta[16], tb[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {                     // unconditional, clamped
            const float* row = partial_in + (long long)min(r0 + u, r_hi - 1) * no;
            ta[u] = ld_sc1(row + o0); tb[u] = ld_sc1(row + o1);
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) { s0 += r0 + u < r_hi ? (double)ta[u] : 0.0; s1 += r0 + u < r_hi ? (double)tb[u] : 0.0; }
    }
    sums[rg * 32 + q] = s0; sums[512 + rg * 32 + q] = s1;
    __syncthreads();
    if (t < 64) {
        const int which = t >> 5;
        double tot = 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k) tot += sums[which * 512 + k * 32 + q];
        if (cm.world > 1) tot = lin_sum_over_ranks(cm, epoch, which ? o1 : o0, tot, status);
        st_sc1(M_out + (which ? o1 : o0), tot);
    }
}

// ---- updater: gradients and loss from (M, parameters), Adam -----------------------------------------------------------------
// R and E are never formed.  With S = E + [diag(s) | 0 | 0 | 0] (samples = S u) and r = Wd^T samples + bd - x + sigma z2:
//     SM = S M                      (L x NF; S has D + 2 non-zeros per row)          Q = E M = SM - diag(s) M[z1 rows]
//     P1 = R M = Wd^T SM - M[x rows] + sigma M[z2 rows] + bd M[one row]              (D x NF)
//     G  = Wd P1                    (L x NF)          dwd[l][d] = S[l,:] . P1[d,:]   (= sum_b samples_bl r_bd)
//     sum_b |r_b|^2 = tr(R M R^T) = sum_{l,d} Wd[l][d] dwd[l][d] + sum_d (-P1[d][x_d] + sigma P1[d][z2_d] + bd_d P1[d][one])
//     sum_b |mu_b|^2 = tr(E M E^T) = sum_l (sum_dd We[dd][l] Q[l][x_dd] + be_l Q[l][one])
// DT, LT > 0: the dimensions at compile time (the metric's 12 / 20: fully unrolled inner products); 0: run-time.
// The parameters and Adam moments of this thread's outputs (idx = t + 512 k) live in its registers; the float64 copies the
// products read live in LDS and are refreshed by the owning thread after each update, so a persistent updater touches global
// memory per step only for M, the loss and (at the end) the results.
constexpr int LKOUT = 4;                              // P + 4 <= 2 048 at L + 2 D + 1 <= 64
template <int NB, int DT, int LT>
struct LinUpd {
    static constexpr int NFP = 16 * NB, NBLK = NB * (NB + 1) / 2;
    int D, L, P, fone, off_be, off_wd, off_bd, off_epsp, off_eps;
    double *Mf, *SM, *P1, *G, *Wed, *Wdd, *bed, *bdd, *sd, *elv, *lvd, *dwd, *red, *epsv;
    float p[LKOUT], m[LKOUT], v[LKOUT];

    __device__ __forceinline__ void carve(const LinArgs& a, char* smem) {
        D = DT ? DT : a.D; L = LT ? LT : a.L; P = a.P; fone = L + 2 * D; off_eps = a.off_eps;
        off_be = D * L; off_wd = off_be + L; off_bd = off_wd + L * D; off_epsp = off_bd + D;
        Mf = reinterpret_cast<double*>(smem);             // [NFP][NFP] symmetric
        SM = Mf + NFP * NFP; P1 = SM + L * NFP; G = P1 + D * NFP;
        Wed = G + L * NFP; Wdd = Wed + D * L; bed = Wdd + L * D; bdd = bed + L; sd = bdd + D; elv = sd + L; lvd = elv + L;
        dwd = lvd + L; red = dwd + L * D; epsv = red + 4 * LNW;
    }
    static __host__ size_t lds_bytes(int D, int L) {
        return sizeof(double) * ((size_t)NFP * NFP + (size_t)(D + 2 * L) * NFP + 3 * (size_t)D * L + 5 * L + D + 4 * LNW + 2);
    }
    __device__ __forceinline__ void load_state(const LinArgs& a) {
        const int t = threadIdx.x;
#pragma unroll
        for (int k = 0; k < LKOUT; ++k) {
            const int idx = min(t + LNT * k, P - 1);
            p[k] = a.params[idx]; m[k] = a.m[idx]; v[k] = a.v[idx];
        }
    }
    // float64 copies of this thread's own parameters into the arrays the products read
    __device__ __forceinline__ void publish_params() {
        const int t = threadIdx.x;
#pragma unroll
        for (int k = 0; k < LKOUT; ++k) {
            const int i = t + LNT * k;
            const double pv = (double)p[k];
            if (i < off_be) Wed[i] = pv;
            else if (i < off_wd) bed[i - off_be] = pv;
            else if (i < off_bd) Wdd[i - off_wd] = pv;
            else if (i < off_epsp) bdd[i - off_bd] = pv;
            else if (i < off_epsp + L) { const double sl = exp(0.5 * pv); sd[i - off_epsp] = sl; elv[i - off_epsp] = sl * sl; lvd[i - off_epsp] = pv; }
            else if (i == off_eps) epsv[0] = pv;
        }
    }
    template <bool SC1>
    __device__ __forceinline__ void expand_M(const double* M_in) {
        for (int e = threadIdx.x; e < NBLK * 256; e += LNT) {
            int k, i, j;
            lin_img_coords(e, k, i, j);
            int b1 = 0, rem = k;
            while (rem >= NB - b1) { rem -= NB - b1; ++b1; }
            const int b2 = b1 + rem;
            const double val = SC1 ? ld_sc1(M_in + e) : M_in[e];
            Mf[(16 * b1 + i) * NFP + 16 * b2 + j] = val;
            Mf[(16 * b2 + j) * NFP + 16 * b1 + i] = val;   // (diagonal blocks are bitwise symmetric: same products, same order)
        }
    }
    // the same in two halves (persistent form): the loads of batch n + 1's M ride under step n when its reducers are already done
    static constexpr int MPT = (NBLK * 256 + LNT - 1) / LNT;
    __device__ __forceinline__ void fetch_M(const double* M_in, double (&r)[MPT]) {
#pragma unroll
        for (int k = 0; k < MPT; ++k) r[k] = ld_sc1(M_in + min((int)threadIdx.x + LNT * k, NBLK * 256 - 1));
    }
    __device__ __forceinline__ void scatter_M(const double (&r)[MPT]) {
#pragma unroll
        for (int k = 0; k < MPT; ++k) {
            const int e = threadIdx.x + LNT * k;
            if (e < NBLK * 256) {
                int blk, i, j;
                lin_img_coords(e, blk, i, j);
                int b1 = 0, rem = blk;
                while (rem >= NB - b1) { rem -= NB - b1; ++b1; }
                const int b2 = b1 + rem;
                Mf[(16 * b1 + i) * NFP + 16 * b2 + j] = r[k];
                Mf[(16 * b2 + j) * NFP + 16 * b1 + i] = r[k];
            }
        }
    }
    // one step: Mf and the parameter copies are in LDS (a barrier behind them); returns with this thread's p / m / v updated,
    // its gradients in gout[], and nothing in LDS that the next publish_params / expand_M may not overwrite after a barrier
    __device__ __forceinline__ void step(const LinArgs& a, int tstep, float (&gout)[LKOUT]) {
        const int t = threadIdx.x;
        const double eps = off_eps >= 0 ? epsv[0] * (double)a.eps_cli : (double)a.eps_cli;
        const double sigma = exp(0.5 * eps), inv_var = 1.0 / (sigma * sigma);       // (one float64 exp on the chain, not two)
        LIN_STAMP(1);
        if constexpr (DT > 0 && LT > 0 && (DT % 2 == 0) && (LT % 2 == 0) && LT * (NFP / 2) <= LNT && LT * DT <= LNT / 2) {
            // The metric's shape.  Each thread owns 2 (SM, P1) or 4 (G) neighbouring columns of one output row, so that the M / SM /
            // P1 operands arrive as 16-byte LDS reads, every phase is ONE round of the workgroup, and all operands of a thread
            // are in registers before its first FMA (hipcc otherwise waits out the LDS latency once per product).
            constexpr int HP = NFP / 2;
            {   // SM = S M: thread (l, f .. f + 1)
                const int l = min(t / HP, LT - 1), f = (t % HP) * 2;
                double w[DT + 2]; d2 mv[DT + 2];
                w[0] = sd[l]; mv[0] = *reinterpret_cast<const d2*>(Mf + l * NFP + f);
                w[1] = bed[l]; mv[1] = *reinterpret_cast<const d2*>(Mf + fone * NFP + f);
#pragma unroll
                for (int dd = 0; dd < DT; ++dd) { w[2 + dd] = Wed[dd * LT + l]; mv[2 + dd] = *reinterpret_cast<const d2*>(Mf + (LT + dd) * NFP + f); }
                __builtin_amdgcn_sched_barrier(0);
                d2 s2 = w[0] * mv[0];
#pragma unroll
                for (int k = 1; k < DT + 2; ++k) s2 += w[k] * mv[k];
                if (t < LT * HP) *reinterpret_cast<d2*>(SM + l * NFP + f) = s2;
            }
            __syncthreads();
            LIN_STAMP(2);
            {   // P1 = R M: thread (d, f .. f + 1)
                const int d = min(t / HP, DT - 1), f = (t % HP) * 2;
                double w[LT + 1]; d2 mv[LT + 3];
                mv[LT] = *reinterpret_cast<const d2*>(Mf + (LT + DT + d) * NFP + f);
                mv[LT + 1] = *reinterpret_cast<const d2*>(Mf + (LT + d) * NFP + f);
                mv[LT + 2] = *reinterpret_cast<const d2*>(Mf + fone * NFP + f);
                w[LT] = bdd[d];
#pragma unroll
                for (int l = 0; l < LT; ++l) { w[l] = Wdd[l * DT + d]; mv[l] = *reinterpret_cast<const d2*>(SM + l * NFP + f); }
                __builtin_amdgcn_sched_barrier(0);
                d2 s2 = sigma * mv[LT] - mv[LT + 1] + w[LT] * mv[LT + 2];
#pragma unroll
                for (int l = 0; l < LT; ++l) s2 += w[l] * mv[l];
                if (t < DT * HP) *reinterpret_cast<d2*>(P1 + d * NFP + f) = s2;
            }
            __syncthreads();
            LIN_STAMP(3);
            if (t < LNT / 2) {   // G = Wd P1: thread (l, f .. f + 3), the lower half of the workgroup
                constexpr int QP = NFP / 4;
                const int l = min(t / QP, LT - 1), f = (t % QP) * 4;
                d2 wv[DT / 2], pa[DT], pb[DT];
#pragma unroll
                for (int d = 0; d < DT / 2; ++d) wv[d] = *reinterpret_cast<const d2*>(Wdd + l * DT + 2 * d);
#pragma unroll
                for (int d = 0; d < DT; ++d) { pa[d] = *reinterpret_cast<const d2*>(P1 + d * NFP + f); pb[d] = *reinterpret_cast<const d2*>(P1 + d * NFP + f + 2); }
                __builtin_amdgcn_sched_barrier(0);
                d2 sa = {0.0, 0.0}, sb = {0.0, 0.0};
#pragma unroll
                for (int d = 0; d < DT; ++d) { const double wd = wv[d / 2][d & 1]; sa += wd * pa[d]; sb += wd * pb[d]; }
                if (t < LT * QP) { *reinterpret_cast<d2*>(G + l * NFP + f) = sa; *reinterpret_cast<d2*>(G + l * NFP + f + 2) = sb; }
            } else {             // dwd[l][d] = S[l,:] . P1[d,:]: the upper half
                const int e = min(t - LNT / 2, LT * DT - 1), l = e / DT, d = e % DT;
                double w[DT]; d2 pv[DT / 2];
                const double s_l = sd[l], b_l = bed[l], p_l = P1[d * NFP + l], p_o = P1[d * NFP + fone];
#pragma unroll
                for (int dd = 0; dd < DT; ++dd) w[dd] = Wed[dd * LT + l];
#pragma unroll
                for (int dd = 0; dd < DT / 2; ++dd) pv[dd] = *reinterpret_cast<const d2*>(P1 + d * NFP + LT + 2 * dd);
                __builtin_amdgcn_sched_barrier(0);
                double sx = s_l * p_l + b_l * p_o;
#pragma unroll
                for (int dd = 0; dd < DT; ++dd) sx += w[dd] * pv[dd / 2][dd & 1];
                if (t - LNT / 2 < LT * DT) dwd[e] = sx;
            }
        } else {
        for (int e = t; e < L * NFP; e += LNT) {              // SM = S M
            const int l = e / NFP, f = e % NFP;
            double s = sd[l] * Mf[l * NFP + f] + bed[l] * Mf[fone * NFP + f];
#pragma unroll
            for (int dd = 0; dd < (DT ? DT : 32); ++dd) if (DT || dd < D) s += Wed[dd * L + l] * Mf[(L + dd) * NFP + f];
            SM[e] = s;
        }
        __syncthreads();
        LIN_STAMP(2);
        for (int e = t; e < D * NFP; e += LNT) {              // P1 = R M
            const int d = e / NFP, f = e % NFP;
            double s = sigma * Mf[(L + D + d) * NFP + f] - Mf[(L + d) * NFP + f] + bdd[d] * Mf[fone * NFP + f];
#pragma unroll
            for (int l = 0; l < (LT ? LT : 32); ++l) if (LT || l < L) s += Wdd[l * D + d] * SM[l * NFP + f];
            P1[e] = s;
        }
        __syncthreads();
        LIN_STAMP(3);
        for (int e = t; e < L * NFP; e += LNT) {              // G = Wd P1
            const int l = e / NFP, f = e % NFP;
            double s = 0.0;
#pragma unroll
            for (int d = 0; d < (DT ? DT : 32); ++d) if (DT || d < D) s += Wdd[l * D + d] * P1[d * NFP + f];
            G[e] = s;
        }
        for (int e = t; e < L * D; e += LNT) {                // dwd[l][d] = S[l,:] . P1[d,:]
            const int l = e / D, d = e % D;
            double s = sd[l] * P1[d * NFP + l] + bed[l] * P1[d * NFP + fone];
#pragma unroll
            for (int dd = 0; dd < (DT ? DT : 32); ++dd) if (DT || dd < D) s += Wed[dd * L + l] * P1[d * NFP + L + dd];
            dwd[e] = s;
        }
        }
        __syncthreads();
        LIN_STAMP(4);
        double ssq = 0.0, musq = 0.0, z2r = 0.0, klc = 0.0;
        if constexpr (DT > 0 && LT > 0 && (DT % 2 == 0) && (LT % 2 == 0) && LT * (NFP / 2) <= LNT && LT * DT <= LNT / 2) {
            // Only d loss / d epsilon and the three loss means need the four scalar sums, and wave 0 owns those outputs (index
            // P - 1 .. P + 2 < LNT + 64): it forms the sums by itself -- strided shares per lane, operands batched, xor-shuffles --
            // while the other waves are already at their gradients and Adam.  No barrier, no workgroup-wide reduction.
            static_assert(DT * LT + LT + LT * DT + DT + LT + 1 + 3 <= LNT + 64, "wave 0 must own the scalar outputs");
            if (t < 64) {
                constexpr int NS = (LT * DT + 63) / 64, NM = (LT * (DT + 1) + 63) / 64;
                double wa[NS], da[NS], qw[NM], qs[NM], qd[NM], qm[NM];
#pragma unroll
                for (int j = 0; j < NS; ++j) { const int e = min(t + 64 * j, LT * DT - 1); wa[j] = Wdd[e]; da[j] = dwd[e]; }
#pragma unroll
                for (int j = 0; j < NM; ++j) {
                    const int e = min(t + 64 * j, LT * (DT + 1) - 1), l = e / (DT + 1), dd = e % (DT + 1), f = dd < DT ? LT + dd : fone;
                    qw[j] = dd < DT ? Wed[dd * LT + l] : bed[l]; qs[j] = SM[l * NFP + f]; qd[j] = sd[l]; qm[j] = Mf[l * NFP + f];
                }
                const int dq = min(t, DT - 1), lq = min(t, LT - 1);
                const double px = P1[dq * NFP + LT + dq], pz = P1[dq * NFP + LT + DT + dq], po = P1[dq * NFP + fone], bq = bdd[dq];
                const double lvq = lvd[lq], evq = elv[lq];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < NS; ++j) ssq += t + 64 * j < LT * DT ? wa[j] * da[j] : 0.0;
#pragma unroll
                for (int j = 0; j < NM; ++j) musq += t + 64 * j < LT * (DT + 1) ? qw[j] * (qs[j] - qd[j] * qm[j]) : 0.0;
                if (t < DT) { ssq += -px + sigma * pz + bq * po; z2r = pz; }
                if (t < LT) klc = 1.0 + lvq - evq;                                     // 1 + lv - e^{lv}
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) {
                    ssq += __shfl_xor(ssq, o, 64); musq += __shfl_xor(musq, o, 64);
                    z2r += __shfl_xor(z2r, o, 64); klc += __shfl_xor(klc, o, 64);
                }
            }
        } else {
        // the four scalar sums: each thread a strided share, lanes by xor-shuffle, the waves in order (all fixed order)
        double p_ssq = 0.0, p_musq = 0.0, p_z2r = 0.0, p_klc = 0.0;
        for (int e = t; e < L * D; e += LNT) p_ssq += Wdd[e] * dwd[e];
        if (t < D) {
            p_ssq += -P1[t * NFP + L + t] + sigma * P1[t * NFP + L + D + t] + bdd[t] * P1[t * NFP + fone];
            p_z2r = P1[t * NFP + L + D + t];
        }
        for (int e = t; e < L * (D + 1); e += LNT) {
            const int l = e / (D + 1), dd = e % (D + 1);
            const int f = dd < D ? L + dd : fone;
            const double q = SM[l * NFP + f] - sd[l] * Mf[l * NFP + f];            // Q = E M
            p_musq += (dd < D ? Wed[dd * L + l] : bed[l]) * q;
        }
        if (t < L) p_klc = 1.0 + lvd[t] - elv[t];                                  // 1 + lv - e^{lv}
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            p_ssq += __shfl_xor(p_ssq, o, 64); p_musq += __shfl_xor(p_musq, o, 64);
            p_z2r += __shfl_xor(p_z2r, o, 64); p_klc += __shfl_xor(p_klc, o, 64);
        }
        if ((t & 63) == 0) { red[t >> 6] = p_ssq; red[LNW + (t >> 6)] = p_musq; red[2 * LNW + (t >> 6)] = p_z2r; red[3 * LNW + (t >> 6)] = p_klc; }
        __syncthreads();
#pragma unroll
        for (int w = 0; w < LNW; ++w) { ssq += red[w]; musq += red[LNW + w]; z2r += red[2 * LNW + w]; klc += red[3 * LNW + w]; }
        }
        LIN_STAMP(5);
        const double inv_bt = (double)a.inv_bt, rows = (double)a.rows, c0 = inv_var * inv_bt;
        const float bc1 = -expm1f((float)tstep * -0.10536051565782628f), bc2 = -expm1f((float)tstep * -0.0010005003335835335f);
#pragma unroll
        for (int k = 0; k < LKOUT; ++k) {
            const int idx = t + LNT * k;
            double gd = 0.0;
            if (idx < off_be) { const int d = idx / L, l = idx % L; gd = c0 * G[l * NFP + L + d] + (SM[l * NFP + L + d] - sd[l] * Mf[l * NFP + L + d]) * inv_bt; }
            else if (idx < off_wd) { const int l = idx - off_be; gd = c0 * G[l * NFP + fone] + (SM[l * NFP + fone] - sd[l] * Mf[l * NFP + fone]) * inv_bt; }
            else if (idx < off_bd) gd = c0 * dwd[idx - off_wd];
            else if (idx < off_epsp) gd = c0 * P1[(idx - off_bd) * NFP + fone];
            else if (idx < off_epsp + L) {
                const int l = idx - off_epsp;
                gd = 0.5 * sd[l] * c0 * G[l * NFP + l] - 0.5 * (1.0 - elv[l]) * (double)a.rows_over_bt;
            } else if (idx == off_eps) {
                gd = (double)a.eps_cli * (-0.5 * ssq * inv_var + 0.5 * rows * D + 0.5 * sigma * z2r * inv_var) * inv_bt;
            } else if (idx >= P && idx < P + 3) {
                const double dkl = (0.5 * musq - 0.5 * rows * klc) * inv_bt;
                const double mse = (0.5 * ssq * inv_var + 0.5 * rows * D * ((double)kLog2Pi + eps)) * inv_bt;
                gd = idx == P ? dkl + mse : (idx == P + 1 ? dkl : mse);
            }
            const float gf = (float)gd;
            gout[k] = gf;
            if (idx == P && a.loss_hist) a.loss_hist[(long long)(tstep - 1) % a.loss_hist_cap] = gf;
            if (idx < P) adam_apply_f(p[k], gf, m[k], v[k], a.lr, bc1, bc2);
        }
        LIN_STAMP(6);
    }
    __device__ __forceinline__ void store_state(const LinArgs& a, const float (&gout)[LKOUT], int tstep) {
        const int t = threadIdx.x;
#pragma unroll
        for (int k = 0; k < LKOUT; ++k) {
            const int idx = t + LNT * k;
            if (idx < P + kExtra) a.grads[idx] = gout[k];
            if (idx < P) { a.params[idx] = p[k]; a.m[idx] = m[k]; a.v[idx] = v[k]; }
        }
        if (t == 0) a.step_dev[0] = tstep;
    }
};

// ---- updater on the float64 matrix cores (persistent form; D <= 16, L <= 32, L + 2 D + 1 <= 48) ------------------------------------
// The same algebra as LinUpd, laid out so that the dependent chain SM -> P1 -> G never leaves the registers: with the FEATURE index
// on the MFMA column (lane & 15) every product is  Out = Weights x Prev  and an accumulator tile of v_mfma_f64_16x16x4_f64
// (lane (col, g), register r = row g + 4 r) is exactly the B operand of the next product's k-step r in natural k order (probed
// with exact integers: tools/mfma_f64_probe.hip; 64 cycles per instruction, dependent or not).  Wave w < 3 owns feature block w:
//     SM[:, blk]  = diag(s) M[z1 rows, blk] + be (x) M[one, blk]  (accumulator init)  +  We^T . M[x rows, blk]        (2 row tiles x KD k-steps)
//     P1[:, blk]  = -M[x rows] + sigma M[z2 rows] + bd (x) M[one]  (init)              +  Wd^T . SM                     (4 + RL1 k-steps)
//     G[:, blk]   =                                                                       Wd . P1                       (2 x KD)
// and every gradient that is an ELEMENT of those tiles is finished in place: dWe / dbe = c0 G + (SM - s M[z1 row]) / B on the x and
// "one" columns, d lv on the diagonal of the z1 columns, dbd = c0 P1[:, one].  Only dWd = S P1^T sums over the feature index, i.e.
// across lanes: the chain waves drop P1 (4.6 KB) into LDS and wave 3 forms dwd^T = P1[:, x cols] We + ... with 2 x KD MFMAs while
// the chain waves run G.  The four scalar sums are wave reductions of values the lanes hold anyway.  Per step: 3 barriers,
// ~17 dependent MFMAs (1.1 k cycles) instead of ~7 k cycles of LDS-bound float64 FMAs.  M arrives straight from the reducers'
// image into registers (17 write-through loads per lane at offsets fixed for the launch, prefetched a step ahead when the
// reducers are ahead): no symmetric copy in LDS.
using d4 = __attribute__((ext_vector_type(4))) double;
// Sum over the wave, the same value in every lane.  Four DPP stages inside each row of 16 lanes (the two dwords of a double move
// separately; xor 1, xor 2, half mirror, mirror: every lane ends with its row's total), then the four rows by v_readlane -- ~25
// instructions where six __shfl_xor stages are twelve ds_bpermute round trips (the three sums of a step cost 1.8 k cycles that way).
template <int CTRL>
__device__ __forceinline__ double lin_dpp_add(double x) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, 0xf, 0xf, false);
    return x + __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lin_wave_sum(double x) {
    x = lin_dpp_add<0xB1>(x);          // quad_perm [1, 0, 3, 2]
    x = lin_dpp_add<0x4E>(x);          // quad_perm [2, 3, 0, 1]
    x = lin_dpp_add<0x141>(x);         // row_half_mirror
    x = lin_dpp_add<0x140>(x);         // row_mirror
    double tot = 0.0;
#pragma unroll
    for (int r = 0; r < 4; ++r)
        tot += __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), 16 * r), __builtin_amdgcn_readlane(__double2loint(x), 16 * r));
    return tot;
}
// e^x in float64 (|x| < 700): k = round(x / ln 2), degree-13 Taylor polynomial on |r| <= ln 2 / 2 (truncation 5e-18), one ldexp.
// ~20 fused multiply-adds on the updater's critical path instead of the library routine's ~3x that.
__device__ __forceinline__ double lin_exp(double x) {
    const double k = rint(x * 1.4426950408889634);
    double r = fma(-k, 6.93147180369123816490e-01, x);
    r = fma(-k, 1.90821492927058770002e-10, r);
    double p = 1.0 / 6227020800.0;
    p = fma(p, r, 1.0 / 479001600.0); p = fma(p, r, 1.0 / 39916800.0); p = fma(p, r, 1.0 / 3628800.0); p = fma(p, r, 1.0 / 362880.0);
    p = fma(p, r, 1.0 / 40320.0); p = fma(p, r, 1.0 / 5040.0); p = fma(p, r, 1.0 / 720.0); p = fma(p, r, 1.0 / 120.0);
    p = fma(p, r, 1.0 / 24.0); p = fma(p, r, 1.0 / 6.0); p = fma(p, r, 0.5); p = fma(p, r, 1.0); p = fma(p, r, 1.0);
    return ldexp(p, (int)k);
}
// element (r, c) of the symmetric moment matrix in the packed image (upper block triangle, accumulator layout)
__device__ __forceinline__ int lin_m_index(int NB, int r, int c) {
    const int br = r >> 4, bc = c >> 4;
    const int b1 = br <= bc ? br : bc, b2 = br <= bc ? bc : br, i = br <= bc ? (r & 15) : (c & 15), j = br <= bc ? (c & 15) : (r & 15);
    return lin_blk(NB, b1, b2) * 256 + (((i >> 2) * 16 + j) << 2) + (i & 3);
}
template <int NB, int DT, int LT>
struct LinUpdM {
    static constexpr int NFP = 16 * NB;
    static constexpr int KD = DT ? (DT + 3) / 4 : 4;                                   // k-steps over the data dimension
    static constexpr int RL1 = LT ? (LT > 16 ? (LT - 16 + 3) / 4 : 0) : 4;             // registers of the second latent row tile that can hold a row
    static constexpr int NM = 4 + RL1 + 2 * KD + 1;                                    // M values per chain lane
    int D, L, P, fone, off_be, off_wd, off_bd, off_epsp, off_eps;
    double *Wed, *Wdd, *bed, *bdd, *sd, *elv, *lvd, *scal, *part, *P1s, *gq;
    float p[LKOUT], m[LKOUT], v[LKOUT];
    int mo[NM];                                                                        // chain lanes: image offsets of their M values

    static __host__ __device__ constexpr bool shape_ok(int D, int L) { return D <= 16 && L <= 32 && L + 2 * D + 1 <= NFP && NB == 3; }
    static __host__ size_t lds_bytes(int D, int L, int P) {
        return sizeof(double) * ((size_t)2 * D * L + 4 * L + D + 8 + 16 + 16 * NFP + (size_t)(P + kExtra + 7));
    }
    __device__ __forceinline__ void carve(const LinArgs& a, char* smem) {
        D = DT ? DT : a.D; L = LT ? LT : a.L; P = a.P; fone = L + 2 * D; off_eps = a.off_eps;
        off_be = D * L; off_wd = off_be + L; off_bd = off_wd + L * D; off_epsp = off_bd + D;
        Wed = reinterpret_cast<double*>(smem); Wdd = Wed + D * L; bed = Wdd + L * D; bdd = bed + L; sd = bdd + D; elv = sd + L; lvd = elv + L;
        scal = lvd + L; part = scal + 8; P1s = part + 16; gq = P1s + 16 * NFP;
    }
    __device__ __forceinline__ void load_state(const LinArgs& a) {
        const int t = threadIdx.x;
#pragma unroll
        for (int k = 0; k < LKOUT; ++k) {
            const int idx = min(t + LNT * k, P - 1);
            p[k] = a.params[idx]; m[k] = a.m[idx]; v[k] = a.v[idx];
        }
        if (off_eps < 0 && t == 0) { const double e = (double)a.eps_cli; scal[0] = e; scal[1] = lin_exp(0.5 * e); scal[2] = lin_exp(-e); }
        // image offsets of this lane's M values (chain waves: wave w = feature block w)
        const int lane = t & 63, j = lane & 15, g = lane >> 4, f = min(16 * (t >> 6) + j, NFP - 1);
        int k = 0;
#pragma unroll
        for (int r = 0; r < 4; ++r) mo[k++] = lin_m_index(NB, g + 4 * r, f);                                   // z1 rows 0 .. 15
#pragma unroll
        for (int r = 0; r < RL1; ++r) mo[k++] = lin_m_index(NB, min(16 + g + 4 * r, NFP - 1), f);             // z1 rows 16 ..
#pragma unroll
        for (int r = 0; r < KD; ++r) mo[k++] = lin_m_index(NB, min(L + g + 4 * r, NFP - 1), f);               // x rows
#pragma unroll
        for (int r = 0; r < KD; ++r) mo[k++] = lin_m_index(NB, min(L + D + g + 4 * r, NFP - 1), f);           // z2 rows
        mo[k++] = lin_m_index(NB, fone, f);
    }
    __device__ __forceinline__ void fetch_M(const double* M_in, double (&r)[NM]) const {
#pragma unroll
        for (int k = 0; k < NM; ++k) r[k] = ld_sc1(M_in + mo[k]);
    }
    // float64 copies of this thread's own parameters into the arrays the products read
    __device__ __forceinline__ void publish_params(const LinArgs& a) {
        const int t = threadIdx.x;
#pragma unroll
        for (int k = 0; k < LKOUT; ++k) {
            const int i = t + LNT * k;
            const double pv = (double)p[k];
            if (i < off_be) Wed[i] = pv;
            else if (i < off_wd) bed[i - off_be] = pv;
            else if (i < off_bd) Wdd[i - off_wd] = pv;
            else if (i < off_epsp) bdd[i - off_bd] = pv;
            else if (i < off_epsp + L || i == off_eps) {
                // e^{lv / 2} and e^{eps / 2}, e^{-eps}: the same instruction stream for the latent lanes and the epsilon lane
                const bool is_eps = i == off_eps;
                const double e = is_eps ? pv * (double)a.eps_cli : pv, h = lin_exp(0.5 * e);
                if (is_eps) {
                    const double q = h * h;
                    double r = __builtin_amdgcn_rcp(q);                // 1 / sigma^2: hardware seed, two Newton steps (full double precision)
                    r = fma(fma(-q, r, 1.0), r, r); r = fma(fma(-q, r, 1.0), r, r);
                    scal[0] = e; scal[1] = h; scal[2] = r;
                }
                else { sd[i - off_epsp] = h; elv[i - off_epsp] = h * h; lvd[i - off_epsp] = pv; }
            }
        }
    }
    // one step.  In: parameters published and a barrier behind them; mreg = this lane's values of the batch's M (chain waves).
    // Out: p / m / v updated, gout[] this thread's gradients.  Ends WITHOUT a barrier: the caller's next publish_params writes arrays
    // that only the phases before this step's last barrier read.
    // next_cnt / M_next: the NEXT batch's reducer counter and M (nullptr: none); if its reducers are done already, its M is loaded
    // into mnext under this step's second half and have_next says so.
    __device__ __forceinline__ void step(const LinArgs& a, int tstep, const double (&mreg)[NM], float (&gout)[LKOUT], const unsigned* next_cnt,
                                         unsigned per_set, const double* M_next, double (&mnext)[NM], bool& have_next) {
        const int t = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63, j = lane & 15, g = lane >> 4;
        LIN_STAMP(1);
        if (t == 64 * (NB + 2)) scal[4] = (next_cnt && __hip_atomic_load(next_cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= per_set) ? 1.0 : 0.0;
        const double eps = scal[0], sigma = scal[1], inv_var = scal[2], inv_bt = (double)a.inv_bt, c0 = inv_var * inv_bt;
        d4 sm[2], p1;
        double musq_p = 0.0, ssq_p = 0.0, z2r_p = 0.0;
        const int f = 16 * wave + j;                                   // chain waves: this lane's feature column
        // The chain's WEIGHT operands (functions of the published parameters only) are read from LDS up front: left where they are
        // used, hipcc issues each ds_read right in front of its MFMA and waits for it -- ds_read, s_waitcnt lgkmcnt(0), v_mfma, 17
        // times: an LDS round trip (~130 cycles) on top of every 64-cycle product of the dependent chain.
        // (SM's and P1's weights here; G's and the dWd wave's operands in ONE batch behind the barrier: held across it they spill)
        double awSMr[KD][2], awP1r[4 + (RL1 ? RL1 : 1)];
        auto awSM = [&](int kk, int lt) -> double& { return awSMr[kk][lt]; };
        auto awP1 = [&](int r) -> double& { return awP1r[r]; };
        if (wave < NB) {
#pragma unroll
            for (int kk = 0; kk < KD; ++kk)
#pragma unroll
                for (int lt = 0; lt < 2; ++lt) {
                    const int dd = 4 * kk + g, l = 16 * lt + j;
                    const double w1 = Wed[min(dd, D - 1) * L + min(l, L - 1)];
                    awSM(kk, lt) = (dd < D && l < L && (lt == 0 || RL1 > 0)) ? w1 : 0.0;
                }
#pragma unroll
            for (int r = 0; r < 4 + RL1; ++r) {
                const int l = (r < 4 ? 4 * r : 16 + 4 * (r - 4)) + g;
                const double w = Wdd[min(l, L - 1) * D + min(j, D - 1)];
                awP1(r) = (j < D && l < L) ? w : 0.0;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (wave < NB) {
            const double* Mz = mreg; const double* Mx = mreg + 4 + RL1; const double* Mz2 = Mx + KD; const double Mone = mreg[NM - 1];
            // ---- SM ----
#pragma unroll
            for (int lt = 0; lt < 2; ++lt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int l = 16 * lt + g + 4 * r;
                    const bool ok = l < L && (lt == 0 || r < RL1);
                    const double sv = sd[min(l, L - 1)], bv = bed[min(l, L - 1)];
                    const double slv = ok ? sv : 0.0, blv = ok ? bv : 0.0;
                    sm[lt][r] = (lt == 0 || r < RL1) ? slv * Mz[lt == 0 ? r : min(4 + r, 3 + RL1)] + blv * Mone : 0.0;
                }
#pragma unroll
            for (int kk = 0; kk < KD; ++kk) {
#pragma unroll
                for (int lt = 0; lt < 2; ++lt) {
                    if (lt == 1 && RL1 == 0) continue;
                    sm[lt] = __builtin_amdgcn_mfma_f64_16x16x4f64(awSM(kk, lt), Mx[kk], sm[lt], 0, 0, 0);
                }
            }
            // ---- P1 ----
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int d = g + 4 * r;
                p1[r] = (r < KD && d < D) ? -Mx[min(r, KD - 1)] + sigma * Mz2[min(r, KD - 1)] + bdd[min(d, D - 1)] * Mone : 0.0;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) p1 = __builtin_amdgcn_mfma_f64_16x16x4f64(awP1(r), sm[0][r], p1, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < RL1; ++r) p1 = __builtin_amdgcn_mfma_f64_16x16x4f64(awP1(4 + r), sm[1][r], p1, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) P1s[(g + 4 * r) * NFP + f] = p1[r];
        } else if (wave == NB + 1) {
            // 1 + lv - e^{lv} summed over the latent dimension (the closed-form KL term of the loss)
            const double kl = lane < L ? 1.0 + lvd[min(lane, L - 1)] - elv[min(lane, L - 1)] : 0.0;
            const double tot = lin_wave_sum(kl);
            if (lane == 0) {
                part[12] = tot;
                // Adam's bias corrections 1 - beta^t for everybody (float, as the other paths compute them)
                part[13] = (double)(-expm1f((float)tstep * -0.10536051565782628f)); part[14] = (double)(-expm1f((float)tstep * -0.0010005003335835335f));
            }
        }
        LIN_STAMP(2);
        __syncthreads();                                               // P1 is in LDS
        LIN_STAMP(3);
        have_next = scal[4] != 0.0;
        if (wave < NB) {
            if (have_next) fetch_M(M_next, mnext);
            // ---- G = Wd P1 ---- (its six weights in one batch of LDS reads first)
            double awGr[KD][2];
#pragma unroll
            for (int kk = 0; kk < KD; ++kk)
#pragma unroll
                for (int lt = 0; lt < 2; ++lt) {
                    const int dd = 4 * kk + g, l = 16 * lt + j;
                    const double w2 = Wdd[min(l, L - 1) * D + min(dd, D - 1)];
                    awGr[kk][lt] = (l < L && dd < D && (lt == 0 || RL1 > 0)) ? w2 : 0.0;
                }
            double sl[2][4], bl[2][4];                                 // (read again: held across the barrier they spill)
#pragma unroll
            for (int lt = 0; lt < 2; ++lt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int l = 16 * lt + g + 4 * r;
                    const bool ok = l < L && (lt == 0 || r < RL1);
                    const double sv = sd[min(l, L - 1)], bv = bed[min(l, L - 1)];
                    sl[lt][r] = ok ? sv : 0.0; bl[lt][r] = ok ? bv : 0.0;
                }
            __builtin_amdgcn_sched_barrier(0);
            d4 G[2];
            G[0] = d4{0.0, 0.0, 0.0, 0.0}; G[1] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int r = 0; r < KD; ++r) {
#pragma unroll
                for (int lt = 0; lt < 2; ++lt) {
                    if (lt == 1 && RL1 == 0) continue;
                    G[lt] = __builtin_amdgcn_mfma_f64_16x16x4f64(awGr[r][lt], p1[r], G[lt], 0, 0, 0);
                }
            }
            // ---- the gradients that are elements of these tiles ----
            const double* Mz = mreg;
            const bool is_x = f >= L && f < L + D, is_z2 = f >= L + D && f < fone, is_one = f == fone, is_z1 = f < L;
            if (is_x || is_one) {
                const int base = is_x ? (f - L) * L : off_be;          // dWe[dd][:] resp. dbe
#pragma unroll
                for (int lt = 0; lt < 2; ++lt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (lt == 1 && r >= RL1) continue;
                        const int l = 16 * lt + g + 4 * r;
                        if (l < L) {
                            const double q = sm[lt][r] - sl[lt][r] * Mz[lt == 0 ? r : min(4 + r, 3 + RL1)];           // Q = E M
                            gq[base + l] = c0 * G[lt][r] + q * inv_bt;
                            musq_p += (is_x ? Wed[base + l] : bl[lt][r]) * q;
                        }
                    }
            }
            if (is_one) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int d = g + 4 * r;
                    if (d < D) { gq[off_bd + d] = c0 * p1[r]; ssq_p += bdd[d] * p1[r]; }
                }
            }
            if (is_z1 && (j & 3) == g) {                               // the lane that holds G[f][f]
                const int lt = f >> 4, r = j >> 2;
                double gll = 0.0;
#pragma unroll
                for (int q2 = 0; q2 < 2; ++q2)
#pragma unroll
                    for (int r2 = 0; r2 < 4; ++r2) gll = (q2 == lt && r2 == r) ? G[q2][r2] : gll;
                gq[off_epsp + f] = 0.5 * sd[f] * c0 * gll - 0.5 * (1.0 - elv[f]) * (double)a.rows_over_bt;
            }
            if (is_x || is_z2) {                                       // the diagonal entries P1[d][x_d], P1[d][z2_d] of the residual sums
                const int d = is_x ? f - L : f - L - D;
                if ((d & 3) == g) {
                    double pd = 0.0;
#pragma unroll
                    for (int r2 = 0; r2 < 4; ++r2) pd = (r2 == (d >> 2)) ? p1[r2] : pd;
                    if (is_x) ssq_p -= pd; else { ssq_p += sigma * pd; z2r_p += pd; }
                }
            }
            musq_p = lin_wave_sum(musq_p); ssq_p = lin_wave_sum(ssq_p); z2r_p = lin_wave_sum(z2r_p);
            if (lane == 0) { part[3 * wave] = ssq_p; part[3 * wave + 1] = musq_p; part[3 * wave + 2] = z2r_p; }
        } else if (wave == NB) {
            // ---- dwd^T[d][l] = s_l P1[d][l] + be_l P1[d][one] + sum_dd P1[d][x_dd] We[dd][l]  (rows d = g + 4 r, columns l = 16 ct + j) ----
            // (every LDS operand first -- We was read in front of the barrier --, then the products back to back: see the chain waves)
            d4 C[2];
            double wsum = 0.0, avk[KD], p1l[2][4], p1o[4], dw_sr[2], dw_br[2], dw_wr[KD][2], dw_dr[2][4];
            auto dw_s = [&](int ct) -> double& { return dw_sr[ct]; };
            auto dw_b = [&](int ct) -> double& { return dw_br[ct]; };
            auto dw_w = [&](int kk, int ct) -> double& { return dw_wr[kk][ct]; };
            auto dw_d = [&](int ct, int r) -> double& { return dw_dr[ct][r]; };
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                const int l = 16 * ct + j, lc = min(l, L - 1);
                const double s_l = sd[lc], b_l = bed[lc];
                dw_s(ct) = l < L ? s_l : 0.0; dw_b(ct) = l < L ? b_l : 0.0;
#pragma unroll
                for (int kk = 0; kk < KD; ++kk) {
                    const int dd = 4 * kk + g;
                    const double w = Wed[min(dd, D - 1) * L + lc];
                    dw_w(kk, ct) = (dd < D && l < L) ? w : 0.0;
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) dw_d(ct, r) = Wdd[lc * D + min(g + 4 * r, D - 1)];
            }
#pragma unroll
            for (int kk = 0; kk < KD; ++kk) avk[kk] = P1s[j * NFP + min(L + 4 * kk + g, NFP - 1)];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                p1o[r] = P1s[(g + 4 * r) * NFP + fone];
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) p1l[ct][r] = P1s[(g + 4 * r) * NFP + min(16 * ct + j, L - 1)];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                if (ct == 1 && RL1 == 0) continue;
                const int l = 16 * ct + j;
#pragma unroll
                for (int r = 0; r < 4; ++r) C[ct][r] = dw_s(ct) * p1l[ct][r] + dw_b(ct) * p1o[r];
#pragma unroll
                for (int kk = 0; kk < KD; ++kk) C[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(avk[kk], dw_w(kk, ct), C[ct], 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int d = g + 4 * r;
                    if (l < L && d < D) { gq[off_wd + l * D + d] = c0 * C[ct][r]; wsum += dw_d(ct, r) * C[ct][r]; }
                }
            }
            wsum = lin_wave_sum(wsum);
            if (lane == 0) part[9] = wsum;
        }
        LIN_STAMP(4);
        __syncthreads();                                               // every gradient and partial sum is in LDS
        LIN_STAMP(5);
        const double rows = (double)a.rows;
        const float bc1 = (float)part[13], bc2 = (float)part[14];
#pragma unroll
        for (int k = 0; k < LKOUT; ++k) {
            const int idx = t + LNT * k;
            if (LNT * k >= P + 3) break;                               // (uniform) nothing lives up here
            double gd = 0.0;
            if (idx < P && idx != off_eps) gd = gq[idx];
            else if (idx == off_eps || (idx >= P && idx < P + 3)) {
                const double ssq = part[0] + part[3] + part[6] + part[9], musq = part[1] + part[4] + part[7], z2r = part[2] + part[5] + part[8], klc = part[12];
                if (idx == off_eps) gd = (double)a.eps_cli * (-0.5 * ssq * inv_var + 0.5 * rows * D + 0.5 * sigma * z2r * inv_var) * inv_bt;
                else {
                    const double dkl = (0.5 * musq - 0.5 * rows * klc) * inv_bt;
                    const double mse = (0.5 * ssq * inv_var + 0.5 * rows * D * ((double)kLog2Pi + eps)) * inv_bt;
                    gd = idx == P ? dkl + mse : (idx == P + 1 ? dkl : mse);
                }
            }
            const float gf = (float)gd;
            gout[k] = gf;
            if (idx == P && a.loss_hist) a.loss_hist[(long long)(tstep - 1) % a.loss_hist_cap] = gf;
            if (idx < P) adam_apply_f(p[k], gf, m[k], v[k], a.lr, bc1, bc2);
        }
        LIN_STAMP(6);
    }
    __device__ __forceinline__ void store_state(const LinArgs& a, const float (&gout)[LKOUT], int tstep) {
        const int t = threadIdx.x;
#pragma unroll
        for (int k = 0; k < LKOUT; ++k) {
            const int idx = t + LNT * k;
            if (idx < P + kExtra) a.grads[idx] = gout[k];
            if (idx < P) { a.params[idx] = p[k]; a.m[idx] = m[k]; a.v[idx] = v[k]; }
        }
        if (t == 0) a.step_dev[0] = tstep;
    }
};

// ---- launch-per-step form ---------------------------------------------------------------------------------------------------------
template <int NB, int DT, int LT>
__global__ __launch_bounds__(LNT) void lin_step_kernel(const LinArgs a) {
    extern __shared__ __attribute__((aligned(16))) char lin_smem[];
    const int b = blockIdx.x, t = threadIdx.x;
    constexpr int NO = NB * (NB + 1) / 2 * 256;
    if (b < a.has_update) {
        LinUpd<NB, DT, LT> u;
        u.carve(a, lin_smem);
        LIN_STAMP(0);
        const int tstep = a.step_dev[0] + 1;
        u.load_state(a);
        u.publish_params();
        u.template expand_M<false>(a.M_in);
        __syncthreads();
        float g[LKOUT];
        u.step(a, tstep, g);
        u.store_state(a, g, tstep);
        LIN_STAMP(7);
    } else if (b < a.has_update + a.n_reduce) {
        lin_reduce<false>(a.partial_in, a.M_out, a.ntiles, lin_smem, b - a.has_update, NO);
    } else {
        const int tile = b - a.has_update - a.n_reduce;
        const int wave = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63;
        const LinTile tl(a.D, a.L, a.T);
        const int stride = max(tl.bytes, lin_scratch_bytes(NB)), v_off = stride, c_off = v_off + 4 * a.T;
        for (int p = wave; p < tl.np; p += LNW) lin_issue_piece(a, tl, a.x, a.z1, a.z2, tile, p, lin_smem, lane);
        lin_write_vcol(reinterpret_cast<float*>(lin_smem + v_off), a.T, (int)min((long long)a.T, (long long)a.B - (long long)tile * a.T), t);
        if (t == 0) *reinterpret_cast<float*>(lin_smem + c_off) = 0.f;
        lin_wait_vmcnt<0>();
        lin_barrier();
        lin_fix_ragged(a, tl, a.x, a.z1, a.z2, tile, lin_smem, t);
        f32x4 acc[NB * (NB + 1) / 2];
        lin_tile_products<NB, 0, LNW>(a, tl, lin_smem, 0, v_off, c_off, acc, lane, wave, [](int) {});
        lin_barrier();                                         // every wave has read its last operand: the slot turns into scratch
        lin_tile_combine<NB, false, LNW>(acc, lin_smem, a.partial_out + (long long)tile * NO, t, wave, lane);
    }
}

// ---- persistent form: up to kLinMaxPersist steps in one launch ----------------------------------------------------------------
// GEN: the batches are not read from HBM but DRAWN by the streamers, tile by tile, straight into the LDS slots: the Philox work
// items of vaek_make_batch (rng_dev.h: same counters, same functions, same bits) for the rows of a tile, dealt to the 512
// threads and issued between the k-steps of an earlier tile exactly where the other form issues its LDS-DMA pieces -- the loop
// body of model.py:221-222 (get_batch, sample_latent, train_step) with no batch ever in HBM.  The RNG step of the batch that
// takes the Adam counter from t to t + 1 is t, as in trainer.GraphLoop / vaek_train_step_gen.
template <int NB, int DT, int LT, int JT, bool GEN>
__global__ __launch_bounds__(LNT, 2) void lin_persist_kernel(const LinArgs a, const std::conditional_t<GEN, BatchArgs, LinPtrs> src) {       // one workgroup per CU (its LDS request sees to that)
    extern __shared__ __attribute__((aligned(16))) char lin_smem[];
    const int b = blockIdx.x, t = threadIdx.x, N = a.n_steps;
    constexpr int NO = NB * (NB + 1) / 2 * 256;
    const int per_set = a.n_reduce / a.sets;                  // reducer workgroups per set
    if (b < a.has_update) {
        // ---- the updater: one workgroup, parameters and Adam state in registers / LDS across all N steps (the host sends only
        // shapes the matrix-core updater covers into this form: lin_persist_supported) -------------------------------------------
        LinUpdM<NB, DT, LT> u;
        u.carve(a, lin_smem);
        int tstep = a.step_dev[0];
        u.load_state(a);
        float g[LKOUT];
#pragma unroll
        for (int k = 0; k < LKOUT; ++k) g[k] = 0.f;
        constexpr int NM = LinUpdM<NB, DT, LT>::NM;
        double mreg[NM], mnext[NM];
        bool have_next = false;
        LIN_STAMP(10);
        for (int n = 0; n < N; ++n) {
            LIN_STAMP(0);
            u.publish_params(a);
            LIN_STAMP(9);
            if (!have_next) {
                lin_wait_count(a.cnt_reduce + n, (unsigned)per_set, a.status, (1u << 28) | ((unsigned)n << 16));     // (also the barrier behind publish_params)
                if (t < 64 * NB) u.fetch_M(a.M_base + (long long)n * NO, mreg);
            } else {
                __syncthreads();
#pragma unroll
                for (int k = 0; k < NM; ++k) mreg[k] = mnext[k];
            }
            LIN_STAMP(8);
            ++tstep;
            u.step(a, tstep, mreg, g, n + 1 < N ? a.cnt_reduce + n + 1 : nullptr, (unsigned)per_set, a.M_base + (long long)(n + 1) * NO, mnext, have_next);
            LIN_STAMP(7);
            { [[maybe_unused]] unsigned long long te = 0; LIN_NOWQ(te); LIN_PUT(64 + n, te); }
        }
        u.store_state(a, g, tstep);
        // every reducer has added to the last batch's counter, every streamer long before: nobody reads or writes the arrival
        // counters any more -- zero them for the next launch (write-through; the kernel boundary orders them)
        for (int k = t; k < N * kLinShards; k += LNT) __hip_atomic_store(a.cnt_stream + k * kLinShardStride, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int k = t; k < N; k += LNT) __hip_atomic_store(a.cnt_reduce + k, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else if (b < a.has_update + a.n_reduce) {
        // ---- reducers: set (rb / per_set) takes batches set, set + 2, ... ------------------------------------------------------
        const int rb = b - a.has_update, set = rb / per_set, ro = rb % per_set;
        const int tstep0 = a.step_dev[0];              // (the updater stores the counter at the very end of the launch)
        [[maybe_unused]] unsigned long long r0 = 0, r1 = 0, r2 = 0, racc_w = 0, racc_r = 0;
        for (int n = set; n < N; n += a.sets) {
            LIN_NOWQ(r0);
            lin_wait_shards(a.cnt_stream + n * kLinShards * kLinShardStride, (unsigned)a.ntiles, a.status, (2u << 28) | ((unsigned)n << 16));
            LIN_NOWQ(r1);
            // 32-output slices, two at a time where there are two (a 128-output form reading 16 bytes per lane with sc1 buffer loads
            // measured 8 % SLOWER per step)
            for (int sub = ro; sub < NO / 32; sub += 2 * per_set) {
                if (sub != ro) __syncthreads();                         // the previous slices' LDS sums have been read
                const float* pin = a.partial_base + (long long)n * a.ntiles * NO;
                const unsigned epoch = (unsigned)(tstep0 + n + 1);          // the batch's Adam step: the tag of its exchange granules
                if (sub + per_set < NO / 32)
                    lin_reduce_pair(pin, a.M_base + (long long)n * NO, a.ntiles, lin_smem, sub, sub + per_set, NO, a.comm, epoch, a.status);
                else lin_reduce<true>(pin, a.M_base + (long long)n * NO, a.ntiles, lin_smem, sub, NO, &a.comm, epoch, a.status);
            }
            lin_wait_vmcnt<0>();                               // every storing wave drains its write-through stores ...
            __syncthreads();                                   // ... before the one lane that signals for the workgroup
            LIN_NOWQ(r2);
            racc_w += r1 - r0; racc_r += r2 - r1;
            if (rb == 5) { LIN_PUT(40, racc_w); LIN_PUT(41, racc_r); }
            if (rb == 5 && n < 3) { LIN_PUT(56 + 2 * n, r1); LIN_PUT(57 + 2 * n, r2); }
            if (t == 0) __hip_atomic_fetch_add(a.cnt_reduce + n, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            { [[maybe_unused]] unsigned long long te = 0; LIN_NOW(te); LIN_PUTMAX(128 + n, te); }
        }
    } else {
        // ---- streamers: workgroup sid takes tiles sid, sid + S, ... of every batch, in batch order, through a ring of THREE LDS
        // slots.  Iteration i multiplies item i while the LDS-DMA pieces of item i + 2 are issued between its k-steps (the issue
        // cost hides behind the matrix pipe) and item i + 1 is in flight.  Per wave the memory operations retire in issue order
        //     ... P(i) | S(i - 2) | P(i + 1) | S(i - 1) | [iteration i:] P(i + 2) | S(i)        (P: pieces, S: the partial image's store)
        // so ONE counted wait at the top of iteration i -- all but P(i + 1) and S(i - 1) -- says that tile i has landed and that
        // this wave's share of image i - 2 has left; the barrier behind it makes both true for the workgroup, and one lane signals
        // image i - 2.  The batch pointers come in as kernel arguments and live in LDS.
        const int sid = b - a.has_update - a.n_reduce, S = a.n_stream;
        { [[maybe_unused]] unsigned long long te = 0; LIN_NOWQ(te); if (sid == 0) LIN_PUT(50, te); if (sid == S / 2) LIN_PUT(51, te); if (sid == S - 1) LIN_PUT(52, te); }
        const int wave = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63;
        const LinTile tl(a.D, a.L, a.T);
        const int stride = max(tl.bytes, lin_scratch_bytes(NB));        // a slot doubles as the combine's cross-wave scratch
        const int v_off = 3 * stride, vr_off = v_off + 4 * a.T, c_off = vr_off + 4 * a.T;
        const float** tab = reinterpret_cast<const float**>(lin_smem + c_off + 16);            // [3][kLinMaxPersist]
        if constexpr (!GEN) {
            if (t < 3 * kLinMaxPersist) {
                const int which = t / kLinMaxPersist, n = t % kLinMaxPersist;
                tab[t] = n < N ? (which == 0 ? src.x : which == 1 ? src.z1 : src.z2)[n] : nullptr;
            }
        }
        [[maybe_unused]] float* const Amat = reinterpret_cast<float*>(lin_smem + c_off + 16);     // GEN: the dataset's mixing matrix (<= 16 x 16), where the other form keeps its pointer tables
        if constexpr (GEN) {
            if (src.kind == 0 && t < src.dd * src.did) Amat[t] = src.A[t];
        }
        const int valid_last = a.B - (a.ntiles - 1) * a.T;              // rows of a batch's last tile
        lin_write_vcol(reinterpret_cast<float*>(lin_smem + v_off), a.T, a.T, t);
        lin_write_vcol(reinterpret_cast<float*>(lin_smem + vr_off), a.T, valid_last, t);
        if (t == 0) { *reinterpret_cast<float*>(lin_smem + c_off) = 0.f; *reinterpret_cast<unsigned*>(lin_smem + c_off + 8) = 0u; }      // the zero word; the draw's ticket counter
        __syncthreads();
        const int per_batch = sid < a.ntiles ? (a.ntiles - sid + S - 1) / S : 0, items = N * per_batch;
        // Wave roles inside a streamer: waves 0 .. 3 (one per SIMD) multiply, waves 4 .. 7 load (LDS-DMA pieces of the tile two ahead,
        // or its Philox draw): with all 8 waves doing both, each wave's ~650 scalar / vector / LDS instructions per tile and its 54
        // MFMAs simply added up (streamers alone 4.7 us per tile; without the MFMAs 3.4, without the LDS-DMA 3.2, without both 2.1:
        // tools/lin_ablate.sh) -- a wave cannot issue anything else while it waits for the matrix pipe to take its next MFMA.
        const bool loader = wave >= kLinCW;
        const int lw = wave - kLinCW;                                                            // loader waves: 0 .. NLW - 1
        constexpr int NLW = LNW - kLinCW;
        auto share = [&](int n) { return lw < n ? (n - lw + NLW - 1) / NLW : 0; };
        const int npw = (GEN || !loader) ? 0 : share(tl.nz1) + share(tl.nx) + share(tl.np - tl.nz1 - tl.nx);  // this wave's pieces of a tile: its share of each tensor's
        const int sw = wave < (NO / 4 + 63) / 64 ? 1 : 0;                                        // does this wave store a share of an image?
        // item i = tile sid + (i % per_batch) S of batch i / per_batch, in slot i % 3: walked by cursors (a run-time integer division
        // costs ~40 instructions, and the loop wanted a dozen per tile)
        struct Pos {
            int idx, n, r, slot;
            __device__ __forceinline__ void advance(int per_batch) { ++idx; slot = slot == 2 ? 0 : slot + 1; if (++r == per_batch) { r = 0; ++n; } }
        };
        auto pos_tile = [&](const Pos& q) { return sid + q.r * S; };
        Pos p_cur{0, 0, 0, 0}, p_load{0, 0, 0, 0}, p_sig{0, 0, 0, 0};      // the item being multiplied / the next to load or draw / the next to signal
        LinTileSrc nxt;                                       // the item whose pieces are being issued
        nxt.on = false;
        auto prepare = [&]() __attribute__((always_inline)) {             // the next item in order (p_load), which it advances
            if constexpr (!GEN) {
                nxt.on = false;
                if (p_load.idx < items) {
                    const int n = p_load.n;
                    nxt.prepare(a, tab[n], tab[kLinMaxPersist + n], tab[2 * kLinMaxPersist + n], pos_tile(p_load), lin_smem + p_load.slot * stride);
                }
                p_load.advance(per_batch);
            }
        };
        auto issue_pieces = [&]() __attribute__((always_inline)) {      // this wave's share of that item
            if constexpr (!GEN) { if (!(VAEK_LIN_ABL & 8) && nxt.on) nxt.issue_tile(tl, lw, NLW, lane); }
        };
        // GEN: the draw of item i in ROUNDS dealt to the GT = 256 threads of the loader waves.  A unit of work is a row of x or one
        // latent block (4 normals: block z = row * nzb + q holds columns 4 q .. of the row's [z1 | z2]): first the rows of x (thread =
        // row, ceil(T / GT) rounds), then rounds that give every thread TWO latent blocks (two independent Philox chains), then what
        // is left, one block per thread.  Rows past the batch end are written as zeros.  WHO draws: a unit of 64 draw threads of one
        // round goes to whichever wave takes the next ticket (an LDS counter) -- the loading waves from the barrier on, the multiplying
        // waves once their products are done: the draw is vector-ALU work (9 Philox blocks + Box-Muller per sample: 8.7 us per tile on
        // four waves), the products keep a wave's matrix pipe busy for 2.2 us of it.  What a unit writes depends on the unit alone.
        constexpr int GT = NLW * 64;
        [[maybe_unused]] const int gD = DT ? DT : a.D, gL = LT ? LT : a.L, nzb = (gL + gD + 3) / 4;
        [[maybe_unused]] const int nlat = a.T * nzb, xr = (a.T + GT - 1) / GT, npair = nlat / (2 * GT), rem = nlat - 2 * GT * npair;
        [[maybe_unused]] const int gen_rounds = xr + npair + (rem + GT - 1) / GT;
        [[maybe_unused]] const unsigned step0 = (unsigned)a.step_dev[0];          // (the updater stores the counter at the very end of the launch)
        // (copies: read through `src` the generator's scalars are re-fetched from the kernel-argument segment inside the loops)
        [[maybe_unused]] int g_kind = 0, g_dd = 0, g_did = 0; [[maybe_unused]] float g_noise = 0.f; [[maybe_unused]] unsigned g_tag = 0;
        if constexpr (GEN) { g_kind = src.kind; g_dd = src.dd; g_did = src.did; g_noise = src.noise_std; g_tag = src.tag; }
        auto gen_round = [&](const Pos& q, int k, int gt) __attribute__((always_inline)) {       // round k of item q, as draw thread gt of GT
            if constexpr (GEN) {
                if (q.idx < items && k < gen_rounds) {
                    const uint2 key = make_uint2((unsigned)src.seed, (unsigned)(src.seed >> 32));
                    const long long row_lo = (long long)pos_tile(q) * a.T;
                    const unsigned step = step0 + (unsigned)q.n;
                    char* slot = lin_smem + q.slot * stride;
                    auto put = [&](int z, const float (&n4)[4]) __attribute__((always_inline)) {      // latent block z of the tile
                        const int r = z / nzb, q = z - r * nzb, c0 = 4 * q;
                        const bool live = row_lo + r < a.B;
                        const f32x4 v4 = live ? f32x4{n4[0], n4[1], n4[2], n4[3]} : f32x4{0.f, 0.f, 0.f, 0.f};
                        float* z1r = reinterpret_cast<float*>(slot) + r * gL;
                        float* z2r = reinterpret_cast<float*>(slot + tl.oZ2) + r * gD;
                        if (c0 + 3 < gL && gL % 4 == 0) *reinterpret_cast<f32x4*>(z1r + c0) = v4;
                        else if (c0 >= gL && (c0 - gL) + 3 < gD && gD % 4 == 0 && gL % 4 == 0) *reinterpret_cast<f32x4*>(z2r + (c0 - gL)) = v4;
                        else {
#pragma unroll
                            for (int c = 0; c < 4; ++c) {
                                if (c0 + c < gL) z1r[c0 + c] = v4[c];
                                else if (c0 + c < gL + gD) z2r[c0 + c - gL] = v4[c];
                            }
                        }
                    };
                    auto draw = [&](int z, float (&n4)[4]) __attribute__((always_inline)) {
                        const int r = z / nzb;
                        latent_block(src, step, src.row0 + row_lo + r, key, z - r * nzb, n4);
                    };
                    if (k >= xr && k < xr + npair) {
                        const int za = 2 * GT * (k - xr) + gt;
                        float na[4], nb[4];
                        draw(za, na); draw(za + GT, nb);
                        put(za, na); put(za + GT, nb);
                    } else if (k >= xr + npair) {
                        const int z = 2 * GT * npair + GT * (k - xr - npair) + gt;
                        if (z < nlat) {
                            float n4[4];
                            draw(z, n4);
                            put(z, n4);
                        }
                    } else {
                        const int row = gt + GT * k;
                        if (row < a.T) {
                            // the row of x (datasets.py:183-195 / :75-84), the arithmetic of rng_dev.h's dataset_cols4 step for step --
                            // with the mixing matrix read from LDS (through the scalar cache its loads cost 2.8 us per tile)
                            const long long lrow = row_lo + row, grow = src.row0 + lrow;
                            const bool live = lrow < a.B;
                            float nrm[16];
                            dataset_normals(src, step, grow, key, nrm);
                            float* xr_ = reinterpret_cast<float*>(slot + tl.oX) + row * gD;
                            float inv = 0.f;
                            if (g_kind == 2) {
                                float nsq = 0.f;
#pragma unroll
                                for (int d = 0; d < 16; ++d) if (d < g_dd) nsq = fmaf(nrm[d], nrm[d], nsq);
                                inv = 1.f / sqrtf(nsq);
                            }
                            for (int c0 = 0; c0 < gD; c0 += 4) {
                                float o[4] = {0.f, 0.f, 0.f, 0.f};
                                if (c0 < g_dd) {
#pragma unroll
                                    for (int c = 0; c < 4; ++c) {
                                        const int d = c0 + c;
                                        float v = 0.f;
                                        if (d < g_dd) {
                                            if (g_kind == 0) {
#pragma unroll
                                                for (int kk = 0; kk < 16; ++kk) if (kk < g_did) v = fmaf(Amat[d * g_did + kk], nrm[kk], v);
                                            } else {
#pragma unroll
                                                for (int kk = 0; kk < 16; ++kk) v = (kk == d) ? nrm[kk] : v;
                                                v *= inv;
                                            }
                                        }
                                        o[c] = v;
                                    }
                                }
                                if (g_kind == 0 && g_noise > 0.f) {      // noise normals: blocks (did+3)/4 .. of the same stream
                                    float n4[4];
                                    normals4(philox4x32_10(make_uint4((unsigned)grow, (unsigned)((g_did + 3) / 4 + c0 / 4), step, g_tag), key), n4);
#pragma unroll
                                    for (int c = 0; c < 4; ++c) o[c] = fmaf(g_noise, n4[c], o[c]);
                                }
                                const f32x4 v4 = live ? f32x4{o[0], o[1], o[2], o[3]} : f32x4{0.f, 0.f, 0.f, 0.f};
                                if (gD % 4 == 0) *reinterpret_cast<f32x4*>(xr_ + c0) = v4;
                                else {
#pragma unroll
                                    for (int c = 0; c < 4; ++c) if (c0 + c < gD) xr_[c0 + c] = v4[c];
                                }
                            }
                        }
                    }
                }
            }
        };
        [[maybe_unused]] unsigned* const ticket = reinterpret_cast<unsigned*>(lin_smem + c_off + 8);       // zeroed between two barriers before every use
        // every wave: units of the draw of q0 (and then of q1, nitems == 2) until the tickets run out
        auto gen_share = [&](const Pos& q0, const Pos& q1, int nitems) __attribute__((always_inline)) {
            if constexpr (GEN) {
                const unsigned units = (unsigned)gen_rounds * NLW, total = units * (unsigned)nitems;
#pragma unroll 1
                for (;;) {
                    unsigned u = 0;
                    if (lane == 0) u = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    u = (unsigned)__builtin_amdgcn_readfirstlane((int)u);
                    if (u >= total) break;
                    const bool second = u >= units;
                    const unsigned v = second ? u - units : u;
                    gen_round(second ? q1 : q0, (int)(v / NLW), (int)(v % NLW) * 64 + lane);
                }
            }
        };
        unsigned* const my_shard = a.cnt_stream + (b & (kLinShards - 1)) * kLinShardStride;
        auto signal = [&]() {                                 // the next image in order is out (ONE lane, behind every wave's drain + a barrier)
            if (t == 0) __hip_atomic_fetch_add(my_shard + p_sig.n * kLinShards * kLinShardStride, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            p_sig.advance(per_batch);
        };
        if constexpr (GEN) {
            Pos q1 = p_load;
            q1.advance(per_batch);
            gen_share(p_load, q1, 2);                          // items 0 and 1: every wave draws
            p_load.advance(per_batch); p_load.advance(per_batch);
            lin_barrier();
            if (t == 0) *ticket = 0u;                          // (the loop's first barrier stands between this and the next ticket)
        } else if (loader) {
            prepare();
            issue_pieces();
            prepare();
            issue_pieces();
        }
        [[maybe_unused]] unsigned long long s0 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0, s5 = 0, s6 = 0, sacc_i = 0, sacc_l = 0, sacc_f = 0, sacc_m = 0, sacc_p = 0, sacc_b = 0;
        for (int i = 0; i < items; ++i) {
            LIN_NOWQ(s0);
            lin_wait_vmcnt_upto((i + 1 < items ? npw : 0) + (i >= 1 ? sw : 0));       // tile i has landed, this wave's share of image i - 2 is out
            LIN_NOWQ(s1);
            lin_barrier();                                     // ... for every wave; and everybody is done with slot (i + 2) % 3 (the scratch of i - 1)
            LIN_NOWQ(s2);
            if (i >= 3) signal();                              // image i - 2 (image 0 left at the end of iteration 0)
            const int n = p_cur.n, tile = pos_tile(p_cur), slot_off = p_cur.slot * stride;
            char* slot = lin_smem + slot_off;
            const bool ragged = tile == a.ntiles - 1 && valid_last < a.T;
            if constexpr (!GEN) {
                if (ragged) lin_fix_ragged(a, tl, tab[n], tab[kLinMaxPersist + n], tab[2 * kLinMaxPersist + n], tile, slot, t);
            }
            LIN_NOWQ(s3);
            f32x4 acc[NB * (NB + 1) / 2];
            if (!loader) lin_tile_products<NB, JT, kLinCW>(a, tl, lin_smem, slot_off, ragged ? vr_off : v_off, c_off, acc, lane, wave, [](int) {});
            if constexpr (GEN) {
                gen_share(p_load, p_load, 1);                  // item i + 2: the loading waves at once, the multiplying waves behind their products
                p_load.advance(per_batch);
            } else if (loader && i > 0) {
                // (iteration 0 issues the pieces of tile 2 only behind its early signal: the drain in front of that signal would
                // otherwise wait for them to land)
                prepare();
                issue_pieces();
            }
            LIN_NOWQ(s5);
            lin_barrier();                                     // every wave has read its last operand: the slot turns into scratch
            if constexpr (GEN) { if (t == 0) *ticket = 0u; }   // (nobody takes a ticket before the next iteration's first barrier)
            LIN_NOWQ(s6);
            lin_tile_combine<NB, true, kLinCW>(acc, slot, a.partial_base + ((long long)n * a.ntiles + tile) * NO, t, wave, lane);
            if (i == 0) {                                      // the launch's first image: out at once (pipeline fill), not two tiles later
                lin_wait_vmcnt<0>();
                lin_barrier();
                signal();
                if (sid == 0) { [[maybe_unused]] unsigned long long te = 0; LIN_NOWQ(te); LIN_PUT(53, te); }
                if constexpr (!GEN) {
                    if (loader) {
                        prepare();
                        issue_pieces();
                    }
                }
            }
            LIN_NOWQ(s4);
            if (i + 1 < items) { sacc_i += s1 - s0; sacc_l += s2 - s1; sacc_f += s3 - s2; sacc_m += s4 - s3; sacc_p += s5 - s3; sacc_b += s6 - s5; }
            if (sid == 7) { LIN_PUT(42, sacc_i); LIN_PUT(43, sacc_l); LIN_PUT(44, sacc_f); LIN_PUT(45, sacc_m); LIN_PUT(46, (unsigned long long)items); LIN_PUT(47, sacc_p); LIN_PUT(48, sacc_b); }
            p_cur.advance(per_batch);
        }
        lin_wait_vmcnt<0>();
        lin_barrier();
        if (items - 2 >= 1) signal();
        if (items - 1 >= 1) signal();
    }
}

// zeroes the arrival counters of a workspace (once per workspace: afterwards every launch leaves them zero) and, unless the
// workspace carries the mark of an earlier initialisation, the sticky status word
constexpr unsigned kLinMagic = 0x4c494e33u;
__global__ void lin_init_kernel(unsigned* cnt, int n_words, unsigned* status) {
    for (int k = threadIdx.x; k < n_words; k += blockDim.x) cnt[k] = 0u;
    if (threadIdx.x == 0 && status[1] != kLinMagic) { status[0] = 0u; status[1] = kLinMagic; }
}

// ---- host side ------------------------------------------------------------------------------------------------------------------
// 16-feature blocks of the kernel instantiation that serves this model: 3 (up to 48 features: the metric's 45) or 4
static int lin_nb(const vaek_ctx* c) { return (c->L + 2 * c->D + 1 + 15) / 16 <= 3 ? 3 : 4; }
static int lin_no(const vaek_ctx* c) { const int NB = lin_nb(c); return NB * (NB + 1) / 2 * 256; }

// the model / batch shapes the moment formulation covers (whatever the number of ranks)
static bool lin_steps_shape_ok(const vaek_ctx* c) {
    return c->cfg.n_enc_hidden == 0 && c->cfg.n_dec_hidden == 0 && !c->cfg.sigmoid_decoder && c->cfg.dtype == VAEK_F32 &&
           c->L + 2 * c->D + 1 <= 64 && (long long)c->B * std::min(c->D, c->L) >= 8;
}
static bool lin_persist_supported(const vaek_ctx* c);
// data parallel: the persistent form only, and only once the P2P communicator (vaek_comm_create / _init) carries the moment region
bool lin_steps_supported(const vaek_ctx* c) {
    if (!lin_steps_shape_ok(c)) return false;
    if (c->cfg.world == 1) return true;
    return lin_persist_supported(c) && c->comm.ready && c->comm.lin_bytes > 0;
}
// LDS of a persistent streamer: three tile slots (each doubling as the combine's cross-wave scratch), the two validity columns,
// the zero word, the batch pointer tables
static size_t lin_ring_bytes(const vaek_ctx* c, int T) {
    const LinTile tl(c->D, c->L, T);
    return 3 * std::max((size_t)tl.bytes, (size_t)lin_scratch_bytes(lin_nb(c))) + 8 * (size_t)T + 16 + 3 * kLinMaxPersist * sizeof(void*) + 64;
}
constexpr size_t kLinMaxLds = 160 * 1024;
// Samples per tile.  256, unless a slightly taller tile lets every streamer of the persistent launch take exactly ONE tile per
// batch (the metric: 65 536 samples = 228 tiles of 288 on the 231 CUs the updater and the reducers leave).
static int lin_tile_rows(const vaek_ctx* c) {
    const int smax = c->n_cu - 1 - kLinReduceSets * kLinReduceWgs;
    if (smax < 16 || c->B <= 256 * smax) return 256;
    const int T = 32 * (int)(((long long)c->B + 32ll * smax - 1) / (32ll * smax));
    const LinTile tl(c->D, c->L, T);
    return T <= 512 && (tl.np + LNW - kLinCW - 1) / (LNW - kLinCW) <= 18 && lin_ring_bytes(c, T) <= kLinMaxLds ? T : 256;
}
static int lin_ntiles(const vaek_ctx* c) { const int T = lin_tile_rows(c); return (c->B + T - 1) / T; }
static size_t lin_slot_stride(const vaek_ctx* c) {      // one tile slot, large enough to double as the combine's cross-wave scratch
    const LinTile tl(c->D, c->L, lin_tile_rows(c));
    return std::max((size_t)tl.bytes, (size_t)lin_scratch_bytes(lin_nb(c)));
}
static size_t lin_lds_need(const vaek_ctx* c) {         // launch-per-step form: one slot + validity column + zero word | updater | reducer sums
    const int NB = lin_nb(c), D = c->D, L = c->L;
    const size_t upd = std::max(NB == 3 ? LinUpd<3, 0, 0>::lds_bytes(D, L) : LinUpd<4, 0, 0>::lds_bytes(D, L), LinUpdM<3, 0, 0>::lds_bytes(D, L, (int)c->P));
    return std::max(lin_slot_stride(c) + 4 * (size_t)lin_tile_rows(c) + 16, std::max(upd, (size_t)2 * 16 * 32 * sizeof(double))) + 64;
}
// The persistent form gives every workgroup a CU of its own (the updater's float64 chains and the streamers' MFMA loops both
// lose a factor ~2 when they share one): each workgroup asks for more than half a CU's LDS -- the streamers need it anyway for
// their ring of tile slots -- and the grid stays within the CU count.
static size_t lin_persist_lds(const vaek_ctx* c) {
    return std::max(std::max(lin_lds_need(c), lin_ring_bytes(c, lin_tile_rows(c))), (size_t)82 * 1024);
}
static bool lin_persist_supported(const vaek_ctx* c) {
    const LinTile tl(c->D, c->L, lin_tile_rows(c));
    return lin_steps_shape_ok(c) && lin_nb(c) == 3 && LinUpdM<3, 0, 0>::shape_ok(c->D, c->L) && lin_persist_lds(c) <= kLinMaxLds &&
           (tl.np + LNW - kLinCW - 1) / (LNW - kLinCW) <= 18 && c->n_cu >= 1 + kLinReduceSets * kLinReduceWgs + 16;
}
// streamer workgroups of the persistent launch: the CUs the updater and the reducers leave, tiles dealt evenly
static int lin_persist_streamers(const vaek_ctx* c) {
    const int ntiles = lin_ntiles(c), smax = c->n_cu - 1 - kLinReduceSets * kLinReduceWgs;
    const int per = (ntiles + smax - 1) / smax;
    return (ntiles + per - 1) / per;
}

// workspace: [cnt_stream: 64 batches x 8 shards x 128 B][cnt_reduce: 64 words][status word, init mark][M slots][partial image slots]
constexpr size_t kLinCntStreamBytes = (size_t)kLinMaxPersist * kLinShards * kLinShardStride * 4, kLinCntBytes = kLinCntStreamBytes + 1024, kLinHeadBytes = kLinCntBytes + 256;
struct LinWs { float* partial; double* M; unsigned* cnt; unsigned* cnt_reduce; unsigned* status; size_t total; };
static LinWs lin_carve(const vaek_ctx* c, char* base) {
    const size_t no = lin_no(c), ntiles = lin_ntiles(c);
    const int slots = lin_persist_supported(c) ? kLinMaxPersist : 2;
    LinWs w{};
    size_t off = 0;
    w.cnt = reinterpret_cast<unsigned*>(base + off); w.cnt_reduce = reinterpret_cast<unsigned*>(base + kLinCntStreamBytes);
    w.status = reinterpret_cast<unsigned*>(base + kLinCntBytes); off += kLinHeadBytes;
    w.M = reinterpret_cast<double*>(base + off); off += (size_t)slots * no * sizeof(double) + 256;
    w.partial = reinterpret_cast<float*>(base + off); off += (size_t)slots * ntiles * no * sizeof(float) + 256;
    w.total = (off + 255) / 256 * 256;
    return w;
}
size_t lin_steps_workspace_bytes(const vaek_ctx* c) { return lin_steps_shape_ok(c) ? lin_carve(c, nullptr).total : 0; }
// bytes of the moment-exchange region of the P2P communicator's buffer (0: this context never exchanges moments)
size_t lin_comm_bytes(const vaek_ctx* c) {
    if (c->cfg.world < 2 || !lin_persist_supported(c)) return 0;
    return (size_t)kLinCommBanks * c->cfg.world * 2 * lin_no(c) * sizeof(unsigned long long);
}

static int lin_fill_common(const vaek_ctx* c, LinArgs& a, float* params, float* grads, float* m, float* v, int32_t* step_dev, float lr) {
    a.B = c->B; a.D = c->D; a.L = c->L; a.T = lin_tile_rows(c); a.ntiles = lin_ntiles(c);
    a.params = params; a.grads = grads; a.m = m; a.v = v; a.step_dev = step_dev; a.lr = lr;
    // data parallel: M is summed over the ranks before the updater sees it, so its "rows" are the GLOBAL batch
    const double rows = c->cfg.world > 1 ? (double)c->Bt : (double)c->B;
    a.inv_bt = (float)(1.0 / (double)c->Bt); a.eps_cli = c->cfg.eps_cli; a.rows = (float)rows;
    a.rows_over_bt = (float)(rows / (double)c->Bt); a.off_eps = (int)c->off_eps; a.P = (int)c->P;
    a.comm = LinComm{};
    a.comm.world = 1;
    if (c->cfg.world > 1 && c->comm.ready && c->comm.lin_bytes > 0) {
        a.comm.world = c->cfg.world; a.comm.rank = c->cfg.rank; a.comm.ng2 = 2 * lin_no(c);
        for (int r = 0; r < c->cfg.world; ++r)
            a.comm.peer[r] = reinterpret_cast<unsigned long long*>(static_cast<char*>(c->comm.peers[r]) + c->comm.lin_off);
    }
    a.loss_hist = c->loss_hist; a.loss_hist_cap = c->loss_hist_cap;
    static const int stagger = getenv("VAEK_LIN_STAGGER") ? atoi(getenv("VAEK_LIN_STAGGER")) : 3;      // diagnostic override
    a.stagger = stagger;
    return 0;
}

// once per (context, workspace): the arrival counters start from zero (every persistent launch leaves them zero again)
static int lin_ensure_init(vaek_ctx* c, const LinWs& w, void* ws, hipStream_t st) {
    if (c->lin_ws_inited == ws && !c->lin_ws_reinit) return VAEK_OK;
    hipLaunchKernelGGL(lin_init_kernel, dim3(1), dim3(1024), 0, st, w.cnt, (int)(kLinCntBytes / 4), w.status);
    VAEK_HIP_CHECK(hipGetLastError());
    c->lin_ws_inited = ws;
    c->lin_ws_reinit = false;
    return VAEK_OK;
}

// the in-launch batch generator serves the datasets a linear VAE can be trained on (the sigmoid dataset brings a second decoder)
bool lin_steps_gen_supported(const vaek_ctx* c, int kind) { return lin_steps_supported(c) && lin_persist_supported(c) && (kind == 0 || kind == 2); }

// gen != nullptr: the batches are drawn inside the launch (xs / z1s / z2s unused)
static int lin_train_steps_impl(vaek_ctx* c, float* params, float* grads, float* m, float* v, int32_t* step_dev, const float* const* xs,
                                const float* const* z1s, const float* const* z2s, const BatchArgs* gen, int n_steps, float lr, void* ws, hipStream_t st) {
    const int NB = lin_nb(c), no = lin_no(c), ntiles = lin_ntiles(c);
    const LinWs w = lin_carve(c, static_cast<char*>(ws) + c->ws_lin);
    static const char* env = getenv("VAEK_LIN_PERSIST");              // diagnostic: 0 forces the launch-per-step form
    const bool persistent = lin_persist_supported(c) && (gen || c->cfg.world > 1 || !(env && atoi(env) == 0));   // data parallel, in-launch draw: persistent form only
    if (gen && !persistent) { set_error("vaek_train_steps_gen: this context has no persistent form"); return VAEK_ERR_INVALID; }
    // the metric's shape with its dimensions (and the k-steps per wave of its 288-row tile) at compile time; every other linear
    // model on the run-time instantiations
    const int which = (c->D == 12 && c->L == 20 && (!persistent || lin_tile_rows(c) == 288)) ? 0 : (NB <= 3 ? 1 : 2);
    const size_t lds = persistent ? lin_persist_lds(c) : lin_lds_need(c);
    int dev = 0;
    VAEK_HIP_CHECK(hipGetDevice(&dev));
    static thread_local unsigned char attr_set[3][3][64] = {};        // hipFuncSetAttribute is per device
    const auto set_attr = [&](const void* fn) -> int {
        unsigned char& done = attr_set[persistent ? (gen ? 2 : 1) : 0][which][dev & 63];
        if (!done) {
            VAEK_HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLinMaxLds));   // a cap, not a request
            done = 1;
        }
        return VAEK_OK;
    };
    if (persistent) {
        typedef void (*LinPersist)(const LinArgs, const LinPtrs);
        typedef void (*LinPersistGen)(const LinArgs, const BatchArgs);
        const LinPersist fn = which == 0 ? lin_persist_kernel<3, 12, 20, 18, false> : lin_persist_kernel<3, 0, 0, 0, false>;
        const LinPersistGen fng = which == 0 ? lin_persist_kernel<3, 12, 20, 18, true> : lin_persist_kernel<3, 0, 0, 0, true>;
        if (int rc = set_attr(gen ? (const void*)fng : (const void*)fn)) return rc;
        if (int rc = lin_ensure_init(c, w, ws, st)) return rc;
        for (int s0 = 0; s0 < n_steps; s0 += kLinMaxPersist) {
            const int n = std::min(kLinMaxPersist, n_steps - s0);
            LinArgs a{};
            lin_fill_common(c, a, params, grads, m, v, step_dev, lr);
            a.persistent = 1; a.n_steps = n;
            a.sets = kLinReduceSets;
            a.has_update = 1; a.n_reduce = kLinReduceSets * std::min(kLinReduceWgs, no / 32); a.n_stream = lin_persist_streamers(c);
            static const int proles = getenv("VAEK_LIN_ROLES") ? atoi(getenv("VAEK_LIN_ROLES")) : 7;    // diagnostic (tools/lin_roles.sh)
            if (!(proles & 4)) a.has_update = 0;
            if (!(proles & 2)) { a.has_update = 0; a.n_reduce = 0; }
            if (!(proles & 1)) a.n_stream = 0;
            if (proles != 7) c->lin_ws_reinit = true;          // nobody re-zeroes the counters without the updater: start over next call
            a.partial_base = w.partial; a.M_base = w.M;
            a.cnt_stream = w.cnt; a.cnt_reduce = w.cnt_reduce; a.status = w.status;
            const dim3 grid((unsigned)(a.has_update + a.n_reduce + a.n_stream));
            if (gen) {
                ProfScope ps("lin_moments_persistent_gen", st);
                launch_k(ps, fng, grid, dim3(LNT), lds, st, a, *gen);
            } else {
                LinPtrs ptrs{};
                for (int i = 0; i < n; ++i) { ptrs.x[i] = xs[s0 + i]; ptrs.z1[i] = z1s[s0 + i]; ptrs.z2[i] = z2s[s0 + i]; }
                ProfScope ps("lin_moments_persistent", st);
                launch_k(ps, fn, grid, dim3(LNT), lds, st, a, ptrs);
            }
        }
        VAEK_HIP_CHECK(hipGetLastError());
        return VAEK_OK;
    }
    typedef void (*LinKernel)(const LinArgs);
    const LinKernel fn = which == 0 ? lin_step_kernel<3, 12, 20> : which == 1 ? lin_step_kernel<3, 0, 0> : lin_step_kernel<4, 0, 0>;
    if (int rc = set_attr((const void*)fn)) return rc;
    static const int roles = getenv("VAEK_LIN_ROLES") ? atoi(getenv("VAEK_LIN_ROLES")) : 7;   // diagnostic: 1 stream, 2 reduce, 4 update
    const size_t pstride = (size_t)ntiles * no;
    for (int n = 0; n < n_steps + 2; ++n) {       // launch n: stream batch n, reduce batch n - 1, update batch n - 2
        LinArgs a{};
        lin_fill_common(c, a, params, grads, m, v, step_dev, lr);
        a.has_update = (n >= 2 && (roles & 4)) ? 1 : 0;
        a.n_reduce = (n >= 1 && n <= n_steps && (roles & 2)) ? no / 32 : 0;
        a.n_stream = (n < n_steps && (roles & 1)) ? ntiles : 0;
        if (a.has_update + a.n_reduce + a.n_stream == 0) continue;
        if (a.n_stream) { a.x = xs[n]; a.z1 = z1s[n]; a.z2 = z2s[n]; a.partial_out = w.partial + (n & 1) * pstride; }
        if (a.n_reduce) { a.partial_in = w.partial + ((n - 1) & 1) * pstride; a.M_out = w.M + ((n - 1) & 1) * no; }
        a.M_in = w.M + (n & 1) * no;               // (n - 2) & 1
        ProfScope ps(a.n_stream ? "lin_moments_step" : "lin_moments_drain", st);
        launch_k(ps, fn, dim3((unsigned)(a.has_update + a.n_reduce + a.n_stream)), dim3(LNT), lds, st, a);
    }
    VAEK_HIP_CHECK(hipGetLastError());
    return VAEK_OK;
}

int lin_train_steps(vaek_ctx* c, float* params, float* grads, float* m, float* v, int32_t* step_dev, const float* const* xs,
                    const float* const* z1s, const float* const* z2s, int n_steps, float lr, void* ws, hipStream_t st) {
    return lin_train_steps_impl(c, params, grads, m, v, step_dev, xs, z1s, z2s, nullptr, n_steps, lr, ws, st);
}
int lin_train_steps_gen(vaek_ctx* c, float* params, float* grads, float* m, float* v, int32_t* step_dev, const BatchArgs& gen, int n_steps,
                        float lr, void* ws, hipStream_t st) {
    return lin_train_steps_impl(c, params, grads, m, v, step_dev, nullptr, nullptr, nullptr, &gen, n_steps, lr, ws, st);
}

// ---- the two halves of ONE step of the launch-per-step form, for a host-side collective between them ------------------------------
// Data parallel without the P2P communicator (GradExchange mode "rccl"): the moment matrix is additive over the ranks' shards
// (the loss is a batch mean, networks.py:97-98), so a rank forms the M of its shard (lin_moments: streamers, then reducers, two
// launches ordered by the stream), torch.distributed sums the 12 KB float64 image over the ranks (RCCL; gloo in rehearsal -- every
// rank receives the same bits), and every rank applies the identical update (lin_update: rows = the GLOBAL batch).
size_t lin_moment_len(const vaek_ctx* c) { return lin_steps_shape_ok(c) ? (size_t)lin_no(c) : 0; }
int lin_moments(vaek_ctx* c, const float* x, const float* z1, const float* z2, double* M_out, void* ws, hipStream_t st) {
    const int NB = lin_nb(c), no = lin_no(c), ntiles = lin_ntiles(c);
    const LinWs w = lin_carve(c, static_cast<char*>(ws) + c->ws_lin);
    const int which = (c->D == 12 && c->L == 20) ? 0 : (NB <= 3 ? 1 : 2);
    typedef void (*LinKernel)(const LinArgs);
    const LinKernel fn = which == 0 ? lin_step_kernel<3, 12, 20> : which == 1 ? lin_step_kernel<3, 0, 0> : lin_step_kernel<4, 0, 0>;
    VAEK_HIP_CHECK(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLinMaxLds));
    const size_t lds = lin_lds_need(c);
    LinArgs a{};
    lin_fill_common(c, a, nullptr, nullptr, nullptr, nullptr, nullptr, 0.f);
    a.n_stream = ntiles; a.x = x; a.z1 = z1; a.z2 = z2; a.partial_out = w.partial;
    { ProfScope ps("lin_moments_stream", st); launch_k(ps, fn, dim3((unsigned)ntiles), dim3(LNT), lds, st, a); }
    a.n_stream = 0; a.n_reduce = no / 32; a.partial_in = w.partial; a.M_out = M_out;
    a.comm = LinComm{}; a.comm.world = 1;                      // the ranks meet on the host, not in the reducers
    { ProfScope ps("lin_moments_reduce", st); launch_k(ps, fn, dim3((unsigned)a.n_reduce), dim3(LNT), lds, st, a); }
    VAEK_HIP_CHECK(hipGetLastError());
    return VAEK_OK;
}
int lin_update(vaek_ctx* c, float* params, float* grads, float* m, float* v, int32_t* step_dev, const double* M_in, float lr, hipStream_t st) {
    const int NB = lin_nb(c);
    const int which = (c->D == 12 && c->L == 20) ? 0 : (NB <= 3 ? 1 : 2);
    typedef void (*LinKernel)(const LinArgs);
    const LinKernel fn = which == 0 ? lin_step_kernel<3, 12, 20> : which == 1 ? lin_step_kernel<3, 0, 0> : lin_step_kernel<4, 0, 0>;
    VAEK_HIP_CHECK(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLinMaxLds));
    LinArgs a{};
    lin_fill_common(c, a, params, grads, m, v, step_dev, lr);
    a.has_update = 1; a.M_in = M_in;
    ProfScope ps("lin_moments_update", st);
    launch_k(ps, fn, dim3(1), dim3(LNT), lin_lds_need(c), st, a);
    VAEK_HIP_CHECK(hipGetLastError());
    return VAEK_OK;
}
bool lin_moments_supported(const vaek_ctx* c) { return lin_steps_shape_ok(c); }

// synchronous: did a bounded wait of the persistent form ever give up (a workgroup that never became resident, a lost store)?
// The word is STICKY -- no launch clears it, and while it is set every wait of every later launch returns at once (the grid
// drains, its results are garbage) -- until this call reads it: read-and-clear.
int lin_steps_status(vaek_ctx* c, void* ws, int* gave_up) {
    *gave_up = 0;
    if (!lin_steps_shape_ok(c)) return VAEK_OK;
    const LinWs w = lin_carve(c, static_cast<char*>(ws) + c->ws_lin);
    if (c->lin_ws_inited != ws) return VAEK_OK;            // no persistent launch has used this workspace yet
    unsigned s = 0;
    VAEK_HIP_CHECK(hipMemcpy(&s, w.status, sizeof(s), hipMemcpyDeviceToHost));
    *gave_up = (int)s;
    if (s) {
        VAEK_HIP_CHECK(hipMemset(w.status, 0, sizeof(unsigned)));
        c->lin_ws_reinit = true;                           // the counters of the launch that gave up are in an unknown state
    }
    return VAEK_OK;
}

}  // namespace vaek

#ifdef VAEK_LIN_STAMPS
extern "C" int vaek_debug_lin_stamps(unsigned long long* buf) {
    return hipMemcpyToSymbol(HIP_SYMBOL(vaek::g_lin_stamp_buf), &buf, sizeof(buf)) == hipSuccess ? 0 : -2;
}
#endif
