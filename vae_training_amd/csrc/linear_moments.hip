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
// Here launch n carries three roles at once:
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
// Numerics: M accumulates exact-f32 products in chains of 64 samples (one wave's 16 MFMA k-steps), summed further in
// float64; everything downstream is float64 until the final rounding of each gradient to float32.  Against the float64
// oracle the loss sits at ~1e-7 relative (tests/test_gpu_steps.py), like the sample-by-sample kernels -- but it is a
// different summation order, so this path is NOT bitwise comparable with vaek_train_step.
#include <stdlib.h>

#include "comm_dev.h"
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
    const float* const* xs; const float* const* z1s; const float* const* z2s;
    float* partial_base; double* M_base;                                          // slot n at + n * ntiles * NO resp. + n * NO
    unsigned* cnt_stream; unsigned* cnt_reduce; unsigned* status;                 // [n_steps], [n_steps], [1]; zeroed before the launch
    // ---- updater
    float* params; float* grads; float* m; float* v; int32_t* step_dev; float lr;
    float inv_bt, eps_cli, rows, rows_over_bt; int off_eps, P;
    float* loss_hist; long long loss_hist_cap;
    LinComm comm;
};

#ifdef VAEK_LIN_STAMPS      // diagnostic build (tools/lin_stamps.sh): s_memtime at the updater's phase boundaries, into a buffer nothing reads
__device__ unsigned long long* g_lin_stamp_buf = nullptr;
#define LIN_STAMP(i)                                                                                         \
    do {                                                                                                     \
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
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v)::"memory"); \
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
// wait until at most n (wave-uniform, <= 12) of this wave's memory operations are outstanding
__device__ __forceinline__ void lin_wait_vmcnt_upto(int n) {
    switch (n) {
        case 1: lin_wait_vmcnt<1>(); break;   case 2: lin_wait_vmcnt<2>(); break;   case 3: lin_wait_vmcnt<3>(); break;
        case 4: lin_wait_vmcnt<4>(); break;   case 5: lin_wait_vmcnt<5>(); break;   case 6: lin_wait_vmcnt<6>(); break;
        case 7: lin_wait_vmcnt<7>(); break;   case 8: lin_wait_vmcnt<8>(); break;   case 9: lin_wait_vmcnt<9>(); break;
        case 10: lin_wait_vmcnt<10>(); break; case 11: lin_wait_vmcnt<11>(); break; case 12: lin_wait_vmcnt<12>(); break;
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
// LDS slot of one T-sample tile: the three tensors' tiles exactly as they lie in HBM (row-major [T][L] / [T][D]), each
// padded to whole 8 KB (one pass of the 512-thread copy: every thread issues the same number of loads), the validity column
// V[T] (the "1" feature; 0 for rows past the batch end) and a zero word.
struct LinSlot {
    int z1_b, x_b, oX, oZ2, oV, oC, bytes, passes;
    __device__ __host__ LinSlot(int D, int L, int T) {
        z1_b = (L * 4 * T + 8191) / 8192 * 8192; x_b = (D * 4 * T + 8191) / 8192 * 8192;
        oX = z1_b; oZ2 = oX + x_b; oV = oZ2 + x_b; oC = oV + 4 * T; bytes = (oC + 16 + 255) / 256 * 256;
        passes = (z1_b + 2 * x_b) / 8192;
    }
};

__device__ __forceinline__ void lin_issue_tile(const LinArgs& a, const LinSlot& sl, const float* x, const float* z1, const float* z2,
                                               int tile, char* slot, int t, int wave) {
    const long long row0 = (long long)tile * a.T;
    auto copy = [&](const float* src, int cols, int bytes, int lds_off) {
        const long long tot = (long long)a.B * cols * 4, base = row0 * cols * 4;
        // the last 16-byte piece this tile may fetch: the tile's own (the slot's padding up to whole 8 KB passes re-reads it -- a
        // line this CU has just fetched -- instead of pulling the NEXT tile's rows through another XCD's L2: 10 % of the HBM reads),
        // and never past the tensor's last whole 16 bytes; what a clamped piece brings lands in LDS nobody reads, in rows that are
        // zeroed afterwards, or in the tensor's last <= 3 floats, which lin_fix_tile rewrites
        const long long last = min(tot & ~15ll, base + (long long)a.T * cols * 4) - 16;
        for (int i = 0; i < bytes / 8192; ++i) {
            long long off = base + i * 8192 + t * 16;
            off = off <= last ? off : last;
            lin_glds16(reinterpret_cast<const char*>(src) + off, slot + lds_off + i * 8192 + wave * 1024);
        }
    };
    copy(z1, a.L, sl.z1_b, 0);
    copy(x, a.D, sl.x_b, sl.oX);
    copy(z2, a.D, sl.x_b, sl.oZ2);
}

// after the tile has landed: the validity column, the zero word, and for the last tile of a ragged batch the rows past the end
__device__ __forceinline__ void lin_fix_tile(const LinArgs& a, const LinSlot& sl, const float* x, const float* z1, const float* z2,
                                             int tile, char* slot, int t) {
    const int T = a.T;
    const long long row0 = (long long)tile * T;
    const int valid = (int)min((long long)T, (long long)a.B - row0), D = a.D, L = a.L;
    if (t < T) reinterpret_cast<float*>(slot + sl.oV)[t] = t < valid ? 1.f : 0.f;
    if (t == 0) *reinterpret_cast<float*>(slot + sl.oC) = 0.f;
    if (valid < T) {
        auto patch_tail = [&](const float* src, int cols, int lds_off) {     // floats behind the tensor's last whole 16 bytes
            const long long tot = (long long)a.B * cols * 4, full = tot & ~15ll, base = row0 * cols * 4;
            if (t < (int)((tot - full) / 4) && full >= base) reinterpret_cast<float*>(slot + lds_off)[(full - base) / 4 + t] = src[full / 4 + t];
        };
        patch_tail(z1, L, 0); patch_tail(x, D, sl.oX); patch_tail(z2, D, sl.oZ2);
        for (int e = valid * L + t; e < T * L; e += LNT) reinterpret_cast<float*>(slot)[e] = 0.f;
        for (int e = valid * D + t; e < T * D; e += LNT) { reinterpret_cast<float*>(slot + sl.oX)[e] = 0.f; reinterpret_cast<float*>(slot + sl.oZ2)[e] = 0.f; }
    }
    lin_barrier();
}

// M_tile = U^T U.  The SAMPLES are dealt to the waves: wave w takes samples (T / 8) w .. -- T / 32 k-steps of 4 samples -- for
// every block of the upper block triangle, so each operand register read from LDS feeds NB (+1) MFMAs, the matrix pipes of
// the four SIMDs carry equal loads and the NBLK accumulator chains are independent.  The eight per-wave images are then
// summed through LDS (in wave order: deterministic) in the tile's own slot, which is dead by then, and leave as one image.
constexpr int lin_scratch_bytes(int NB) { return LNW * (NB * (NB + 1) / 2) * 1024; }
template <int NB, bool SC1>
__device__ __forceinline__ void lin_multiply_tile(const LinArgs& a, const LinSlot& sl, char* slot, float* out, int t, int wave) {
    constexpr int NBLK = NB * (NB + 1) / 2;
    const int lane = t & 63, g = lane >> 4, D = a.D, L = a.L;
    // where this lane's feature 16 u + (lane & 15) lives (byte offset of sample 0, byte stride per sample)
    int fb[NB], fs[NB];
#pragma unroll
    for (int u = 0; u < NB; ++u) {
        const int f = 16 * u + (lane & 15);
        if (f < L) { fb[u] = f * 4; fs[u] = L * 4; }
        else if (f < L + D) { fb[u] = sl.oX + (f - L) * 4; fs[u] = D * 4; }
        else if (f < L + 2 * D) { fb[u] = sl.oZ2 + (f - L - D) * 4; fs[u] = D * 4; }
        else if (f == L + 2 * D) { fb[u] = sl.oV; fs[u] = 4; }
        else { fb[u] = sl.oC; fs[u] = 0; }
    }
    f32x4 acc[NBLK];
#pragma unroll
    for (int k = 0; k < NBLK; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    // Wave w takes samples (T / 8) w .. + T / 8 - 1 in J = T / 32 k-steps of 4.  Within a run of 16 samples lane group g takes sample
    // s + 4 g in step s: four rows apart, i.e. 16 banks with 80- and 48-byte rows, so the lane groups one ds_read_b32 services
    // together never collide; a run shorter than 16 (T / 8 not a multiple of 16: its last r < 4 steps) takes s + r g.
    // Operands run one step ahead in a second register set, and the scheduler is told to keep it that way: left alone hipcc
    // reuses one register set and waits out the full LDS latency before every MFMA.
    const int J = a.T >> 5, wbase = (a.T >> 3) * wave, jfull = J & ~3;
    float op[2][NB];
    auto fetch = [&](int set, int j) {
        const int sample = wbase + (j < jfull ? 16 * (j >> 2) + (j & 3) + 4 * g : 4 * jfull + (j - jfull) + (J - jfull) * g);
#pragma unroll
        for (int u = 0; u < NB; ++u) op[set][u] = *reinterpret_cast<const float*>(slot + fb[u] + sample * fs[u]);
    };
    auto products = [&](int set) {
        int k = 0;
#pragma unroll
        for (int b1 = 0; b1 < NB; ++b1)
#pragma unroll
            for (int b2 = b1; b2 < NB; ++b2, ++k)
                acc[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(op[set][b1], op[set][b2], acc[k], 0, 0, 0);
    };
    fetch(0, 0);
    for (int j = 0; j < J; j += 2) {
        if (j + 1 < J) fetch(1, j + 1);
        __builtin_amdgcn_sched_barrier(0);                 // the reads of step j + 1 are issued before the MFMAs of step j
        products(0);
        __builtin_amdgcn_sched_barrier(0);
        if (j + 1 < J) {
            if (j + 2 < J) fetch(0, j + 2);
            __builtin_amdgcn_sched_barrier(0);
            products(1);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    lin_barrier();                                         // every wave has read its last operand: the slot turns into scratch
    float* scr = reinterpret_cast<float*>(slot);           // [wave][block][lane][4]: 16-byte lane-linear writes
#pragma unroll
    for (int k = 0; k < NBLK; ++k) *reinterpret_cast<f32x4*>(scr + ((wave * NBLK + k) * 64 + lane) * 4) = acc[k];
    lin_barrier();
    for (int o = t; o < NBLK * 256; o += LNT) {            // image element (row, col) of block k <- lane (row / 4) * 16 + col, register row % 4
        const int k = o >> 8, row = (o >> 4) & 15, col = o & 15, src = ((k * 64 + (row >> 2) * 16 + col) << 2) + (row & 3);
        float sum = 0.f;
#pragma unroll
        for (int w = 0; w < LNW; ++w) sum += scr[w * NBLK * 256 + src];
        if (SC1) st_sc1(out + o, sum); else out[o] = sum;
    }
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
            const int k = e >> 8, i = (e >> 4) & 15, j = e & 15;
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
                const int blk = e >> 8, i = (e >> 4) & 15, j = e & 15;
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
        const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
        const LinSlot sl(a.D, a.L, a.T);
        lin_issue_tile(a, sl, a.x, a.z1, a.z2, tile, lin_smem, t, wave);
        lin_wait_vmcnt<0>();
        __syncthreads();
        lin_fix_tile(a, sl, a.x, a.z1, a.z2, tile, lin_smem, t);
        lin_multiply_tile<NB, false>(a, sl, lin_smem, a.partial_out + (long long)tile * NO, t, wave);
    }
}

// ---- persistent form: up to kLinMaxPersist steps in one launch ----------------------------------------------------------------
template <int NB, int DT, int LT>
__global__ __launch_bounds__(LNT, 2) void lin_persist_kernel(const LinArgs a) {       // one workgroup per CU (its LDS request sees to that)
    extern __shared__ __attribute__((aligned(16))) char lin_smem[];
    const int b = blockIdx.x, t = threadIdx.x, N = a.n_steps;
    constexpr int NO = NB * (NB + 1) / 2 * 256;
    const int per_set = a.n_reduce / a.sets;                  // reducer workgroups per set
    if (b < a.has_update) {
        // ---- the updater: one workgroup, parameters and Adam state in registers / LDS across all N steps -------------------
        LinUpd<NB, DT, LT> u;
        u.carve(a, lin_smem);
        int tstep = a.step_dev[0];
        u.load_state(a);
        float g[LKOUT];
#pragma unroll
        for (int k = 0; k < LKOUT; ++k) g[k] = 0.f;
        LIN_STAMP(10);
        [[maybe_unused]] unsigned long long tw0 = 0, tw1 = 0, twait = 0, twait_max = 0;
        double mreg[LinUpd<NB, DT, LT>::MPT];
        bool pre_ok = false;                                   // (uniform) mreg already holds this batch's M
        for (int n = 0; n < N; ++n) {
            LIN_STAMP(0);
            u.publish_params();
            LIN_STAMP(9);
            LIN_NOW(tw0);
            if (!pre_ok) {
                lin_wait_count(a.cnt_reduce + n, (unsigned)per_set, a.status, (1u << 28) | ((unsigned)n << 16));     // (also the barrier behind publish_params)
                u.fetch_M(a.M_base + (long long)n * NO, mreg);
            } else {
                __syncthreads();
            }
            LIN_NOW(tw1);
            twait += tw1 - tw0; twait_max = tw1 - tw0 > twait_max ? tw1 - tw0 : twait_max;
            if (n == 0) LIN_PUT(13, tw1 - tw0);
            LIN_PUT(11, twait); LIN_PUT(12, twait_max);
            LIN_STAMP(8);
            u.scatter_M(mreg);
            // are the reducers of the next batch done already?  (they usually are: the streamers run ahead.)  Then its M is loaded
            // now and travels under this step's arithmetic; otherwise the top of the next iteration waits as usual.
            if (t == 0)
                u.epsv[1] = (n + 1 < N && __hip_atomic_load(a.cnt_reduce + n + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned)per_set) ? 1.0 : 0.0;
            __syncthreads();
            pre_ok = u.epsv[1] != 0.0;
            if (pre_ok) u.fetch_M(a.M_base + (long long)(n + 1) * NO, mreg);
            ++tstep;
            u.step(a, tstep, g);
            __syncthreads();                                   // everybody is done reading the LDS copies before they are refreshed
            LIN_STAMP(7);
            { [[maybe_unused]] unsigned long long te = 0; LIN_NOW(te); LIN_PUT(64 + n, te); }
        }
        u.store_state(a, g, tstep);
    } else if (b < a.has_update + a.n_reduce) {
        // ---- reducers: set (rb / per_set) takes batches set, set + 2, ... ------------------------------------------------------
        const int rb = b - a.has_update, set = rb / per_set, ro = rb % per_set;
        const int tstep0 = a.step_dev[0];              // (the updater stores the counter at the very end of the launch)
        [[maybe_unused]] unsigned long long r0 = 0, r1 = 0, r2 = 0, racc_w = 0, racc_r = 0;
        for (int n = set; n < N; n += a.sets) {
            LIN_NOWQ(r0);
            lin_wait_count(a.cnt_stream + n, (unsigned)a.ntiles, a.status, (2u << 28) | ((unsigned)n << 16));
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
        // ---- streamers: workgroup sid takes tiles sid, sid + S, ... of every batch, in batch order, through a ring of two LDS
        // slots: the loads of work item i + 1 are in flight while item i is multiplied.  The batch pointers live in LDS (one
        // fetch per launch: a scalar load per batch would sit on the critical path with the memory system busy).
        const int sid = b - a.has_update - a.n_reduce, S = a.n_stream;
        { [[maybe_unused]] unsigned long long te = 0; LIN_NOWQ(te); if (sid == 0) LIN_PUT(50, te); if (sid == S / 2) LIN_PUT(51, te); if (sid == S - 1) LIN_PUT(52, te); }
        const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
        const LinSlot sl(a.D, a.L, a.T);
        const int stride = max(sl.bytes, lin_scratch_bytes(NB));        // a slot doubles as the multiply's cross-wave scratch
        const float** tab = reinterpret_cast<const float**>(lin_smem + 2 * stride);            // [3][kLinMaxPersist]
        if (t < 3 * kLinMaxPersist) {
            const int which = t / kLinMaxPersist, n = t % kLinMaxPersist;
            tab[t] = n < N ? (which == 0 ? a.xs : which == 1 ? a.z1s : a.z2s)[n] : nullptr;
        }
        __syncthreads();
        const int per_batch = sid < a.ntiles ? (a.ntiles - sid + S - 1) / S : 0, items = N * per_batch;
        auto item_batch = [&](int i) { return i / per_batch; };
        auto item_tile = [&](int i) { return sid + (i % per_batch) * S; };
        auto issue = [&](int i) {
            const int n = item_batch(i);
            lin_issue_tile(a, sl, tab[n], tab[kLinMaxPersist + n], tab[2 * kLinMaxPersist + n], item_tile(i), lin_smem + (i & 1) * stride, t, wave);
        };
        if (items > 0) issue(0);
        [[maybe_unused]] unsigned long long s0 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0, sacc_i = 0, sacc_l = 0, sacc_f = 0, sacc_m = 0;
        for (int i = 0; i < items; ++i) {
            // (the slot item i + 1 lands in was released by the barrier that closed iteration i - 1)
            LIN_NOWQ(s0);
            if (i + 1 < items) { issue(i + 1); LIN_NOWQ(s1); lin_wait_vmcnt_upto(sl.passes); } else lin_wait_vmcnt<0>();
            // tile i has landed, and -- the counter retires in order -- so have this wave's write-through stores of item i - 1
            lin_barrier();
            LIN_NOWQ(s2);
            if (i >= 1 && t == 0) __hip_atomic_fetch_add(a.cnt_stream + item_batch(i - 1), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (i == 1) { [[maybe_unused]] unsigned long long te = 0; LIN_NOWQ(te); if (sid == 0) LIN_PUT(53, te); if (sid == S / 2) LIN_PUT(54, te); if (sid == S - 1) LIN_PUT(55, te); }
            const int n = item_batch(i), tile = item_tile(i);
            char* slot = lin_smem + (i & 1) * stride;
            lin_fix_tile(a, sl, tab[n], tab[kLinMaxPersist + n], tab[2 * kLinMaxPersist + n], tile, slot, t);
            LIN_NOWQ(s3);
            lin_multiply_tile<NB, true>(a, sl, slot, a.partial_base + ((long long)n * a.ntiles + tile) * NO, t, wave);
            lin_barrier();                                     // every wave's products are done: the slot is free for item i + 2
            LIN_NOWQ(s4);
            if (i + 1 < items) { sacc_i += s1 - s0; sacc_l += s2 - s1; sacc_f += s3 - s2; sacc_m += s4 - s3; }
            if (sid == 7) { LIN_PUT(42, sacc_i); LIN_PUT(43, sacc_l); LIN_PUT(44, sacc_f); LIN_PUT(45, sacc_m); LIN_PUT(46, (unsigned long long)items); }
        }
        lin_wait_vmcnt<0>();
        __syncthreads();
        if (items > 0 && t == 0) __hip_atomic_fetch_add(a.cnt_stream + item_batch(items - 1), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// device tables of batch pointers (the persistent kernel reads them; 1.5 KB of kernarg).  The
// table launch of a persistent launch also zeroes its arrival counters and status word (a kernel of this stream rather
// than a memset node: it stays an ordinary kernel node when the call is captured into a hipGraph).
struct LinTable { const float* x[kLinMaxPersist]; const float* z1[kLinMaxPersist]; const float* z2[kLinMaxPersist]; int n; unsigned* zero; };
__global__ void lin_table_kernel(const LinTable tb, const float** xs, const float** z1s, const float** z2s) {
    const int i = threadIdx.x;
    if (i < tb.n) { xs[i] = tb.x[i]; z1s[i] = tb.z1[i]; z2s[i] = tb.z2[i]; }
    if (tb.zero)
        for (int k = i; k < 1024; k += 64) tb.zero[k] = 0u;
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
static size_t lin_ring_bytes(const vaek_ctx* c, int T) {        // the streamers' two tile slots (each doubling as the multiply's scratch) + pointer tables
    const LinSlot sl(c->D, c->L, T);
    return 2 * std::max((size_t)sl.bytes, (size_t)lin_scratch_bytes(lin_nb(c))) + 3 * kLinMaxPersist * sizeof(void*) + 64;
}
// Samples per tile.  256, unless a slightly taller tile lets every streamer of the persistent launch take exactly ONE tile per
// batch (the metric: 65 536 samples = 228 tiles of 288 on the 231 CUs the updater and the reducers leave).
static int lin_tile_rows(const vaek_ctx* c) {
    const int smax = c->n_cu - 1 - kLinReduceSets * kLinReduceWgs;
    if (smax < 16 || c->B <= 256 * smax) return 256;
    const int T = 32 * (int)(((long long)c->B + 32ll * smax - 1) / (32ll * smax));
    const LinSlot sl(c->D, c->L, T);
    return T <= 512 && sl.passes <= 12 && lin_ring_bytes(c, T) <= 160 * 1024 ? T : 256;
}
static int lin_ntiles(const vaek_ctx* c) { const int T = lin_tile_rows(c); return (c->B + T - 1) / T; }
static size_t lin_slot_stride(const vaek_ctx* c) {      // one tile slot, large enough to double as the multiply's cross-wave scratch
    const LinSlot sl(c->D, c->L, lin_tile_rows(c));
    return std::max((size_t)sl.bytes, (size_t)lin_scratch_bytes(lin_nb(c)));
}
static size_t lin_lds_need(const vaek_ctx* c) {
    const int NB = lin_nb(c), D = c->D, L = c->L;
    const size_t upd = NB == 3 ? LinUpd<3, 0, 0>::lds_bytes(D, L) : LinUpd<4, 0, 0>::lds_bytes(D, L);
    return std::max(lin_slot_stride(c), std::max(upd, (size_t)16 * 32 * sizeof(double))) + 64;
}
// The persistent form gives every workgroup a CU of its own (the updater's float64 chains and the streamers' MFMA loops both
// lose a factor ~2 when they share one): each workgroup asks for more than half a CU's LDS -- the streamers need it anyway for
// their ring of two tile slots + the batch pointer tables -- and the grid stays within the CU count.
static size_t lin_persist_lds(const vaek_ctx* c) {
    return std::max(std::max(lin_lds_need(c), lin_ring_bytes(c, lin_tile_rows(c))), (size_t)82 * 1024);
}
static bool lin_persist_supported(const vaek_ctx* c) {
    const LinSlot sl(c->D, c->L, lin_tile_rows(c));
    return lin_steps_shape_ok(c) && lin_nb(c) == 3 && lin_persist_lds(c) <= 160 * 1024 && sl.passes <= 12 &&
           c->n_cu >= 1 + kLinReduceSets * kLinReduceWgs + 16;
}
// streamer workgroups of the persistent launch: the CUs the updater and the reducers leave, tiles dealt evenly
static int lin_persist_streamers(const vaek_ctx* c) {
    const int ntiles = lin_ntiles(c), smax = c->n_cu - 1 - kLinReduceSets * kLinReduceWgs;
    const int per = (ntiles + smax - 1) / smax;
    return (ntiles + per - 1) / per;
}

struct LinWs { float* partial; double* M; unsigned* cnt; const float** tab; size_t total; };
static LinWs lin_carve(const vaek_ctx* c, char* base) {
    const size_t no = lin_no(c), ntiles = lin_ntiles(c);
    const int slots = lin_persist_supported(c) ? kLinMaxPersist : 2;
    LinWs w{};
    size_t off = 0;
    w.cnt = reinterpret_cast<unsigned*>(base + off); off += 4096;                 // cnt_stream[64] | cnt_reduce[64] | status, zeroed by lin_table_kernel
    w.tab = reinterpret_cast<const float**>(base + off); off += 3 * kLinMaxPersist * sizeof(void*) + 256;
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

typedef void (*LinKernel)(const LinArgs);
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
    return 0;
}

int lin_train_steps(vaek_ctx* c, float* params, float* grads, float* m, float* v, int32_t* step_dev, const float* const* xs,
                    const float* const* z1s, const float* const* z2s, int n_steps, float lr, void* ws, hipStream_t st) {
    const int NB = lin_nb(c), no = lin_no(c), ntiles = lin_ntiles(c);
    const LinWs w = lin_carve(c, static_cast<char*>(ws) + c->ws_lin);
    // the metric's shape with its dimensions at compile time; every other linear model on the run-time instantiations
    const int which = (c->D == 12 && c->L == 20) ? 0 : (NB <= 3 ? 1 : 2);
    static const char* env = getenv("VAEK_LIN_PERSIST");              // diagnostic: 0 forces the launch-per-step form
    const bool persistent = lin_persist_supported(c) && (c->cfg.world > 1 || !(env && atoi(env) == 0));   // data parallel: persistent form only
    const size_t lds = persistent ? lin_persist_lds(c) : lin_lds_need(c);
    const LinKernel fn = persistent ? (which == 0 ? lin_persist_kernel<3, 12, 20> : lin_persist_kernel<3, 0, 0>)
                                    : (which == 0 ? lin_step_kernel<3, 12, 20> : which == 1 ? lin_step_kernel<3, 0, 0> : lin_step_kernel<4, 0, 0>);
    static thread_local bool attr_set[2][3] = {};
    if (!attr_set[persistent ? 1 : 0][which]) {
        VAEK_HIP_CHECK(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));   // a cap, not a request
        attr_set[persistent ? 1 : 0][which] = true;
    }
    if (persistent) {
        for (int s0 = 0; s0 < n_steps; s0 += kLinMaxPersist) {
            const int n = std::min(kLinMaxPersist, n_steps - s0);
            const float** txs = w.tab; const float** tz1 = w.tab + kLinMaxPersist; const float** tz2 = w.tab + 2 * kLinMaxPersist;
            {
                LinTable tb{};
                tb.n = n; tb.zero = w.cnt;
                for (int i = 0; i < n; ++i) { tb.x[i] = xs[s0 + i]; tb.z1[i] = z1s[s0 + i]; tb.z2[i] = z2s[s0 + i]; }
                hipLaunchKernelGGL(lin_table_kernel, dim3(1), dim3(64), 0, st, tb, txs, tz1, tz2);
            }
            LinArgs a{};
            lin_fill_common(c, a, params, grads, m, v, step_dev, lr);
            a.persistent = 1; a.n_steps = n;
            a.sets = kLinReduceSets;
            a.has_update = 1; a.n_reduce = kLinReduceSets * std::min(kLinReduceWgs, no / 32); a.n_stream = lin_persist_streamers(c);
            static const int proles = getenv("VAEK_LIN_ROLES") ? atoi(getenv("VAEK_LIN_ROLES")) : 7;    // diagnostic (tools/lin_roles.sh)
            if (!(proles & 4)) a.has_update = 0;
            if (!(proles & 2)) { a.has_update = 0; a.n_reduce = 0; }
            a.xs = txs; a.z1s = tz1; a.z2s = tz2;
            a.partial_base = w.partial; a.M_base = w.M;
            a.cnt_stream = w.cnt; a.cnt_reduce = w.cnt + 256; a.status = w.cnt + 512;
            ProfScope ps("lin_moments_persistent", st);
            launch_k(ps, fn, dim3((unsigned)(a.has_update + a.n_reduce + a.n_stream)), dim3(LNT), lds, st, a);
        }
        VAEK_HIP_CHECK(hipGetLastError());
        return VAEK_OK;
    }
    {   // no in-launch waits in this form: the status word reads "never gave up"
        LinTable tb{};
        tb.zero = w.cnt;
        hipLaunchKernelGGL(lin_table_kernel, dim3(1), dim3(64), 0, st, tb, w.tab, w.tab + kLinMaxPersist, w.tab + 2 * kLinMaxPersist);
    }
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

// synchronous: did a bounded wait of the persistent form ever give up (a workgroup that never became resident, a lost store)?
int lin_steps_status(vaek_ctx* c, void* ws, int* gave_up) {
    *gave_up = 0;
    if (!lin_steps_shape_ok(c)) return VAEK_OK;
    const LinWs w = lin_carve(c, static_cast<char*>(ws) + c->ws_lin);
    unsigned s = 0;
    VAEK_HIP_CHECK(hipMemcpy(&s, w.cnt + 512, sizeof(s), hipMemcpyDeviceToHost));
    *gave_up = (int)s;
    return VAEK_OK;
}

}  // namespace vaek

#ifdef VAEK_LIN_STAMPS
extern "C" int vaek_debug_lin_stamps(unsigned long long* buf) {
    return hipMemcpyToSymbol(HIP_SYMBOL(vaek::g_lin_stamp_buf), &buf, sizeof(buf)) == hipSuccess ? 0 : -2;
}
#endif
