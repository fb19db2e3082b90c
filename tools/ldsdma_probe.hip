// How fast can the loading waves of a streamer workgroup pull HBM into LDS?  G workgroups (one per CU) x W waves issue 1 KB LDS-DMA
// pieces (global_load_lds, 16 bytes per lane) of a chunk per "step" -- workgroup g's chunk of step s lies at (s * G + g) * CHUNK, so a
// step is one contiguous G * CHUNK bytes, as a batch of the metric is -- keeping at most Q pieces per wave in flight (counted
// s_waitcnt), and either never draining (DRAIN 0) or draining + a workgroup barrier at the end of every chunk (DRAIN 1: what a
// ring of two tile slots forces).  Prints TB/s.     hipcc -O3 --offload-arch=gfx950 tools/ldsdma_probe.hip -o /tmp/ldsdma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) void glb_void_t;
extern __shared__ __attribute__((aligned(1024))) char smem[];

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

typedef float f32x4 __attribute__((ext_vector_type(4)));
// waves 0 .. W-1 load, waves W .. W+C-1 multiply (108 MFMAs of 16x16x4 f32 per step on 6 accumulators: a streamer's share of a tile)
template <int W, int C, int Q, int DRAIN, int PRIO>
__global__ __launch_bounds__((W + C) * 64) void stream(const char* src, long long total, int chunk_pieces, int steps, unsigned* sink) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int G = gridDim.x, g = blockIdx.x;
    if (wave >= W) {
        if (PRIO == 2) __builtin_amdgcn_s_setprio(0);
        f32x4 acc[6] = {};
        float a = lane * 1e-3f, b = 1.f - lane * 1e-3f;
        for (int s = 0; s < steps; ++s) {
#pragma unroll 1
            for (int k = 0; k < 18; ++k) {
#pragma unroll
                for (int j = 0; j < 6; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j], 0, 0, 0);
                a += 1e-6f;
            }
            if (DRAIN) asm volatile("s_barrier" ::: "memory");
        }
        f32x4 t = acc[0] + acc[1] + acc[2] + acc[3] + acc[4] + acc[5];
        if (t[0] + t[1] + t[2] + t[3] == 12345.f) sink[1024 + g] = 1;
        __syncthreads();
        return;
    }
    if (PRIO) __builtin_amdgcn_s_setprio(3);
    const int ppw = (chunk_pieces + W - 1) / W;                 // pieces per wave and chunk
    char* ring = smem + wave * (Q + 1) * 1024;                  // Q + 1 piece slots per wave
    int slot = 0;
    for (int s = 0; s < steps; ++s) {
        const char* base = src + (((long long)s * G + g) * chunk_pieces * 1024ll) % total;
        for (int k = 0; k < ppw; ++k) {
            const int piece = wave + W * k;
            if (piece < chunk_pieces) {
                wait_vm<Q>();                                   // (counts every earlier piece but Q: the slot about to be overwritten is free)
                __builtin_amdgcn_global_load_lds((glb_void_t*)(base + piece * 1024 + lane * 16), (lds_void_t*)(ring + slot * 1024), 16, 0, 0);
                slot = slot == Q ? 0 : slot + 1;
            }
        }
        if (DRAIN) { wait_vm<0>(); asm volatile("s_barrier" ::: "memory"); }
    }
    wait_vm<0>();
    __syncthreads();
    if (threadIdx.x == 0) sink[g] = *reinterpret_cast<unsigned*>(smem + 64);
}

// The streamers' ring as built: a step's chunk moves as two sub-chunks through four LDS slots, the loading waves issue sub-chunk j + 3,
// wait (counted) until sub-chunk j + 1 has landed, and meet the multiplying waves (54 MFMAs per sub-chunk, READS operand reads from the
// slot first) at ONE barrier per sub-chunk.  STREAMS 3: a sub-chunk comes from three tensors (11.25 + 6.75 + 6.75 KB, as z1 | x | z2).
template <int READS, int STREAMS>
__global__ __launch_bounds__(512) void ring(const char* src, long long total, int steps, unsigned* sink) {
    constexpr int W = 4, SUB = 25 * 1024;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int G = gridDim.x, g = blockIdx.x, J = 2 * steps;
    auto issue = [&](int j) {
        if (j >= J) return;
        char* slot = smem + (j & 3) * 26 * 1024;
        const int s = j >> 1, h = j & 1;
        if (STREAMS == 1) {
            const char* base = src + (((long long)s * G + g) * 2 + h) * SUB % total;
            for (int p = wave - W; p < 25; p += W)
                __builtin_amdgcn_global_load_lds((glb_void_t*)(base + p * 1024 + lane * 16), (lds_void_t*)(slot + p * 1024), 16, 0, 0);
        } else {
            // three tensors of 80 / 48 / 48 bytes per sample, 144 samples per sub-chunk; batch s lives at s * 3 tensors
            const long long tb = ((long long)s * 3) * (65536ll * 80) % (total - 3 * 65536ll * 80);
            const long long row0 = ((long long)g * 2 + h) * 144;
            const char* z1 = src + tb + row0 * 80; const char* x = src + tb + 65536ll * 80 + row0 * 48; const char* z2 = src + tb + 65536ll * 128 + row0 * 48;
            for (int p = wave - W; p < 12; p += W) __builtin_amdgcn_global_load_lds((glb_void_t*)(z1 + min(p * 1024 + lane * 16, 11520 - 16)), (lds_void_t*)(slot + p * 1024), 16, 0, 0);
            for (int p = wave - W; p < 7; p += W) __builtin_amdgcn_global_load_lds((glb_void_t*)(x + min(p * 1024 + lane * 16, 6912 - 16)), (lds_void_t*)(slot + (12 + p) * 1024), 16, 0, 0);
            for (int p = wave - W; p < 7; p += W) __builtin_amdgcn_global_load_lds((glb_void_t*)(z2 + min(p * 1024 + lane * 16, 6912 - 16)), (lds_void_t*)(slot + (19 + p) * 1024), 16, 0, 0);
        }
    };
    if (wave < W) {
        f32x4 acc[6] = {};
        float a = lane * 1e-3f, b = 1.f - lane * 1e-3f;
        asm volatile("s_barrier" ::: "memory");
        for (int j = 0; j < J; ++j) {
            const float* slot = reinterpret_cast<const float*>(smem + (j & 3) * 26 * 1024);
#pragma unroll 1
            for (int k = 0; k < 9; ++k) {
                if (READS) { a = slot[(wave * 36 + k * 4 + (lane >> 4)) * 20 + (lane & 15)]; b = slot[3072 + (wave * 36 + k * 4 + (lane >> 4)) * 12 + (lane & 7)]; }
#pragma unroll
                for (int q = 0; q < 6; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[q], 0, 0, 0);
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
        f32x4 t = acc[0] + acc[1] + acc[2] + acc[3] + acc[4] + acc[5];
        if (t[0] + t[1] + t[2] + t[3] == 12345.f) sink[1024 + g] = 1;
        return;
    }
    const int lw = wave - W;
    const int ppw = STREAMS == 1 ? (lw < 1 ? 7 : 6) : ((lw < 12 ? (12 - lw + 3) / 4 : 0) + 2 * (lw < 7 ? (7 - lw + 3) / 4 : 0));
    issue(0); issue(1); issue(2);
    if (ppw == 7) wait_vm<14>(); else if (ppw == 6) wait_vm<12>(); else wait_vm<10>();
    asm volatile("s_barrier" ::: "memory");
    for (int j = 0; j < J; ++j) {
        issue(j + 3);
        if (j + 3 < J) { if (ppw == 7) wait_vm<14>(); else if (ppw == 6) wait_vm<12>(); else wait_vm<10>(); }
        else wait_vm<0>();
        asm volatile("s_barrier" ::: "memory");
    }
}
template <int READS, int STREAMS>
static void run_ring(const char* buf, long long total, int G, int steps, unsigned* sink) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const size_t lds = 4 * 26 * 1024 + 48 * 1024;
    hipFuncSetAttribute((const void*)ring<READS, STREAMS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((ring<READS, STREAMS>), dim3(G), dim3(512), lds, 0, buf, total, steps, sink);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
    }
    printf("ring of 4 sub-chunk slots, 3 ahead, one barrier per sub-chunk, operand reads %d, tensors %d: %.2f TB/s  (%.2f us per step)\n", READS, STREAMS, (double)G * 50 * 1024.0 * steps / best * 1e-9, best * 1e3 / steps);
}

template <int W, int C, int Q, int DRAIN, int PRIO>
static void run(const char* buf, long long total, int G, int chunk_pieces, int steps, unsigned* sink) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const size_t lds = (size_t)(W ? W : 1) * (Q + 1) * 1024 + 82 * 1024;     // (more than half a CU's LDS: one workgroup per CU)
    hipFuncSetAttribute((const void*)stream<W, C, Q, DRAIN, PRIO>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((stream<W, C, Q, DRAIN, PRIO>), dim3(G), dim3((W + C) * 64), lds, 0, buf, total, chunk_pieces, steps, sink);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
    }
    const double bytes = (double)G * chunk_pieces * 1024.0 * steps;
    printf("G %3d  loading waves %d  multiplying waves %d  in flight per wave %2d  drain + barrier per chunk %d  prio %d: %.2f TB/s  (%.2f us per step)\n", G, W, C, Q, DRAIN, PRIO, W ? bytes / best * 1e-9 : 0.0, best * 1e3 / steps);
}

// the same stream through the ordinary path: global_load_dwordx4 into registers (UNR loads in flight per lane), nothing stored
template <int W, int UNR>
__global__ __launch_bounds__(W * 64) void stream_regs(const char* src, long long total, int chunk_pieces, int steps, unsigned* sink) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int G = gridDim.x, g = blockIdx.x;
    unsigned acc = 0;
    for (int s = 0; s < steps; ++s) {
        const char* base = src + (((long long)s * G + g) * chunk_pieces * 1024ll) % total;
        for (int k = wave; k < chunk_pieces; k += W * UNR) {
            uint4 v[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) v[u] = *reinterpret_cast<const uint4*>(base + min(k + W * u, chunk_pieces - 1) * 1024 + lane * 16);
#pragma unroll
            for (int u = 0; u < UNR; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
        }
    }
    if (acc == 0x12345678u) sink[g] = acc;
}
template <int W, int UNR>
static void run_regs(const char* buf, long long total, int G, int chunk_pieces, int steps, unsigned* sink, const char* what) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((stream_regs<W, UNR>), dim3(G), dim3(W * 64), 0, 0, buf, total, chunk_pieces, steps, sink);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
    }
    printf("%s, registers: %d waves x %d loads in flight: %.2f TB/s\n", what, W, UNR, (double)G * chunk_pieces * 1024.0 * steps / best * 1e-9);
}

int main(int argc, char** argv) {
    const long long total = 4ll << 30;
    char* buf; unsigned* sink;
    if (hipMalloc(&buf, total) != hipSuccess) { printf("no memory\n"); return 1; }
    hipMemset(buf, 1, total); hipMalloc(&sink, 16384);
    const int steps = 256, G = 228;
    run<4, 0, 13, 1, 0>(buf, total, G, 50, steps, sink);
    run<4, 0, 13, 0, 0>(buf, total, G, 50, steps, sink);
    run<0, 4, 13, 0, 0>(buf, total, G, 50, steps, sink);
    run<0, 4, 13, 1, 0>(buf, total, G, 50, steps, sink);
    run<4, 4, 13, 1, 0>(buf, total, G, 50, steps, sink);
    run<4, 4, 13, 0, 0>(buf, total, G, 50, steps, sink);
    run<4, 4, 26, 0, 0>(buf, total, G, 50, steps, sink);
    run<4, 4, 13, 1, 1>(buf, total, G, 50, steps, sink);
    run<4, 4, 13, 0, 1>(buf, total, G, 50, steps, sink);
    run<4, 4, 13, 1, 2>(buf, total, G, 50, steps, sink);
    run<2, 4, 26, 1, 0>(buf, total, G, 50, steps, sink);
    run<2, 4, 26, 0, 0>(buf, total, G, 50, steps, sink);
    run<1, 4, 50, 0, 0>(buf, total, G, 50, steps, sink);
    // the same chunk again and again: every step after the first is served by the XCD's L2
    const long long resident = (long long)G * 50 * 1024;
    printf("L2-resident source (%lld MB re-read every step):\n", resident >> 20);
    run<4, 0, 13, 0, 0>(buf, resident, G, 50, steps, sink);
    run<8, 0, 13, 0, 0>(buf, resident, G, 50, steps, sink);
    run<4, 4, 13, 0, 0>(buf, resident, G, 50, steps, sink);
    run_regs<4, 4>(buf, resident, G, 50, steps, sink, "L2-resident");
    run_regs<8, 4>(buf, resident, G, 50, steps, sink, "L2-resident");
    run_regs<8, 7>(buf, resident, G, 50, steps, sink, "L2-resident");
    run_regs<4, 4>(buf, total, G, 50, steps, sink, "HBM");
    run_regs<8, 4>(buf, total, G, 50, steps, sink, "HBM");
    run_regs<8, 7>(buf, total, G, 50, steps, sink, "HBM");
    run_ring<0, 1>(buf, total, G, steps, sink);
    run_ring<1, 1>(buf, total, G, steps, sink);
    run_ring<0, 3>(buf, total, G, steps, sink);
    run_ring<1, 3>(buf, total, G, steps, sink);
    return 0;
}
