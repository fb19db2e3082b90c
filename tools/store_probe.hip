// How fast do plain 16-byte stores fill 64 MB?  One row of 1 KB per wave and store instruction (the epilogues' pattern), G workgroups of
// 256 threads, rows dealt round-robin (MODE 0) or in one contiguous span per workgroup (MODE 1); NT: __builtin_nontemporal_store.
//   hipcc -O3 --offload-arch=gfx950 tools/store_probe.hip -o tools/store_probe.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int MODE, int NT>
__global__ __launch_bounds__(256) void fill(u32x4* dst, long long rows) {       // a row = 64 lanes x 16 bytes
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long nw = (long long)gridDim.x * 4, w = (long long)blockIdx.x * 4 + wave;
    const u32x4 v = {1u, 2u, 3u, (unsigned)w};
    if (MODE == 0) {
        for (long long r = w; r < rows; r += nw) { if (NT) __builtin_nontemporal_store(v, dst + r * 64 + lane); else dst[r * 64 + lane] = v; }
    } else {
        const long long per = (rows + nw - 1) / nw, r0 = w * per, r1 = r0 + per < rows ? r0 + per : rows;
        for (long long r = r0; r < r1; ++r) { if (NT) __builtin_nontemporal_store(v, dst + r * 64 + lane); else dst[r * 64 + lane] = v; }
    }
}
template <int MODE, int NT>
static void run(u32x4* buf, long long bytes, int G) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 6; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((fill<MODE, NT>), dim3(G), dim3(256), 0, 0, buf, bytes / 1024);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
    }
    printf("%4lld MB  G %5d  %s  %s: %.1f us  %.2f TB/s\n", bytes >> 20, G, MODE ? "contiguous span per wave" : "rows round-robin       ", NT ? "nontemporal" : "plain      ", best * 1e3, bytes / best * 1e-9);
}
int main() {
    u32x4* buf; hipMalloc(&buf, 1ll << 30);
    for (long long bytes : {64ll << 20, 512ll << 20})
        for (int G : {256, 512, 2048, 8192}) {
            run<0, 0>(buf, bytes, G); run<1, 0>(buf, bytes, G); run<0, 1>(buf, bytes, G); run<1, 1>(buf, bytes, G);
        }
    return 0;
}
