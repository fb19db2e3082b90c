// Probe for the float64 matrix-core updater of vaek_train_steps (csrc/linear_moments.hip): checks the operand / result lane
// maps of v_mfma_f64_16x16x4_f64 with exact integer data (asymmetric B), the "accumulator tile as the next product's B operand"
// chaining in natural k order, and times issue interval / dependent latency with s_memtime.
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_f64_probe.hip -o gpurun_out/mfma_f64_probe && gpurun_out/mfma_f64_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

using d4 = __attribute__((ext_vector_type(4))) double;

__global__ void layout_kernel(const double* A, const double* B, const double* W, double* C, double* C2) {
    // C = A (16x8) B (8x16): two k-steps; then C2 = W (16x16) C: four k-steps with C's registers as the B operand
    const int lane = threadIdx.x, i = lane & 15, g = lane >> 4;
    d4 acc = {0, 0, 0, 0};
    for (int kk = 0; kk < 2; ++kk)
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A[i * 8 + 4 * kk + g], B[(4 * kk + g) * 16 + i], acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) C[(g + 4 * r) * 16 + i] = acc[r];          // documented map: row = (lane >> 4) + 4 reg, col = lane & 15
    d4 acc2 = {0, 0, 0, 0};
    for (int r = 0; r < 4; ++r)                                             // k-step r: k = 4 r + g on both operands
        acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(W[i * 16 + 4 * r + g], acc[r], acc2, 0, 0, 0);
    for (int r = 0; r < 4; ++r) C2[(g + 4 * r) * 16 + i] = acc2[r];
}

template <int NACC>
__global__ void timing_kernel(unsigned long long* out, double seed) {
    d4 acc[NACC];
    for (int k = 0; k < NACC; ++k) acc[k] = d4{seed, seed, seed, seed};
    const double a = seed * 1e-3 + threadIdx.x * 1e-6, b = seed * 1e-3;
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
    for (int it = 0; it < 64; ++it) {
#pragma unroll
        for (int k = 0; k < NACC; ++k) acc[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[k], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    double s = 0;
    for (int k = 0; k < NACC; ++k) s += acc[k][0] + acc[k][3];
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = (unsigned long long)(s != 12345.0); }
}

int main() {
    std::vector<double> A(16 * 8), B(8 * 16), W(16 * 16), C(256), C2(256);
    for (int i = 0; i < 16; ++i) for (int k = 0; k < 8; ++k) A[i * 8 + k] = (i + 1) * 3 + k * 7 % 5;
    for (int k = 0; k < 8; ++k) for (int j = 0; j < 16; ++j) B[k * 16 + j] = (k * 5 + 1) * (j % 3 + 1) + j * j;       // asymmetric
    for (int i = 0; i < 16; ++i) for (int k = 0; k < 16; ++k) W[i * 16 + k] = (i * 2 - k) % 7 + (i == k ? 3 : 0);
    double *dA, *dB, *dW, *dC, *dC2;
    hipMalloc(&dA, A.size() * 8); hipMalloc(&dB, B.size() * 8); hipMalloc(&dW, W.size() * 8); hipMalloc(&dC, 2048); hipMalloc(&dC2, 2048);
    hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(dW, W.data(), W.size() * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(layout_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dW, dC, dC2);
    hipMemcpy(C.data(), dC, 2048, hipMemcpyDeviceToHost); hipMemcpy(C2.data(), dC2, 2048, hipMemcpyDeviceToHost);
    int bad = 0, bad2 = 0;
    std::vector<double> R(256, 0.0);
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
        double s = 0; for (int k = 0; k < 8; ++k) s += A[i * 8 + k] * B[k * 16 + j];
        R[i * 16 + j] = s; bad += (s != C[i * 16 + j]);
    }
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
        double s = 0; for (int k = 0; k < 16; ++k) s += W[i * 16 + k] * R[k * 16 + j];
        bad2 += (s != C2[i * 16 + j]);
    }
    printf("f64 16x16x4 lane maps: %d wrong of 256; accumulator-as-B chaining: %d wrong of 256\n", bad, bad2);
    unsigned long long* dt; hipMalloc(&dt, 16); unsigned long long t[2];
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(timing_kernel<1>, dim3(1), dim3(64), 0, 0, dt, 1.0); hipMemcpy(t, dt, 16, hipMemcpyDeviceToHost);
        if (rep) printf("1 accumulator (dependent chain): %.1f ticks per MFMA\n", t[0] / 64.0);
        hipLaunchKernelGGL(timing_kernel<4>, dim3(1), dim3(64), 0, 0, dt, 1.0); hipMemcpy(t, dt, 16, hipMemcpyDeviceToHost);
        if (rep) printf("4 accumulators, one wave: %.1f ticks per MFMA\n", t[0] / 256.0);
        hipLaunchKernelGGL(timing_kernel<4>, dim3(1), dim3(512), 0, 0, dt, 1.0); hipMemcpy(t, dt, 16, hipMemcpyDeviceToHost);
        if (rep) printf("4 accumulators, 8 waves on the CU (2 per SIMD): %.1f ticks per MFMA per wave\n", t[0] / 256.0);
    }
    return bad + bad2 ? 1 : 0;
}
