// Times one Philox4x32-10 block per lane in three formulations of its 32 x 32 -> 64 multiplies (s_memtime ticks per block, one wave
// and eight waves on a CU): (a) as the compiler lowers (unsigned long long)a * b  (v_mad_u64_u32), (b) __umulhi + low product,
// (c) 16-bit limbs on the full-rate 24-bit multipliers.   hipcc -O3 --offload-arch=gfx950 tools/philox_probe.hip -o tools/philox_probe.bin
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int MODE>
__device__ __forceinline__ void mulhilo(unsigned a, unsigned m, unsigned& hi, unsigned& lo) {
    if (MODE == 0) { const unsigned long long p = (unsigned long long)a * m; hi = (unsigned)(p >> 32); lo = (unsigned)p; }
    else if (MODE == 1) { hi = __umulhi(a, m); lo = a * m; }
    else {
        const unsigned a0 = a & 0xffffu, a1 = a >> 16, m0 = m & 0xffffu, m1 = m >> 16;
        const unsigned p00 = __umul24(a0, m0), p01 = __umul24(a0, m1), p10 = __umul24(a1, m0), p11 = __umul24(a1, m1);
        const unsigned mid = p01 + p10, midc = mid < p01 ? 0x10000u : 0u;            // carry out of the 32-bit middle sum
        lo = p00 + (mid << 16);
        const unsigned c0 = lo < p00 ? 1u : 0u;
        hi = p11 + (mid >> 16) + midc + c0;
    }
}
template <int MODE>
__device__ __forceinline__ uint4 philox(uint4 c, uint2 k) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        unsigned h0, l0, h1, l1;
        mulhilo<MODE>(c.x, 0xD2511F53u, h0, l0);
        mulhilo<MODE>(c.z, 0xCD9E8D57u, h1, l1);
        c = make_uint4(h1 ^ c.y ^ k.x, l1, h0 ^ c.w ^ k.y, l0);
        k.x += 0x9E3779B9u; k.y += 0xBB67AE85u;
    }
    return c;
}
template <int MODE>
__global__ void k(unsigned long long* out, unsigned* sink, unsigned seed) {
    uint4 c = make_uint4(threadIdx.x, seed, 3, 4);
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
    for (int it = 0; it < 64; ++it) c = philox<MODE>(c, make_uint2(seed, it));
    __builtin_amdgcn_sched_barrier(0);
    sink[threadIdx.x] = c.x ^ c.y ^ c.z ^ c.w;
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[0] = t1 - t0;
}
int main() {
    unsigned long long* dt; unsigned* sink; hipMalloc(&dt, 16); hipMalloc(&sink, 4096);
    unsigned long long t; unsigned s[3][4];
    for (int threads : {64, 512}) {
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(k<0>, dim3(1), dim3(threads), 0, 0, dt, sink, 7u); hipMemcpy(&t, dt, 8, hipMemcpyDeviceToHost); hipMemcpy(s[0], sink, 16, hipMemcpyDeviceToHost);
            if (rep) printf("%d threads: v_mad_u64_u32 form  %.0f ticks per Philox block\n", threads, t / 64.0);
            hipLaunchKernelGGL(k<1>, dim3(1), dim3(threads), 0, 0, dt, sink, 7u); hipMemcpy(&t, dt, 8, hipMemcpyDeviceToHost); hipMemcpy(s[1], sink, 16, hipMemcpyDeviceToHost);
            if (rep) printf("%d threads: mul_hi + mul_lo     %.0f\n", threads, t / 64.0);
            hipLaunchKernelGGL(k<2>, dim3(1), dim3(threads), 0, 0, dt, sink, 7u); hipMemcpy(&t, dt, 8, hipMemcpyDeviceToHost); hipMemcpy(s[2], sink, 16, hipMemcpyDeviceToHost);
            if (rep) printf("%d threads: 16-bit limbs (u24)  %.0f   same bits: %d\n", threads, t / 64.0, s[0][1] == s[1][1] && s[1][1] == s[2][1] && s[0][3] == s[2][3]);
        }
    }
    return 0;
}
