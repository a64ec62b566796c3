// Micro-benchmark: what does s_memtime count, and what clock does the chip hold under a sustained fp32-MFMA load?
// Every wave runs a chain of v_mfma_f32_32x32x2_f32 on NACC accumulators with NONZERO operands and records its s_memtime ticks
// and its wall_clock64 (100 MHz, constant) ticks around the chain.  ticks per MFMA and SIMD = 64 x (waves per SIMD) if s_memtime
// counts shader cycles; ticks / wall = the shader clock the kernel ran at.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC> __global__ __launch_bounds__(512) void chain(unsigned long long* rec, int iters, float s) {
    f32x16 acc[NACC];
    for (int j = 0; j < NACC; ++j) for (int q = 0; q < 16; ++q) acc[j][q] = 0.f;
    float a = 0.37f + s * (threadIdx.x % 61), b = -0.73f + s * (threadIdx.x % 53);
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), w0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
        a = -a;
    }
    float r = 0; for (int j = 0; j < NACC; ++j) for (int q = 0; q < 16; ++q) r += acc[j][q];
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), w1 = wall_clock64();
    const int w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if ((threadIdx.x & 63) == 0) { rec[2 * w] = c1 - c0; rec[2 * w + 1] = w1 - w0; }
    if (r == 12345.f) rec[0] = 1;
}
int main() {
    unsigned long long* d; hipMalloc(&d, 4096 * 16);
    std::vector<unsigned long long> h(4096 * 2);
    for (int blocks : {1, 256}) for (int thr : {256, 512}) for (int iters : {200, 2000, 20000, 200000}) {
        const int nw = blocks * thr / 64, wps = thr / 256;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0); chain<2><<<blocks, thr>>>(d, iters, 1e-3f); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h.data(), d, nw * 16, hipMemcpyDeviceToHost);
        double tk = 0, wl = 0; for (int w = 0; w < nw; ++w) { tk += h[2 * w]; wl += h[2 * w + 1]; }
        tk /= nw; wl /= nw;
        const double nm = (double)iters * 16;   // MFMAs per wave
        printf("blocks=%3d waves/SIMD=%d mfma/wave=%8.0f: event %.3f ms | s_memtime ticks per MFMA and wave %.2f (x%d waves = %.1f per SIMD) | tick rate %.3f GHz | %.1f TF\n",
               blocks, wps, nm, ms, tk / nm, wps, tk / nm * wps > 0 ? tk / nm : 0.0, tk / (wl * 10.0), nw * nm * 4096.0 / (wl * 10.0) * 1e-3);
    }
    return 0;
}
