// Micro-benchmark: sustained v_mfma_f32_32x32x2_f32 / 16x16x4 rate vs waves per SIMD and accumulators per wave.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NACC> __global__ __launch_bounds__(512) void k32(float* out, int iters, float s) {
    f32x16 acc[NACC];
    for (int j = 0; j < NACC; ++j) for (int q = 0; q < 16; ++q) acc[j][q] = 0.f;
    float a = s * threadIdx.x, b = s + 1.f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
    }
    float r = 0; for (int j = 0; j < NACC; ++j) for (int q = 0; q < 16; ++q) r += acc[j][q];
    if (r == 12345.f) out[0] = r;
}
template <int NACC> __global__ __launch_bounds__(512) void k16(float* out, int iters, float s) {
    f32x4 acc[NACC];
    for (int j = 0; j < NACC; ++j) for (int q = 0; q < 4; ++q) acc[j][q] = 0.f;
    float a = s * threadIdx.x, b = s + 1.f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j], 0, 0, 0);
    }
    float r = 0; for (int j = 0; j < NACC; ++j) for (int q = 0; q < 4; ++q) r += acc[j][q];
    if (r == 12345.f) out[0] = r;
}
template <typename F> float timeit(F f) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    hipEventRecord(a); for (int i = 0; i < 5; ++i) f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / 5;
}
int main() {
    float* p; hipMalloc(&p, 4);
    const int iters = 2000;
    for (int thr : {256, 512}) for (int blocks : {256, 512}) {
        const double waves = (double)blocks * thr / 64;
        float ms;
        ms = timeit([&] { k32<1><<<blocks, thr>>>(p, iters, 0.f); }); printf("32x32x2 acc=1 blocks=%d thr=%d: %.1f TF\n", blocks, thr, waves * iters * 8 * 1 * 4096.0 / ms * 1e-9);
        ms = timeit([&] { k32<2><<<blocks, thr>>>(p, iters, 0.f); }); printf("32x32x2 acc=2 blocks=%d thr=%d: %.1f TF\n", blocks, thr, waves * iters * 8 * 2 * 4096.0 / ms * 1e-9);
        ms = timeit([&] { k32<4><<<blocks, thr>>>(p, iters, 0.f); }); printf("32x32x2 acc=4 blocks=%d thr=%d: %.1f TF\n", blocks, thr, waves * iters * 8 * 4 * 4096.0 / ms * 1e-9);
        ms = timeit([&] { k16<1><<<blocks, thr>>>(p, iters, 0.f); }); printf("16x16x4 acc=1 blocks=%d thr=%d: %.1f TF\n", blocks, thr, waves * iters * 8 * 1 * 2048.0 / ms * 1e-9);
        ms = timeit([&] { k16<4><<<blocks, thr>>>(p, iters, 0.f); }); printf("16x16x4 acc=4 blocks=%d thr=%d: %.1f TF\n", blocks, thr, waves * iters * 8 * 4 * 2048.0 / ms * 1e-9);
    }
    // nonzero data (DVFS): s = 1e-3
    float ms = timeit([&] { k32<2><<<256, 512>>>(p, iters, 1e-3f); }); printf("32x32x2 acc=2 blocks=256 thr=512 nonzero data: %.1f TF\n", 4096.0 * 256 * 8 * iters * 8 * 2 / ms * 1e-9);
    return 0;
}
