// Micro-benchmark: sustained v_mfma_f32_32x32x16_bf16 rate (nonzero data) next to v_mfma_f32_32x32x2_f32 -- is an fp32 product
// split into bf16 partial products worth the matrix cores' 16x nominal rate once the chip's power management has its say?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int NACC> __global__ __launch_bounds__(512) void kb(float* out, int iters, float s) {
    f32x16 acc[NACC];
    for (int j = 0; j < NACC; ++j) for (int q = 0; q < 16; ++q) acc[j][q] = 0.f;
    bf16x8 a, b;
    for (int q = 0; q < 8; ++q) { a[q] = (__bf16)(s * (threadIdx.x + q) + 0.37f); b[q] = (__bf16)(s * q + 1.13f); }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j], 0, 0, 0);
    }
    float r = 0; for (int j = 0; j < NACC; ++j) for (int q = 0; q < 16; ++q) r += acc[j][q];
    if (r == 12345.f) out[0] = r;
}
template <int NACC> __global__ __launch_bounds__(512) void kf(float* out, int iters, float s) {
    f32x16 acc[NACC];
    for (int j = 0; j < NACC; ++j) for (int q = 0; q < 16; ++q) acc[j][q] = 0.f;
    float a = s * threadIdx.x + 0.37f, b = s + 1.13f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
    }
    float r = 0; for (int j = 0; j < NACC; ++j) for (int q = 0; q < 16; ++q) r += acc[j][q];
    if (r == 12345.f) out[0] = r;
}
template <typename F> float timeit(F f, int reps) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    hipEventRecord(a); for (int i = 0; i < reps; ++i) f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / reps;
}
int main() {
    float* p; hipMalloc(&p, 4);
    for (int iters : {2000, 20000}) {
        for (int blocks : {256, 512}) {
            const double waves = (double)blocks * 8;
            float ms;
            ms = timeit([&] { kb<2><<<blocks, 512>>>(p, iters, 1e-3f); }, 5);
            printf("bf16 32x32x16 acc=2 blocks=%d iters=%d: %.0f TF  (%.3f ms)\n", blocks, iters, waves * iters * 8 * 2 * 32768.0 / ms * 1e-9, ms);
            ms = timeit([&] { kb<4><<<blocks, 512>>>(p, iters, 1e-3f); }, 5);
            printf("bf16 32x32x16 acc=4 blocks=%d iters=%d: %.0f TF  (%.3f ms)\n", blocks, iters, waves * iters * 8 * 4 * 32768.0 / ms * 1e-9, ms);
            ms = timeit([&] { kf<2><<<blocks, 512>>>(p, iters / 4, 1e-3f); }, 5);
            printf("fp32 32x32x2  acc=2 blocks=%d iters=%d: %.1f TF  (%.3f ms)\n", blocks, iters / 4, waves * (iters / 4) * 8 * 2 * 4096.0 / ms * 1e-9, ms);
        }
    }
    return 0;
}
