// Micro-benchmark: what does ONE kernel cost inside a replayed hipGraph chain of dependent kernels (the step of config 2 is 41
// of them)?  Empty kernels of several shapes, and a kernel that streams a few MB, captured 40 in a row.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <functional>
__global__ void k_empty(float* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1.f; }
__global__ void k_lds(float* p) {
    extern __shared__ float lds[];
    if (threadIdx.x == 0) lds[0] = 1.f;
    __syncthreads();
    if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += lds[0];
}
__global__ void k_stream(const float4* a, float4* b, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float4 v = a[i]; v.x += 1.f; b[i] = v;
    }
}
static float graph_us(std::function<void(hipStream_t)> enqueue, int chain = 40, int reps = 200) {
    hipStream_t s; hipStreamCreate(&s);
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
    for (int i = 0; i < chain; ++i) enqueue(s);
    hipStreamEndCapture(s, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    for (int i = 0; i < 10; ++i) hipGraphLaunch(ge, s);
    hipStreamSynchronize(s);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a, s);
    for (int i = 0; i < reps; ++i) hipGraphLaunch(ge, s);
    hipEventRecord(b, s); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return 1e3f * ms / reps / chain;
}
int main() {
    float* p; hipMalloc(&p, 4); hipMemset(p, 0, 4);
    hipFuncSetAttribute((const void*)k_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    printf("us per kernel inside a replayed graph chain of 40 dependent kernels:\n");
    for (int blocks : {1, 64, 236, 2048}) for (int thr : {256, 512})
        printf("  empty   blocks=%5d thr=%3d            : %.2f\n", blocks, thr, graph_us([&](hipStream_t s) { k_empty<<<blocks, thr, 0, s>>>(p); }));
    for (int kb : {0, 72, 148}) printf("  lds=%3dKB blocks=236 thr=512          : %.2f\n", kb, graph_us([&](hipStream_t s) { k_lds<<<236, 512, kb * 1024, s>>>(p); }));
    for (long mb : {1, 8, 32}) {
        float4 *a, *b; long n = mb * 1024 * 1024 / 16;
        hipMalloc(&a, n * 16); hipMalloc(&b, n * 16); hipMemset(a, 0, n * 16);
        printf("  stream %2ld MB in + out, 1024 blocks x 256 : %.2f\n", mb, graph_us([&](hipStream_t s) { k_stream<<<1024, 256, 0, s>>>(a, b, n); }));
        hipFree(a); hipFree(b);
    }
    return 0;
}
