// Micro-benchmark: duration of near-empty kernels vs. block shape / LDS footprint / kernarg size (launch-floor study).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
struct Big { int v[128]; };
__global__ void k_small(float* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1.f; }
__global__ void k_lds(float* p) {
    extern __shared__ float lds[];
    if (threadIdx.x == 0) lds[0] = 1.f;
    __syncthreads();
    if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += lds[0];
}
__global__ void k_big(float* p, Big b) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += b.v[5]; }
template <typename F> float timeit(F f, int n = 200) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 20; ++i) f();
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < n; ++i) f();
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return 1e3f * ms / n;
}
int main() {
    float* p; hipMalloc(&p, 4); hipMemset(p, 0, 4);
    hipFuncSetAttribute((const void*)k_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    Big b{}; 
    printf("back-to-back launch period (us), includes dependent-kernel boundary:\n");
    for (int blocks : {1, 236, 1947}) for (int thr : {256, 512}) printf("small  blocks=%5d thr=%3d : %.2f\n", blocks, thr, timeit([&] { k_small<<<blocks, thr>>>(p); }));
    for (int kb : {0, 32, 64, 140, 158}) printf("lds=%3dKB blocks=236 thr=512 : %.2f\n", kb, timeit([&] { k_lds<<<236, 512, kb * 1024>>>(p); }));
    for (int kb : {0, 64, 158}) printf("lds=%3dKB blocks=944 thr=512 : %.2f\n", kb, timeit([&] { k_lds<<<944, 512, kb * 1024>>>(p); }));
    printf("kernarg 512B blocks=236 thr=512 : %.2f\n", timeit([&] { k_big<<<236, 512>>>(p, b); }));
    return 0;
}
