// Micro-benchmark: what do vector instructions INSIDE a wave's own fp32 MFMA stream cost the matrix pipe?  (Round 5: would a flush
// that is interleaved with the next tile's multiply -- instead of running while the SIMD partner streams alone -- pay?)
// One 512-thread workgroup per CU = two waves per SIMD, every wave runs the same loop: a "chunk" of 8 MFMAs (two accumulator
// chains, as gemm_nt's CT = 2) followed by N vector instructions of one kind (independent of the MFMAs); optionally one 16-byte
// store per chunk.  Reported: shader cycles per chunk and wave (s_memtime); the matrix pipe needs 2 waves x 8 x 64 = 1024.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
enum { V_FMA, V_DPP, V_MULHI, NKIND };
static const char* kname[] = {"v_fma (4 independent chains)", "v_mov dpp quad_perm", "v_mad_u64_u32 (Philox round)"};
template <int KIND, int N, int STORE> __global__ __launch_bounds__(512) void k(unsigned long long* rec, float* buf, int nchunk) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    f32x16 acc0, acc1;
    for (int q = 0; q < 16; ++q) { acc0[q] = 0.f; acc1[q] = 0.f; }
    float a = 0.37f + 1e-3f * (lane % 61), b = -0.73f + 1e-3f * (lane % 53);
    float x0 = lane, x1 = lane + 1, x2 = lane + 2, x3 = lane + 3;
    unsigned long long u0 = lane * 2654435761ull;
    f32x4 v = {x0, x1, x2, x3};
    float* p = buf + ((size_t)(blockIdx.x * 8 + wave) * 64 + lane) * 4;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < nchunk; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc1, 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < N / 4; ++j) {
            if (KIND == V_FMA) asm volatile("v_fma_f32 %0, %0, %0, %0\n\tv_fma_f32 %1, %1, %1, %1\n\tv_fma_f32 %2, %2, %2, %2\n\tv_fma_f32 %3, %3, %3, %3" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
            if (KIND == V_DPP) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %1, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %2, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %3, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
            if (KIND == V_MULHI) { u0 = u0 * 0xD2511F53ull + 1; u0 = (u0 >> 32) * 0xCD9E8D57ull + 3; u0 = u0 * 0xD2511F53ull + 1; u0 = (u0 >> 32) * 0xCD9E8D57ull + 3; asm volatile("" : "+v"(u0)); }
        }
        if (STORE) asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
        __builtin_amdgcn_sched_barrier(0);
        a = -a;
    }
    float r = 0; for (int q = 0; q < 16; ++q) r += acc0[q] + acc1[q];
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) rec[blockIdx.x * 8 + wave] = c1 - c0;
    if (r + x0 + x1 + x2 + x3 + (float)u0 == 12345.f) rec[0] = 1;
}
template <int KIND, int N, int STORE> void run(unsigned long long* d, float* buf) {
    const int blocks = 256, nchunk = 4000;
    std::vector<unsigned long long> h(blocks * 8);
    k<KIND, N, STORE><<<blocks, 512>>>(d, buf, nchunk); hipDeviceSynchronize();
    k<KIND, N, STORE><<<blocks, 512>>>(d, buf, nchunk); hipDeviceSynchronize();
    hipMemcpy(h.data(), d, blocks * 8 * 8, hipMemcpyDeviceToHost);
    double cyc = 0; for (auto x : h) cyc += (double)x; cyc /= blocks * 8;
    printf("%-30s N=%3d store=%d : %.0f cycles per chunk and wave (matrix pipe alone: 1024) = pipe occupancy %.3f\n", kname[KIND], N, STORE, cyc / nchunk, 1024.0 / (cyc / nchunk));
}
int main() {
    unsigned long long* d; hipMalloc(&d, 256 * 8 * 8); float* buf; hipMalloc(&buf, 256 * 8 * 64 * 16);
    run<V_FMA, 0, 0>(d, buf); run<V_FMA, 8, 0>(d, buf); run<V_FMA, 16, 0>(d, buf); run<V_FMA, 32, 0>(d, buf); run<V_FMA, 64, 0>(d, buf); run<V_FMA, 128, 0>(d, buf);
    run<V_FMA, 32, 1>(d, buf); run<V_FMA, 0, 1>(d, buf);
    run<V_DPP, 16, 0>(d, buf); run<V_DPP, 32, 0>(d, buf); run<V_DPP, 64, 0>(d, buf);
    run<V_MULHI, 16, 0>(d, buf); run<V_MULHI, 32, 0>(d, buf); run<V_MULHI, 64, 0>(d, buf);
    return 0;
}
