// Micro-benchmark: what does the OTHER wave of a SIMD cost a wave that streams v_mfma_f32_32x32x2_f32?
// One workgroup of 512 threads per CU: waves 0-3 (one per SIMD) run a fixed chain of MFMAs on two accumulators and time it with
// s_memtime; waves 4-7 (their SIMD partners) run a filler loop of one instruction kind until wave 0 raises a flag in LDS, and
// count what they got through.  cost = (MFMA cycles with filler - alone) / filler instructions executed meanwhile.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
enum { F_NONE, F_VALU, F_VALU_DEP, F_SALU, F_LDS, F_STORE, F_LOAD, F_MFMA, F_VALU_PK, F_DPP, F_NKIND };
static const char* kname[] = {"none", "v_fma (4 independent)", "v_fma (dependent chain)", "s_add", "ds_read_b128", "global_store_dwordx4", "global_load_dwordx4", "mfma too", "v_pk_fma", "v_mov dpp"};
template <int KIND, int PRIO = 0> __global__ __launch_bounds__(512) void k(unsigned long long* rec, float* buf, int nmfma) {
    __shared__ int flag;
    __shared__ __attribute__((aligned(16))) float tile[4096];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (threadIdx.x == 0) flag = 0;
    for (int i = threadIdx.x; i < 4096; i += 512) tile[i] = 0.001f * i;
    __syncthreads();
    if (wave < 4) {
        f32x16 acc0, acc1;
        for (int q = 0; q < 16; ++q) { acc0[q] = 0.f; acc1[q] = 0.f; }
        float a = 0.37f + 1e-3f * (lane % 61), b = -0.73f + 1e-3f * (lane % 53);
        const unsigned long long c0 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < nmfma / 8; ++i) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc1, 0, 0, 0);
            }
            a = -a;
        }
        float r = 0; for (int q = 0; q < 16; ++q) r += acc0[q] + acc1[q];
        const unsigned long long c1 = __builtin_amdgcn_s_memtime();
        __atomic_store_n(&flag, 1, __ATOMIC_RELAXED);
        if (lane == 0) rec[(blockIdx.x * 8 + wave) * 2] = c1 - c0;
        if (r == 12345.f) rec[0] = 1;
    } else {
        unsigned long long n = 0;
        if (PRIO) __builtin_amdgcn_s_setprio(3);
        float x0 = lane, x1 = lane + 1, x2 = lane + 2, x3 = lane + 3; int s = 0;
        f32x4 v = {x0, x1, x2, x3};
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        f32x2 pk0 = {x0, x1}, pk1 = {x2, x3};
        f32x16 acc; for (int q = 0; q < 16; ++q) acc[q] = 0.f;
        float* p = buf + ((size_t)(blockIdx.x * 8 + wave) * 64 + lane) * 4;
        if (KIND != F_NONE) while (__atomic_load_n(&flag, __ATOMIC_RELAXED) == 0) {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                if (KIND == F_VALU) { asm volatile("v_fma_f32 %0, %0, %0, %0\n\tv_fma_f32 %1, %1, %1, %1\n\tv_fma_f32 %2, %2, %2, %2\n\tv_fma_f32 %3, %3, %3, %3" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3)); n += 4; }
                if (KIND == F_VALU_DEP) { asm volatile("v_fma_f32 %0, %0, %0, %0\n\tv_fma_f32 %0, %0, %0, %0\n\tv_fma_f32 %0, %0, %0, %0\n\tv_fma_f32 %0, %0, %0, %0" : "+v"(x0)); n += 4; }
                if (KIND == F_VALU_PK) { asm volatile("v_pk_fma_f32 %0, %0, %0, %0\n\tv_pk_fma_f32 %1, %1, %1, %1" : "+v"(pk0), "+v"(pk1)); n += 2; }
                if (KIND == F_DPP) { asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %1, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf" : "+v"(x0), "+v"(x1)); n += 2; }
                if (KIND == F_SALU) { asm volatile("s_add_i32 %0, %0, 1\n\ts_add_i32 %0, %0, 1\n\ts_add_i32 %0, %0, 1\n\ts_add_i32 %0, %0, 1" : "+s"(s)); n += 4; }
                if (KIND == F_LDS) { f32x4 t = *reinterpret_cast<volatile f32x4*>(tile + ((lane * 4 + u * 256) & 4095)); x0 += t[0]; n += 1; }
                if (KIND == F_STORE) { asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" : : "v"(p), "v"(v) : "memory"); n += 1; }
                if (KIND == F_LOAD) { f32x4 t; asm volatile("global_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(t) : "v"(p) : "memory"); x0 += t[0]; n += 1; }
                if (KIND == F_MFMA) { acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x0, x1, acc, 0, 0, 0); n += 1; }
            }
        }
        if (lane == 0) rec[(blockIdx.x * 8 + wave) * 2] = n;
        if (x0 + x1 + x2 + x3 + s + v[0] + v[2] + pk0[0] + pk1[1] + acc[0] == 12345.f) rec[1] = 1;
    }
}
template <int KIND, int PRIO = 0> void run(unsigned long long* d, float* buf, int blocks, double base[1]) {
    const int nmfma = 40000;
    std::vector<unsigned long long> h(blocks * 16);
    hipMemset(d, 0, blocks * 16 * 8);
    k<KIND, PRIO><<<blocks, 512>>>(d, buf, nmfma); hipDeviceSynchronize();
    k<KIND, PRIO><<<blocks, 512>>>(d, buf, nmfma); hipDeviceSynchronize();
    hipMemcpy(h.data(), d, blocks * 16 * 8, hipMemcpyDeviceToHost);
    double cyc = 0, nf = 0;
    for (int b = 0; b < blocks; ++b) for (int w = 0; w < 8; ++w) (w < 4 ? cyc : nf) += (double)h[(b * 8 + w) * 2];
    cyc /= blocks * 4; nf /= blocks * 4;
    if (KIND == F_NONE) base[0] = cyc;
    printf("blocks=%3d prio=%d partner: %-26s MFMA wave: %.1f cycles per MFMA | partner got %.0f instr = %.2f per MFMA | pipe cycles lost per partner instr %.1f\n", blocks, PRIO,
           kname[KIND], cyc / nmfma, nf, nf / nmfma, nf > 0 ? (cyc - base[0]) / nf : 0.0);
}
int main() {
    unsigned long long* d; hipMalloc(&d, 256 * 16 * 8); float* buf; hipMalloc(&buf, 256 * 8 * 64 * 16);
    for (int blocks : {256}) {
        double base[1] = {0};
        run<F_NONE>(d, buf, blocks, base); run<F_VALU>(d, buf, blocks, base); run<F_VALU_DEP>(d, buf, blocks, base); run<F_VALU_PK>(d, buf, blocks, base);
        run<F_DPP>(d, buf, blocks, base); run<F_SALU>(d, buf, blocks, base); run<F_LDS>(d, buf, blocks, base); run<F_STORE>(d, buf, blocks, base);
        run<F_LOAD>(d, buf, blocks, base); run<F_MFMA>(d, buf, blocks, base);
        run<F_VALU, 1>(d, buf, blocks, base); run<F_VALU_DEP, 1>(d, buf, blocks, base); run<F_DPP, 1>(d, buf, blocks, base); run<F_STORE, 1>(d, buf, blocks, base); run<F_LDS, 1>(d, buf, blocks, base); run<F_MFMA, 1>(d, buf, blocks, base);
    }
    return 0;
}
