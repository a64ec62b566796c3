// Standalone timing harness for launch_gemm_nt (links libpfn_hip.so): isolates the tall-skinny GEMM from the model.
//   gemm_nt_bench M K N nterm ngroup [iters]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <math.h>
#include <stdlib.h>
#include <vector>
#include "../../poweflownet_amd/csrc/pfn_internal.hpp"
using namespace pfn;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
#ifdef NT_EXP_TS
extern "C" int pfn_debug_nt_ts(unsigned long long*, int);
#endif
int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 15104, K = argc > 2 ? atoi(argv[2]) : 129, N = argc > 3 ? atoi(argv[3]) : 129;
    const int nterm = argc > 4 ? atoi(argv[4]) : 1, ngroup = argc > 5 ? atoi(argv[5]) : 1, iters = argc > 6 ? atoi(argv[6]) : 20;
    const int lda = ld_of(K), ldc = ld_of(N);
    std::vector<float> hA((size_t)M * lda), hW((size_t)N * K);
    for (auto& v : hA) v = (rand() % 2001 - 1000) * 1e-3f;
    for (size_t i = 0; i < hA.size(); ++i) if ((int)(i % lda) >= K) hA[i] = 0.f;
    for (auto& v : hW) v = (rand() % 2001 - 1000) * 1e-3f;
    float *dA, *dW, *dP, *dC;
    CK(hipMalloc(&dA, hA.size() * 4 * nterm)); CK(hipMalloc(&dW, hW.size() * 4));
    for (int t = 0; t < nterm; ++t) CK(hipMemcpy(dA + (size_t)t * hA.size(), hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dW, hW.data(), hW.size() * 4, hipMemcpyHostToDevice));
    const size_t pf = packed_floats(K, ldc);
    CK(hipMalloc(&dP, pf * 4)); CK(hipMalloc(&dC, (size_t)M * ldc * 4 * ngroup));
    PackJob pj; pj.src = dW; pj.dst = dP; pj.ldw = K; pj.wk0 = 0; pj.wn0 = 0; pj.trans = 1; pj.K = K; pj.ncols = N; pj.ld_out = ldc;
    if (launch_pack(&pj, 1, nullptr, 0) != 0) { printf("pack failed: %s\n", pfn_last_error()); return 1; }
    GemmArgs a; memset(&a, 0, sizeof(a));
    a.M = M; a.ncols = N; a.ldc = ldc; a.ngroup = ngroup; a.gate_scale = 1.f; a.bias_group = -1; a.nterm = nterm;
    for (int g = 0; g < ngroup; ++g) a.C[g] = dC + (size_t)g * M * ldc;
    for (int t = 0; t < nterm; ++t) { a.term[t].A = dA + (size_t)t * hA.size(); a.term[t].Bp = dP; a.term[t].lda = lda; a.term[t].K = K; a.term[t].group = ngroup > 1 ? t * ngroup / nterm : 0; }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) if (launch_gemm_nt(a, 0) != 0) { printf("gemm failed: %s\n", pfn_last_error()); return 1; }
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; ++i) launch_gemm_nt(a, 0);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = 1e3 * ms / iters, flops = 2.0 * M * K * N * nterm;
    // spot check a few entries against a host dot product (first group, first term only meaningful when nterm == ngroup == 1)
    // check every 97th row of group 0 against a host dot product (meaningful when every term uses the same A and W)
    std::vector<float> hC((size_t)M * ldc);
    CK(hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost));
    double maxerr = 0; int bad_r = -1, bad_c = -1;
    for (int r = 0; r < M; r += (r < 128 ? 1 : 97)) for (int c = 0; c < N; c += 1) {
        double ref = 0; for (int k = 0; k < K; ++k) ref += (double)hA[(size_t)r * lda + k] * hW[(size_t)c * K + k];
        ref *= (ngroup > 1 ? 1 : nterm);
        const double err = fabs(ref - hC[(size_t)r * ldc + c]); if (err > maxerr) { maxerr = err; if (err > 1e-3 && bad_r < 0) { bad_r = r; bad_c = c; } }
    }
    if (bad_r >= 0) printf("  first bad element: row %d col %d\n", bad_r, bad_c);
    if (getenv("PFN_MAP") && M <= 128) {
        for (int r = 0; r < M; ++r) { printf("  r%3d ", r); for (int c = 0; c < N; ++c) {
            double ref = 0; for (int k = 0; k < K; ++k) ref += (double)hA[(size_t)r * lda + k] * hW[(size_t)c * K + k];
            ref *= (ngroup > 1 ? 1 : nterm); putchar(fabs(ref - hC[(size_t)r * ldc + c]) > 1e-3 ? 'X' : '.'); } putchar('\n'); }
    }
    printf("M=%d K=%d N=%d nterm=%d ngroup=%d : %.2f us  %.1f TFLOP/s  (spot max err %.2e)\n", M, K, N, nterm, ngroup, us, flops / us * 1e-6, maxerr);
#ifdef NT_EXP_TS   /* built against a -DNT_EXP_TS library (tools/ubench/run_gemm_nt_ts.sh): phase timestamps of the LAST launch */
    {
        static unsigned long long ts[4096 * 4];
        CK(hipDeviceSynchronize());
        pfn_debug_nt_ts(ts, 4096 * 4);
        unsigned long long t0 = ~0ull;
        int nb = 0;
        for (int b = 0; b < 4096; ++b) if (ts[b * 4 + 3] != 0) { nb = b + 1; if (ts[b * 4] < t0) t0 = ts[b * 4]; }
        double s0 = 0, s1 = 0, s2 = 0, s3 = 0, m0 = 0, m3 = 0; int cnt = 0;
        for (int b = 0; b < nb; ++b) {
            if (ts[b * 4 + 3] == 0) continue;
            const double a0 = (ts[b * 4] - t0) * 0.01, a1 = (ts[b * 4 + 1] - t0) * 0.01, a2 = (ts[b * 4 + 2] - t0) * 0.01, a3 = (ts[b * 4 + 3] - t0) * 0.01;
            s0 += a0; s1 += a1 - a0; s2 += a2 - a1; s3 += a3 - a2; if (a0 > m0) m0 = a0; if (a3 > m3) m3 = a3; ++cnt;
        }
        printf("    %d workgroups: start mean %.2f max %.2f | to first barrier %.2f | multiply %.2f | flush %.2f | last end %.2f us\n", cnt,
               s0 / cnt, m0, s1 / cnt, s2 / cnt, s3 / cnt, m3);
    }
#endif
    return 0;
}
