// Standalone timing harness for launch_gemm_nt (links libpfn_hip.so): isolates the tall-skinny GEMM from the model.
//   gemm_nt_bench M K N nterm ngroup [iters]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <math.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#include <utility>
#include "../../poweflownet_amd/csrc/pfn_internal.hpp"
using namespace pfn;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
#ifdef NT_EXP_TS
extern "C" int pfn_debug_nt_ts(unsigned long long*, int);
#endif
#ifdef NT_EXP_TS2
extern "C" int pfn_debug_nt_ts2(unsigned long long*, int);
extern "C" int pfn_debug_nt_ts2_dump();
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
#ifdef NT_EXP_TS2   /* built against a -DNT_EXP_TS2 library (tools/ubench/run_gemm_nt_ts2.sh): per-wave cycle accounting of the LAST launch */
    {
        static unsigned long long ts[4096 * 8];
        CK(hipDeviceSynchronize());
        pfn_debug_nt_ts2(ts, 4096 * 8);
        double mul = 0, fl = 0, np = 0, nf = 0, tot = 0, totmax = 0; int cnt = 0;
        for (int w = 0; w < 4096; ++w) {
            if (ts[w * 8 + 4] == 0) continue;
            mul += ts[w * 8]; fl += ts[w * 8 + 1]; np += ts[w * 8 + 2]; nf += ts[w * 8 + 3]; tot += ts[w * 8 + 4];
            if ((double)ts[w * 8 + 4] > totmax) totmax = (double)ts[w * 8 + 4]; ++cnt;
        }
        printf("    %d waves: %.1f pieces, %.1f flushes per wave | cycles per piece in multiply %.0f | per flush %.0f | loop total mean %.0f max %.0f | elsewhere per piece %.0f\n",
               cnt, np / cnt, nf / cnt, mul / np, fl / nf, tot / cnt, totmax, (tot - mul - fl) / np);
        pfn_debug_nt_ts2_dump();
        // where do the slow waves sit?  wave w of block b (grid x = blocks per slice): by XCD (b mod 8), by wave slot, by tile count
        double xs[8] = {0}, xn[8] = {0}, ws[8] = {0}, wn[8] = {0}, ps[64] = {0}, pn[64] = {0}, cyc_pp[8] = {0};
        for (int w = 0; w < 4096; ++w) {
            if (ts[w * 8 + 4] == 0) continue;
            const int b = w / 8, x = b % 8, wi = w % 8; const double t = (double)ts[w * 8 + 4]; const int npc = (int)ts[w * 8 + 2];
            xs[x] += t; xn[x] += 1; ws[wi] += t; wn[wi] += 1; if (npc < 64) { ps[npc] += t; pn[npc] += 1; } cyc_pp[x] += t / (npc > 0 ? npc : 1);
        }
        printf("    loop cycles by XCD (block mod 8):"); for (int x = 0; x < 8; ++x) printf(" %.0f", xn[x] ? xs[x] / xn[x] : 0.0); printf("\n");
        printf("    loop cycles per piece by XCD    :"); for (int x = 0; x < 8; ++x) printf(" %.0f", xn[x] ? cyc_pp[x] / xn[x] : 0.0); printf("\n");
        printf("    loop cycles by wave slot        :"); for (int x = 0; x < 8; ++x) printf(" %.0f", wn[x] ? ws[x] / wn[x] : 0.0); printf("\n");
        printf("    loop cycles by pieces per wave  :"); for (int x = 0; x < 64; ++x) if (pn[x]) printf(" [%d: %.0f waves, %.0f]", x, pn[x], ps[x] / pn[x]); printf("\n");
        if (getenv("PFN_TS2_GX")) {   // loop cycles by slice (blockIdx.y) when the caller knows the grid's x extent
            const int gx = atoi(getenv("PFN_TS2_GX")); double ys[16] = {0}, yn[16] = {0}, ymax[16] = {0};
            for (int w = 0; w < 4096; ++w) { if (ts[w * 8 + 4] == 0) continue; const int y = (w / 8) / gx; if (y < 16) { ys[y] += (double)ts[w * 8 + 4]; yn[y] += 1; if ((double)ts[w * 8 + 4] > ymax[y]) ymax[y] = (double)ts[w * 8 + 4]; } }
            printf("    loop cycles by slice (mean / max):"); for (int y = 0; y < 16; ++y) if (yn[y]) printf(" [%d: %.0f / %.0f]", y, ys[y] / yn[y], ymax[y]); printf("\n");
        }
        // the ten slowest waves
        std::vector<std::pair<double, int>> v; for (int w = 0; w < 4096; ++w) if (ts[w * 8 + 4]) v.push_back({(double)ts[w * 8 + 4], w});
        std::sort(v.begin(), v.end());
        printf("    slowest:"); for (int i = 0; i < 10 && i < (int)v.size(); ++i) { auto& e = v[v.size() - 1 - i]; printf(" (b%d w%d %.0f)", e.second / 8, e.second % 8, e.first); } printf("\n");
        printf("    fastest:"); for (int i = 0; i < 6 && i < (int)v.size(); ++i) { auto& e = v[i]; printf(" (b%d w%d %.0f)", e.second / 8, e.second % 8, e.first); } printf("\n");
        printf("    percentiles of the loop total: p10 %.0f p50 %.0f p90 %.0f p99 %.0f max %.0f\n", v[v.size() / 10].first, v[v.size() / 2].first, v[v.size() * 9 / 10].first, v[v.size() * 99 / 100].first, v.back().first);
    }
#endif
    return 0;
}
