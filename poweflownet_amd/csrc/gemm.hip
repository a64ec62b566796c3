// fp32 MFMA weight-gradient GEMM (dW = dY^T X) of the hot path (gfx950).  The forward / input-gradient GEMM lives in gemm_nt.hip.
//
// These carry what the reference runs as torch addmm/mm per EDGE (EdgeAggregation.edge_aggr,
// networks/MPN.py:17-21,:28) and per node (TAGConv.lins, mask_embd :491-495), restructured to per-NODE
// products.  All shapes are "tall-skinny": M = nodes (1e4..1e6), K and N <= a few hundred,
// exact fp32 via v_mfma_f32_16x16x4_f32 (there is no TF32/xf32 on gfx950).
//
//  gemm_tn : weight gradients dW = dY^T X (reduction over the node dimension), 9 waves x (48 x 48) output
//            tiles per block, operands staged through LDS as whole rows with register prefetch of the next
//            stage, split over M and reduced in a second, ordered pass (deterministic; no atomics).  Bias
//            gradients ride along as a virtual ones-column of X.
#include <stdlib.h>

#include <algorithm>

#include "pfn_internal.hpp"

namespace pfn {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int CB = 144;               // weight-gradient macro block: 3 x 3 waves of 48 x 48

// ============================================================================================ TN
constexpr int TN_MB = 32;          // node rows per LDS stage
constexpr int TN_LD = CB + 4;      // 148
constexpr int TN_THREADS = 576;    // 9 waves: 3 x 3 macro tiles of 48 x 48
constexpr int TN_MAX_PAIRS = 8;
constexpr int TN_MAX_BLOCKS = 64;  // output macro blocks (144 x 144) per launch
constexpr int TN_Q = CB / 4;       // float4 per staged row (36)

struct TnArgs {
    TnPair pair[TN_MAX_PAIRS];
    int npairs, M, rows_per_split, nsplit, nblocks;
    float* partial;   // [nsplit][nblocks][CB*CB]
    unsigned short blk_pair[TN_MAX_BLOCKS], blk_a0[TN_MAX_BLOCKS], blk_b0[TN_MAX_BLOCKS];
};

// column of the macro block that carries the bias gradient (virtual ones-column), or -1
__device__ __host__ __forceinline__ int tn_bias_col(const TnPair& pr, int b0) {
    if (!pr.bias_out) return -1;
    const int j = pr.nb - b0;              // first column past the real ones
    return (j >= 0 && j < CB) ? j : -1;
}

__global__ __launch_bounds__(TN_THREADS) void gemm_tn_kernel(const TnArgs a) {
    __shared__ __attribute__((aligned(16))) float ldsA[TN_MB * TN_LD];
    __shared__ __attribute__((aligned(16))) float ldsB[TN_MB * TN_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, c = lane >> 4;
    const int by = blockIdx.y;
    const TnPair pr = a.pair[a.blk_pair[by]];
    const int a0 = a.blk_a0[by], b0 = a.blk_b0[by];
    const int wo = wave / 3, wi = wave - 3 * wo;
    const int bcol = tn_bias_col(pr, b0);
    const int na_here = min(CB, pr.na - a0);
    const int nb_here = min(CB, pr.nb - b0) + (bcol >= 0 ? 1 : 0);
    const bool wave_active = (48 * wo < na_here) && (48 * wi < nb_here);
    const int lda4 = (pr.na + 3) & ~3, ldb4 = (pr.nb + 3) & ~3;

    f32x4 acc[3][3];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int u = 0; u < 3; ++u) acc[t][u] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int m_beg = blockIdx.x * a.rows_per_split;
    const int m_end = min(a.M, m_beg + a.rows_per_split);
    // each thread stages two float4 of A and two of B per 32-row stage (32 * 36 = 1152 = 2 * 576)
    float4 pa[2], pb[2];
    auto fetch = [&](int m0) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int i = tid + h * TN_THREADS;
            const int m = i / TN_Q, q = i - m * TN_Q;
            const int gm = m0 + m;
            pa[h] = make_float4(0.f, 0.f, 0.f, 0.f);
            pb[h] = pa[h];
            if (gm < m_end) {
                if (a0 + 4 * q < lda4) pa[h] = *reinterpret_cast<const float4*>(pr.A + (size_t)gm * pr.lda + a0 + 4 * q);
                if (b0 + 4 * q < ldb4) pb[h] = *reinterpret_cast<const float4*>(pr.B + (size_t)gm * pr.ldb + b0 + 4 * q);
                if (bcol >= 0 && (bcol >> 2) == q) {
                    const float one = pr.bias_rowscale ? pr.bias_rowscale[gm] : 1.0f;
                    const int bi = bcol & 3;   // selects, not a runtime-indexed store (that would go to scratch)
                    pb[h].x = bi == 0 ? one : pb[h].x;
                    pb[h].y = bi == 1 ? one : pb[h].y;
                    pb[h].z = bi == 2 ? one : pb[h].z;
                    pb[h].w = bi == 3 ? one : pb[h].w;
                }
            }
        }
    };
    if (m_beg < m_end) fetch(m_beg);
    for (int m0 = m_beg; m0 < m_end; m0 += TN_MB) {
        __syncthreads();   // previous stage fully consumed
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int i = tid + h * TN_THREADS;
            const int m = i / TN_Q, q = i - m * TN_Q;
            *reinterpret_cast<float4*>(ldsA + m * TN_LD + 4 * q) = pa[h];
            *reinterpret_cast<float4*>(ldsB + m * TN_LD + 4 * q) = pb[h];
        }
        __syncthreads();
        if (m0 + TN_MB < m_end) fetch(m0 + TN_MB);   // next stage's loads fly under this stage's MFMAs
        if (wave_active) {
#pragma unroll
            for (int g = 0; g < TN_MB / 16; ++g) {
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const int m = 16 * g + 4 * c + s;   // k-permutation over node rows: rows 4 apart per lane group
                    float av[3], bv[3];
#pragma unroll
                    for (int t = 0; t < 3; ++t) {
                        av[t] = ldsA[m * TN_LD + 48 * wo + 16 * t + r];
                        bv[t] = ldsB[m * TN_LD + 48 * wi + 16 * t + r];
                    }
#pragma unroll
                    for (int t = 0; t < 3; ++t)
#pragma unroll
                        for (int u = 0; u < 3; ++u)
                            acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t], bv[u], acc[t][u], 0, 0, 0);
                }
            }
        }
    }
    if (!wave_active) return;
    float* out = a.partial + ((size_t)blockIdx.x * a.nblocks + by) * (CB * CB);
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int u = 0; u < 3; ++u)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int i = 48 * wo + 16 * t + 4 * c + reg, j = 48 * wi + 16 * u + r;
                if (i < na_here && j < nb_here) out[i * CB + j] = acc[t][u][reg];
            }
}

// second stage: ordered sum over splits, scatter into the nn.Linear gradient layout (+ bias column)
__global__ __launch_bounds__(256) void tn_reduce_kernel(const TnArgs a) {
    const int by = blockIdx.y;
    const TnPair pr = a.pair[a.blk_pair[by]];
    const int a0 = a.blk_a0[by], b0 = a.blk_b0[by];
    const int bcol = tn_bias_col(pr, b0);
    const int na_here = min(CB, pr.na - a0);
    const int nb_real = min(CB, pr.nb - b0);
    const int nb_here = nb_real + (bcol >= 0 ? 1 : 0);
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= na_here * nb_here) return;
    const int i = idx / nb_here, j = idx - i * nb_here;
    const float* p = a.partial + (size_t)by * (CB * CB) + i * CB + j;
    const size_t stride = (size_t)a.nblocks * (CB * CB);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int sp = 0;
    for (; sp + 4 <= a.nsplit; sp += 4) {
        s0 += p[(size_t)sp * stride];
        s1 += p[(size_t)(sp + 1) * stride];
        s2 += p[(size_t)(sp + 2) * stride];
        s3 += p[(size_t)(sp + 3) * stride];
    }
    for (; sp < a.nsplit; ++sp) s0 += p[(size_t)sp * stride];
    const float acc = (s0 + s1) + (s2 + s3);
    if (j < nb_real) pr.G[(size_t)(pr.gn0 + a0 + i) * pr.ldg + pr.gk0 + b0 + j] = acc;
    else pr.bias_out[a0 + i] = acc;
}

// fallback column sums for the (rare) case where the bias column has no room in its macro block
struct ColsumArgs {
    const float* A;
    const float* rowscale;
    float* out;
    int lda, ncols, M;
};
__global__ __launch_bounds__(256) void colsum_fallback_kernel(const ColsumArgs a) {
    __shared__ float red[256];
    const int col = blockIdx.x;
    float acc = 0.f;
    for (int m = threadIdx.x; m < a.M; m += 256) {
        const float v = a.A[(size_t)m * a.lda + col];
        acc += a.rowscale ? a.rowscale[m] * v : v;
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) a.out[col] = red[0];
}

static void tn_split(int64_t M, int nblocks, int& rows_per_split, int& nsplit) {
    // aim for ~one 9-wave block per CU: fewer, longer splits keep the partial-sum traffic (83 KB per block, written
    // then re-read by tn_reduce) below the operand traffic; at least 4 stages (128 rows) per split
    int64_t want = std::max<int64_t>(1, 256 / std::max(1, nblocks));
    int64_t s = std::min<int64_t>(want, (M + 4 * TN_MB - 1) / (4 * TN_MB));
    s = std::max<int64_t>(1, std::min<int64_t>(s, 128));
    rows_per_split = (int)round_up((M + s - 1) / s, TN_MB);
    nsplit = rows_per_split > 0 ? (int)std::max<int64_t>(1, (M + rows_per_split - 1) / rows_per_split) : 1;
}

size_t reduce_ws_floats(int64_t M, int max_na, int max_nb, int max_pairs) {
    (void)M; (void)max_na; (void)max_nb; (void)max_pairs;
    // nsplit * nblocks <= max(512, 64) macro blocks of partials for every split choice of tn_split
    return (size_t)(512 + TN_MAX_BLOCKS) * CB * CB + 1024;
}

int launch_weight_grads(const TnPair* pairs, int npairs, int64_t M, ReduceWs ws, hipStream_t s) {
    int p = 0;
    while (p < npairs) {
        TnArgs ta;
        ta.npairs = 0;
        ta.nblocks = 0;
        ta.M = (int)M;
        ta.partial = ws.partial;
        while (p < npairs && ta.npairs < TN_MAX_PAIRS) {
            const TnPair& pr = pairs[p];
            if (pr.lda % 4 || pr.ldb % 4) {
                set_error("launch_weight_grads: row strides must be multiples of 4");
                return PFN_EINVAL;
            }
            const int nb_blocks = ((pr.na + CB - 1) / CB) * ((pr.nb + CB - 1) / CB);
            if (nb_blocks > TN_MAX_BLOCKS) {
                set_error("launch_weight_grads: a %d x %d weight needs %d macro blocks (> %d)", pr.na, pr.nb, nb_blocks,
                          TN_MAX_BLOCKS);
                return PFN_EINVAL;
            }
            if (ta.nblocks + nb_blocks > TN_MAX_BLOCKS) break;
            for (int a0 = 0; a0 < pr.na; a0 += CB)
                for (int b0 = 0; b0 < pr.nb; b0 += CB) {
                    ta.blk_pair[ta.nblocks] = (unsigned short)ta.npairs;
                    ta.blk_a0[ta.nblocks] = (unsigned short)a0;
                    ta.blk_b0[ta.nblocks] = (unsigned short)b0;
                    ++ta.nblocks;
                }
            ta.pair[ta.npairs++] = pr;
            ++p;
        }
        if (ta.nblocks == 0) continue;
        tn_split(M, ta.nblocks, ta.rows_per_split, ta.nsplit);
        if ((size_t)ta.nsplit * ta.nblocks * CB * CB > ws.floats) {
            set_error("launch_weight_grads: reduction workspace %zu < %zu floats", ws.floats,
                      (size_t)ta.nsplit * ta.nblocks * CB * CB);
            return PFN_ENOSPACE;
        }
        if (M > 0) {
            double flops = 0.0, bytes = 0.0;
            for (int q = 0; q < ta.npairs; ++q) {
                flops += 2.0 * (double)M * ta.pair[q].na * ta.pair[q].nb;
                bytes += 4.0 * (double)M * (ta.pair[q].na + ta.pair[q].nb);
            }
            ProfScope ps("gemm_tn", bytes, flops, s);
            gemm_tn_kernel<<<dim3(ta.nsplit, ta.nblocks), TN_THREADS, 0, s>>>(ta);
            PFN_CHECK_LAUNCH();
        } else {
            ta.nsplit = 0;
        }
        {
            ProfScope ps("tn_reduce", 0.0, 0.0, s);
            tn_reduce_kernel<<<dim3((CB * (CB + 1) + 255) / 256, ta.nblocks), 256, 0, s>>>(ta);
            PFN_CHECK_LAUNCH();
        }
        for (int q = 0; q < ta.npairs; ++q) {   // bias column without room in its macro block (nb % 144 == 0)
            const TnPair& pr = ta.pair[q];
            if (pr.bias_out && pr.nb % CB == 0) {
                ColsumArgs ca{pr.A, pr.bias_rowscale, pr.bias_out, pr.lda, pr.na, (int)M};
                colsum_fallback_kernel<<<pr.na, 256, 0, s>>>(ca);
                PFN_CHECK_LAUNCH();
            }
        }
    }
    return PFN_OK;
}

}  // namespace pfn
