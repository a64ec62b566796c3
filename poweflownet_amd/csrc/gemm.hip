// fp32 MFMA GEMMs for the per-node dense contractions of the hot path (gfx950).
//
// These carry what the reference runs as torch addmm/mm per EDGE (EdgeAggregation.edge_aggr,
// networks/MPN.py:17-21,:28) and per node (TAGConv.lins, mask_embd :491-495), restructured to per-NODE
// products.  All shapes are "tall-skinny": M = nodes (1e4..1e6), K and N <= a few hundred,
// exact fp32 via v_mfma_f32_16x16x4_f32 (there is no TF32/xf32 on gfx950).
//
//  gemm_nt : C = sum_t A_t * B_t (+ epilogue).  One wave owns 16 rows x up to 144 columns (9 accumulator
//            tiles); its A fragment comes straight from global memory as float4 (rows are private to the
//            wave, so LDS staging would buy nothing), using a k-permutation inside each 16-wide k chunk:
//            lane group c = lane>>4 supplies k = 16j + 4c + i at MFMA step i, so one 16-byte load feeds four
//            steps.  The weight tile B (shared by the 4 waves of a block) is staged through LDS with a row
//            stride of 148 floats (= 4 mod 8), which makes the permuted B fragment reads bank-conflict free.
//  gemm_tn : weight gradients dW = dY^T X (reduction over the node dimension), 9 waves x (48 x 48) output
//            tiles per block, operands staged through LDS as whole rows, split over M and reduced in a
//            second, ordered pass (deterministic; no atomics).
#include <algorithm>

#include "pfn_internal.hpp"

namespace pfn {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int NT = 9;            // 16-column tiles per wave  -> 144 columns per column block
constexpr int CB = NT * 16;      // 144
constexpr int BK = 32;           // k rows of B per LDS stage
constexpr int LDB = CB + 4;      // 148: (4 * LDB) % 32 == 16 -> the 4 lane groups hit disjoint banks
constexpr int ROWS_PER_BLOCK = 64;

__global__ __launch_bounds__(256) void gemm_nt_kernel(const GemmArgs a) {
    __shared__ __attribute__((aligned(16))) float ldsB[2][BK * LDB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, c = lane >> 4;
    const int group = blockIdx.y / a.ncb, cb = blockIdx.y - group * a.ncb;
    const int n0 = cb * CB;
    const int row0 = blockIdx.x * ROWS_PER_BLOCK + wave * 16;
    const int arow = row0 + r;
    const bool arow_ok = arow < a.M;
    int ntile = (a.ldc - n0 + 15) / 16;
    ntile = ntile > NT ? NT : ntile;

    f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};

    int stage = 0;
    for (int ti = 0; ti < a.nterm; ++ti) {
        const GemmTerm tm = a.term[ti];
        if (tm.group != group) continue;
        const float* Arow = tm.A + (size_t)arow * tm.lda;
        const int K4 = (tm.K + 3) & ~3;
        for (int k0 = 0; k0 < tm.K; k0 += BK, ++stage) {
            float* B = ldsB[stage & 1];
            // ---- stage the weight tile: rows k0..k0+31, columns n0..n0+143 (zero outside the weight)
            if (tm.trans) {
                for (int i = tid; i < BK * CB; i += 256) {
                    const int k = i & (BK - 1), n = i >> 5;
                    const int gk = k0 + k, gn = n0 + n;
                    B[k * LDB + n] = (gk < tm.K && gn < a.ncols) ? tm.W[(size_t)(tm.wn0 + gn) * tm.ldw + tm.wk0 + gk] : 0.f;
                }
            } else {
                for (int i = tid; i < BK * CB; i += 256) {
                    const int k = i / CB, n = i - k * CB;
                    const int gk = k0 + k, gn = n0 + n;
                    B[k * LDB + n] = (gk < tm.K && gn < a.ncols) ? tm.W[(size_t)(tm.wk0 + gk) * tm.ldw + tm.wn0 + gn] : 0.f;
                }
            }
            // ---- this wave's A fragment for the stage: two 16-wide k chunks, 16 B per lane each
            float4 a4[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int kk = k0 + 16 * j + 4 * c;
                a4[j] = (arow_ok && kk < K4) ? *reinterpret_cast<const float4*>(Arow + kk) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
            __syncthreads();
            const float* Bl = B + (4 * c) * LDB + r;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const float av[4] = {a4[j].x, a4[j].y, a4[j].z, a4[j].w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (k0 + 16 * j + i < tm.K) {   // block-uniform: skip MFMA steps that only see zero padding
#pragma unroll
                        for (int t = 0; t < NT; ++t) {
                            if (t < ntile) {
                                const float b = Bl[(16 * j + i) * LDB + 16 * t];
                                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], b, acc[t], 0, 0, 0);
                            }
                        }
                    }
                }
            }
        }
    }

    // ---- epilogue.  Lane holds D[row = 4c + reg][col = r] of every tile.
    float* C = a.C[group];
    uint64_t seed = 0, offset = 0;
    if (a.act == ACT_DROPOUT_RELU) {
        seed = a.rng[0];
        offset = a.rng[1];
    }
    const float keep_scale = a.act == ACT_DROPOUT_RELU ? 1.0f / (1.0f - a.p_drop) : 1.0f;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        if (t >= ntile) continue;
        const int col = n0 + 16 * t + r;
        if (col >= a.ldc) continue;
        const bool real = col < a.ncols;
        const float bias = (real && a.bias && (a.bias_group < 0 || a.bias_group == group)) ? a.bias[col] : 0.f;
        const float rbias = (real && a.rowscale) ? a.rowbias[col] : 0.f;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int row = row0 + 4 * c + reg;
            if (row >= a.M) continue;
            float v = 0.f;
            if (real) {
                v = acc[t][reg] + bias;
                if (a.rowscale) v = fmaf(a.rowscale[row], rbias, v);
                if (a.resid) v += a.resid[(size_t)row * a.ldr + col];
                if (a.act == ACT_RELU) {
                    v = fmaxf(v, 0.f);
                } else if (a.act == ACT_DROPOUT_RELU) {
                    const float u = uniform_hash(seed, offset, a.rng_stream, (uint64_t)row * a.ncols + col);
                    v = (u >= a.p_drop && v > 0.f) ? v * keep_scale : 0.f;
                }
                if (a.gate) v = a.gate[(size_t)row * a.ldg + col] > 0.f ? v * a.gate_scale : 0.f;
            }
            C[(size_t)row * a.ldc + col] = v;
        }
    }
}

int launch_gemm_nt(const GemmArgs& a_in, hipStream_t s) {
    if (a_in.M == 0) return PFN_OK;
    GemmArgs a = a_in;
    a.ncb = (a.ldc + CB - 1) / CB;
    for (int t = 0; t < a.nterm; ++t) {
        if (a.term[t].lda % 4 != 0 || a.term[t].lda < ((a.term[t].K + 3) & ~3)) {
            set_error("gemm_nt: operand row stride %d must be a multiple of 4 and >= roundup(K=%d, 4)", a.term[t].lda,
                      a.term[t].K);
            return PFN_EINVAL;
        }
    }
    dim3 grid((a.M + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK, a.ncb * a.ngroup);
    double flops = 0.0, bytes = (double)a.ngroup * a.M * a.ncols * 4.0;
    for (int t = 0; t < a.nterm; ++t) {
        flops += 2.0 * a.M * a.term[t].K * a.ncols;
        bytes += (double)a.M * a.term[t].K * 4.0;
    }
    ProfScope ps("gemm_nt", bytes, flops, s);
    gemm_nt_kernel<<<grid, 256, 0, s>>>(a);
    PFN_CHECK_LAUNCH();
    return PFN_OK;
}

// ============================================================================================ TN
constexpr int TN_MB = 32;          // node rows per LDS stage
constexpr int TN_LD = CB + 4;      // 148
constexpr int TN_THREADS = 576;    // 9 waves: 3 x 3 macro tiles of 48 x 48
constexpr int TN_MAX_PAIRS = 8;
constexpr int TN_MAX_JOBS = 4;

constexpr int TN_MAX_BLOCKS = 64;   // output macro blocks (144 x 144) per launch
struct TnArgs {
    TnPair pair[TN_MAX_PAIRS];
    int npairs, M, rows_per_split, nsplit, nblocks;
    float* partial;   // [nsplit][nblocks][CB*CB]
    unsigned short blk_pair[TN_MAX_BLOCKS], blk_a0[TN_MAX_BLOCKS], blk_b0[TN_MAX_BLOCKS];
};

__global__ __launch_bounds__(TN_THREADS) void gemm_tn_kernel(const TnArgs a) {
    __shared__ __attribute__((aligned(16))) float ldsA[TN_MB * TN_LD];
    __shared__ __attribute__((aligned(16))) float ldsB[TN_MB * TN_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, c = lane >> 4;
    const int by = blockIdx.y;
    const TnPair pr = a.pair[a.blk_pair[by]];
    const int a0 = a.blk_a0[by], b0 = a.blk_b0[by];
    const int wo = wave / 3, wi = wave - 3 * wo;
    const int na_here = min(CB, pr.na - a0), nb_here = min(CB, pr.nb - b0);
    const bool wave_active = (48 * wo < na_here) && (48 * wi < nb_here);
    const int lda4 = (pr.na + 3) & ~3, ldb4 = (pr.nb + 3) & ~3;

    f32x4 acc[3][3];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int u = 0; u < 3; ++u) acc[t][u] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int m_beg = blockIdx.x * a.rows_per_split;
    const int m_end = min(a.M, m_beg + a.rows_per_split);
    for (int m0 = m_beg; m0 < m_end; m0 += TN_MB) {
        __syncthreads();   // previous stage fully consumed
        for (int i = tid; i < TN_MB * (CB / 4); i += TN_THREADS) {
            const int m = i / (CB / 4), q = i - m * (CB / 4);
            const int gm = m0 + m;
            float4 va = make_float4(0.f, 0.f, 0.f, 0.f), vb = va;
            if (gm < m_end) {
                if (a0 + 4 * q < lda4) va = *reinterpret_cast<const float4*>(pr.A + (size_t)gm * pr.lda + a0 + 4 * q);
                if (b0 + 4 * q < ldb4) vb = *reinterpret_cast<const float4*>(pr.B + (size_t)gm * pr.ldb + b0 + 4 * q);
            }
            *reinterpret_cast<float4*>(ldsA + m * TN_LD + 4 * q) = va;
            *reinterpret_cast<float4*>(ldsB + m * TN_LD + 4 * q) = vb;
        }
        __syncthreads();
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

// second stage: ordered sum over splits, scatter into the nn.Linear gradient layout
__global__ __launch_bounds__(256) void tn_reduce_kernel(const TnArgs a) {
    const int by = blockIdx.y;
    const TnPair pr = a.pair[a.blk_pair[by]];
    const int a0 = a.blk_a0[by], b0 = a.blk_b0[by];
    const int na_here = min(CB, pr.na - a0), nb_here = min(CB, pr.nb - b0);
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= na_here * nb_here) return;
    const int i = idx / nb_here, j = idx - i * nb_here;
    float acc = 0.f;
    for (int sp = 0; sp < a.nsplit; ++sp) acc += a.partial[((size_t)sp * a.nblocks + by) * (CB * CB) + i * CB + j];
    pr.G[(size_t)(pr.gn0 + a0 + i) * pr.ldg + pr.gk0 + b0 + j] = acc;
}

// column sums (bias gradients), optionally row-weighted: out[n] = sum_m rowscale[m] * A[m][n]
struct ColsumArgs {
    ColsumJob job[TN_MAX_JOBS];
    int njobs, M, rows_per_split, nsplit;
    float* partial;   // [njobs][nsplit][maxcols]
    int maxcols;
};
__global__ __launch_bounds__(256) void colsum_kernel(const ColsumArgs a) {
    __shared__ float red[4][64];
    const ColsumJob jb = a.job[blockIdx.y];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int m_beg = blockIdx.x * a.rows_per_split, m_end = min(a.M, m_beg + a.rows_per_split);
    for (int c0 = 0; c0 < jb.ncols; c0 += 64) {
        const int col = c0 + tx;
        float acc = 0.f;
        if (col < jb.ncols) {
            for (int m = m_beg + ty; m < m_end; m += 4) {
                const float v = jb.A[(size_t)m * jb.lda + col];
                acc += jb.rowscale ? jb.rowscale[m] * v : v;
            }
        }
        red[ty][tx] = acc;
        __syncthreads();
        if (ty == 0 && col < jb.ncols)
            a.partial[((size_t)blockIdx.y * a.nsplit + blockIdx.x) * a.maxcols + col] =
                (red[0][tx] + red[1][tx]) + (red[2][tx] + red[3][tx]);
        __syncthreads();
    }
}
__global__ __launch_bounds__(256) void colsum_reduce_kernel(const ColsumArgs a) {
    const ColsumJob jb = a.job[blockIdx.y];
    const int col = blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= jb.ncols) return;
    float acc = 0.f;
    for (int sp = 0; sp < a.nsplit; ++sp) acc += a.partial[((size_t)blockIdx.y * a.nsplit + sp) * a.maxcols + col];
    jb.out[col] = acc;
}

static int tn_nsplit(int64_t M) {
    int64_t s = (M + 4 * TN_MB - 1) / (4 * TN_MB);   // >= 128 node rows per split
    if (s < 1) s = 1;
    if (s > 64) s = 64;
    return (int)s;
}
constexpr int COLSUM_SPLIT = 128;

size_t reduce_ws_floats(int64_t M, int max_na, int max_nb, int max_pairs) {
    (void)max_pairs;
    const size_t tn = (size_t)tn_nsplit(M) * TN_MAX_BLOCKS * CB * CB;   // one launch's worth of partials
    const size_t cs = (size_t)TN_MAX_JOBS * COLSUM_SPLIT * (size_t)round_up(std::max(max_na, max_nb), 64);
    return tn + cs + 1024;
}

int launch_weight_grads(const TnPair* pairs, int npairs, const ColsumJob* jobs, int njobs, int64_t M, ReduceWs ws,
                        hipStream_t s) {
    if (njobs > TN_MAX_JOBS) {
        set_error("launch_weight_grads: too many column-sum jobs (%d)", njobs);
        return PFN_EINVAL;
    }
    const int nsplit0 = tn_nsplit(M);
    const int rows_per_split = (int)round_up((M + nsplit0 - 1) / nsplit0, TN_MB);
    const int nsplit = rows_per_split > 0 ? (int)std::max<int64_t>(1, (M + rows_per_split - 1) / rows_per_split) : 1;
    const size_t tn_floats = (size_t)nsplit * TN_MAX_BLOCKS * CB * CB;
    ColsumArgs ca;
    ca.njobs = njobs;
    ca.M = (int)M;
    ca.nsplit = (int)std::min<int64_t>(COLSUM_SPLIT, std::max<int64_t>(1, (M + 63) / 64));
    ca.rows_per_split = (int)((M + ca.nsplit - 1) / ca.nsplit);
    ca.maxcols = 0;
    for (int j = 0; j < njobs; ++j) {
        ca.job[j] = jobs[j];
        ca.maxcols = std::max(ca.maxcols, jobs[j].ncols);
    }
    ca.partial = ws.partial + tn_floats;
    const size_t need = tn_floats + (size_t)njobs * ca.nsplit * ca.maxcols;
    if (need > ws.floats) {
        set_error("launch_weight_grads: reduction workspace %zu < %zu floats", ws.floats, need);
        return PFN_ENOSPACE;
    }
    // pairs are batched so that one launch covers at most TN_MAX_BLOCKS output macro blocks / TN_MAX_PAIRS pairs
    int p = 0;
    while (p < npairs) {
        TnArgs ta;
        ta.npairs = 0;
        ta.nblocks = 0;
        ta.M = (int)M;
        ta.rows_per_split = rows_per_split;
        ta.nsplit = nsplit;
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
        if (ta.nblocks > 0) {
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
            ProfScope ps("tn_reduce", 0.0, 0.0, s);
            tn_reduce_kernel<<<dim3((CB * CB + 255) / 256, ta.nblocks), 256, 0, s>>>(ta);
            PFN_CHECK_LAUNCH();
        }
    }
    if (njobs > 0 && ca.maxcols > 0) {
        ProfScope ps("colsum", 0.0, 0.0, s);
        if (M > 0) {
            colsum_kernel<<<dim3(ca.nsplit, njobs), 256, 0, s>>>(ca);
            PFN_CHECK_LAUNCH();
        } else {
            ca.nsplit = 0;
        }
        colsum_reduce_kernel<<<dim3((ca.maxcols + 255) / 256, njobs), 256, 0, s>>>(ca);
        PFN_CHECK_LAUNCH();
    }
    return PFN_OK;
}

}  // namespace pfn
