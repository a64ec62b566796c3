// fp32 MFMA weight-gradient GEMM (dW = dY^T X) of the hot path (gfx950).  The forward / input-gradient GEMM lives in gemm_nt.hip.
//
// Carries what the reference's autograd runs as one `mm` per nn.Linear weight (EdgeAggregation.edge_aggr
// networks/MPN.py:17-21, TAGConv.lins, mask_embd :491-495), restructured to per-NODE operands:
//   dW[i][j] = sum_m A[m][i] * B[m][j]     A = gradient of the layer output (M x na), B = layer input (M x nb),
// M = nodes (1e4..1e6) is the REDUCTION dimension, na, nb <= a few hundred.  Exact fp32 on v_mfma_f32_32x32x2_f32.
//
// Both operands are row-major with the reduction index as the row, which is exactly the MFMA operand order: at step s
// lane (c = lane & 31, kh = lane >> 5) supplies row m + 2s + kh.  A lane loads TWO adjacent columns (8 bytes: columns
// 2c, 2c+1 of a 64-column quadrant), so a wave load covers 2 rows x 256 contiguous bytes and the wave owns a 64 x 64
// output quadrant as 2 x 2 interleaved 32 x 32 accumulator tiles (tile (sa, sb) = rows 2i + sa, columns 2j + sb).
// No LDS staging, no barriers in the loop: every wave streams its own row range straight from global memory
// (hand double-buffered batches of 16 rows), 4 MFMAs per 2 loads.  H = 129 = 2*64 + 1: the odd row / column and the
// bias gradient (a virtual ones- or rowscale-column of B) never get a tile -- they are VALU dot products off the
// fragments the wave already holds.
//
// Work split: a block's 8 waves = the (up to 4) quadrants of ONE pair x 2..8 consecutive row ranges, so the quadrants
// share every operand row through the CU's L1 (one quadrant per block re-read each row 2x from L2/HBM: measured 2.6x
// slower); the ranges are summed by a fixed LDS tree; when a group is split over several blocks they publish partials
// and tn_combine_kernel sums them in block order (deterministic, no float atomics) into the nn.Linear gradient layout.
// (A last-arriver combine inside the kernel was tried: one CU pulling ~100 partials of 73 KB serialises -- 270 us.)
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <type_traits>

#include "pfn_internal.hpp"

namespace pfn {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int TN_THREADS = 512;
constexpr int TN_WAVES = TN_THREADS / 64;
constexpr int TN_U = 8;                    // MFMA steps (row pairs) per batch: 16 rows
constexpr int TN_QUADS = 18;               // float4 per lane of a partial: 16 (four 32x32 tiles) + 2 (VALU extras)
constexpr int TN_PART = TN_QUADS * 64 * 4; // floats per partial
constexpr int TN_MAX_PAIRS = 8;
constexpr int TN_MAX_TASKS = 64;
constexpr int TN_MAX_PARTIALS = 2048;
enum { TNF_XCOL = 1, TNF_BIAS = 2, TNF_XROW = 4 };

struct TnTask { short pair, qi, qj, flags; };
struct TnGroup { TnTask task[4]; int ng, gshift; };   // ng = 1, 2 or 4 quadrants of one pair; gshift = log2(ng)
struct TnArgs {
    TnPair pair[TN_MAX_PAIRS];
    TnGroup group[TN_MAX_TASKS / 2];
    int M, rows_per_wave, nblk_x, ntasks;   // ntasks = number of GROUPS
    float* partial;   // [ngroups][nblk_x][4][TN_QUADS][64] float4
};

struct TnBatch {
    f32x2 a[TN_U], b[TN_U];
    // side operands of the VALU extras, ONE value per lane for the whole batch: lane l holds row (l & 15) of the batch; step s
    // fetches row 2s + kh through the LDS crossbar (ds_bpermute).  As three broadcast loads per step they made the kernel
    // VMEM-issue-bound: 5 vector-memory instructions per 4 MFMAs, 8 waves per CU.
    float xv16, rs16, yv16;
};

// Scatter one float4 quad of a quadrant partial into the nn.Linear gradient layout.  Quad Q < 16 = accumulator registers
// 4(Q&3)..+3 of tile (sa = Q >> 3, sb = (Q >> 2) & 1): element e is row 64 qi + 2 (e + 8 (Q&3) + 4 kh) + sa, column
// 64 qj + 2 c + sb.  Q = 16: {odd column[2c], [2c+1], bias[2c], [2c+1]};  Q = 17: {odd row[2c], [2c+1], corner, corner bias}.
__device__ __forceinline__ void tn_emit(const TnPair& pr, const TnTask tk, int Q, float4 s, int c, int kh, int lane) {
    const bool f_xcol = tk.flags & TNF_XCOL, f_bias = tk.flags & TNF_BIAS, f_xrow = tk.flags & TNF_XROW;
    const int na_main = pr.na - ((pr.na % 64 == 1 && pr.na > 1) ? 1 : 0), nb_main = pr.nb - ((pr.nb % 64 == 1 && pr.nb > 1) ? 1 : 0);
    const float e4[4] = {s.x, s.y, s.z, s.w};
    if (Q < 16) {
        const int sa = Q >> 3, sb = (Q >> 2) & 1;
        const int j = 64 * tk.qj + 2 * c + sb;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int i = 64 * tk.qi + 2 * (e + 8 * (Q & 3) + 4 * kh) + sa;
            if (i < na_main && j < nb_main) pr.G[(size_t)(pr.gn0 + i) * pr.ldg + pr.gk0 + j] = e4[e];
        }
    } else if (Q == 16) {
        if (kh == 0) {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int i = 64 * tk.qi + 2 * c + e;
                if (i < na_main) {
                    if (f_xcol) pr.G[(size_t)(pr.gn0 + i) * pr.ldg + pr.gk0 + pr.nb - 1] = e4[e];
                    if (f_bias) pr.bias_out[i] = e4[2 + e];
                }
            }
        }
    } else if (kh == 0 && f_xrow) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int j = 64 * tk.qj + 2 * c + e;
            if (j < nb_main) pr.G[(size_t)(pr.gn0 + pr.na - 1) * pr.ldg + pr.gk0 + j] = e4[e];
        }
        if (lane == 0) {
            if (f_xcol) pr.G[(size_t)(pr.gn0 + pr.na - 1) * pr.ldg + pr.gk0 + pr.nb - 1] = e4[2];
            if (f_bias) pr.bias_out[pr.na - 1] = e4[3];
        }
    }
}

__global__ __launch_bounds__(TN_THREADS, 1) void gemm_tn_kernel(const TnArgs a) {
    __shared__ __attribute__((aligned(16))) float4 red[4][TN_QUADS][64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 31, kh = lane >> 5;
    // 1-D grid, XCD-aware: workgroups are dealt round-robin to the 8 XCDs (each with its own L2), and all tasks of one row
    // range read the same rows (a pair's quadrants; the pairs of a TAGConv share A, dP/dQ share B) -> they get linear ids
    // that agree mod 8:  id = (bx / 8) * 8 * ntasks + task * 8 + bx % 8
    const int sup = blockIdx.x / (8 * a.ntasks), rem = blockIdx.x - sup * 8 * a.ntasks;
    const int by = rem >> 3, bx = sup * 8 + (rem & 7);
    if (bx >= a.nblk_x) return;
    const int ng = a.group[by].ng, gshift = a.group[by].gshift;
    const int wq = wave & (ng - 1), wr = wave >> gshift, nr = TN_WAVES >> gshift;   // quadrant / row range of this wave
    const TnTask tk = a.group[by].task[wq];
    const TnPair pr = a.pair[tk.pair];
    const bool f_xcol = tk.flags & TNF_XCOL, f_bias = tk.flags & TNF_BIAS, f_xrow = tk.flags & TNF_XROW;
    const bool f_any = tk.flags != 0;
    // columns of this lane: clamped into the row so the 8-byte read is always legal; columns past na / nb produce
    // outputs that are never stored
    const int acol = min(64 * tk.qi + 2 * c, pr.lda - 2), bcol = min(64 * tk.qj + 2 * c, pr.ldb - 2);
    const float* Ap = pr.A + acol;
    const float* Bp = pr.B + bcol;
    const float* Xc = pr.B + (pr.nb - 1);     // the odd column of B (f_xcol)
    const float* Yr = pr.A + (pr.na - 1);     // the odd column of A = odd row of dW (f_xrow)
    const float* Rs = pr.bias_rowscale;
    // The block owns ONE contiguous row range; its nr row-waves take the 32-row chunks of it round-robin, so the block
    // reads a single stream per operand (all 8 waves touch the same DRAM pages / L1 lines at about the same time).
    const int R0 = min(a.M, bx * nr * a.rows_per_wave), R1 = min(a.M, R0 + nr * a.rows_per_wave);
    const int nchunks = (R1 - R0) >> 5;
    const int mlast = a.M - 1;

    f32x16 acc[2][2];
#pragma unroll
    for (int sa = 0; sa < 2; ++sa)
#pragma unroll
        for (int sb = 0; sb < 2; ++sb)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[sa][sb][q] = 0.f;
    float xc[2] = {0.f, 0.f}, bs[2] = {0.f, 0.f}, xr[2] = {0.f, 0.f}, cn = 0.f, cb = 0.f;

    // The streaming loop exists in two instantiations chosen ONCE per wave -- with and without the VALU extras: as
    // runtime flags inside load()/compute() they put a scalar branch in front of every load.  With extras all three side
    // operands are loaded unconditionally from always-valid addresses (results a task does not own are never emitted).
    const float* Rs2 = Rs ? Rs : pr.A;                 // no rowscale: any valid address, the value is replaced by 1
    const size_t rs_stride = Rs ? 1 : (size_t)pr.lda;
    const bool has_rs = Rs != nullptr;
    auto stream = [&](auto ex_c) {
        constexpr bool EX = decltype(ex_c)::value;
        // Addresses = wave-uniform 64-bit row base (scalar ALU) + a per-lane 32-bit offset that never changes: no vector
        // address math in the loop (with per-load 64-bit multiply-adds the kernel was VALU-bound at 31 % of the MFMA
        // peak even when every load hit L1).  A batch that would run past the matrix is moved back as a whole (uniform
        // clamp); the only caller that can see moved rows is the prefetch of a batch that is never consumed.
        const uint32_t voA = (uint32_t)(kh * pr.lda) * 4u, voB = (uint32_t)(kh * pr.ldb) * 4u;
        const int l16 = lane & 15;
        const uint32_t voX16 = (uint32_t)(l16 * pr.ldb) * 4u, voY16 = (uint32_t)(l16 * pr.lda) * 4u, voR16 = (uint32_t)(l16 * (int)rs_stride) * 4u;
        auto load = [&](TnBatch& t, int m0) {   // rows m0 .. m0+15
            const int mu = max(0, min(m0, a.M - 2 * TN_U));
            const char* rowA = reinterpret_cast<const char*>(Ap) + (size_t)mu * pr.lda * 4;
            const char* rowB = reinterpret_cast<const char*>(Bp) + (size_t)mu * pr.ldb * 4;
            const char* rowX = reinterpret_cast<const char*>(Xc) + (size_t)mu * pr.ldb * 4;
            const char* rowY = reinterpret_cast<const char*>(Yr) + (size_t)mu * pr.lda * 4;
            const char* rowR = reinterpret_cast<const char*>(Rs2) + (size_t)mu * rs_stride * 4;
#pragma unroll
            for (int s = 0; s < TN_U; ++s) {
                t.a[s] = *reinterpret_cast<const f32x2*>(rowA + (size_t)(2 * s) * pr.lda * 4 + voA);
                t.b[s] = *reinterpret_cast<const f32x2*>(rowB + (size_t)(2 * s) * pr.ldb * 4 + voB);
            }
            if (EX) {
                t.xv16 = *reinterpret_cast<const float*>(rowX + voX16);
                t.yv16 = *reinterpret_cast<const float*>(rowY + voY16);
                const float r = *reinterpret_cast<const float*>(rowR + voR16);
                t.rs16 = has_rs ? r : 1.f;
            }
        };
        auto load_tail = [&](TnBatch& t, int m0) {   // per-lane clamped rows (ragged tail only)
#pragma unroll
            for (int s = 0; s < TN_U; ++s) {
                const int row = min(m0 + 2 * s + kh, mlast);
                t.a[s] = *reinterpret_cast<const f32x2*>(Ap + (size_t)row * pr.lda);
                t.b[s] = *reinterpret_cast<const f32x2*>(Bp + (size_t)row * pr.ldb);
            }
            if (EX) {
                const int row = min(m0 + l16, mlast);
                t.xv16 = Xc[(size_t)row * pr.ldb];
                t.yv16 = Yr[(size_t)row * pr.lda];
                const float r = Rs2[(size_t)row * rs_stride];
                t.rs16 = has_rs ? r : 1.f;
            }
        };
        auto compute = [&](const TnBatch& t) {
#pragma unroll
            for (int s = 0; s < TN_U; ++s) {
#pragma unroll
                for (int sa = 0; sa < 2; ++sa)
#pragma unroll
                    for (int sb = 0; sb < 2; ++sb)
                        acc[sa][sb] = __builtin_amdgcn_mfma_f32_32x32x2f32(t.a[s][sa], t.b[s][sb], acc[sa][sb], 0, 0, 0);
                if (EX) {
                    const float xv = __shfl(t.xv16, 2 * s + kh), rs = __shfl(t.rs16, 2 * s + kh), yv = __shfl(t.yv16, 2 * s + kh);
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        xc[e] = fmaf(t.a[s][e], xv, xc[e]);
                        bs[e] = fmaf(t.a[s][e], rs, bs[e]);
                        xr[e] = fmaf(yv, t.b[s][e], xr[e]);
                    }
                    cn = fmaf(yv, xv, cn);
                    cb = fmaf(yv, rs, cb);
                }
            }
        };
        // ---- full 32-row double batches: the next batch's loads fly under the current batch's MFMAs (two register
        // sets, swapped by unrolling -- no copies)
        TnBatch t0, t1;
        int j = wr;
        if (j < nchunks) load(t0, R0 + 32 * j);
        for (; j < nchunks; j += nr) {
            const int m = R0 + 32 * j;
            load(t1, m + 16);
            compute(t0);
            load(t0, m + 32 * nr);   // this wave's next chunk (past the range on the last round: clamped rows, never used)
            compute(t1);
        }
        // ---- ragged tail of the block's range (< 32 rows), taken by the wave whose turn it is: rows past the end
        // contribute zeros through B and the side operands
        if (wr == nchunks % nr) {
            for (int m = R0 + 32 * nchunks; m < R1; m += 16) {
                load_tail(t0, m);
#pragma unroll
                for (int s = 0; s < TN_U; ++s) {
                    const bool ok = m + 2 * s + kh < R1;
                    t0.b[s][0] = ok ? t0.b[s][0] : 0.f;
                    t0.b[s][1] = ok ? t0.b[s][1] : 0.f;
                }
                if (EX) {
                    const bool ok16 = m + l16 < R1;
                    t0.xv16 = ok16 ? t0.xv16 : 0.f;
                    t0.rs16 = ok16 ? t0.rs16 : 0.f;
                    t0.yv16 = ok16 ? t0.yv16 : 0.f;
                }
                compute(t0);
            }
        }
    };
    if (f_any) stream(std::true_type{});
    else stream(std::false_type{});
    // the two k halves of the VALU extras
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        xc[e] += __shfl_xor(xc[e], 32);
        bs[e] += __shfl_xor(bs[e], 32);
        xr[e] += __shfl_xor(xr[e], 32);
    }
    cn += __shfl_xor(cn, 32);
    cb += __shfl_xor(cb, 32);

    // ---- everything a lane holds, as 18 float4: quad Q < 16 = registers 4(Q&3) .. +3 of tile (sa = Q >> 3, sb = (Q >> 2) & 1)
    float4 v[TN_QUADS];
#pragma unroll
    for (int Q = 0; Q < 16; ++Q) {
        const f32x16& t = acc[Q >> 3][(Q >> 2) & 1];
        v[Q] = make_float4(t[4 * (Q & 3)], t[4 * (Q & 3) + 1], t[4 * (Q & 3) + 2], t[4 * (Q & 3) + 3]);
    }
    v[16] = make_float4(xc[0], xc[1], bs[0], bs[1]);
    v[17] = make_float4(xr[0], xr[1], cn, cb);
    // ---- fixed tree over the block's row ranges (per quadrant), e.g. nr = 8: ((0+4)+(2+6)) + ((1+5)+(3+7))
    for (int stride = nr >> 1; stride >= 1; stride >>= 1) {
        if (wr >= stride && wr < 2 * stride) {
#pragma unroll
            for (int Q = 0; Q < TN_QUADS; ++Q) red[((wr - stride) << gshift) + wq][Q][lane] = v[Q];
        }
        __syncthreads();
        if (wr < stride) {
#pragma unroll
            for (int Q = 0; Q < TN_QUADS; ++Q) {
                const float4 o = red[(wr << gshift) + wq][Q][lane];
                v[Q].x += o.x; v[Q].y += o.y; v[Q].z += o.z; v[Q].w += o.w;
            }
        }
        __syncthreads();
    }
    // ---- one block: write the gradient; several: publish the partial for tn_combine_kernel
    if (wr != 0) return;
    if (a.nblk_x == 1) {
#pragma unroll
        for (int Q = 0; Q < TN_QUADS; ++Q) tn_emit(pr, tk, Q, v[Q], c, kh, lane);
        return;
    }
    float4* mine = reinterpret_cast<float4*>(a.partial) + (((size_t)by * a.nblk_x + bx) * 4 + wq) * (TN_QUADS * 64);
#pragma unroll
    for (int Q = 0; Q < TN_QUADS; ++Q) mine[Q * 64 + lane] = v[Q];
}

// Second stage (only when a group was split over several blocks): block = one float4 quad Q of one quadrant, 64 lanes x 8
// slices of the block list; every slice sums its blocks in block order, then a fixed tree over the slices -> deterministic.
__global__ __launch_bounds__(512) void tn_combine_kernel(const TnArgs a) {
    __shared__ float4 red[8][64];
    const int lane = threadIdx.x & 63, slice = threadIdx.x >> 6;
    const int Q = blockIdx.x, by = blockIdx.y >> 2, wq = blockIdx.y & 3;
    if (wq >= a.group[by].ng) return;
    const TnTask tk = a.group[by].task[wq];
    const TnPair pr = a.pair[tk.pair];
    const size_t qstride = (size_t)TN_QUADS * 64;
    const float4* src = reinterpret_cast<const float4*>(a.partial) + ((size_t)by * a.nblk_x * 4 + wq) * qstride + Q * 64 + lane;
    float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0;
    int b = slice;
    for (; b + 8 < a.nblk_x; b += 16) {
        const float4 p0 = src[(size_t)b * 4 * qstride], p1 = src[(size_t)(b + 8) * 4 * qstride];
        s0.x += p0.x; s0.y += p0.y; s0.z += p0.z; s0.w += p0.w;
        s1.x += p1.x; s1.y += p1.y; s1.z += p1.z; s1.w += p1.w;
    }
    if (b < a.nblk_x) {
        const float4 p0 = src[(size_t)b * 4 * qstride];
        s0.x += p0.x; s0.y += p0.y; s0.z += p0.z; s0.w += p0.w;
    }
    red[slice][lane] = make_float4(s0.x + s1.x, s0.y + s1.y, s0.z + s1.z, s0.w + s1.w);
    __syncthreads();
    for (int off = 4; off >= 1; off >>= 1) {
        if (slice < off) {
            const float4 o = red[slice + off][lane];
            float4& m = red[slice][lane];
            m.x += o.x; m.y += o.y; m.z += o.z; m.w += o.w;
        }
        __syncthreads();
    }
    if (slice == 0) tn_emit(pr, tk, Q, red[0][lane], lane & 31, lane >> 5, lane);
}

size_t reduce_ws_floats(int64_t M, int max_na, int max_nb, int max_pairs) {
    (void)M; (void)max_na; (void)max_nb; (void)max_pairs;
    return (size_t)TN_MAX_PARTIALS * TN_PART + 256;   // partials + tickets
}

int launch_weight_grads(const TnPair* pairs, int npairs, int64_t M, ReduceWs ws, hipStream_t s) {
    if (ws.floats < (size_t)TN_PART + 256) {
        set_error("launch_weight_grads: reduction workspace too small");
        return PFN_ENOSPACE;
    }
    const size_t max_partials = (ws.floats - 256) / TN_PART;
    int p = 0;
    while (p < npairs) {
        TnArgs ta;
        memset(&ta, 0, sizeof(ta));
        ta.M = (int)M;
        ta.partial = ws.partial;
        int np_here = 0;
        while (p < npairs && np_here < TN_MAX_PAIRS) {
            const TnPair& pr = pairs[p];
            if (pr.lda % 4 || pr.ldb % 4 || pr.lda < 2 || pr.ldb < 2) {
                set_error("launch_weight_grads: row strides must be multiples of 4");
                return PFN_EINVAL;
            }
            const bool xrow = pr.na % 64 == 1 && pr.na > 1, xcol = pr.nb % 64 == 1 && pr.nb > 1;
            const int QA = std::max(1, (pr.na - (xrow ? 1 : 0) + 63) / 64), QB = std::max(1, (pr.nb - (xcol ? 1 : 0) + 63) / 64);
            if (QA * QB > TN_MAX_TASKS) {
                set_error("launch_weight_grads: a %d x %d weight needs %d quadrant tasks (> %d)", pr.na, pr.nb, QA * QB, TN_MAX_TASKS);
                return PFN_EINVAL;
            }
            const int nq = QA * QB, ngr = (nq + 3) / 4 + ((nq % 4) == 3 ? 1 : 0);   // groups of 4, then 2, then 1
            if (ta.ntasks + ngr > TN_MAX_TASKS / 2) break;
            TnTask all[TN_MAX_TASKS];
            int n_all = 0;
            for (int qi = 0; qi < QA; ++qi)
                for (int qj = 0; qj < QB; ++qj) {
                    TnTask& t = all[n_all++];
                    t.pair = (short)np_here;
                    t.qi = (short)qi;
                    t.qj = (short)qj;
                    t.flags = (short)(((xcol && qj == QB - 1) ? TNF_XCOL : 0) | ((pr.bias_out && qj == 0) ? TNF_BIAS : 0) |
                                      ((xrow && qi == QA - 1) ? TNF_XROW : 0));
                }
            for (int t0 = 0; t0 < n_all;) {
                const int left = n_all - t0, ng = left >= 4 ? 4 : (left >= 2 ? 2 : 1);
                TnGroup& gr = ta.group[ta.ntasks++];
                gr.ng = ng;
                gr.gshift = ng == 4 ? 2 : (ng == 2 ? 1 : 0);
                for (int q = 0; q < ng; ++q) gr.task[q] = all[t0 + q];
                t0 += ng;
            }
            ta.pair[np_here++] = pr;
            ++p;
        }
        if (ta.ntasks == 0) continue;
        // ~512 blocks per launch, at least 32 rows per wave; every wave range is a multiple of 32 rows.  A block covers
        // at least 2 ranges (4 quadrants) -- the row split below assumes the smallest; blocks with more ranges just get
        // further into M and the ones past the end idle.
        static const int want_blocks = getenv("PFN_TN_BLOCKS") ? atoi(getenv("PFN_TN_BLOCKS")) : 256;   // tuning aid
        int nbx = std::max(1, std::min(128, (want_blocks + ta.ntasks - 1) / ta.ntasks));
        nbx = (int)std::min<size_t>(nbx, std::max<size_t>(1, std::min<size_t>(max_partials, TN_MAX_PARTIALS) / (4 * ta.ntasks)));
        int min_nr = TN_WAVES;
        for (int g2 = 0; g2 < ta.ntasks; ++g2) min_nr = std::min(min_nr, TN_WAVES / ta.group[g2].ng);
        const int64_t rpw = std::max<int64_t>(32, round_up((M + (int64_t)nbx * min_nr - 1) / ((int64_t)nbx * min_nr), 32));
        ta.rows_per_wave = (int)rpw;
        ta.nblk_x = (int)std::max<int64_t>(1, (M + rpw * min_nr - 1) / (rpw * min_nr));
        double flops = 0.0, bytes = 0.0;
        for (int q = 0; q < np_here; ++q) {
            flops += 2.0 * (double)M * ta.pair[q].na * ta.pair[q].nb;
            bytes += 4.0 * (double)M * (ta.pair[q].na + ta.pair[q].nb);
        }
        {
            ProfScope ps("gemm_tn", bytes, flops, s);
            gemm_tn_kernel<<<(ta.nblk_x + 7) / 8 * 8 * ta.ntasks, TN_THREADS, 0, s>>>(ta);
            PFN_CHECK_LAUNCH();
        }
        if (ta.nblk_x > 1) {
            ProfScope ps("tn_reduce", 0.0, 0.0, s);
            tn_combine_kernel<<<dim3(TN_QUADS, 4 * ta.ntasks), 512, 0, s>>>(ta);
            PFN_CHECK_LAUNCH();
        }
    }
    return PFN_OK;
}

}  // namespace pfn
