// EdgeAggregation for batches of SMALL graphs (the metric configuration: case118 x 128), graph-resident in LDS.
//
// A batch of disjoint graphs of `seg` nodes each (pfn_graph_segments) never sends a message across a graph boundary, and the
// edge walk acts on every hidden column independently.  So a workgroup that owns the rows of whole graphs AND one 32-column
// quarter of the hidden width can run the node GEMM that produces its P | Q columns, keep them in LDS, and walk the edges
// right there -- what the generic path does as two or three launches with an N x H round trip through HBM between them
// (networks/MPN.py:17-21,:28: per-edge Linear -> ReLU -> Linear -> sum, restructured per node as in edge.hip):
//
//   forward   P | Q quarter = x W1i^T + b1 | x W1j^T   (MFMA, one 32 x 32 tile per wave)      \  ea_seg_fwd_kernel
//             S[i] = sum_{e -> i} relu(P[i] + Q[src e] + a_e We)                (LDS walk)     /  (was gemm_nt + edge_fwd)
//   backward  dS quarter = gout W2                         (MFMA; last layer: from the 16-byte gout rows, VALU)   \
//             dP[i] = sum_{e -> i} dh_e, dWe partials ; dQ[j] = sum_{e: src = j} dh_e          (LDS walks)          /  ea_seg_bwd_kernel
//                                                                                               (was gemm_nt + edge_bwd)
// P, Q, S, dP, dQ are still written to HBM: the backward pass and the weight gradients read them.  The MFMA tiles use the
// same packed weight images and the same k order as gemm_nt and the walks the same edge order as edge.hip, so the results
// are bit-identical to the generic path except for the order of the dWe partial sums.
// With 128 graphs x 4 quarters = 512 workgroups (two per CU, 4 waves per SIMD) a launch is ~one MFMA tile and one short
// LDS walk per wave deep.  Graphs whose rows do not fit (ea_seg_fit) take the generic kernels.
#include <stdlib.h>

#include <algorithm>

#include "pfn_internal.hpp"
#include "seg_tile.hpp"

namespace pfn {

struct SegCsr {      // one adjacency slice in LDS: rp[rows + 1] (relative), nb[cap] (row index inside the block), ea[cap] (a_e)
    int* rp;
    int* nb;
    float2* ea;
    bool in_lds;
    int e0, ne;
};

// The block's slice of one CSR (by destination or by source) goes to LDS once; `ea_slot` = the edge attributes already in slot
// order (SlotEa, gathered once per forward), so the chain is two loads deep: rowptr, then indices | attributes.
// Issue / commit are split so that EVERY global load of a kernel's prologue is requested before its first LDS store: written
// as load-store loops, each store waited for all loads before it (VMEM returns in order) and the next loop's loads went out
// only then -- eight serial round trips, 8.4 of ea_seg_bwd's 25 us (phase timestamps, tools/ubench/run_ea_seg_ts.sh).
// rows + 1 <= 129 and cap <= 512 = SG_THREADS: one row pointer and one slot per thread.
struct CsrRegs { int rp, nb; float2 ea; };
__device__ __forceinline__ void csr_issue1(SegCsr& c, CsrRegs& r, int r0, int rows, const int* __restrict__ rowptr) {
    c.e0 = rowptr[r0];
    c.ne = rowptr[r0 + rows] - c.e0;
    r.rp = (int)threadIdx.x <= rows ? rowptr[r0 + threadIdx.x] : 0;
}
__device__ __forceinline__ void csr_issue2(SegCsr& c, CsrRegs& r, int cap, const int* __restrict__ nbr,
                                           const float* __restrict__ ea_slot) {
    c.in_lds = c.ne <= cap;
    r.nb = 0;
    r.ea = make_float2(0.f, 0.f);
    if (c.in_lds && (int)threadIdx.x < c.ne) {
        r.nb = nbr[c.e0 + threadIdx.x];
        r.ea = reinterpret_cast<const float2*>(ea_slot)[c.e0 + threadIdx.x];
    }
}
__device__ __forceinline__ void csr_commit(const SegCsr& c, const CsrRegs& r, int r0, int rows) {
    if ((int)threadIdx.x <= rows) c.rp[threadIdx.x] = r.rp - c.e0;
    if (c.in_lds && (int)threadIdx.x < c.ne) {
        c.nb[threadIdx.x] = r.nb - r0;
        c.ea[threadIdx.x] = r.ea;
    }
}
__device__ __forceinline__ void csr_slot(const SegCsr& c, int p, int r0, const int* __restrict__ nbr, const float* __restrict__ ea_slot,
                                         int& ls, float2& a2) {
    if (c.in_lds) {
        ls = c.nb[p];
        a2 = c.ea[p];
    } else {   // a block with more edges than the LDS slice holds: indices from global memory
        ls = nbr[c.e0 + p] - r0;
        a2 = reinterpret_cast<const float2*>(ea_slot)[c.e0 + p];
    }
}

// The up to 4 trailing columns (H = 129 = 4 * 32 + 1) never get an MFMA tile: the LAST quarter's block forms them as VALU dot
// products, thread t = (row t >> PL2, k part t & (2^PL2 - 1)); a part adds its k groups in order (NB groups per batch, all loads
// of a batch requested before the first multiply), the parts are added by a fixed xor tree.  Two images (P and Q) share the row
// loads.  v1[c] = sum_k A[row][k] * image1_rem[k][c], v2 likewise; seg_rem_store drops them into tile columns 32.. (+ bias).
// Forward: all 512 threads (4 parts per row) while the MFMA operands are in flight; backward: the four waves without a tile
// (2 parts per row) while the other four multiply.
template <bool TWO, int PL2, int NB>
__device__ __forceinline__ void seg_rem_dots(int t, const float* __restrict__ A, int lda, int K, int r0, int rows,
                                             const float* __restrict__ Bp1, const float* __restrict__ Bp2, int nq, int nreal,
                                             float (&v1)[4], float (&v2)[4]) {
    constexpr int PARTS = 1 << PL2;
    const int lr = t >> PL2, part = t & (PARTS - 1);
    const int G = ((K + 7) & ~7) >> 2;
    const float* arow = A + (size_t)(r0 + min(lr, rows - 1)) * lda;
    const size_t roff = (size_t)nq * G * 128;
    const int gmax = (lda >> 2) - 1;                                 // k groups past the row multiply zero image rows
#pragma unroll
    for (int c = 0; c < 4; ++c) v1[c] = v2[c] = 0.f;
    if (nreal == 1) {   // H = 128 + 1
        for (int g0 = part; g0 < G; g0 += NB * PARTS) {
            float4 xa[NB], w1[NB], w2[TWO ? NB : 1];
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const int g = min(g0 + PARTS * j, G - 1);
                const float4 v = sg_ld4(arow + 4 * min(g, gmax));
                xa[j] = g0 + PARTS * j < G ? v : make_float4(0.f, 0.f, 0.f, 0.f);   // (past G: zeroed by a select, no divergent branch)
                w1[j] = sg_ld4(Bp1 + roff + (size_t)g * 16);
                if (TWO) w2[j] = sg_ld4(Bp2 + roff + (size_t)g * 16);
            }
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                v1[0] = fmaf(xa[j].w, w1[j].w, fmaf(xa[j].z, w1[j].z, fmaf(xa[j].y, w1[j].y, fmaf(xa[j].x, w1[j].x, v1[0]))));
                if (TWO) v2[0] = fmaf(xa[j].w, w2[j].w, fmaf(xa[j].z, w2[j].z, fmaf(xa[j].y, w2[j].y, fmaf(xa[j].x, w2[j].x, v2[0]))));
            }
        }
    } else {
        for (int g = part; g < G; g += PARTS) {
            const float4 xa = sg_ld4(arow + 4 * min(g, gmax));
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (c < nreal) {
                    const float4 w = sg_ld4(Bp1 + roff + (size_t)g * 16 + c * 4);
                    v1[c] = fmaf(xa.w, w.w, fmaf(xa.z, w.z, fmaf(xa.y, w.y, fmaf(xa.x, w.x, v1[c]))));
                    if (TWO) {
                        const float4 w2 = sg_ld4(Bp2 + roff + (size_t)g * 16 + c * 4);
                        v2[c] = fmaf(xa.w, w2.w, fmaf(xa.z, w2.z, fmaf(xa.y, w2.y, fmaf(xa.x, w2.x, v2[c]))));
                    }
                }
            }
        }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
#pragma unroll
        for (int off = 1; off < PARTS; off <<= 1) {
            v1[c] += __shfl_xor(v1[c], off);
            if (TWO) v2[c] += __shfl_xor(v2[c], off);
        }
    }
}
template <bool TWO, int PL2>
__device__ __forceinline__ void seg_rem_store(int t, int rows, int nq, int nreal, const float* __restrict__ bias,
                                              const float (&v1)[4], const float (&v2)[4], float* tile1, float* tile2) {
    const int lr = t >> PL2, part = t & ((1 << PL2) - 1);
    if (part == 0 && lr < rows) {   // (columns past the real ones: zero, like every pad column)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            tile1[(size_t)lr * SG_TW + 32 + c] = c < nreal ? v1[c] + (bias ? bias[32 * nq + c] : 0.f) : 0.f;
            if (TWO) tile2[(size_t)lr * SG_TW + 32 + c] = c < nreal ? v2[c] : 0.f;
        }
    }
}

// residue columns of W1 for the slice: we[f][tile column] = W1[col][2Fi + f] (zero past H); 2 * SG_TW = 72 values, one per thread
__device__ __forceinline__ float we_issue(const SegCols& c, const float* __restrict__ w1, int h, int fi) {
    const int i = threadIdx.x, ldw = 2 * fi + 2;
    if (i >= 2 * SG_TW) return 0.f;
    const int f = i / SG_TW, t = i - f * SG_TW;
    const int col = seg_col_of_tile(c, t);
    return (col >= 0 && col < h) ? w1[(size_t)col * ldw + 2 * fi + f] : 0.f;
}
__device__ __forceinline__ void we_commit(float* s_we, float v) {
    if (threadIdx.x < 2 * SG_TW) s_we[threadIdx.x] = v;
}

// LDS carve-up (floats): tiles first (16-byte aligned), then the CSR slices
constexpr int SG_W2A_CH = 34;        // float4 chunks per W2 row kept for the MSELoss tail (ld <= 8 * SG_NCH = 136 floats)
struct SegLds {
    float* P;
    float* Q;
    float* D;      // backward only: dS
    float* we;     // [2][SG_TW]
    float* w2s;    // backward, last layer only: W2 slice [4][SG_TW]
    float4* w2a;   // backward, last layer with the MSELoss tail: all of W2 as [4][SG_W2A_CH] float4 chunks
    float* B0;     // forward: the quarter of W1i^T | W1j^T (2 x 34 x 128 floats); backward: the quarter of W2 -- ALIASES the dS tile
    float* B1;
    float4* part;  // backward only: dWe partials [8 waves][16 chunk lanes][2]
    SegCsr in, out;
};
__device__ __forceinline__ SegLds seg_lds(float* base, int trows, int rows_pb, int cap, bool bwd) {
    SegLds l;
    float* p = base;
    l.P = p; p += (size_t)trows * SG_TW;
    l.Q = p; p += (size_t)trows * SG_TW;
    l.D = p; if (bwd) p += (size_t)trows * SG_TW;
    l.B0 = bwd ? l.D : p; if (!bwd) p += SG_NCH * 256;
    l.B1 = p; if (!bwd) p += SG_NCH * 256;
    l.we = p; p += 2 * SG_TW;
    l.w2s = p; if (bwd) p += 4 * SG_TW;
    l.w2a = reinterpret_cast<float4*>(p); if (bwd) p += 4 * SG_W2A_CH * 4;
    l.part = reinterpret_cast<float4*>(p); if (bwd) p += 8 * 16 * 2 * 4;
    l.in.ea = reinterpret_cast<float2*>(p); p += 2 * cap;
    l.out.ea = reinterpret_cast<float2*>(p); if (bwd) p += 2 * cap;
    int* ip = reinterpret_cast<int*>(p);
    l.in.rp = ip; ip += rows_pb + 1;
    l.in.nb = ip; ip += cap;
    l.out.rp = ip; if (bwd) ip += rows_pb + 1;
    l.out.nb = ip;
    return l;
}
static size_t seg_lds_bytes(int trows, int rows_pb, int cap, bool bwd) {
    size_t f = (size_t)(bwd ? 3 : 2) * trows * SG_TW + (bwd ? 0 : 2 * SG_NCH * 256) + 2 * SG_TW + (bwd ? 4 * SG_TW + 4 * SG_W2A_CH * 4 + 8 * 16 * 2 * 4 : 0) + (size_t)(bwd ? 2 : 1) * 2 * cap;
    size_t i = (size_t)(bwd ? 2 : 1) * (rows_pb + 1 + cap);
    return (f + i) * 4 + 16;
}

// ------------------------------------------------------------------------------------------------ forward
__global__ __launch_bounds__(SG_THREADS) __attribute__((amdgpu_waves_per_eu(4, 4)))
void ea_seg_fwd_kernel(int n, int rows_pb, int trows, int cap, const int* __restrict__ rowptr, const int* __restrict__ nbr,
                       const EaSegFwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sg_smem[];
    const SegLds l = seg_lds(sg_smem, trows, rows_pb, cap, false);
    const int r0 = blockIdx.x * rows_pb, rows = min(rows_pb, n - r0);
    const SegCols sc = seg_cols(a.ld);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // ---- P | Q tiles: wave w < 4 owns row tile w and multiplies its ONE A fragment with both weight quarters (P, then Q); as two
    // waves per row tile every fragment was fetched twice, and the prologue is bound by the bytes it pulls through L2 -> L1.
    // (rows <= SG_MAX_ROWS = 128: waves 4..7 carry no tile)
    const int nrt = (rows + 31) >> 5;
    const bool mfma_on = wave < nrt;
    const int K8 = (a.K + 7) & ~7;
    // (K8 == 136, every hidden layer: the fragment goes out behind the staging and is consumed while it arrives, seg_tile.hpp)
    const bool async_a = K8 == 8 * SG_NCH;
    SegA ta;
    if (mfma_on && !async_a) seg_load_a(ta, a.x, a.ldx, K8, r0 + 32 * wave, r0 + rows - 1, lane);
    seg_copy_b(l.B0, a.Bi, sc.q, K8, wave, lane);
    seg_copy_b(l.B1, a.Bj, sc.q, K8, wave, lane);
    SegCsr cin = l.in;
    CsrRegs cri;
    csr_issue1(cin, cri, r0, rows, rowptr);
    const float wev = we_issue(sc, a.w1, a.h, a.fi);
    csr_issue2(cin, cri, cap, nbr, a.ea_in);
    csr_commit(cin, cri, r0, rows);
    we_commit(l.we, wev);
    if (sc.rem) {   // the trailing columns (tile columns 32..: the tiles write 0..31), in the shadow of the operand loads; all 512
        float v1[4], v2[4];   // threads, four k parts per row (on waves 4..7 alone, two parts per row, it took 2 us longer)
        const int nreal = min(sc.remv, a.h - 32 * sc.nq);
        seg_rem_dots<true, 2, 3>(threadIdx.x, a.x, a.ldx, a.K, r0, rows, a.Bi, a.Bj, sc.nq, nreal, v1, v2);
        seg_rem_store<true, 2>(threadIdx.x, rows, sc.nq, nreal, a.b1, v1, v2, l.P, l.Q);
    }
    if (async_a) {
        seg_drain_visible();
        if (mfma_on) seg_load_a_async(ta, a.x, a.ldx, r0 + 32 * wave, r0 + rows - 1, lane);
        seg_lds_barrier();
        if (mfma_on) {
            const f32x16 accp = seg_mma_async(ta, l.B0, lane);
            seg_store_tile(accp, sc.q, a.b1, a.h, l.P, 32 * wave, lane);
            const f32x16 accq = seg_mma_t<true>(ta, l.B1, K8, lane);
            seg_store_tile(accq, sc.q, nullptr, a.h, l.Q, 32 * wave, lane);
        }
    } else {
        seg_dma_wait();
        __syncthreads();
        if (mfma_on) {
            const f32x16 accp = seg_mma(ta, l.B0, K8, lane);
            seg_store_tile(accp, sc.q, a.b1, a.h, l.P, 32 * wave, lane);
            const f32x16 accq = seg_mma(ta, l.B1, K8, lane);
            seg_store_tile(accq, sc.q, nullptr, a.h, l.Q, 32 * wave, lane);
        }
    }
    __syncthreads();
    // ---- P, Q out (the backward pass recomputes the pre-activation from them), and the walk
    for (int it = threadIdx.x; it < rows * sc.cw; it += SG_THREADS) {
        const int lr = it / sc.cw, lc = it - lr * sc.cw;
        const int tc = seg_tcol(sc, lc), gc = seg_gcol(sc, lc);
        const float4 p4 = sg_ld4(l.P + (size_t)lr * SG_TW + tc);
        const size_t o = (size_t)(r0 + lr) * a.ld + gc;
        sg_st4_wt(a.P + o, p4);
        sg_st4_wt(a.Q + o, sg_ld4(l.Q + (size_t)lr * SG_TW + tc));
        const float4 w0 = sg_ld4(l.we + tc), w1 = sg_ld4(l.we + SG_TW + tc);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        const int beg = cin.rp[lr], end = cin.rp[lr + 1];
        if (cin.in_lds) {   // four slots per trip (slots past the row's end re-read its last edge and are not added): the walk is a
            const int last = end - 1;   // chain of dependent LDS reads (index -> tile), four independent chains at a time
            for (int p = beg; p < end; p += 4) {
                int s_[4];
                float2 a_[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int q = min(p + u, last);
                    s_[u] = cin.nb[q];
                    a_[u] = cin.ea[q];
                }
                float4 q_[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) q_[u] = sg_ld4(l.Q + (size_t)s_[u] * SG_TW + tc);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    float4 v = sg_add4(p4, q_[u]);
                    v = sg_fma4(a_[u].x, w0, v);
                    v = sg_fma4(a_[u].y, w1, v);
                    const float4 r = sg_add4(acc, sg_relu4(v));
                    const bool k = p + u < end;
                    acc.x = k ? r.x : acc.x;
                    acc.y = k ? r.y : acc.y;
                    acc.z = k ? r.z : acc.z;
                    acc.w = k ? r.w : acc.w;
                }
            }
        } else {
            for (int p = beg; p < end; ++p) {
                int ls;
                float2 a2;
                csr_slot(cin, p, r0, nbr, a.ea_in, ls, a2);
                float4 v = sg_add4(p4, sg_ld4(l.Q + (size_t)ls * SG_TW + tc));
                v = sg_fma4(a2.x, w0, v);
                v = sg_fma4(a2.y, w1, v);
                acc = sg_add4(acc, sg_relu4(v));
            }
        }
        sg_st4_wt(a.S + o, acc);
    }
}


// ------------------------------------------------------------------------------------------------ forward, layer 0
// The network's FRONT (mask_embd + residual, networks/MPN.py:533-537) and the whole first EdgeAggregation edge stage in ONE launch
// for batches of small graphs -- what front.hip's row-per-wave kernel and the generic edge walk did as two (18.8 + 9.2 us at
// case118v2 x 128).  Layer 0's node "GEMM" has K = 4: no matrix core, just the front's fma chains.  Block = (whole graph(s), one
// 32-column quarter), as everywhere in this file:
//   * every wave takes rows of the block round-robin, ONE ROW PER WAVE with front_fwd_wave_body's lane = four-unit chunk layout,
//     weight slices in registers, the same fma chains and the same xor-butterfly row sum -> x0 carries the bits of the front
//     kernel (all four quarter-blocks of a graph compute it: 2 kFLOP per row; the block of quarter 0 stores x0 and maskf);
//   * the lanes whose chunk lies in the block's quarter store their me_h units (the backward front reads them), form P | Q of
//     their chunk from x0 (the front's chains), drop them into the LDS tiles and store them (ea_seg_bwd reads them);
//   * the walk over the incoming edges runs on the tiles exactly as in ea_seg_fwd_kernel (edge attributes through the edge ids:
//     the slot-ordered copy is written by the pack blocks of this very launch).
// The weight re-layout ("pack") blocks of the forward pass ride behind the graph blocks, as they did behind the front's.
__device__ __forceinline__ float4 sg_wave_sum4(float4 v) {   // (front.hip wave_sum4: fixed butterfly, every lane ends with the same sum)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        v.x += __shfl_xor(v.x, off);
        v.y += __shfl_xor(v.y, off);
        v.z += __shfl_xor(v.z, off);
        v.w += __shfl_xor(v.w, off);
    }
    return v;
}
__global__ __launch_bounds__(SG_THREADS) __attribute__((amdgpu_waves_per_eu(4, 4)))
void front_seg_fwd_kernel(const FrontFwdArgs f, const PackArgs pa, int nseg_x, int nseg_y, int pack_bx, int rows_pb, int trows, int cap,
                          int e_stored, const int* __restrict__ rowptr, const int* __restrict__ nbr, const int* __restrict__ eid,
                          const float* __restrict__ ea, float* __restrict__ S, int ld) {
    extern __shared__ __attribute__((aligned(16))) float sg_smem[];
    const int tid = threadIdx.x;
    // riders of the forward pass's first launch (front.hip front_pack_kernel): dropout stream, workspace stamp, slot-ordered attributes
    if (pa.rng_advance && blockIdx.x == 0 && tid == 0) pa.rng_advance[1] += 1;
    if (pa.stamp && blockIdx.x == 0 && tid == 0) *pa.stamp = pa.stamp_value;
    slot_ea_body(pa.slot_ea, (int64_t)blockIdx.x * blockDim.x + tid, (int64_t)gridDim.x * blockDim.x);
    // the weight re-layout jobs: ONE workgroup per job, FIRST in the grid (every workgroup of this launch holds half a CU's LDS:
    // behind the graph blocks the jobs would start when those retire; dealt to the graph blocks they cost every block a loop
    // over all jobs, +10 us)
    // pack_bx > 0: the jobs have workgroups of their own, FIRST in the grid (few graph blocks: a small batch); pack_bx < 0: every
    // job is dealt to -pack_bx consecutive GRAPH blocks, which run their share after their walk (a few elements per thread) --
    // every workgroup of this launch holds half a CU's LDS, so job workgroups push a 512-block grid past one round of the chip,
    // and one workgroup per job is 36 dependent gather -> store trips (31 us)
    const int npack = pack_bx > 0 ? pack_bx * pa.njobs : 0;
    if ((int)blockIdx.x < npack) {
        const int job = (int)blockIdx.x / pack_bx;
        pack_job_body(pa.job[job], (int)blockIdx.x - job * pack_bx, pack_bx);
        return;
    }
    const int sb = (int)blockIdx.x - npack;
    const int bx = sb % nseg_x, by = sb / nseg_x;
    const SegLds l = seg_lds(sg_smem, trows, rows_pb, cap, false);
    const int n = f.n, h = f.h, ldw1 = f.ldw1;
    const int r0 = bx * rows_pb, rows = min(rows_pb, n - r0);
    const SegCols sc = seg_cols(ld, by);
    const int nchunk = ld >> 2;
    const float4 bb4 = make_float4(f.bb[0], f.bb[1], f.bb[2], f.bb[3]);
    // LDS behind the tiles / CSR slice: x0 and mask rows of the block, and the front weights of this block's (<= 9) chunks
    float4* s_x0 = reinterpret_cast<float4*>(l.B0);              // [rows_pb]   (B0 / B1: the weight-image room, unused here)
    float4* s_m = s_x0 + rows_pb;                                // [rows_pb]
    float* s_w = reinterpret_cast<float*>(s_m + rows_pb);        // [9 chunks][4 units][14]: wa[4] | ba | w1[8] | b1
    // ---- adjacency slice (by destination) and its edge attributes, through the edge ids
    SegCsr cin = l.in;
    cin.e0 = rowptr[r0];
    cin.ne = rowptr[r0 + rows] - cin.e0;
    cin.in_lds = cin.ne <= cap;
    const int rpv = tid <= rows ? rowptr[r0 + tid] : 0;
    const float wev = we_issue(sc, f.w1, h, 4);
    int nbv = 0, idv = 0;
    if (cin.in_lds && tid < cin.ne) {
        nbv = nbr[cin.e0 + tid];
        idv = eid[cin.e0 + tid];
    }
    // front weights of the block's chunks: value t = (local chunk lc, unit i, slot k)
    float fwv = 0.f;
    if (tid < sc.cw * 56) {
        const int lc = tid / 56, r = tid - lc * 56, i = r / 14, k = r - i * 14;
        const int u = seg_gcol(sc, lc) + i;
        if (u < h) fwv = k < 4 ? f.wa[(size_t)u * 4 + k] : k == 4 ? f.ba[u] : k < 13 ? f.w1[(size_t)u * ldw1 + (k - 5)] : f.b1[u];
    }
    float2 eav = make_float2(0.f, 0.f);
    if (cin.in_lds && tid < cin.ne) eav = *reinterpret_cast<const float2*>(ea + (size_t)(idv >= e_stored ? idv - e_stored : idv) * 2);
    if (tid <= rows) cin.rp[tid] = rpv - cin.e0;
    if (cin.in_lds && tid < cin.ne) {
        cin.nb[tid] = nbv - r0;
        cin.ea[tid] = eav;
    }
    we_commit(l.we, wev);
    if (tid < sc.cw * 56) s_w[tid] = fwv;
    // mask_embd's weights as one 48-byte record per hidden unit: wa[4] | ba, wb[0..2] | wb[3], 0, 0, 0  (three 16-byte LDS reads
    // per unit; as scalar loads inside the per-row loop every unit waited ~300 cycles for its own loads: 40 us)
    float4* s_me = reinterpret_cast<float4*>(s_w + 9 * 56);      // [h][3]
    int* s_cnt = reinterpret_cast<int*>(s_me + 3 * h);           // [2]: the block's mask census (FrontFwdArgs::mask_counts)
    float4* s_tab = s_me + 3 * h + 1;                            // [16]: the residual term of the 16 binary mask patterns
    if (tid < 2) s_cnt[tid] = 0;
    if (tid < h) {
        const float4 a4 = sg_ld4(f.wa + (size_t)tid * 4);
        const float b0 = f.ba[tid], w0_ = f.wb[tid], w1_ = f.wb[h + tid], w2_ = f.wb[2 * h + tid], w3_ = f.wb[3 * h + tid];
        s_me[3 * tid] = a4;
        s_me[3 * tid + 1] = make_float4(b0, w0_, w1_, w2_);
        s_me[3 * tid + 2] = make_float4(w3_, 0.f, 0.f, 0.f);
    }
    seg_lds_barrier();
    // ---- x0 = x + bb + Wb relu(Wa mask + ba): FOUR THREADS PER ROW.  The chunk sums a[c] (four hidden units each, an fma chain
    // from zero) are added in the order of the row-per-wave kernel's xor butterfly (front.hip wave_sum4, lane 0's tree:
    // s[c] = a[c] + a[c + 32]; t = s[c] + s[c + 16]; u = t[c] + t[c + 8]; v = u[c] + u[c + 4]; w = v[c] + v[c + 2]; w[0] + w[1]), so
    // x0 carries the bits that kernel stores: thread p of a row forms v[p], two quad shuffles finish the tree.  (As a butterfly
    // per row in every one of a graph's four blocks the LDS crossbar was the whole kernel: 24 ds_bpermute per row.)
    // ... and a TABLE for binary masks: the residual term Wb relu(Wa m + ba) depends on the row's four mask entries only, and
    // pred_mask is 0 / 1 (datasets/PowerFlowData.py:193: one pattern per bus type), so the first wave evaluates the tree ONCE for
    // each of the 16 patterns and every row looks its sum up -- the same operands in the same order give the same bits as the
    // per-row evaluation, which remains the path of a row whose mask holds anything else.  (Per row the phase was 36 hidden-unit
    // records and ~800 vector instructions per thread, in every one of a graph's four blocks: 7.6-9.8 of the launch's 25 us.)
    {
        const int pq = tid & 3;
        // v[pq] of one row's tree, then the two quad shuffles: call with all four lanes of a quad
        auto x0_tree = [&](const float4& m) -> float4 {
            auto chunk_sum = [&](int c) -> float4 {
                float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
                if (c < nchunk) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int u = 4 * c + i;
                        if (u < h) {
                            const float4 ra = s_me[3 * u], rb = s_me[3 * u + 1], rc = s_me[3 * u + 2];
                            float v = rb.x;
                            v = fmaf(ra.x, m.x, v); v = fmaf(ra.y, m.y, v); v = fmaf(ra.z, m.z, v); v = fmaf(ra.w, m.w, v);
                            v = fmaxf(v, 0.f);
                            acc.x = fmaf(rb.y, v, acc.x); acc.y = fmaf(rb.z, v, acc.y);
                            acc.z = fmaf(rb.w, v, acc.z); acc.w = fmaf(rc.x, v, acc.w);
                        }
                    }
                }
                return acc;
            };
            float4 u0 = make_float4(0.f, 0.f, 0.f, 0.f), vp = u0;
#pragma unroll
            for (int c2 = 0; c2 < 2; ++c2) {
                float4 t0 = make_float4(0.f, 0.f, 0.f, 0.f), tt = t0;
#pragma unroll
                for (int c3 = 0; c3 < 2; ++c3) {
                    const int c = pq + 4 * c2 + 8 * c3;
                    const float4 sa = sg_add4(chunk_sum(c), chunk_sum(c + 32)), sb = sg_add4(chunk_sum(c + 16), chunk_sum(c + 48));
                    const float4 t = sg_add4(sa, sb);
                    if (c3 == 0) t0 = t; else tt = sg_add4(t0, t);
                }
                if (c2 == 0) u0 = tt; else vp = sg_add4(u0, tt);
            }
            // w[p & 1] = v[p & 1] + v[(p & 1) + 2] (partner: lane ^ 2), then w[0] + w[1] (partner: lane ^ 1): lower index first
            float4 vo;
            vo.x = __shfl_xor(vp.x, 2); vo.y = __shfl_xor(vp.y, 2); vo.z = __shfl_xor(vp.z, 2); vo.w = __shfl_xor(vp.w, 2);
            const float4 wp = (pq & 2) ? sg_add4(vo, vp) : sg_add4(vp, vo);
            float4 wo;
            wo.x = __shfl_xor(wp.x, 1); wo.y = __shfl_xor(wp.y, 1); wo.z = __shfl_xor(wp.z, 1); wo.w = __shfl_xor(wp.w, 1);
            return (pq & 1) ? sg_add4(wo, wp) : sg_add4(wp, wo);
        };
        // the row's inputs are requested before the table is built
        const int lr = tid >> 2;
        const bool on = lr < rows;
        const int row = r0 + min(lr, rows - 1);
        float4 m;
        if (f.mask_dtype == 0) {
            const int64_t* mp = static_cast<const int64_t*>(f.mask) + (size_t)row * 4;
            m = make_float4((float)mp[0], (float)mp[1], (float)mp[2], (float)mp[3]);
        } else {
            m = sg_ld4(static_cast<const float*>(f.mask) + (size_t)row * 4);
        }
        const float4 xi = sg_ld4(f.x + (size_t)row * 4);
        if (tid < 64) {   // pattern k = tid >> 2: bit j set = mask entry j is 1
            const int k = tid >> 2;
            const float4 mk = make_float4((k & 1) ? 1.f : 0.f, (k & 2) ? 1.f : 0.f, (k & 4) ? 1.f : 0.f, (k & 8) ? 1.f : 0.f);
            const float4 sk = x0_tree(mk);
            if (pq == 0) s_tab[k] = sk;
        }
        seg_lds_barrier();
        const bool b0 = m.x == 0.f || m.x == 1.f, b1 = m.y == 0.f || m.y == 1.f, b2 = m.z == 0.f || m.z == 1.f, b3 = m.w == 0.f || m.w == 1.f;
        float4 s4;
        if (b0 && b1 && b2 && b3) s4 = s_tab[(m.x != 0.f ? 1 : 0) | (m.y != 0.f ? 2 : 0) | (m.z != 0.f ? 4 : 0) | (m.w != 0.f ? 8 : 0)];
        else s4 = x0_tree(m);      // (the four threads of a row hold the same mask: the quad takes this branch together)
        const float4 o = make_float4(xi.x + (s4.x + bb4.x), xi.y + (s4.y + bb4.y), xi.z + (s4.z + bb4.z), xi.w + (s4.w + bb4.w));
        int c1 = 0, c0 = 0;
        if (on && pq == 0) {
            s_x0[lr] = o;
            s_m[lr] = m;
            if (by == 0) {   // (one of the graph's blocks stores the 4-wide tensors)
                sg_st4_wt(f.maskf + (size_t)row * 4, m);
                sg_st4_wt(f.x0 + (size_t)row * 4, o);
            }
            // the block's mask census for a Masked_L2_loss riding in the backward pass (MseTail::counts): integer sums, any order
            c1 = (m.x != 0.f) + (m.y != 0.f) + (m.z != 0.f) + (m.w != 0.f);
            c0 = (1.f - m.x != 0.f) + (1.f - m.y != 0.f) + (1.f - m.z != 0.f) + (1.f - m.w != 0.f);
        }
        if (f.mask_counts && by == 0) {
            for (int off = 32; off > 0; off >>= 1) {
                c1 += __shfl_xor(c1, off);
                c0 += __shfl_xor(c0, off);
            }
            if ((tid & 63) == 0) {
                atomicAdd(&s_cnt[0], c1);
                atomicAdd(&s_cnt[1], c0);
            }
        }
    }
    seg_lds_barrier();
    if (f.mask_counts && by == 0 && tid < 2) f.mask_counts[2 * bx + tid] = s_cnt[tid];
    // ---- me_h, P | Q of the block's chunks: item = (row, chunk), the row-per-wave kernel's fma chains
    for (int it = tid; it < rows * sc.cw; it += SG_THREADS) {
        const int lr = it / sc.cw, lc = it - lr * sc.cw;
        const int tcw = seg_tcol(sc, lc), gc = seg_gcol(sc, lc), row = r0 + lr;
        const float4 m = s_m[lr], o = s_x0[lr];
        const float* wv = s_w + lc * 56;
        float hv[4], pv[4], qv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float* w14 = wv + i * 14;
            float v = w14[4];
            v = fmaf(w14[0], m.x, v); v = fmaf(w14[1], m.y, v); v = fmaf(w14[2], m.z, v); v = fmaf(w14[3], m.w, v);
            hv[i] = fmaxf(v, 0.f);
            float pa_ = w14[13];
            pa_ = fmaf(w14[5], o.x, pa_); pa_ = fmaf(w14[6], o.y, pa_); pa_ = fmaf(w14[7], o.z, pa_); pa_ = fmaf(w14[8], o.w, pa_);
            float qb = 0.f;
            qb = fmaf(w14[9], o.x, qb); qb = fmaf(w14[10], o.y, qb); qb = fmaf(w14[11], o.z, qb); qb = fmaf(w14[12], o.w, qb);
            pv[i] = pa_;
            qv[i] = qb;
        }
        if (f.me_h) sg_st4_wt(f.me_h + (size_t)row * ld + gc, make_float4(hv[0], hv[1], hv[2], hv[3]));
        const float4 p4 = make_float4(pv[0], pv[1], pv[2], pv[3]), q4 = make_float4(qv[0], qv[1], qv[2], qv[3]);
        sg_st4(l.P + (size_t)lr * SG_TW + tcw, p4);
        sg_st4(l.Q + (size_t)lr * SG_TW + tcw, q4);
        sg_st4_wt(f.P + (size_t)row * ld + gc, p4);
        sg_st4_wt(f.Q + (size_t)row * ld + gc, q4);
    }
    seg_lds_barrier();
    // ---- the walk (ea_seg_fwd_kernel's: four slots per trip, edge-id order)
    for (int it = tid; it < rows * sc.cw; it += SG_THREADS) {
        const int lr = it / sc.cw, lc = it - lr * sc.cw;
        const int tcw = seg_tcol(sc, lc), gc = seg_gcol(sc, lc);
        const float4 p4 = sg_ld4(l.P + (size_t)lr * SG_TW + tcw);
        const float4 w0 = sg_ld4(l.we + tcw), w1 = sg_ld4(l.we + SG_TW + tcw);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        const int beg = cin.rp[lr], end = cin.rp[lr + 1];
        if (cin.in_lds) {
            const int last = end - 1;
            for (int p = beg; p < end; p += 4) {
                int s_[4];
                float2 a_[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int q = min(p + u, last);
                    s_[u] = cin.nb[q];
                    a_[u] = cin.ea[q];
                }
                float4 q_[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) q_[u] = sg_ld4(l.Q + (size_t)s_[u] * SG_TW + tcw);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    float4 v = sg_add4(p4, q_[u]);
                    v = sg_fma4(a_[u].x, w0, v);
                    v = sg_fma4(a_[u].y, w1, v);
                    const float4 r = sg_add4(acc, sg_relu4(v));
                    const bool k = p + u < end;
                    acc.x = k ? r.x : acc.x;
                    acc.y = k ? r.y : acc.y;
                    acc.z = k ? r.z : acc.z;
                    acc.w = k ? r.w : acc.w;
                }
            }
        } else {
            for (int p = beg; p < end; ++p) {
                const int ls = nbr[cin.e0 + p] - r0;
                int id = eid[cin.e0 + p];
                id = id >= e_stored ? id - e_stored : id;
                const float2 a2 = *reinterpret_cast<const float2*>(ea + (size_t)id * 2);
                float4 v = sg_add4(p4, sg_ld4(l.Q + (size_t)ls * SG_TW + tcw));
                v = sg_fma4(a2.x, w0, v);
                v = sg_fma4(a2.y, w1, v);
                acc = sg_add4(acc, sg_relu4(v));
            }
        }
        sg_st4_wt(S + (size_t)(r0 + lr) * ld + gc, acc);
    }
    if (pack_bx < 0) {
        const int per = -pack_bx, job = sb / per;
        if (job < pa.njobs) pack_job_body(pa.job[job], sb - job * per, per);
    }
}

// ------------------------------------------------------------------------------------------------ backward
// One row's two backward walks for one column chunk:
//   by destination: dP[i] = sum_{e -> i} dh_e, dWe[f] += a_e[f] dh_e ;  by source: dQ[j] = sum_{e: src(e) = j} dh_e
//   dh_e = dS[dst e] where the recomputed pre-activation P[dst] + Q[src] + a_e We is > 0
// BOTH walks advance together, two slots each per trip (slots past a row's end re-read the block's last slot and contribute an
// exact zero): the walk is a chain of dependent LDS reads (index -> tile), and four independent chains per trip instead of one
// cut it from 9.2 to ~4 us per launch.  Sums stay in slot (= edge id) order.
__device__ __forceinline__ void seg_bwd_row(const SegLds& l, const SegCsr& cin, const SegCsr& cout, int lr, int tc, float4 w0,
                                            float4 w1, float4& accP, float4& accQ, float4& dwe0, float4& dwe1) {
    const float4 p4 = sg_ld4(l.P + (size_t)lr * SG_TW + tc), q4 = sg_ld4(l.Q + (size_t)lr * SG_TW + tc);
    const float4 g4 = sg_ld4(l.D + (size_t)lr * SG_TW + tc);
    accP = make_float4(0.f, 0.f, 0.f, 0.f);
    accQ = accP;
    const int b1 = cin.rp[lr], e1 = cin.rp[lr + 1], b2 = cout.rp[lr], e2 = cout.rp[lr + 1];
    const int nit = max(e1 - b1, e2 - b2);
    const int last1 = cin.ne - 1, last2 = cout.ne - 1;     // (nit > 0 implies both lists are non-empty for an undirected batch;
    for (int t = 0; t < nit; t += 2) {                     //  a clamped -1 is excluded by the max below)
        int s_[2], d_[2];
        float2 ai[2], ao[2];
        bool ki[2], ko[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int pi = max(min(b1 + t + u, last1), 0), po = max(min(b2 + t + u, last2), 0);
            ki[u] = b1 + t + u < e1;
            ko[u] = b2 + t + u < e2;
            s_[u] = cin.nb[pi];
            ai[u] = cin.ea[pi];
            d_[u] = cout.nb[po];
            ao[u] = cout.ea[po];
        }
        float4 qs[2], pd[2], gd[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            qs[u] = sg_ld4(l.Q + (size_t)s_[u] * SG_TW + tc);
            pd[u] = sg_ld4(l.P + (size_t)d_[u] * SG_TW + tc);
            gd[u] = sg_ld4(l.D + (size_t)d_[u] * SG_TW + tc);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            float4 v = sg_add4(p4, qs[u]);
            v = sg_fma4(ai[u].x, w0, v);
            v = sg_fma4(ai[u].y, w1, v);
            float4 dh;
            dh.x = (ki[u] && v.x > 0.f) ? g4.x : 0.f;
            dh.y = (ki[u] && v.y > 0.f) ? g4.y : 0.f;
            dh.z = (ki[u] && v.z > 0.f) ? g4.z : 0.f;
            dh.w = (ki[u] && v.w > 0.f) ? g4.w : 0.f;
            accP = sg_add4(accP, dh);
            dwe0 = sg_fma4(ai[u].x, dh, dwe0);
            dwe1 = sg_fma4(ai[u].y, dh, dwe1);
            float4 z = sg_add4(pd[u], q4);
            z = sg_fma4(ao[u].x, w0, z);
            z = sg_fma4(ao[u].y, w1, z);
            accQ.x += (ko[u] && z.x > 0.f) ? gd[u].x : 0.f;
            accQ.y += (ko[u] && z.y > 0.f) ? gd[u].y : 0.f;
            accQ.z += (ko[u] && z.z > 0.f) ? gd[u].z : 0.f;
            accQ.w += (ko[u] && z.w > 0.f) ? gd[u].w : 0.f;
        }
    }
}
// One direction only (the trailing chunk's second pass gives a row's two walks to two threads), four slots per trip
__device__ __forceinline__ void seg_bwd_row_dst(const SegLds& l, const SegCsr& cin, int lr, int tc, float4 w0, float4 w1,
                                                float4& accP, float4& dwe0, float4& dwe1) {
    const float4 p4 = sg_ld4(l.P + (size_t)lr * SG_TW + tc), g4 = sg_ld4(l.D + (size_t)lr * SG_TW + tc);
    accP = make_float4(0.f, 0.f, 0.f, 0.f);
    const int beg = cin.rp[lr], end = cin.rp[lr + 1], last = end - 1;
    for (int p = beg; p < end; p += 4) {
        int s_[4];
        float2 ai[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int q = min(p + u, last);
            s_[u] = cin.nb[q];
            ai[u] = cin.ea[q];
        }
        float4 qs[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) qs[u] = sg_ld4(l.Q + (size_t)s_[u] * SG_TW + tc);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float4 v = sg_add4(p4, qs[u]);
            v = sg_fma4(ai[u].x, w0, v);
            v = sg_fma4(ai[u].y, w1, v);
            const bool k = p + u < end;
            float4 dh;
            dh.x = (k && v.x > 0.f) ? g4.x : 0.f;
            dh.y = (k && v.y > 0.f) ? g4.y : 0.f;
            dh.z = (k && v.z > 0.f) ? g4.z : 0.f;
            dh.w = (k && v.w > 0.f) ? g4.w : 0.f;
            accP = sg_add4(accP, dh);
            dwe0 = sg_fma4(ai[u].x, dh, dwe0);
            dwe1 = sg_fma4(ai[u].y, dh, dwe1);
        }
    }
}
__device__ __forceinline__ void seg_bwd_row_src(const SegLds& l, const SegCsr& cout, int lr, int tc, float4 w0, float4 w1,
                                                float4& accQ) {
    const float4 q4 = sg_ld4(l.Q + (size_t)lr * SG_TW + tc);
    accQ = make_float4(0.f, 0.f, 0.f, 0.f);
    const int beg = cout.rp[lr], end = cout.rp[lr + 1], last = end - 1;
    for (int p = beg; p < end; p += 4) {
        int d_[4];
        float2 ao[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int q = min(p + u, last);
            d_[u] = cout.nb[q];
            ao[u] = cout.ea[q];
        }
        float4 pd[4], gd[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            pd[u] = sg_ld4(l.P + (size_t)d_[u] * SG_TW + tc);
            gd[u] = sg_ld4(l.D + (size_t)d_[u] * SG_TW + tc);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float4 z = sg_add4(pd[u], q4);
            z = sg_fma4(ao[u].x, w0, z);
            z = sg_fma4(ao[u].y, w1, z);
            const bool k = p + u < end;
            accQ.x += (k && z.x > 0.f) ? gd[u].x : 0.f;
            accQ.y += (k && z.y > 0.f) ? gd[u].y : 0.f;
            accQ.z += (k && z.z > 0.f) ? gd[u].z : 0.f;
            accQ.w += (k && z.w > 0.f) ? gd[u].w : 0.f;
        }
    }
}
// the same with the indices read from global memory (a block with more edges than its LDS slice holds)
__device__ __forceinline__ void seg_bwd_row_slow(const SegLds& l, const SegCsr& cin, const SegCsr& cout, int lr, int tc, int r0,
                                                 const int* __restrict__ in_src, const int* __restrict__ out_dst,
                                                 const float* __restrict__ ea_in, const float* __restrict__ ea_out, float4 w0,
                                                 float4 w1, float4& accP, float4& accQ, float4& dwe0, float4& dwe1) {
    const float4 p4 = sg_ld4(l.P + (size_t)lr * SG_TW + tc), q4 = sg_ld4(l.Q + (size_t)lr * SG_TW + tc);
    const float4 g4 = sg_ld4(l.D + (size_t)lr * SG_TW + tc);
    accP = make_float4(0.f, 0.f, 0.f, 0.f);
    accQ = accP;
    for (int p = cin.rp[lr]; p < cin.rp[lr + 1]; ++p) {
        int ls;
        float2 a2;
        csr_slot(cin, p, r0, in_src, ea_in, ls, a2);
        float4 v = sg_add4(p4, sg_ld4(l.Q + (size_t)ls * SG_TW + tc));
        v = sg_fma4(a2.x, w0, v);
        v = sg_fma4(a2.y, w1, v);
        float4 dh;
        dh.x = v.x > 0.f ? g4.x : 0.f;
        dh.y = v.y > 0.f ? g4.y : 0.f;
        dh.z = v.z > 0.f ? g4.z : 0.f;
        dh.w = v.w > 0.f ? g4.w : 0.f;
        accP = sg_add4(accP, dh);
        dwe0 = sg_fma4(a2.x, dh, dwe0);
        dwe1 = sg_fma4(a2.y, dh, dwe1);
    }
    for (int p = cout.rp[lr]; p < cout.rp[lr + 1]; ++p) {
        int ld_;
        float2 a2;
        csr_slot(cout, p, r0, out_dst, ea_out, ld_, a2);
        float4 v = sg_add4(sg_ld4(l.P + (size_t)ld_ * SG_TW + tc), q4);
        v = sg_fma4(a2.x, w0, v);
        v = sg_fma4(a2.y, w1, v);
        const float4 gd = sg_ld4(l.D + (size_t)ld_ * SG_TW + tc);
        accQ.x += v.x > 0.f ? gd.x : 0.f;
        accQ.y += v.y > 0.f ? gd.y : 0.f;
        accQ.z += v.z > 0.f ? gd.z : 0.f;
        accQ.w += v.w > 0.f ? gd.w : 0.f;
    }
}

template <bool DSG, bool LOSS = false>   // LOSS (with DSG): the MSELoss tail -- out, loss and grad_out formed here (MseTail)
__global__ __launch_bounds__(SG_THREADS) __attribute__((amdgpu_waves_per_eu(4, 4)))
void ea_seg_bwd_kernel(int n, int rows_pb, int trows, int cap, const int* __restrict__ rp_in, const int* __restrict__ in_src,
                       const int* __restrict__ rp_out, const int* __restrict__ out_dst, const EaSegBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sg_smem[];
    const SegLds l = seg_lds(sg_smem, trows, rows_pb, cap, true);
    const int r0 = blockIdx.x * rows_pb, rows = min(rows_pb, n - r0);
    const SegCols sc = seg_cols(a.ld);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // ---- dS slice: the MFMA waves request their operands first (the staging below is several dependent loads deep)
    const int nrt = (rows + 31) >> 5;
    const bool mfma_on = !DSG && wave < nrt;
    const int K8 = (a.fo + 7) & ~7;
    const bool async_a = !DSG && K8 == 8 * SG_NCH;   // (the fragment behind the staging, consumed while it arrives: seg_tile.hpp)
    SegA ta;
    if (mfma_on && !async_a) seg_load_a(ta, a.gout, a.ldgo, K8, r0 + 32 * wave, r0 + rows - 1, lane);
    if (!DSG) seg_copy_b(l.B0, a.Bd, sc.q, K8, wave, lane);
    SegCsr cin = l.in, cout = l.out;
    CsrRegs cri, cro;
    csr_issue1(cin, cri, r0, rows, rp_in);
    csr_issue1(cout, cro, r0, rows, rp_out);
    const float wev = we_issue(sc, a.w1, a.h, a.fi);
    // P, Q slices (rows * cw <= 1152 items: three per thread at most), requested with everything else
    float4 pv[3], qv[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int it = threadIdx.x + j * SG_THREADS;
        pv[j] = qv[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (it < rows * sc.cw) {
            const int lr = it / sc.cw, lc = it - lr * sc.cw;
            const size_t o = (size_t)(r0 + lr) * a.ld + seg_gcol(sc, lc);
            pv[j] = sg_ld4(a.P + o);
            qv[j] = sg_ld4(a.Q + o);
        }
    }
    csr_issue2(cin, cri, cap, in_src, a.ea_in);
    csr_issue2(cout, cro, cap, out_dst, a.ea_out);
    // MSELoss tail: thread (row lr = t >> 2, part pq = t & 3) requests the S chunks c = pq + 4 k of its row (nine at most)
    const int nch = a.ld >> 2;
    const int mlr = threadIdx.x >> 2, mpq = threadIdx.x & 3;
    const int mrow = r0 + min(mlr, rows - 1);
    float4 sv[9], yv = make_float4(0.f, 0.f, 0.f, 0.f), mkv = yv;
    float dgv = 0.f;
    if (LOSS) {   // (requested with the rest of the prologue: behind its commits they were one more exposed round trip)
#pragma unroll
        for (int k = 0; k < 9; ++k) sv[k] = sg_ld4(a.mse.S + (size_t)mrow * a.ld + 4 * min(mpq + 4 * k, nch - 1));
        if (mpq == 0) {
            yv = sg_ld4(a.mse.y + (size_t)mrow * 4);
            dgv = a.mse.deg[mrow];
            if (a.mse.maskf) mkv = sg_ld4(a.mse.maskf + (size_t)mrow * 4);
        }
    }
    csr_commit(cin, cri, r0, rows);
    csr_commit(cout, cro, r0, rows);
    we_commit(l.we, wev);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int it = threadIdx.x + j * SG_THREADS;
        if (it < rows * sc.cw) {
            const int lr = it / sc.cw, lc = it - lr * sc.cw;
            const int tc = seg_tcol(sc, lc);
            sg_st4(l.P + (size_t)lr * SG_TW + tc, pv[j]);
            sg_st4(l.Q + (size_t)lr * SG_TW + tc, qv[j]);
        }
    }
    // the trailing columns of dS: VALU dot products on the four waves that carry no tile (rows <= 128), in the shadow of the
    // prologue's loads; kept in registers until the dS tile exists (it is still the weight quarter)
    float v1[4], v2[4];
    const int nreal = min(sc.remv, a.h - 32 * sc.nq);
    const bool rem_helper = !DSG && sc.rem && wave >= 4;
    // (where the multiply runs while its fragment arrives, these dot products run UNDER it, behind the publishing barrier: in the
    //  prologue they held that barrier -- of the block that is the launch's critical path -- back by their ~4 us)
    if (rem_helper && !async_a) seg_rem_dots<false, 1, 5>(threadIdx.x - 256, a.gout, a.ldgo, a.fo, r0, rows, a.Bd, nullptr, sc.nq, nreal, v1, v2);
    if (DSG) {   // last layer: dS[row][u] = sum_o gout[row][o] W2[o][u] from the 16-byte gout rows (edge.hip ds_row)
        // (the gout rows are requested BEFORE the barrier that publishes the W2 slice: behind it they were one more exposed round trip)
        float4 gv[3];
        if (LOSS) {
            if (a.mse.maskf) {   // Masked_L2_loss: the batch's mask census = the sum of the row blocks' (integers: any order)
                int c1 = 0, c0 = 0;
                for (int i = threadIdx.x; i < a.mse.count_blocks; i += SG_THREADS) {
                    c1 += a.mse.counts[2 * i];
                    c0 += a.mse.counts[2 * i + 1];
                }
                for (int off = 32; off > 0; off >>= 1) {
                    c1 += __shfl_xor(c1, off);
                    c0 += __shfl_xor(c0, off);
                }
                if (lane == 0) {
                    int* cw = reinterpret_cast<int*>(l.part) + 768 + 2 * wave;
                    cw[0] = c1;
                    cw[1] = c0;
                }
            }
            for (int i = threadIdx.x; i < 4 * SG_W2A_CH; i += SG_THREADS) {
                const int o = i / SG_W2A_CH, c = i - o * SG_W2A_CH;
                float w[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int u = 4 * c + j;
                    const float v = a.w2[(size_t)min(o, a.fo - 1) * a.h + min(u, a.h - 1)];
                    w[j] = (o < a.fo && u < a.h) ? v : 0.f;
                }
                l.w2a[i] = make_float4(w[0], w[1], w[2], w[3]);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int it = threadIdx.x + j * SG_THREADS;
                gv[j] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (it < rows * sc.cw) gv[j] = sg_ld4(a.gout + (size_t)(r0 + it / sc.cw) * 4);
            }
        }
        for (int i = threadIdx.x; i < 4 * SG_TW; i += SG_THREADS) {
            const int o = i / SG_TW, t = i - o * SG_TW;
            const int col = seg_col_of_tile(sc, t);
            l.w2s[i] = (o < a.fo && col >= 0 && col < a.h) ? a.w2[(size_t)o * a.h + col] : 0.f;
        }
        __syncthreads();
        if (LOSS) {
            // out[row][o] = sum_u S[row][u] W2[o][u] + deg b2[o] with lin_out4_wave_kernel's bits: chunk products as its fma chain, the
            // chunk sums added in the order of its xor butterfly (lane 0's tree: a[c] + a[c + 32], + 16, + 8, + 4 inside a part --
            // slots k, k + 8 | + 4 | + 2 | + 1 --, then the parts by two quad shuffles); see front_seg_fwd_kernel's x0 phase
            float4* gs = l.part;                                   // [SG_MAX_ROWS] grad_out rows (the dWe partials come last)
            float* sq = reinterpret_cast<float*>(l.part + SG_MAX_ROWS);   // [SG_MAX_ROWS] squared errors per row
            auto slot = [&](int k) -> float4 {
                const int c = mpq + 4 * k;
                if (k >= 9 || c >= nch) return make_float4(0.f, 0.f, 0.f, 0.f);
                const float4 s4 = sv[k < 9 ? k : 0];
                const float4 r0_ = l.w2a[c], r1_ = l.w2a[SG_W2A_CH + c], r2_ = l.w2a[2 * SG_W2A_CH + c], r3_ = l.w2a[3 * SG_W2A_CH + c];
                float4 acc;
                acc.x = fmaf(s4.w, r0_.w, fmaf(s4.z, r0_.z, fmaf(s4.y, r0_.y, s4.x * r0_.x)));
                acc.y = fmaf(s4.w, r1_.w, fmaf(s4.z, r1_.z, fmaf(s4.y, r1_.y, s4.x * r1_.x)));
                acc.z = fmaf(s4.w, r2_.w, fmaf(s4.z, r2_.z, fmaf(s4.y, r2_.y, s4.x * r2_.x)));
                acc.w = fmaf(s4.w, r3_.w, fmaf(s4.z, r3_.z, fmaf(s4.y, r3_.y, s4.x * r3_.x)));
                return acc;
            };
            float4 u0 = make_float4(0.f, 0.f, 0.f, 0.f), vp = u0;
#pragma unroll
            for (int k1 = 0; k1 < 2; ++k1) {
                float4 t0 = make_float4(0.f, 0.f, 0.f, 0.f), tt = t0;
#pragma unroll
                for (int k2 = 0; k2 < 2; ++k2) {
                    const int k = k1 + 2 * k2;
                    const float4 sa = sg_add4(slot(k), slot(k + 8)), sb = sg_add4(slot(k + 4), slot(k + 12));
                    const float4 t = sg_add4(sa, sb);
                    if (k2 == 0) t0 = t; else tt = sg_add4(t0, t);
                }
                if (k1 == 0) u0 = tt; else vp = sg_add4(u0, tt);
            }
            float4 vo;
            vo.x = __shfl_xor(vp.x, 2); vo.y = __shfl_xor(vp.y, 2); vo.z = __shfl_xor(vp.z, 2); vo.w = __shfl_xor(vp.w, 2);
            const float4 wp = (mpq & 2) ? sg_add4(vo, vp) : sg_add4(vp, vo);
            float4 wo;
            wo.x = __shfl_xor(wp.x, 1); wo.y = __shfl_xor(wp.y, 1); wo.z = __shfl_xor(wp.z, 1); wo.w = __shfl_xor(wp.w, 1);
            const float4 t4 = (mpq & 1) ? sg_add4(wo, wp) : sg_add4(wp, wo);
            if (mpq == 0) {
                const float b0 = a.mse.b2[0], b1_ = a.fo > 1 ? a.mse.b2[1] : 0.f, b2_ = a.fo > 2 ? a.mse.b2[2] : 0.f, b3_ = a.fo > 3 ? a.mse.b2[3] : 0.f;
                const float4 o4 = make_float4(fmaf(dgv, b0, t4.x), a.fo > 1 ? fmaf(dgv, b1_, t4.y) : 0.f,
                                              a.fo > 2 ? fmaf(dgv, b2_, t4.z) : 0.f, a.fo > 3 ? fmaf(dgv, b3_, t4.w) : 0.f);
                const bool on = mlr < rows;
                const float4 d4 = make_float4(o4.x - yv.x, o4.y - yv.y, o4.z - yv.z, o4.w - yv.w);
                float4 g4 = make_float4(0.f, 0.f, 0.f, 0.f);
                float q1 = 0.f, q0 = 0.f;
                if (a.mse.maskf) {
                    // Masked_L2_loss: masked_l2_grad_kernel's expressions (model.hip) on the batch-wide counts
                    const int* cw = reinterpret_cast<const int*>(l.part) + 768;
                    int c1 = 0, c0 = 0;
#pragma unroll
                    for (int w = 0; w < SG_WAVES; ++w) { c1 += cw[2 * w]; c0 += cw[2 * w + 1]; }
                    const float g1 = 2.f / (float)c1, g0 = a.mse.regularize ? 2.f * a.mse.regcoeff / (float)c0 : 0.f;
                    const float dv[4] = {d4.x, d4.y, d4.z, d4.w}, mv[4] = {mkv.x, mkv.y, mkv.z, mkv.w};
                    float gv4[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float d = dv[j], m = mv[j];
                        float g = 0.f;
                        if (m != 0.f) { g += g1 * d; q1 = fmaf(d, d, q1); }
                        if (1.f - m != 0.f) {
                            if (a.mse.regularize) g += g0 * d;
                            q0 = fmaf(d, d, q0);
                        }
                        gv4[j] = g;
                    }
                    if (on) g4 = make_float4(gv4[0], gv4[1], gv4[2], gv4[3]);
                } else {
                    if (on) g4 = make_float4(2.f * d4.x * a.mse.inv_n, 2.f * d4.y * a.mse.inv_n, 2.f * d4.z * a.mse.inv_n, 2.f * d4.w * a.mse.inv_n);
                    q1 = fmaf(d4.w, d4.w, fmaf(d4.z, d4.z, fmaf(d4.y, d4.y, d4.x * d4.x)));
                }
                if (mlr < SG_MAX_ROWS) {
                    gs[mlr] = g4;
                    sq[mlr] = on ? q1 : 0.f;
                    sq[SG_MAX_ROWS + mlr] = on ? q0 : 0.f;
                }
                if (on && sc.q == max(0, sc.nq - 2)) {   // ONE quarter per graph stores the 4-wide tensors (an early, light one)
                    sg_st4_wt(a.mse.out + (size_t)mrow * 4, o4);
                    sg_st4_wt(a.mse.gout + (size_t)mrow * 4, g4);
                }
            }
            seg_lds_barrier();
            // the block's loss partial: rows l and l + 64, then a fixed xor tree (wave 0; once per block)
            if (wave == 0 && sc.q == max(0, sc.nq - 2)) {
                float v = sq[lane] + sq[lane + 64], v0 = sq[SG_MAX_ROWS + lane] + sq[SG_MAX_ROWS + lane + 64];
                for (int off = 32; off > 0; off >>= 1) {
                    v += __shfl_xor(v, off);
                    v0 += __shfl_xor(v0, off);
                }
                if (lane == 0) {   // sc1: write-through
                    if (a.mse.maskf) {
                        __hip_atomic_store(a.mse.partial + 2 * blockIdx.x, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(a.mse.partial + 2 * blockIdx.x + 1, v0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    } else {
                        __hip_atomic_store(a.mse.partial + blockIdx.x, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int it = threadIdx.x + j * SG_THREADS;
                gv[j] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (it < rows * sc.cw) gv[j] = gs[it / sc.cw];
            }
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int it = threadIdx.x + j * SG_THREADS;
            if (it < rows * sc.cw) {
                const int lr = it / sc.cw, lc = it - lr * sc.cw;
                const int tc = seg_tcol(sc, lc);
                const float4 g = gv[j];
                const float4 x0 = sg_ld4(l.w2s + tc);
                float4 r = make_float4(g.x * x0.x, g.x * x0.y, g.x * x0.z, g.x * x0.w);
                r = sg_fma4(g.y, sg_ld4(l.w2s + SG_TW + tc), r);
                r = sg_fma4(g.z, sg_ld4(l.w2s + 2 * SG_TW + tc), r);
                r = sg_fma4(g.w, sg_ld4(l.w2s + 3 * SG_TW + tc), r);
                sg_st4(l.D + (size_t)lr * SG_TW + tc, r);
            }
        }
    } else {
        // (the W2 quarter sits where the dS tile goes: every wave is done reading it before the first one writes)
        f32x16 acc;
        if (async_a) {
            seg_drain_visible();
            if (mfma_on) seg_load_a_async(ta, a.gout, a.ldgo, r0 + 32 * wave, r0 + rows - 1, lane);
            seg_lds_barrier();
            if (mfma_on) acc = seg_mma_async(ta, l.B0, lane);
            else if (rem_helper) seg_rem_dots<false, 1, 5>(threadIdx.x - 256, a.gout, a.ldgo, a.fo, r0, rows, a.Bd, nullptr, sc.nq, nreal, v1, v2);
        } else {
            seg_dma_wait();
            __syncthreads();
            if (mfma_on) acc = seg_mma(ta, l.B0, K8, lane);
        }
        __syncthreads();
        if (mfma_on) seg_store_tile(acc, sc.q, nullptr, a.h, l.D, 32 * wave, lane);
        else if (rem_helper) seg_rem_store<false, 1>(threadIdx.x - 256, rows, sc.nq, nreal, nullptr, v1, v2, l.D, nullptr);
    }
    __syncthreads();
    // ---- walks.  Thread = (chunk lane lc = tid & 7, row lane ty = tid >> 3): a thread keeps ONE column chunk for all its rows, so
    // the dWe partial sums stay in registers and the block emits one ordered partial per column.  The block that also owns the
    // trailing columns walks that ninth chunk in a second, short pass (two threads per row, one per direction): as a ninth lane it halved the row lanes and
    // made those blocks -- the launch's critical path -- twice as long.
    auto walk_rows = [&](int lr0, int lr_step, int tc, int gc, float4& dwe0, float4& dwe1) {
        const float4 w0 = sg_ld4(l.we + tc), w1 = sg_ld4(l.we + SG_TW + tc);
        for (int lr = lr0; lr < rows; lr += lr_step) {
            float4 accP, accQ;
            if (cin.in_lds && cout.in_lds)
                seg_bwd_row(l, cin, cout, lr, tc, w0, w1, accP, accQ, dwe0, dwe1);
            else
                seg_bwd_row_slow(l, cin, cout, lr, tc, r0, in_src, out_dst, a.ea_in, a.ea_out, w0, w1, accP, accQ, dwe0, dwe1);
            sg_st4_wt(a.dP + (size_t)(r0 + lr) * a.ld + gc, accP);
            sg_st4_wt(a.dQ + (size_t)(r0 + lr) * a.ld + gc, accQ);
        }
    };
    const int cwm = sc.cw - (sc.rem ? 1 : 0);             // chunks of the 32-column quarter itself (<= 8)
    const int lc = threadIdx.x & 7, ty = threadIdx.x >> 3;
    float4 dwe0 = make_float4(0.f, 0.f, 0.f, 0.f), dwe1 = dwe0, rwe0 = dwe0, rwe1 = dwe0;
    if (lc < cwm) walk_rows(ty, SG_THREADS / 8, 4 * lc, sc.col0 + 4 * lc, dwe0, dwe1);
    if (sc.rem) {   // the trailing chunk: a row's two walks go to two threads (t >> 1 = row, t & 1 = direction)
        const int lr = threadIdx.x >> 1, gc = 32 * sc.nq;
        if (lr < rows) {
            if (cin.in_lds && cout.in_lds) {
                const float4 w0 = sg_ld4(l.we + 32), w1 = sg_ld4(l.we + SG_TW + 32);
                float4 acc;
                if (threadIdx.x & 1) {
                    seg_bwd_row_src(l, cout, lr, 32, w0, w1, acc);
                    sg_st4_wt(a.dQ + (size_t)(r0 + lr) * a.ld + gc, acc);
                } else {
                    seg_bwd_row_dst(l, cin, lr, 32, w0, w1, acc, rwe0, rwe1);
                    sg_st4_wt(a.dP + (size_t)(r0 + lr) * a.ld + gc, acc);
                }
            } else if (!(threadIdx.x & 1)) {
                walk_rows(lr, SG_THREADS, 32, gc, rwe0, rwe1);
            }
        }
    }
    // ordered reduction of the dWe partials over the row lanes: inside a wave by a fixed xor tree (over the lanes that share a chunk:
    // lane bits 3..5; all six bits for the trailing chunk), then over the 8 waves in wave order -> the block's partial [2][ld]
    auto xor_sum = [&](float4& v, int off) {
        v.x += __shfl_xor(v.x, off); v.y += __shfl_xor(v.y, off); v.z += __shfl_xor(v.z, off); v.w += __shfl_xor(v.w, off);
    };
    for (int off = 32; off >= 8; off >>= 1) { xor_sum(dwe0, off); xor_sum(dwe1, off); }
    if (sc.rem)
        for (int off = 32; off >= 1; off >>= 1) { xor_sum(rwe0, off); xor_sum(rwe1, off); }
    if (lane < 8) {
        l.part[(wave * 16 + lane) * 2] = dwe0;
        l.part[(wave * 16 + lane) * 2 + 1] = dwe1;
    } else if (lane == 8) {
        l.part[(wave * 16 + 8) * 2] = rwe0;
        l.part[(wave * 16 + 8) * 2 + 1] = rwe1;
    }
    seg_lds_barrier();   // (not __syncthreads(): the dP / dQ stores of the walks drain meanwhile)
    if (threadIdx.x < 32) {
        const int f = threadIdx.x >> 4, c2 = threadIdx.x & 15;
        if (c2 < sc.cw) {
            float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int w = 0; w < SG_WAVES; ++w) s = sg_add4(s, l.part[(w * 16 + c2) * 2 + f]);
            sg_st4_wt(a.dWe_partial + ((size_t)blockIdx.x * 2 + f) * a.ld + seg_gcol(sc, c2), s);
        }
    }
    if (LOSS && wave == 0 && sc.q == max(0, sc.nq - 2)) {
        // the loss: the last of the partial-owning blocks to get here sums the partials in BLOCK order (deterministic).  Same
        // hand-off as mse_kernel (model.hip): write-through partial, drained, then the ticket; agent-scope loads on the consumer
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "the loss hand-off relies on gfx950 semantics (sc1 write-through stores drained by s_waitcnt vmcnt(0))"
#endif
        int last = 0;
        if (lane == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            last = __hip_atomic_fetch_add(a.mse.counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.x - 1;
        }
        last = __shfl(last, 0);
        if (last) {
            const int stride = a.mse.maskf ? 2 : 1;
            float v = 0.f, v0 = 0.f;
            for (int i = lane; i < (int)gridDim.x; i += 64) {
                v += __hip_atomic_load(a.mse.partial + stride * i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (a.mse.maskf) v0 += __hip_atomic_load(a.mse.partial + 2 * i + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            int c1 = 0, c0 = 0;
            if (a.mse.maskf)
                for (int i = lane; i < a.mse.count_blocks; i += 64) { c1 += a.mse.counts[2 * i]; c0 += a.mse.counts[2 * i + 1]; }
            for (int off = 32; off > 0; off >>= 1) {
                v += __shfl_xor(v, off);
                v0 += __shfl_xor(v0, off);
                c1 += __shfl_xor(c1, off);
                c0 += __shfl_xor(c0, off);
            }
            if (lane == 0) {
                if (a.mse.maskf) {   // masked_l2_reduce_kernel's closing expressions (0 / 0 = NaN: torch's mean of an empty selection)
                    float lv = v / (float)c1;
                    if (a.mse.regularize) lv += a.mse.regcoeff * (v0 / (float)c0);
                    a.mse.loss[0] = lv;
                } else {
                    a.mse.loss[0] = v * a.mse.inv_n;
                }
                *a.mse.counter = 0;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ host
struct SegPlan { int rows_pb, trows, cap, nblocks, ny; };
static bool seg_plan(int seg, int n, int ld, SegPlan& p, bool bwd_limits = true) {
    if (seg <= 0 || seg > SG_MAX_ROWS || n <= 0 || n % seg != 0) return false;
    const int gpb = std::max(1, SG_MAX_ROWS / seg);
    p.rows_pb = gpb * seg;
    p.trows = (p.rows_pb + 31) / 32 * 32;
    p.cap = p.rows_pb * 4;
    p.nblocks = (n + p.rows_pb - 1) / p.rows_pb;
    int remv, nq;
    col_plan(ld, remv, nq);
    p.ny = nq;   // one block column per 32-column quarter
    return p.ny >= 1 && ld <= 8 * SG_NCH &&   // (one k piece: every layer width of the model is <= ld)
           (!bwd_limits || p.nblocks <= 1024) &&   // (the dWe partial buffer holds 1024 row blocks)
           seg_lds_bytes(p.trows, p.rows_pb, p.cap, true) <= (size_t)SG_LDS_BYTES &&
           seg_lds_bytes(p.trows, p.rows_pb, p.cap, false) <= (size_t)SG_LDS_BYTES;
}
// `bwd`: the backward kernel's dWe partial buffer bounds the number of row blocks; the forward kernel has no such bound (a big
// inference batch takes it, and a training batch beyond the bound pairs it with the generic backward: P, Q, S mean the same)
bool ea_seg_fit(int seg, int n, int fe, int ld, bool bwd) {
    static const bool off = diag_env("PFN_NO_SEG_EA") != nullptr;   // A/B switch: the generic gemm_nt + edge kernels
    SegPlan p;
    // Only in the latency regime: with a few blocks per CU a launch is one MFMA tile and one walk deep and the saved launches
    // and HBM round trips win (case118 x 128: -5 % of the step); with many blocks per CU the weight-stationary persistent
    // gemm_nt amortises its prologue and is the faster GEMM (case118 x 2048 inference: 3.39 vs 3.11 ms, measured)
    const long per_cu = 4L;
    return !off && fe == 2 && seg_plan(seg, n, ld, p, bwd) && (long)p.nblocks * p.ny <= per_cu * device_cus();
}
int ea_seg_blocks(int seg, int n, int ld) {
    SegPlan p;
    return seg_plan(seg, n, ld, p) ? p.nblocks : 0;
}

// the front + first edge stage in one launch (front_seg_fwd_kernel): latency regime, nfeature_dim 4, Fe = 2
bool front_seg_fit(int seg, int n, int h, int fe) {
    static const bool off = diag_env("PFN_NO_SEG_FRONT") != nullptr;   // A/B switch: front.hip's launch + the generic edge walk
    const int ld = ld_of(h);
    return !off && ld / 4 <= 64 && front_latency_regime(h, n) && ea_seg_fit(seg, n, fe, ld, false);
}
int launch_front_seg_fwd(const GraphView& g, const FrontFwdArgs& f, const PackJob* jobs, int njobs, uint64_t* rng_advance,
                         const SlotEa* slot_ea, int* stamp, int stamp_value, const float* ea, float* S, int seg, hipStream_t s) {
    const int ld = ld_of(f.h);
    SegPlan p;
    if (!seg_plan(seg, g.n, ld, p, false)) {
        set_error("front_seg_fwd: %d-row graphs do not fit", seg);
        return PFN_EINVAL;
    }
    if (f.mask_dtype != 0 && f.mask_dtype != 1) {
        set_error("pred_mask dtype code %d unsupported (0: int64, 1: float32)", f.mask_dtype);
        return PFN_EINVAL;
    }
    PackArgs pa;
    if (slot_ea) pa.slot_ea = *slot_ea;
    pa.njobs = std::min(njobs, PACK_MAX_JOBS);
    pa.rng_advance = rng_advance;
    pa.stamp = stamp;
    pa.stamp_value = stamp_value;
    pa.mask = nullptr;
    pa.maskf = nullptr;
    pa.mask_count = 0;
    pa.mask_dtype = 0;
    long biggest = 0;
    for (int j = 0; j < pa.njobs; ++j) {
        pa.job[j] = jobs[j];
        biggest = std::max<long>(biggest, (long)packed_floats(jobs[j].K, jobs[j].ld_out));
    }
    (void)biggest;
    const int nseg = p.nblocks * p.ny;
    const int pack_bx = pa.njobs == 0 ? 0 : nseg >= 4 * pa.njobs ? -(nseg / pa.njobs) : 8;
    static std::atomic<uint64_t> raised{0};
    PFN_TRY(ensure_dynamic_lds(reinterpret_cast<const void*>(front_seg_fwd_kernel), SG_LDS_BYTES, raised));
    ProfScope ps("front_seg_fwd+pack", 0.0, 0.0, s);
    const int nblocks = nseg + (pack_bx > 0 ? pack_bx * pa.njobs : 0);
    front_seg_fwd_kernel<<<nblocks, SG_THREADS, seg_lds_bytes(p.trows, p.rows_pb, p.cap, false), s>>>(
        f, pa, p.nblocks, p.ny, pack_bx, p.rows_pb, p.trows, p.cap, g.e_stored, g.rowptr_in, g.in_src, g.in_eid, ea, S, ld);
    PFN_CHECK_LAUNCH();
    if (njobs > pa.njobs) return launch_pack(jobs + pa.njobs, njobs - pa.njobs, nullptr, s);   // (deep networks: > 64 weights)
    return PFN_OK;
}

int launch_ea_seg_fwd(const GraphView& g, const EaSegFwdArgs& a, int seg, hipStream_t s) {
    SegPlan p;
    if (!seg_plan(seg, g.n, a.ld, p, false)) {
        set_error("ea_seg_fwd: %d-row graphs do not fit", seg);
        return PFN_EINVAL;
    }
    static std::atomic<uint64_t> raised{0};
    PFN_TRY(ensure_dynamic_lds(reinterpret_cast<const void*>(ea_seg_fwd_kernel), SG_LDS_BYTES, raised));
    ProfScope ps("ea_seg_fwd", 0.0, 4.0 * g.n * (double)a.K * a.h, s);   // (the P | Q GEMM's flops; the walk is LDS work)
    ea_seg_fwd_kernel<<<dim3(p.nblocks, p.ny), SG_THREADS, seg_lds_bytes(p.trows, p.rows_pb, p.cap, false), s>>>(
        g.n, p.rows_pb, p.trows, p.cap, g.rowptr_in, g.in_src, a);
    PFN_CHECK_LAUNCH();
    return PFN_OK;
}

int launch_ea_seg_bwd(const GraphView& g, const EaSegBwdArgs& a, int seg, hipStream_t s) {
    SegPlan p;
    if (!seg_plan(seg, g.n, a.ld, p)) {
        set_error("ea_seg_bwd: %d-row graphs do not fit", seg);
        return PFN_EINVAL;
    }
    const size_t lds = seg_lds_bytes(p.trows, p.rows_pb, p.cap, true);
    const bool dsg = a.Bd == nullptr;
    static std::atomic<uint64_t> raised0{0}, raised1{0}, raised2{0};
    if (a.mse.y && !(dsg && a.ld / 4 <= SG_W2A_CH && a.mse.S && a.mse.b2 && a.mse.deg && a.mse.out && a.mse.gout && a.mse.partial &&
                     a.mse.counter && a.mse.loss && (!a.mse.maskf || (a.mse.counts && a.mse.count_blocks == p.nblocks)))) {
        set_error("ea_seg_bwd: the MSELoss tail needs the last layer's form and every MseTail pointer (internal)");
        return PFN_EINVAL;
    }
    ProfScope ps(a.mse.y ? (a.mse.maskf ? "ea_seg_bwd+out+masked_l2" : "ea_seg_bwd+out+mse") : "ea_seg_bwd", 0.0, a.Bd ? 2.0 * g.n * (double)a.fo * a.h : 0.0, s);
    if (a.mse.y) {
        PFN_TRY(ensure_dynamic_lds(reinterpret_cast<const void*>(ea_seg_bwd_kernel<true, true>), SG_LDS_BYTES, raised2));
        ea_seg_bwd_kernel<true, true><<<dim3(p.nblocks, p.ny), SG_THREADS, lds, s>>>(g.n, p.rows_pb, p.trows, p.cap, g.rowptr_in, g.in_src,
                                                                                   g.rowptr_out, g.out_dst, a);
    } else if (dsg) {
        PFN_TRY(ensure_dynamic_lds(reinterpret_cast<const void*>(ea_seg_bwd_kernel<true>), SG_LDS_BYTES, raised1));
        ea_seg_bwd_kernel<true><<<dim3(p.nblocks, p.ny), SG_THREADS, lds, s>>>(g.n, p.rows_pb, p.trows, p.cap, g.rowptr_in, g.in_src,
                                                                             g.rowptr_out, g.out_dst, a);
    } else {
        PFN_TRY(ensure_dynamic_lds(reinterpret_cast<const void*>(ea_seg_bwd_kernel<false>), SG_LDS_BYTES, raised0));
        ea_seg_bwd_kernel<false><<<dim3(p.nblocks, p.ny), SG_THREADS, lds, s>>>(g.n, p.rows_pb, p.trows, p.cap, g.rowptr_in, g.in_src,
                                                                              g.rowptr_out, g.out_dst, a);
    }
    PFN_CHECK_LAUNCH();
    return PFN_OK;
}

}  // namespace pfn

