// A per-node Linear and the K TAGConv hops that consume its output, for batches of SMALL graphs, in ONE launch (gfx950).
//
// In the reference's E -> act -> T -> act loop (networks/MPN.py:541-547) every EdgeAggregation is followed by a TAGConv, whose
// first act is K normalised hops x^(k) = A_hat x^(k-1) (PyG TAGConv.propagate); in the backward pass every EdgeAggregation's input
// gradient dx = dP W1i + dQ W1j is the output gradient of the TAGConv before it, whose backward first hops it over A_hat^T
// (model.hip tag_backward).  A hop acts on every column independently and never leaves a graph, so a workgroup that owns the
// rows of a whole graph AND one 32-column quarter can form its quarter of the Linear (one 32 x 32 MFMA tile per wave and term,
// gemm_nt's packed images and k order), apply the epilogue, and run the K hops on the tile right there in LDS:
//
//   forward   y = act(S W2^T + deg b2)        ; x^(k) = A_hat x^(k-1),      k = 1..K    (was gemm_nt + fused_hops: two launches)
//   backward  g = (dP W1i + dQ W1j) [y_in > 0]; g^(k) = A_hat^T g^(k-1),    k = 1..K    (was gemm_nt + fused_hops)
//
// y / g and every hop result are written out (they are GEMM operands of the next launch and saved for the weight gradients).
// Same arithmetic in the same order as the kernels it replaces -- gemm_nt_kernel's MFMA k order, term order and trailing-column
// chains, its epilogue expressions, fused_hops_kernel's edge order -- so outputs and gradients are BIT-IDENTICAL to the two-launch
// path (tests/test_gpu_parity.py::test_fused_linear_hops_are_bit_identical_to_two_launches; PFN_NO_SEG_LIN_HOPS=1 gives the latter).
#include <stdlib.h>

#include <algorithm>

#include "pfn_internal.hpp"
#include "seg_tile.hpp"

namespace pfn {

__device__ __forceinline__ float4 slh_sel4(bool k, float4 a, float4 b) {   // (element-wise: a ?: on the struct goes through scratch)
    return make_float4(k ? a.x : b.x, k ? a.y : b.y, k ? a.z : b.z, k ? a.w : b.w);
}

constexpr int SLH_REM_FLOATS = SG_NCH * 32;   // trailing-column image of one term: 34 k groups x [4 columns][4 k's]

struct SlhLds {
    float* t0;      // [trows][SG_TW]: the Linear's tile, then hop ping
    float* t1;      // hop pong -- ALIASES the weight images (dead once the tiles are multiplied)
    float* B[2];    // [SG_NCH * 256] weight quarter per term
    float* R[2];    // [SLH_REM_FLOATS] trailing-column image per term
    float* bias;    // [SG_TW] rowbias of the slice's columns (zero past ncols)
    float* dinv;    // [rows_pb]
    float* rsc;     // [rows_pb] rowscale
    float* rq;      // [4 row tiles][64 lanes]: the trailing column's half-chains between the terms
    int* rp;        // [rows_pb + 1]
    int* nb;        // [cap]
};
__host__ __device__ inline size_t slh_union_floats(int trows, int nterm) {
    const size_t tile = (size_t)trows * SG_TW, img = (size_t)nterm * SG_NCH * 256;
    return tile > img ? tile : img;
}
__device__ __forceinline__ SlhLds slh_lds(float* base, int trows, int rows_pb, int nterm) {
    SlhLds l;
    float* p = base;
    l.t0 = p; p += (size_t)trows * SG_TW;
    l.t1 = p;
    l.B[0] = p;
    l.B[1] = p + SG_NCH * 256;
    p += slh_union_floats(trows, nterm);
    l.R[0] = p; p += SLH_REM_FLOATS;
    l.R[1] = p; if (nterm > 1) p += SLH_REM_FLOATS;
    l.bias = p; p += SG_TW;
    l.dinv = p; p += rows_pb;
    l.rsc = p; p += rows_pb;
    l.rq = p; p += 256;
    l.rp = reinterpret_cast<int*>(p);
    l.nb = l.rp + rows_pb + 1;
    return l;
}
static size_t slh_lds_bytes(int trows, int rows_pb, int cap, int nterm) {
    return ((size_t)trows * SG_TW + slh_union_floats(trows, nterm) + (size_t)nterm * SLH_REM_FLOATS + SG_TW + 2 * (size_t)rows_pb + 256 +
            (size_t)rows_pb + 1 + cap) * 4 + 16;
}

// One term's 32 x 32 tile on top of `acc` (gemm_nt's chunk / step order: chunk m, step i, lane half kh supplies k = 8m + 4kh + i;
// the last chunk of K = 129 carries ONE real step, LS = 1, like gemm_nt's variant 0), plus -- REM -- the trailing column's
// chain off the same fragment (racc: this lane half's k's; the halves are added after the last term, as gemm_nt's flush does).
template <int M>
__device__ __forceinline__ void slh_wait(const SegA& t, int m) {   // seg_wait_chunk<m> for the unrolled loop's constant m
    if constexpr (M < SG_NCH) {
        if (m == M) seg_wait_chunk<M>(t);
        else slh_wait<M + 1>(t, m);
    }
}
template <int LS, bool REM, bool ASYNC>   // ASYNC: the fragment was requested with seg_load_a_async (seg_tile.hpp): waited for chunk by chunk
__device__ __forceinline__ void slh_mma(f32x16& acc, float& racc, const SegA& t, const float* bl, const float* rl, int lane) {
    const int r32 = lane & 31, kh = lane >> 5;
    const float* bp = bl + kh * 128 + r32 * 4;
    const float* rp = rl + kh * 16;
    f32x4 b = *reinterpret_cast<const f32x4*>(bp);
    f32x4 r = f32x4{0.f, 0.f, 0.f, 0.f};
    if (REM) r = *reinterpret_cast<const f32x4*>(rp);
#pragma unroll
    for (int m = 0; m < SG_NCH; ++m) {
        f32x4 bn = b, rn = r;
        if (m + 1 < SG_NCH) {
            bn = *reinterpret_cast<const f32x4*>(bp + (m + 1) * 256);
            if (REM) rn = *reinterpret_cast<const f32x4*>(rp + (m + 1) * 32);
        }
        if (ASYNC) slh_wait<0>(t, m);
#pragma unroll
        for (int i = 0; i < (m == SG_NCH - 1 ? LS : 4); ++i) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(t.av[m][i], b[i], acc, 0, 0, 0);
            if (REM) racc = fmaf(t.av[m][i], r[i], racc);
        }
        b = bn;
        r = rn;
        // (left alone, the scheduler sinks the trailing column's 65 fmas and their LDS reads BEHIND the last MFMA as one serial
        //  chain: +2 us on the block that owns the column -- the launch's critical path; measured with phase timestamps)
        if (REM) __builtin_amdgcn_sched_barrier(0);
    }
}
// the raw accumulator tile of one wave <-> LDS (register 4g + e of lane (r32, kh) = row 8g + 4kh + e, column r32)
__device__ __forceinline__ void slh_load_tile(f32x16& acc, const float* tile, int trow0, int lane) {
    const int r32 = lane & 31, kh = lane >> 5;
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = tile[(size_t)(trow0 + 8 * (j >> 2) + 4 * kh + (j & 3)) * SG_TW + r32];
}

template <int NTERM>
__global__ __launch_bounds__(SG_THREADS) __attribute__((amdgpu_waves_per_eu(4, 4)))
void seg_lin_hops_kernel(int n, int rows_pb, int trows, int cap, const int* __restrict__ rowptr, const int* __restrict__ nbr,
                         const float* __restrict__ dinv, const SegLinHopsArgs a) {
    extern __shared__ __attribute__((aligned(16))) float slh_smem[];
    const SlhLds l = slh_lds(slh_smem, trows, rows_pb, NTERM);
    const int r0 = blockIdx.x * rows_pb, rows = min(rows_pb, n - r0);
    const SegCols sc = seg_cols(a.ld);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // one 32 x 32 MFMA task per wave: row tile (wave & 3) of term (wave >> 2); rows <= SG_MAX_ROWS = 128: four row tiles at most
    const int nrt = (rows + 31) >> 5;
    const int mtile = wave & 3, mterm = wave >> 2;
    const bool mfma_on = mtile < nrt && mterm < NTERM;
    const int K8 = 8 * SG_NCH;                       // (the launcher admits K8 == 136 only: straight-line multiply)
    // ---- prologue: EVERY global load is requested before the first LDS store (cf. ea_seg.hip)
    SegA ta;
    // (the fragment: behind the staging, consumed while it arrives -- below)
    seg_copy_b(l.B[0], a.B0, sc.q, K8, wave, lane);
    if (NTERM > 1) seg_copy_b(l.B[1], a.B1, sc.q, K8, wave, lane);
    const int e0 = rowptr[r0], ne = rowptr[r0 + rows] - e0;
    const bool nb_in_lds = ne <= cap;
    const int rpv = tid <= rows ? rowptr[r0 + tid] : 0;
    const float dv = tid < rows ? dinv[r0 + tid] : 0.f;
    const float rsv = (a.rowscale && tid < rows) ? a.rowscale[r0 + tid] : 0.f;
    float4 rem0 = make_float4(0.f, 0.f, 0.f, 0.f), rem1 = rem0;
    if (sc.rem && tid < SLH_REM_FLOATS / 4) {       // trailing-column images: behind the nq quarters of the packed image
        const size_t roff = (size_t)sc.nq * (K8 >> 2) * 128;
        rem0 = sg_ld4(a.B0 + roff + tid * 4);
        if (NTERM > 1) rem1 = sg_ld4(a.B1 + roff + tid * 4);
    }
    float bv = 0.f;
    if (tid < SG_TW) {
        const int col = seg_col_of_tile(sc, tid);
        bv = (a.rowbias && col >= 0 && col < a.ncols) ? a.rowbias[col] : 0.f;
    }
    const int nitems = rows * sc.cw;                // <= 128 * 9 = 1152: three per thread at most
    float4 gv[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int it = tid + j * SG_THREADS;
        gv[j] = make_float4(1.f, 1.f, 1.f, 1.f);
        if (a.gate && it < nitems) {
            const int lr = it / sc.cw, lc = it - lr * sc.cw;
            gv[j] = sg_ld4(a.gate + (size_t)(r0 + lr) * a.ldg + seg_gcol(sc, lc));
        }
    }
    const int nbv = (nb_in_lds && tid < ne) ? nbr[e0 + tid] : 0;   // (second level: needs e0)
    if (tid <= rows) l.rp[tid] = rpv - e0;
    if (tid < rows) {
        l.dinv[tid] = dv;
        l.rsc[tid] = rsv;
    }
    if (nb_in_lds && tid < ne) l.nb[tid] = nbv - r0;
    if (tid < SG_TW) l.bias[tid] = bv;
    if (sc.rem && tid < SLH_REM_FLOATS / 4) {
        sg_st4(l.R[0] + tid * 4, rem0);
        if (NTERM > 1) sg_st4(l.R[1] + tid * 4, rem1);
    }
    // the gate is only ever asked `> 0`: four bits per item instead of four registers through the multiply phase
    unsigned gbits = 0u;
#pragma unroll
    for (int j = 0; j < 3; ++j)
        gbits |= ((gv[j].x > 0.f ? 1u : 0u) | (gv[j].y > 0.f ? 2u : 0u) | (gv[j].z > 0.f ? 4u : 0u) | (gv[j].w > 0.f ? 8u : 0u)) << (4 * j);
    // the operand fragment goes out NOW -- staging consumed, weight DMAs landed -- and stays in flight across the barrier: the
    // multiply runs while it arrives (protocol: seg_tile.hpp seg_load_a_async)
    seg_drain_visible();
    if (mfma_on) seg_load_a_async(ta, (NTERM > 1 && mterm) ? a.A1 : a.A0, a.lda, r0 + 32 * mtile, r0 + rows - 1, lane);
    seg_lds_barrier();
    // ---- the Linear's tiles.  The terms of a tile go into ONE accumulator chain in term order (gemm_nt's order: bit-identical
    // sums): the wave that owns (tile, term 1) takes over the accumulators -- and the trailing column's two half-chains -- that
    // the wave of (tile, term 0) leaves in LDS.  (One wave running both terms needs the second fragment refilled in place under
    // hand-counted waits: 128 VGPRs, 55 of them spilled into the multiply loop.)
    for (int term = 0; term < NTERM; ++term) {
        if (mfma_on && mterm == term) {
            f32x16 acc;
            float racc = 0.f;
            if (term == 0) {
#pragma unroll
                for (int j = 0; j < 16; ++j) acc[j] = 0.f;
            } else {
                slh_load_tile(acc, l.t0, 32 * mtile, lane);
                if (sc.rem) racc = l.rq[mtile * 64 + lane];
            }
            if (sc.rem) slh_mma<1, true, true>(acc, racc, ta, l.B[term], l.R[term], lane);
            else slh_mma<1, false, true>(acc, racc, ta, l.B[term], l.R[term], lane);
            seg_store_tile(acc, sc.q, nullptr, a.ncols, l.t0, 32 * mtile, lane);
            if (sc.rem) {
                if (term + 1 < NTERM) {
                    l.rq[mtile * 64 + lane] = racc;
                } else {
                    const float tot = racc + __shfl_xor(racc, 32);           // the two k halves
                    if (lane < 32) sg_st4(l.t0 + (size_t)(32 * mtile + lane) * SG_TW + 32, make_float4(tot, 0.f, 0.f, 0.f));
                }
            }
        }
        __syncthreads();
    }
    // ---- epilogue, item = (row, float4 chunk): gemm_nt's expressions element for element; result -> y and back into the tile
    DropKey dk = DropKey{0u, 0u, 0u, 0u};
    float keep_scale = 1.f;
    if (a.act == ACT_DROPOUT_RELU) {
        dk = drop_key(a.rng[0], a.rng[1], a.rng_stream);
        keep_scale = 1.0f / (1.0f - a.p_drop);
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int it = tid + j * SG_THREADS;
        if (it < nitems) {
            const int lr = it / sc.cw, lc = it - lr * sc.cw;
            const int tc = seg_tcol(sc, lc), gc = seg_gcol(sc, lc);
            const float4 v4 = sg_ld4(l.t0 + (size_t)lr * SG_TW + tc);
            float v[4] = {v4.x, v4.y, v4.z, v4.w};
            if (a.rowscale) {
                const float rs = l.rsc[lr];
                const float4 cb = sg_ld4(l.bias + tc);
                v[0] = fmaf(rs, cb.x, v[0]); v[1] = fmaf(rs, cb.y, v[1]); v[2] = fmaf(rs, cb.z, v[2]); v[3] = fmaf(rs, cb.w, v[3]);
            }
            if (a.act == ACT_RELU) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
            } else if (a.act == ACT_DROPOUT_RELU) {
                float u[4];
                dropout_uniform4(dk, (uint32_t)(r0 + lr), (uint32_t)(gc >> 2), u);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = (u[e] >= a.p_drop && v[e] > 0.f) ? v[e] * keep_scale : 0.f;
            }
            if (a.gate) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = ((gbits >> (4 * j + e)) & 1u) ? v[e] * a.gate_scale : 0.f;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = gc + e < a.ncols ? v[e] : 0.f;   // (pad columns stay zero: the layout invariant)
            const float4 o4 = make_float4(v[0], v[1], v[2], v[3]);
            sg_st4(l.t0 + (size_t)lr * SG_TW + tc, o4);
            sg_st4_wt(a.y + (size_t)(r0 + lr) * a.ld + gc, o4);
        }
    }
    seg_lds_barrier();   // (not __syncthreads(): the y stores drain while the hops run)
    // ---- K hops, ping-pong between the two tiles (fused_hops_kernel's walk: four slots per trip, edge-id order).  A row's first
    // four slots -- all of most rows of a power grid -- are planned ONCE for the K hops: tile offsets of the neighbour rows and the
    // edge weights dinv[src] * dinv[dst] in registers, so a hop is four independent tile reads and four fmas per item instead of
    // a chain of dependent LDS reads (row pointer -> index -> weight | row) per hop: 5-6 us of a launch were three such hops.
    // Items past the end are clamped to the last item (branch-free: the three items' reads interleave) and not stored.
    // (The tiles are addressed as 32-bit float offsets into the block's LDS: two swapped `float*` are flat 64-bit addresses.)
    uint32_t cur = 0, nxt = (uint32_t)(trows * SG_TW);
    // one item's hop, unplanned (the third item of the few threads that have one, and blocks whose indices are not in LDS)
    auto walk_item = [&](int it, float* gout, bool last) {
        const int lr = it / sc.cw, lc = it - lr * sc.cw;
        const int tc = seg_tcol(sc, lc), gc = seg_gcol(sc, lc);
        const float di = l.dinv[lr];
        const int beg = l.rp[lr], end = l.rp[lr + 1], lastp = end - 1;
        float4 h = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int p = beg; p < end; p += 4) {
            int s_[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) s_[u] = nb_in_lds ? l.nb[min(p + u, lastp)] : nbr[e0 + min(p + u, lastp)] - r0;
            float w_[4];
            float4 x_[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                w_[u] = l.dinv[s_[u]] * di;
                x_[u] = sg_ld4(slh_smem + cur + (uint32_t)(s_[u] * SG_TW + tc));
            }
            h = sg_fma4(w_[0], x_[0], h);
            h = slh_sel4(p + 1 < end, sg_fma4(w_[1], x_[1], h), h);
            h = slh_sel4(p + 2 < end, sg_fma4(w_[2], x_[2], h), h);
            h = slh_sel4(p + 3 < end, sg_fma4(w_[3], x_[3], h), h);
        }
        if (!last) sg_st4(slh_smem + nxt + (uint32_t)(lr * SG_TW + tc), h);
        sg_st4_wt(gout + (size_t)(r0 + lr) * a.ld + gc, h);
    };
    if (nb_in_lds) {
        uint32_t h_to[2], h_go[2], h_so[2][4];
        int h_tc[2], h_beg[2], h_cnt[2];
        float h_di[2], h_w[2][4];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int it = min(tid + j * SG_THREADS, nitems - 1);
            const int lr = it / sc.cw, lc = it - lr * sc.cw;
            h_tc[j] = seg_tcol(sc, lc);
            h_to[j] = (uint32_t)(lr * SG_TW + h_tc[j]);
            h_go[j] = (uint32_t)((r0 + lr) * a.ld + seg_gcol(sc, lc));
            h_beg[j] = l.rp[lr];
            h_cnt[j] = l.rp[lr + 1] - h_beg[j];
            h_di[j] = l.dinv[lr];
#pragma unroll
            for (int u = 0; u < 4; ++u) {   // (slots past the row's end re-read its last edge; an empty row reads slot 0 of the block, unused)
                const int sv = l.nb[max(h_beg[j] + min(u, h_cnt[j] - 1), 0)];
                h_so[j][u] = (uint32_t)(sv * SG_TW + h_tc[j]);
                h_w[j][u] = l.dinv[sv] * h_di[j];
            }
        }
        for (int k = 1; k <= a.nhops; ++k) {
            const bool last = k == a.nhops;
            float* gout = a.xk + (size_t)(k - 1) * a.stride;
            float4 v_[2][4];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int u = 0; u < 4; ++u) v_[j][u] = sg_ld4(slh_smem + cur + h_so[j][u]);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
                float4 h = slh_sel4(h_cnt[j] > 0, sg_fma4(h_w[j][0], v_[j][0], z), z);
                h = slh_sel4(h_cnt[j] > 1, sg_fma4(h_w[j][1], v_[j][1], h), h);
                h = slh_sel4(h_cnt[j] > 2, sg_fma4(h_w[j][2], v_[j][2], h), h);
                h = slh_sel4(h_cnt[j] > 3, sg_fma4(h_w[j][3], v_[j][3], h), h);
                if (h_cnt[j] > 4) {   // the rest of a longer row: the generic walk
                    const int end = h_beg[j] + h_cnt[j], lastp = end - 1;
                    for (int p = h_beg[j] + 4; p < end; p += 4) {
                        int s_[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) s_[u] = l.nb[min(p + u, lastp)];
                        float w_[4];
                        float4 x_[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            w_[u] = l.dinv[s_[u]] * h_di[j];
                            x_[u] = sg_ld4(slh_smem + cur + (uint32_t)(s_[u] * SG_TW + h_tc[j]));
                        }
                        h = sg_fma4(w_[0], x_[0], h);
                        h = slh_sel4(p + 1 < end, sg_fma4(w_[1], x_[1], h), h);
                        h = slh_sel4(p + 2 < end, sg_fma4(w_[2], x_[2], h), h);
                        h = slh_sel4(p + 3 < end, sg_fma4(w_[3], x_[3], h), h);
                    }
                }
                if (tid + j * SG_THREADS < nitems) {
                    if (!last) sg_st4(slh_smem + nxt + h_to[j], h);
                    sg_st4_wt(gout + h_go[j], h);
                }
            }
            for (int it = tid + 2 * SG_THREADS; it < nitems; it += SG_THREADS) walk_item(it, gout, last);
            seg_lds_barrier();
            const uint32_t t = cur;
            cur = nxt;
            nxt = t;
        }
    } else {   // a block with more edges than its LDS slice holds: indices from global memory, hop by hop
        for (int k = 1; k <= a.nhops; ++k) {
            float* gout = a.xk + (size_t)(k - 1) * a.stride;
            for (int it = tid; it < nitems; it += SG_THREADS) walk_item(it, gout, k == a.nhops);
            seg_lds_barrier();
            const uint32_t t = cur;
            cur = nxt;
            nxt = t;
        }
    }
}

// ------------------------------------------------------------------------------------------------ host
struct SlhPlan { int rows_pb, trows, cap, nblocks, ny; };
static bool slh_plan(int seg, int n, int ld, int nterm, SlhPlan& p) {
    if (seg <= 0 || seg > SG_MAX_ROWS || n <= 0 || n % seg != 0) return false;
    const int gpb = std::max(1, SG_MAX_ROWS / seg);
    p.rows_pb = gpb * seg;
    p.trows = (p.rows_pb + 31) / 32 * 32;
    p.cap = std::min(p.rows_pb * 4, SG_THREADS);          // (one slot per thread in the prologue)
    p.nblocks = (n + p.rows_pb - 1) / p.rows_pb;
    int remv, nq;
    col_plan(ld, remv, nq);
    p.ny = nq;
    return p.ny >= 1 && slh_lds_bytes(p.trows, p.rows_pb, p.cap, nterm) <= (size_t)SG_LDS_BYTES;
}

// Same regime as the graph-resident EdgeAggregation kernels (ea_seg_fit): a few workgroups per CU, where a launch is one tile and
// one walk deep and every launch removed is worth its fixed cost; K = H = 129-shaped operands only (one 136-k piece, one real
// step in the last chunk, one trailing column: what the MaskEmbdMultiMPN layers between a TAGConv and an EdgeAggregation are)
bool seg_lin_hops_fit(int seg, int n, int ld, int K, int ncols, int nhops, int nterm) {
    static const bool off = diag_env("PFN_NO_SEG_LIN_HOPS") != nullptr;   // A/B switch: gemm_nt + fused_hops, two launches
    const long per_cu = 4L;   // (ea_seg_fit's bound)
    SlhPlan p;
    int remv, nq;
    col_plan(ld, remv, nq);
    return !off && nhops > 0 && nterm >= 1 && nterm <= 2 && ((K + 7) & ~7) == 8 * SG_NCH && K - (8 * SG_NCH - 8) == 1 &&
           ncols <= ld && ld == ld_of(ncols) && remv == 4 && ncols - 32 * nq == 1 && fused_hops_fit(seg, ld, n) &&
           slh_plan(seg, n, ld, nterm, p) && (long)p.nblocks * p.ny <= per_cu * device_cus();
}

int launch_seg_lin_hops(const GraphView& g, const SegLinHopsArgs& a, int seg, hipStream_t s) {
    const int nterm = a.A1 ? 2 : 1;
    SlhPlan p;
    if (!slh_plan(seg, g.n, a.ld, nterm, p)) {
        set_error("seg_lin_hops: %d-row graphs do not fit", seg);
        return PFN_EINVAL;
    }
    const size_t lds = slh_lds_bytes(p.trows, p.rows_pb, p.cap, nterm);
    const int* rp = a.adjt ? g.rowptr_out : g.rowptr_in;
    const int* nb = a.adjt ? g.out_dst : g.in_src;
    static std::atomic<uint64_t> raised1{0}, raised2{0};
    ProfScope ps(a.adjt ? "seg_lin_hops_bwd" : "seg_lin_hops_fwd", 0.0, 2.0 * g.n * (double)a.K * a.ncols * nterm, s);
    if (nterm == 1) {
        PFN_TRY(ensure_dynamic_lds(reinterpret_cast<const void*>(seg_lin_hops_kernel<1>), SG_LDS_BYTES, raised1));
        seg_lin_hops_kernel<1><<<dim3(p.nblocks, p.ny), SG_THREADS, lds, s>>>(g.n, p.rows_pb, p.trows, p.cap, rp, nb, g.dinv, a);
    } else {
        PFN_TRY(ensure_dynamic_lds(reinterpret_cast<const void*>(seg_lin_hops_kernel<2>), SG_LDS_BYTES, raised2));
        seg_lin_hops_kernel<2><<<dim3(p.nblocks, p.ny), SG_THREADS, lds, s>>>(g.n, p.rows_pb, p.trows, p.cap, rp, nb, g.dinv, a);
    }
    PFN_CHECK_LAUNCH();
    return PFN_OK;
}

}  // namespace pfn

