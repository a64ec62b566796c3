// The whole E -> act -> T -> act -> E chain of a batch of SMALL graphs in ONE persistent launch per direction (gfx950).
//
// networks/MPN.py:541-547 is a loop  x = act(EdgeAggregation(x)) ; x = act(TAGConv(x)) ; ...  In this library every link of that
// loop is either ROW-local (a per-node Linear: needs all columns of a row) or COLUMN-local and graph-local (an edge walk, a
// TAGConv hop: needs all rows of a graph, one column at a time).  Rounds 2-4 ran the loop as graph-resident launches whose
// workgroups own (whole graph, one 32-column quarter) -- ea_seg.hip, seg_lin_hops.hip -- plus gemm_nt for the TAGConv product,
// with a KERNEL BOUNDARY wherever a Linear needs the columns the other three quarter-workgroups of its graph produced: three
// launches per (E, T) pair and direction, each re-staging the same 118-node adjacency, re-fetching weights, and running its
// phases in lockstep across one round of 512 workgroups (profiles/r04: 13 such launches = 46 % of the config-2 step at an
// MFMA-busy of 0.16-0.32).
//
// Here the boundary is a PER-GRAPH barrier instead: the four quarter-workgroups of a graph (placed on ONE XCD, same L2) hand
// their columns to each other through memory -- write-through (sc1) stores, every storing wave drained, ONE agent-scope
// arrival per workgroup on the graph's counter, the consumers poll that one word and read with sc1 loads
// (MI355X_MICROARCH.md "inter-workgroup visibility"; cdna_hip_programming.md Guideline 16 R1) -- and a workgroup walks the whole
// chain without leaving its CU:
//
//   forward, per (E_i second half, T_{i+1}, E_{i+2} first half):
//     A  y = act(S W2^T + deg b2) ; x^(k) = A_hat x^(k-1), k = 1..K            (was seg_lin_hops_kernel<1>)
//        -- graph barrier --
//     B  h = act(sum_k x^(k) W_k^T + bias)                                       (was gemm_nt_kernel, 4 terms)
//        -- graph barrier --
//     C  P | Q = h W1i^T + b1 | h W1j^T ; S = sum_{e -> i} relu(P_i + Q_src + a_e We)   (was ea_seg_fwd_kernel)
//        -- graph barrier --   (next pair)
//
// The adjacency slice is staged ONCE per launch, the weight quarters of the next phase are LDS-DMA'd while the current phase
// computes, graphs drift apart instead of marching in lockstep (one workgroup's hops overlap its CU partner's MFMA stream), and
// nine launches become one.  Arithmetic, k order, term order, edge order and epilogue expressions are those of the kernels
// replaced (gemm_nt's chunk / step order with the one-step tail of K = 129, the trailing column as two half-chains off the MFMA
// waves' own fragments, fused_hops_kernel's / ea_seg's slot order), so every output is BIT-IDENTICAL to the multi-launch path
// (tests/test_gpu_parity.py::test_chain_kernels_are_bit_identical_to_the_launch_per_phase_path; the chain is
// opt-in, PFN_SEG_CHAIN=1).  Shape: hidden 129 (four MFMA quarters + one trailing column), K = 3, <= 120 rows of whole graphs per workgroup.
//
// LDS (80 KiB per workgroup, two per CU): four 17-KiB SLOTS that are weight-quarter images or 120 x 36 tiles by turns, two
// trailing-column images, one 4-KiB region that is two more of those (phase B) or the walk's edge attributes (phase C), and the
// adjacency slice.  Residency: all workgroups of the launch must be co-resident (they wait for each other): the launcher
// checks grid <= 2 x CUs and the occupancy query; every spin is bounded, a timed-out workgroup poisons its output with NaN.
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <type_traits>

#include "pfn_internal.hpp"
#include "seg_tile.hpp"

namespace pfn {

constexpr int CH_SLOT = SG_NCH * 256;                 // 4352 floats: one weight-quarter image, or a tile of <= 120 rows
constexpr int CH_MAX_ROWS = CH_SLOT / SG_TW;          // 120
constexpr int CH_REM = SG_NCH * 32;                   // 544 floats: a trailing-column image
constexpr int CH_U = 2 * CH_REM;                      // region U: two trailing-column images | edge attributes + residue weights
constexpr int CH_CAP = 480;                           // adjacency slots staged per workgroup (>= 4 per row)
constexpr int CH_LDS_FLOATS = 4 * CH_SLOT + 2 * CH_REM + CH_U + SG_TW + CH_MAX_ROWS + (CH_MAX_ROWS + 4) + CH_CAP;
constexpr int CH_LDS_BYTES = CH_LDS_FLOATS * 4 + 16;   // (+ the hand-off flag word)
static_assert(CH_LDS_BYTES <= 80 * 1024, "two chain workgroups per CU");
static_assert(2 * CH_CAP + 2 * SG_TW <= CH_U, "edge attributes + residue weights fit region U");
constexpr unsigned long long CH_SPIN_TICKS = 20ull * 1000 * 1000;   // 0.2 s of the 100 MHz wall clock: a bounded wait

typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));

struct ChainLds {
    float* X[4];
    float* rem[2];
    float* U;
    float* bias;
    float* dinv;
    int* rp;
    int* nb;
    int* flag;     // the polling lane's verdict, broadcast to the workgroup (in the DYNAMIC region: a static __shared__ word in front
                   // of it would cost the region its 16-byte alignment, cdna_hip_programming.md Guideline 17)
};
__device__ __forceinline__ ChainLds chain_lds(float* base) {
    ChainLds l;
    float* p = base;
    for (int i = 0; i < 4; ++i) { l.X[i] = p; p += CH_SLOT; }
    l.rem[0] = p; p += CH_REM;
    l.rem[1] = p; p += CH_REM;
    l.U = p; p += CH_U;
    l.bias = p; p += SG_TW;
    l.dinv = p; p += CH_MAX_ROWS;
    l.rp = reinterpret_cast<int*>(p); p += CH_MAX_ROWS + 4;
    l.nb = reinterpret_cast<int*>(p); p += CH_CAP;
    l.flag = reinterpret_cast<int*>(p);
    return l;
}

template <int M, int END, class F>
__device__ __forceinline__ void ch_static_for(F&& f) {
    if constexpr (M < END) {
        f(std::integral_constant<int, M>{});
        ch_static_for<M + 1, END>(f);
    }
}

// ---- inter-workgroup hand-off primitives
// sc1 (agent-scope) 16-byte load, compiler-visible (hipcc counts its wait): the consumer side of a hand-off whose producer
// stored write-through.  rsrc = a raw buffer descriptor over the whole tensor, built from wave-uniform values.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t ch_rsrc(const float* p, size_t floats) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, (int)std::min<size_t>(floats * 4, 0x7fffffffu), 0x00020000);
}
__device__ __forceinline__ f32x4 ch_ld4(__amdgpu_buffer_rsrc_t r, uint32_t byte_off, int soff = 0) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, soff, 16 /* sc1 */));
}
__device__ __forceinline__ float4 ch_ld4f(__amdgpu_buffer_rsrc_t r, uint32_t byte_off) {
    const f32x4 v = ch_ld4(r, byte_off);
    return make_float4(v[0], v[1], v[2], v[3]);
}
// the same load HIDDEN from hipcc (inline asm), for the streamed multiply below whose waits are counted by hand
template <int OFF>
__device__ __forceinline__ void ch_ld4_asm(f32x4& dst, const char* sbase, uint32_t voff) {
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3 sc1" : "+v"(dst) : "v"(voff), "s"(sbase), "n"(OFF) : "memory");
}
// (first load of a fragment: output only -- as "+v" the registers' previous contents count as an input, i.e. stay alive from
//  the fragment's last use in the stage before: 68 registers carried around the whole stage loop)
template <int OFF>
__device__ __forceinline__ void ch_ld4_asm_first(f32x4& dst, const char* sbase, uint32_t voff) {
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3 sc1" : "=v"(dst) : "v"(voff), "s"(sbase), "n"(OFF) : "memory");
}
template <int N>
__device__ __forceinline__ void ch_wait(f32x4& v) { asm volatile("s_waitcnt vmcnt(%1)" : "+v"(v) : "n"(N)); }
__device__ __forceinline__ void ch_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void ch_bar() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }   // (LDS only: loads stay in flight)

struct ChainSync {
    int* cnt;            // the graph block's arrival counter
    int* status;         // [0]: set to 1 by a workgroup whose wait timed out
    int ny;
    bool failed;
};
// every storing wave has drained (ch_drain) and the workgroup has met at a barrier before this
__device__ __forceinline__ void ch_arrive(const ChainSync& sy) {
    if (threadIdx.x == 0) __hip_atomic_fetch_add(sy.cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// all ny workgroups of the graph block have arrived `epoch` times.  ONE lane polls ONE word, relaxed, with a sleep in between;
// the workgroup then meets at a barrier and reads the handed-over columns with sc1 loads.
__device__ __forceinline__ void ch_wait_graph(ChainSync& sy, int epoch, int* s_flag) {
    if (threadIdx.x == 0) {
        int ok = 1;
        if (!sy.failed) {
            const int want = epoch * sy.ny;
            const unsigned long long t0 = wall_clock64();
            while (__hip_atomic_load(sy.cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
                __builtin_amdgcn_s_sleep(2);
                if (wall_clock64() - t0 > CH_SPIN_TICKS) {
                    ok = 0;
                    __hip_atomic_store(sy.status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
            }
        } else {
            ok = 0;
        }
        *s_flag = ok;
    }
    __syncthreads();
    if (*s_flag == 0) sy.failed = true;
}


// ---- the MFMA tile products
// A fragment of one 32-row tile, K8 = 136: chunk m of lane (r32, kh) = A[row][8m + 4kh .. + 3] (seg_tile.hpp SegA); the last
// chunk's read position is clamped into the row (its k's past 128 multiply zero image rows).
struct ChainFrag {
    uint32_t voff, voff_last;   // the lane's byte offsets: chunks 0..15 sit at voff + 32 m, chunk 16 at voff_last
};
__device__ __forceinline__ ChainFrag ch_frag(int row, int lda, int lane) {
    const int kh = lane >> 5;
    ChainFrag f;
    f.voff = (uint32_t)(((size_t)row * lda + 4 * kh) * 4);
    f.voff_last = (uint32_t)(((size_t)row * lda + min(8 * (SG_NCH - 1) + 4 * kh, lda - 4)) * 4);
    return f;
}
// compiler-visible fragment load (phases A and C: one term, nothing to stream)
__device__ __forceinline__ void ch_load_a(SegA& t, __amdgpu_buffer_rsrc_t r, const ChainFrag& f) {
#pragma unroll
    for (int m = 0; m < SG_NCH; ++m) t.av[m] = m + 1 < SG_NCH ? ch_ld4(r, f.voff, 32 * m) : ch_ld4(r, f.voff_last);
}
// hidden fragment load, chunk order (the streamed multiply counts on exactly these 17 loads, in this order)
__device__ __forceinline__ void ch_load_a_asm(f32x4 (&a)[SG_NCH], const char* base, const ChainFrag& f) {
    ch_static_for<0, SG_NCH>([&](auto mc) {
        constexpr int m = decltype(mc)::value;
        if constexpr (m + 1 < SG_NCH) ch_ld4_asm_first<32 * m>(a[m], base, f.voff);
        else ch_ld4_asm_first<0>(a[m], base, f.voff_last);
    });
}
// One term's 32 x 32 tile on top of `acc`, K = 129 (gemm_nt's variant 0: chunk m, step i, lane half kh supplies k = 8m + 4kh + i;
// the last chunk carries ONE real step), plus -- REM -- the trailing column's chain off the same fragment.  Visible form.
template <bool REM>
__device__ __forceinline__ void ch_mma(f32x16& acc, float& racc, const SegA& t, const float* bl, const float* rl, int lane) {
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
#pragma unroll
        for (int i = 0; i < (m == SG_NCH - 1 ? 1 : 4); ++i) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(t.av[m][i], b[i], acc, 0, 0, 0);
            if (REM) racc = fmaf(t.av[m][i], r[i], racc);
        }
        b = bn;
        r = rn;
        if (REM) __builtin_amdgcn_sched_barrier(0);   // (seg_lin_hops.hip slh_mma: else the trailing column trails the last MFMA)
    }
}
// The same with the fragment STREAMED (phase B's four terms in one accumulator chain): while term t is multiplied, the chunks it
// has consumed are refilled IN PLACE with term t + 1's, four at a time (gemm_nt.hip nt_multiply: a 128-byte line holds four
// consecutive chunks).  Every load is hidden from hipcc and every wait counted by hand: VMEM returns in order, so chunk m of a
// term is complete when at most the loads issued after it are outstanding -- 16 - (m & 3) with refills going out (the 16 - m
// rest of its own term + the m - (m & 3) refills already issued for the next), 16 - m without.  The wave issues NO other
// vector-memory instruction between the first fragment load and the last multiply.
template <bool REM, bool REFILL>
__device__ __forceinline__ void ch_mma_stream(f32x16& acc, float& racc, f32x4 (&a)[SG_NCH], const float* bl, const float* rl, int lane,
                                              const char* nbase, const ChainFrag& nf) {
    const int r32 = lane & 31, kh = lane >> 5;
    const float* bp = bl + kh * 128 + r32 * 4;
    const float* rp = rl + kh * 16;
    f32x4 b_nxt = *reinterpret_cast<const f32x4*>(bp);
    ch_static_for<0, SG_NCH>([&](auto mc) {
        constexpr int m = decltype(mc)::value;
        const f32x4 b = b_nxt;
        // (the trailing column's four weights of this chunk: read now, used behind the chunk's MFMAs -- not pipelined a chunk ahead
        //  like b: four registers the streamed fragment needs)
        f32x4 r = f32x4{0.f, 0.f, 0.f, 0.f};
        if (REM) r = *reinterpret_cast<const f32x4*>(rp + m * 32);
        if (m + 1 < SG_NCH) b_nxt = *reinterpret_cast<const f32x4*>(bp + (m + 1) * 256);
        ch_wait<REFILL ? SG_NCH - 1 - (m & 3) : SG_NCH - 1 - m>(a[m]);
#pragma unroll
        for (int i = 0; i < (m == SG_NCH - 1 ? 1 : 4); ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m][i], b[i], acc, 0, 0, 0);
        if (REM) {
#pragma unroll
            for (int i = 0; i < (m == SG_NCH - 1 ? 1 : 4); ++i) racc = fmaf(a[m][i], r[i], racc);
            // (pinned here: a pure chain with one use at the very end, the optimiser otherwise sinks all 65 fmas of a term behind the
            //  last multiply and keeps COPIES of every fragment chunk for them -- 462 spills)
            asm volatile("" : "+v"(racc));
        }
        if constexpr (REFILL && ((m & 3) == 3 || m == SG_NCH - 1)) {
            ch_static_for<(m & ~3), m + 1>([&](auto mmc) {
                constexpr int mm = decltype(mmc)::value;
                if constexpr (mm + 1 < SG_NCH) ch_ld4_asm<32 * mm>(a[mm], nbase, nf.voff);
                else ch_ld4_asm<0>(a[mm], nbase, nf.voff_last);
            });
        }
        __builtin_amdgcn_sched_barrier(0);
    });
}
// accumulator register 4g + e of lane (r32, kh) = row 8g + 4kh + e, column r32 -> LDS tile (rows past the slot are dropped: a
// slot holds 120 rows, the last row tile reaches 127)
__device__ __forceinline__ void ch_store_tile(const f32x16& acc, int q, const float* __restrict__ bias, int ncols, float* tile,
                                              int trow0, int lane) {
    const int r32 = lane & 31, kh = lane >> 5;
    const int col = 32 * q + r32;
    const float cb = (bias && col < ncols) ? bias[col] : 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int tr = trow0 + 8 * (j >> 2) + 4 * kh + (j & 3);
        if (tr < CH_MAX_ROWS) tile[(size_t)tr * SG_TW + r32] = acc[j] + cb;
    }
}

// residue columns of W1 for the slice (ea_seg.hip we_issue): we[f][tile column] = W1[col][2Fi + f], zero past H; one per thread
__device__ __forceinline__ float we_issue_ch(const SegCols& c, const float* __restrict__ w1, int h, int fi, int i) {
    const int ldw = 2 * fi + 2;
    if (i >= 2 * SG_TW) return 0.f;
    const int f = i / SG_TW, t = i - f * SG_TW;
    const int col = seg_col_of_tile(c, t);
    return (col >= 0 && col < h) ? w1[(size_t)col * ldw + 2 * fi + f] : 0.f;
}

__device__ __forceinline__ float4 ch_sel4(bool k, float4 a, float4 b) {
    return make_float4(k ? a.x : b.x, k ? a.y : b.y, k ? a.z : b.z, k ? a.w : b.w);
}

// ------------------------------------------------------------------------------------------------ forward
struct ChainFwdStage {
    const float* S_in;        // [N][ld]: the edge stage's sums of the EdgeAggregation whose second Linear opens this stage
    const float* w2_img;      // packed image of that W2^T (H -> H)
    const float* b2;
    float* y;                 // act(S W2^T + deg b2): that layer's output (= hop 0)
    float* xk;                // the TAGConv's K hop buffers
    const float* tag_img[4];  // packed images of W_0 .. W_3
    const float* tag_bias;
    float* h;                 // the TAGConv's output (post-activation)
    const float* w1i_img;     // the next EdgeAggregation: images of W1[:, :H]^T, W1[:, H:2H]^T, its b1 and raw W1 (residue columns)
    const float* w1j_img;
    const float* b1;
    const float* w1;
    float* P;
    float* Q;
    float* S_out;
    uint32_t stream_y, stream_h;   // dropout stream ids of the two activations (= layer indices)
};
constexpr int CH_MAX_STAGES = 6;
struct ChainFwdArgs {
    int n, rows_pb, nblocks, ny, ld, h, nhops, nstage, act, store_pq;
    float p_drop;
    const uint64_t* rng;
    const int* rowptr;
    const int* nbr;
    const float* dinv;
    const float* deg;
    const float* ea_in;       // edge attributes in by-destination slot order (SlotEa)
    int* cnt;                 // [nblocks] arrival counters, zero at launch
    int* zero_words;          // optional: nblocks words this launch zeroes for the NEXT chain launch (the backward pass's counters)
    int* status;
    size_t xk_stride;
    ChainFwdStage st[CH_MAX_STAGES];
};

// shape constants of the chain kernels (seg_chain_fit admits nothing else): hidden 129 = four 32-column MFMA quarters + ONE trailing
// column, padded row stride 132, one 136-k piece per term, K = 3 hops (four TAGConv terms)
constexpr int CH_H = 129, CH_LD = 132, CH_NQ = 4, CH_K8 = 8 * SG_NCH, CH_HOPS = 3;
constexpr size_t CH_ROFF = (size_t)CH_NQ * (CH_K8 >> 2) * 128;   // trailing-column image: behind the four quarters of a packed image

// a phase's view of the thread id: opaque, so that what a phase derives from it (item geometry, hop plans, fragment offsets) is
// neither hoisted out of the stage loop nor shared between phases -- either way it would stay alive through the streamed multiply
// of phase B, whose 68-register fragment must not spill (a spilled hidden load stores the register before the data has landed)
__device__ __forceinline__ int ch_tid() {
    int t = threadIdx.x;
    asm volatile("" : "+v"(t));
    return t;
}
struct ChainBlock {      // wave-uniform geometry of the workgroup
    int r0, rows, q, cw, e0, ne;
    bool rem, nb_in_lds;
};
__device__ __forceinline__ int ch_tcol(const ChainBlock& b, int lc) { return (b.rem && lc == b.cw - 1) ? 32 : 4 * lc; }
__device__ __forceinline__ int ch_gcol(const ChainBlock& b, int lc) { return (b.rem && lc == b.cw - 1) ? 32 * CH_NQ : 32 * b.q + 4 * lc; }
// global column of LDS tile column t (0..31: the quarter; 32..35: the trailing columns), -1: not in this block / past H
__device__ __forceinline__ int ch_col_of_tile(const ChainBlock& b, int t) {
    const int c = t < 32 ? 32 * b.q + t : (b.rem ? 32 * CH_NQ + t - 32 : -1);
    return c < CH_H ? c : -1;
}
__device__ __forceinline__ ChainFrag ch_frag_of(const ChainBlock& b, int wave, int lane) {
    return ch_frag(min(b.r0 + 32 * wave + (lane & 31), b.r0 + b.rows - 1), CH_LD, lane);   // (clamped: pad rows of the tile)
}

// epilogue of a Linear on the tile: item = (row, float4 chunk); gemm_nt's expressions element for element
//   ROWSCALE: v = fma(deg[row], bias[col], v)  (EdgeAggregation's second Linear) ; else v += bias[col] (TAGConv)
template <bool ROWSCALE, bool TO_TILE>
__device__ __forceinline__ void ch_epilogue(const ChainBlock& b, const ChainLds& l, float* tile, float* __restrict__ out, const float* __restrict__ deg,
                                            int act, float p_drop, float keep_scale, const DropKey& dk, int tid) {
    const int nitems = b.rows * b.cw;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int it = tid + j * SG_THREADS;
        if (it < nitems) {
            const int lr = it / b.cw, lc = it - lr * b.cw;
            const int tc = ch_tcol(b, lc), gc = ch_gcol(b, lc);
            const float4 v4 = sg_ld4(tile + (size_t)lr * SG_TW + tc);
            const float4 cb = sg_ld4(l.bias + tc);
            float v[4] = {v4.x, v4.y, v4.z, v4.w};
            if (ROWSCALE) {
                const float rs = deg[b.r0 + lr];
                v[0] = fmaf(rs, cb.x, v[0]); v[1] = fmaf(rs, cb.y, v[1]); v[2] = fmaf(rs, cb.z, v[2]); v[3] = fmaf(rs, cb.w, v[3]);
            } else {
                v[0] += cb.x; v[1] += cb.y; v[2] += cb.z; v[3] += cb.w;
            }
            if (act == ACT_RELU) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
            } else if (act == ACT_DROPOUT_RELU) {
                float u[4];
                dropout_uniform4(dk, (uint32_t)(b.r0 + lr), (uint32_t)(gc >> 2), u);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = (u[e] >= p_drop && v[e] > 0.f) ? v[e] * keep_scale : 0.f;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = gc + e < CH_H ? v[e] : 0.f;   // (pad columns stay zero: the layout invariant)
            const float4 o4 = make_float4(v[0], v[1], v[2], v[3]);
            if (TO_TILE) sg_st4(tile + (size_t)lr * SG_TW + tc, o4);
            sg_st4_wt(out + (size_t)(b.r0 + lr) * CH_LD + gc, o4);
        }
    }
}

// K normalised hops on the tile pair (t0 holds hop 0), every hop written to xk + (k - 1) * stride: seg_lin_hops_kernel's walk
// (four slots per trip in edge-id order; a row's first four slots planned once for the K hops)
__device__ __forceinline__ void ch_hops(const ChainBlock& b, const ChainLds& l, float* smem, float* t0, float* t1, float* __restrict__ xk,
                                        size_t stride, const int* __restrict__ nbr, int tid) {
    const int nitems = b.rows * b.cw;
    uint32_t cur = (uint32_t)(t0 - smem), nxt = (uint32_t)(t1 - smem);
    auto walk_item = [&](int it, float* gout, bool last) {
        const int lr = it / b.cw, lc = it - lr * b.cw;
        const int tc = ch_tcol(b, lc), gc = ch_gcol(b, lc);
        const float di = l.dinv[lr];
        const int beg = l.rp[lr], end = l.rp[lr + 1], lastp = end - 1;
        float4 hh = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int p = beg; p < end; p += 4) {
            int s_[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) s_[u] = b.nb_in_lds ? l.nb[min(p + u, lastp)] : nbr[b.e0 + min(p + u, lastp)] - b.r0;
            float w_[4];
            float4 x_[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                w_[u] = l.dinv[s_[u]] * di;
                x_[u] = sg_ld4(smem + cur + (uint32_t)(s_[u] * SG_TW + tc));
            }
            hh = sg_fma4(w_[0], x_[0], hh);
            hh = ch_sel4(p + 1 < end, sg_fma4(w_[1], x_[1], hh), hh);
            hh = ch_sel4(p + 2 < end, sg_fma4(w_[2], x_[2], hh), hh);
            hh = ch_sel4(p + 3 < end, sg_fma4(w_[3], x_[3], hh), hh);
        }
        if (!last) sg_st4(smem + nxt + (uint32_t)(lr * SG_TW + tc), hh);
        sg_st4_wt(gout + (size_t)(b.r0 + lr) * CH_LD + gc, hh);
    };
    if (b.nb_in_lds) {
        uint32_t h_to[2], h_go[2], h_so[2][4];
        int h_tc[2], h_beg[2], h_cnt[2];
        float h_di[2], h_w[2][4];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int it = min(tid + j * SG_THREADS, nitems - 1);
            const int lr = it / b.cw, lc = it - lr * b.cw;
            h_tc[j] = ch_tcol(b, lc);
            h_to[j] = (uint32_t)(lr * SG_TW + h_tc[j]);
            h_go[j] = (uint32_t)((b.r0 + lr) * CH_LD + ch_gcol(b, lc));
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
        for (int k = 1; k <= CH_HOPS; ++k) {
            const bool last = k == CH_HOPS;
            float* gout = xk + (size_t)(k - 1) * stride;
            float4 v_[2][4];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int u = 0; u < 4; ++u) v_[j][u] = sg_ld4(smem + cur + h_so[j][u]);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
                float4 hh = ch_sel4(h_cnt[j] > 0, sg_fma4(h_w[j][0], v_[j][0], z), z);
                hh = ch_sel4(h_cnt[j] > 1, sg_fma4(h_w[j][1], v_[j][1], hh), hh);
                hh = ch_sel4(h_cnt[j] > 2, sg_fma4(h_w[j][2], v_[j][2], hh), hh);
                hh = ch_sel4(h_cnt[j] > 3, sg_fma4(h_w[j][3], v_[j][3], hh), hh);
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
                            x_[u] = sg_ld4(smem + cur + (uint32_t)(s_[u] * SG_TW + h_tc[j]));
                        }
                        hh = sg_fma4(w_[0], x_[0], hh);
                        hh = ch_sel4(p + 1 < end, sg_fma4(w_[1], x_[1], hh), hh);
                        hh = ch_sel4(p + 2 < end, sg_fma4(w_[2], x_[2], hh), hh);
                        hh = ch_sel4(p + 3 < end, sg_fma4(w_[3], x_[3], hh), hh);
                    }
                }
                if (tid + j * SG_THREADS < nitems) {
                    if (!last) sg_st4(smem + nxt + h_to[j], hh);
                    sg_st4_wt(gout + h_go[j], hh);
                }
            }
            for (int it = tid + 2 * SG_THREADS; it < nitems; it += SG_THREADS) walk_item(it, gout, last);
            seg_lds_barrier();
            const uint32_t t = cur;
            cur = nxt;
            nxt = t;
        }
    } else {   // a block with more edges than its LDS slice holds: indices from global memory, hop by hop
        for (int k = 1; k <= CH_HOPS; ++k) {
            float* gout = xk + (size_t)(k - 1) * stride;
            for (int it = tid; it < nitems; it += SG_THREADS) walk_item(it, gout, k == CH_HOPS);
            seg_lds_barrier();
            const uint32_t t = cur;
            cur = nxt;
            nxt = t;
        }
    }
}

// The trailing column (k-th term's chain) of a streamed product, on a thread of its own: (row, lane half kh) adds what that lane
// half of the MFMA wave would -- k = 8m + 4kh + i in chunk / step order, one real step in the last chunk -- so the two half-sums
// and their total carry gemm_nt's bits.  Waves 4..7 run this while waves 0..3 stream the tiles (phase B: their fragment must not
// share registers with a second chain).
__device__ __forceinline__ void ch_rem_term(float& racc, __amdgpu_buffer_rsrc_t rA, uint32_t rowb, int kh, const float* rl) {
    const float* rp = rl + kh * 16;
    f32x4 av[SG_NCH];                                  // the whole term in flight: one round trip per term (these waves hold nothing else)
#pragma unroll
    for (int m = 0; m < SG_NCH; ++m) av[m] = ch_ld4(rA, rowb + 4u * (uint32_t)min(8 * m + 4 * kh, CH_LD - 4));
#pragma unroll
    for (int m = 0; m < SG_NCH; ++m) {
        const f32x4 r = *reinterpret_cast<const f32x4*>(rp + m * 32);
#pragma unroll
        for (int i = 0; i < (m == SG_NCH - 1 ? 1 : 4); ++i) racc = fmaf(av[m][i], r[i], racc);
    }
}

__global__ __launch_bounds__(SG_THREADS) __attribute__((amdgpu_waves_per_eu(4, 4)))
void seg_chain_fwd_kernel(const ChainFwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float ch_smem[];
    const ChainLds l = chain_lds(ch_smem);
    // block -> (graph block, quarter): the four quarter-workgroups of a graph block share the id mod 8, i.e. ONE XCD and its L2
    // (a speed matter only: the hand-offs are agent-scope whatever the placement); the quarter that also owns the trailing
    // column -- the longest -- first
    ChainBlock b;
    {
        const int bx8 = (int)blockIdx.x & 7, bt = (int)blockIdx.x >> 3;
        const int gb = bx8 + 8 * (bt / CH_NQ);
        if (gb >= a.nblocks) return;
        b.q = CH_NQ - 1 - (bt % CH_NQ);
        b.rem = b.q == CH_NQ - 1;
        b.cw = 8 + (b.rem ? 1 : 0);
        b.r0 = gb * a.rows_pb;
        b.rows = min(a.rows_pb, a.n - b.r0);
        b.e0 = a.rowptr[b.r0];
        b.ne = a.rowptr[b.r0 + b.rows] - b.e0;
        b.nb_in_lds = b.ne <= CH_CAP;
    }
    const int gb_ = b.r0 / a.rows_pb;
    ChainSync sy{a.cnt + gb_, a.status, CH_NQ, false};
    const int nrt = (b.rows + 31) >> 5;
    const size_t nld = (size_t)a.n * CH_LD;
    float keep_scale = 1.f;
    if (a.act == ACT_DROPOUT_RELU) keep_scale = 1.0f / (1.0f - a.p_drop);
    // ---- once per launch: the adjacency slice (by destination), deg^-1/2, and the first stage's W2 quarter
    {
        const int tid = ch_tid(), lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        if (a.zero_words && b.q == 0 && tid == 0) a.zero_words[gb_] = 0;
        const int rpv = tid <= b.rows ? a.rowptr[b.r0 + tid] : 0;
        const float dv = tid < b.rows ? a.dinv[b.r0 + tid] : 0.f;
        const int nbv = (b.nb_in_lds && tid < b.ne) ? a.nbr[b.e0 + tid] : 0;
        if (tid <= b.rows) l.rp[tid] = rpv - b.e0;
        if (tid < b.rows) l.dinv[tid] = dv;
        if (b.nb_in_lds && tid < b.ne) l.nb[tid] = nbv - b.r0;
        seg_copy_b(l.X[1], a.st[0].w2_img, b.q, CH_K8, wave, lane);
        if (b.rem && tid < CH_REM / 4) sg_st4(l.rem[0] + tid * 4, sg_ld4(a.st[0].w2_img + CH_ROFF + tid * 4));
        seg_dma_wait();
        __syncthreads();
    }

    for (int s = 0; s < a.nstage; ++s) {
        const ChainFwdStage& st = a.st[s];
        // =========================================================================================== phase A
        // y = act(S W2^T + deg b2), then the K hops.  (The W2 quarter and its trailing image are in LDS and visible: the barrier
        // above / the stage before's c2.)
        if (s > 0) ch_wait_graph(sy, 3 * s, l.flag);
        {
            const int tid = ch_tid(), lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
            if (tid < SG_TW) {
                const int col = ch_col_of_tile(b, tid);
                l.bias[tid] = col >= 0 ? st.b2[col] : 0.f;
            }
            if (wave < nrt) {
                const ChainFrag fr = ch_frag_of(b, wave, lane);
                SegA ta;
                ch_load_a(ta, ch_rsrc(st.S_in, nld), fr);
                f32x16 acc;
#pragma unroll
                for (int j = 0; j < 16; ++j) acc[j] = 0.f;
                float racc = 0.f;
                if (b.rem) ch_mma<true>(acc, racc, ta, l.X[1], l.rem[0], lane);
                else ch_mma<false>(acc, racc, ta, l.X[1], l.rem[0], lane);
                ch_store_tile(acc, b.q, nullptr, CH_H, l.X[2], 32 * wave, lane);
                if (b.rem) {
                    const float tot = racc + __shfl_xor(racc, 32);           // the two k halves
                    const int tr = 32 * wave + lane;
                    if (lane < 32 && tr < CH_MAX_ROWS) sg_st4(l.X[2] + (size_t)tr * SG_TW + 32, make_float4(tot, 0.f, 0.f, 0.f));
                }
            }
            __syncthreads();                     // a1: tile in X2; X0, X1, rem0, rem1, U are free
        }
        {
            const int tid = ch_tid(), lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
            // the TAGConv's weight quarters 0, 1 (the other two follow when the hop tiles retire) and its four trailing-column images
            seg_copy_b(l.X[0], st.tag_img[0], b.q, CH_K8, wave, lane);
            seg_copy_b(l.X[1], st.tag_img[1], b.q, CH_K8, wave, lane);
            // (the four trailing-column images: requested now, stored behind the epilogue -- their round trip was exposed)
            float4 r0v = make_float4(0.f, 0.f, 0.f, 0.f), r1v = r0v, r2v = r0v, r3v = r0v;
            if (b.rem && tid < CH_REM / 4) {
                r0v = sg_ld4(st.tag_img[0] + CH_ROFF + tid * 4); r1v = sg_ld4(st.tag_img[1] + CH_ROFF + tid * 4);
                r2v = sg_ld4(st.tag_img[2] + CH_ROFF + tid * 4); r3v = sg_ld4(st.tag_img[3] + CH_ROFF + tid * 4);
            }
            DropKey dk = DropKey{0u, 0u, 0u, 0u};
            if (a.act == ACT_DROPOUT_RELU) dk = drop_key(a.rng[0], a.rng[1], st.stream_y);
            ch_epilogue<true, true>(b, l, l.X[2], st.y, a.deg, a.act, a.p_drop, keep_scale, dk, tid);
            if (b.rem && tid < CH_REM / 4) {
                sg_st4(l.rem[0] + tid * 4, r0v);
                sg_st4(l.rem[1] + tid * 4, r1v);
                sg_st4(l.U + tid * 4, r2v);
                sg_st4(l.U + CH_REM + tid * 4, r3v);
            }
            seg_lds_barrier();
            ch_hops(b, l, ch_smem, l.X[2], l.X[3], st.xk, a.xk_stride, a.nbr, tid);
            ch_drain();                          // y, x^(k) written through and acknowledged; the DMAs of quarters 0, 1 landed
            __syncthreads();                     // a2: the hop tiles X2, X3 retire
            ch_arrive(sy);
            seg_copy_b(l.X[2], st.tag_img[2], b.q, CH_K8, wave, lane);
            seg_copy_b(l.X[3], st.tag_img[3], b.q, CH_K8, wave, lane);
            if (tid < SG_TW) {                   // the TAGConv's bias for this block's columns (read after b2)
                const int col = ch_col_of_tile(b, tid);
                l.bias[tid] = col >= 0 ? st.tag_bias[col] : 0.f;
            }
        }
        // =========================================================================================== phase B
        // h = act(sum_k x^(k) W_k^T + bias): four terms in ONE accumulator chain per tile, the fragment streamed
        ch_wait_graph(sy, 3 * s + 1, l.flag);
        {
            const int tid = ch_tid(), lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
            if (wave < nrt) {
                // (one block with the barrier inside: the fragment and the accumulators live nowhere else.  The wave's own DMAs
                //  above are OLDER than its fragment loads: complete before the first counted wait passes.)
                const ChainFrag fr = ch_frag_of(b, wave, lane);
                f32x16 acc;
#pragma unroll
                for (int j = 0; j < 16; ++j) acc[j] = 0.f;
                float racc = 0.f;
                f32x4 fa[SG_NCH];
                ch_load_a_asm(fa, reinterpret_cast<const char*>(st.y), fr);
                const char* x1 = reinterpret_cast<const char*>(st.xk);
                const char* x2 = reinterpret_cast<const char*>(st.xk + a.xk_stride);
                const char* x3 = reinterpret_cast<const char*>(st.xk + 2 * a.xk_stride);
                // The trailing column rides in the stream, as in gemm_nt (on waves of its own it starves behind the MFMA streams of
                // the CU: 0.12 vector-memory instructions per MFMA get through, measured).  EVERY workgroup runs the chain -- the
                // blocks without the column on whatever their trailing-image slots hold, result unused: two instantiations of the
                // stream, chosen per workgroup, made the allocator keep two fragments apart (52 spills).
                ch_mma_stream<true, true>(acc, racc, fa, l.X[0], l.rem[0], lane, x1, fr);
                ch_mma_stream<true, true>(acc, racc, fa, l.X[1], l.rem[1], lane, x2, fr);
                ch_bar();                        // b1: quarters 2, 3 visible
                ch_mma_stream<true, true>(acc, racc, fa, l.X[2], l.U, lane, x3, fr);
                ch_mma_stream<true, false>(acc, racc, fa, l.X[3], l.U + CH_REM, lane, x3, fr);
                ch_store_tile(acc, b.q, nullptr, CH_H, l.X[0], 32 * wave, lane);
                if (b.rem) {
                    const float tot = racc + __shfl_xor(racc, 32);           // the two k halves
                    const int tr = 32 * wave + lane;
                    if (lane < 32 && tr < CH_MAX_ROWS) sg_st4(l.X[0] + (size_t)tr * SG_TW + 32, make_float4(tot, 0.f, 0.f, 0.f));
                }
            } else {
                ch_drain();                      // this wave's share of the quarters 2, 3 has landed
                ch_bar();                        // b1 (the MFMA waves' shares landed before their first counted wait passed)
                // X1 retired: the next EdgeAggregation's W1j quarter, by the four waves that never carry a tile
                if (wave >= 4) seg_copy_b(l.X[1], st.w1j_img, b.q, CH_K8, wave - 4, lane, 4);
            }
            __syncthreads();                     // b2: h tile in X0; X2, X3, U retired
        }
        {
            const int tid = ch_tid(), lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
            seg_copy_b(l.X[2], st.w1i_img, b.q, CH_K8, wave, lane);
            DropKey dk = DropKey{0u, 0u, 0u, 0u};
            if (a.act == ACT_DROPOUT_RELU) dk = drop_key(a.rng[0], a.rng[1], st.stream_h);
            ch_epilogue<false, false>(b, l, l.X[0], st.h, nullptr, a.act, a.p_drop, keep_scale, dk, tid);
            // the walk's edge attributes and residue weights -> region U (retired at b2)
            float2* s_ea = reinterpret_cast<float2*>(l.U);
            float* s_we = l.U + 2 * CH_CAP;
            float wev = 0.f;
            if (tid < 2 * SG_TW) {
                const int f = tid / SG_TW, t = tid - f * SG_TW;
                const int col = ch_col_of_tile(b, t);
                wev = col >= 0 ? st.w1[(size_t)col * (2 * CH_H + 2) + 2 * CH_H + f] : 0.f;
            }
            float2 eav = make_float2(0.f, 0.f);
            if (b.nb_in_lds && tid < b.ne) eav = reinterpret_cast<const float2*>(a.ea_in)[b.e0 + tid];
            if (b.nb_in_lds && tid < b.ne) s_ea[tid] = eav;
            if (tid < 2 * SG_TW) s_we[tid] = wev;
            ch_drain();
            __syncthreads();                     // b3: X0 retires; W1i, W1j quarters landed
            ch_arrive(sy);
        }
        // =========================================================================================== phase C
        // P | Q = h W1i^T + b1 | h W1j^T, then the edge walk (ea_seg_fwd_kernel's)
        ch_wait_graph(sy, 3 * s + 2, l.flag);
        {
            const int tid = ch_tid(), lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
            const __amdgpu_buffer_rsrc_t rh = ch_rsrc(st.h, nld);
            if (wave < nrt) {
                const ChainFrag fr = ch_frag_of(b, wave, lane);
                SegA ta;
                ch_load_a(ta, rh, fr);
                const f32x16 accp = seg_mma(ta, l.X[2], CH_K8, lane);
                ch_store_tile(accp, b.q, st.b1, CH_H, l.X[3], 32 * wave, lane);
                const f32x16 accq = seg_mma(ta, l.X[1], CH_K8, lane);
                ch_store_tile(accq, b.q, nullptr, CH_H, l.X[0], 32 * wave, lane);
            }
            if (b.rem && wave >= 4) {
                // the trailing columns of P | Q (ea_seg.hip seg_rem_dots<true, 2, 3>: four k parts per row added by a fixed xor tree, a
                // part's k groups in order) -- on the four waves without a tile, two (row, part) items per thread, every load of an
                // item requested before its first multiply (one round trip; the sums' order is the per-phase launch's)
                constexpr int G = CH_K8 >> 2;          // 34 k groups: part p adds g = p, p + 4, ...
                constexpr int gmax = (CH_LD >> 2) - 1;
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    const int idx = (tid - 256) + 256 * half;
                    const int lr = idx >> 2, part = idx & 3;
                    const uint32_t rowb = (uint32_t)((size_t)(b.r0 + min(lr, b.rows - 1)) * CH_LD * 4);
                    float4 xa[9];
#pragma unroll
                    for (int j = 0; j < 9; ++j) {
                        const int g = min(part + 4 * j, G - 1);
                        const float4 v = ch_ld4f(rh, rowb + 16u * min(g, gmax));
                        xa[j] = part + 4 * j < G ? v : make_float4(0.f, 0.f, 0.f, 0.f);
                    }
                    float v1 = 0.f, v2 = 0.f;
#pragma unroll
                    for (int j0 = 0; j0 < 9; j0 += 3) {   // (the weights: cache hits, three groups at a time)
                        float4 w1[3], w2[3];
#pragma unroll
                        for (int j = 0; j < 3; ++j) {
                            const int g = min(part + 4 * (j0 + j), G - 1);
                            w1[j] = sg_ld4(st.w1i_img + CH_ROFF + (size_t)g * 16);
                            w2[j] = sg_ld4(st.w1j_img + CH_ROFF + (size_t)g * 16);
                        }
#pragma unroll
                        for (int j = 0; j < 3; ++j) {
                            const float4 x = xa[j0 + j];
                            v1 = fmaf(x.w, w1[j].w, fmaf(x.z, w1[j].z, fmaf(x.y, w1[j].y, fmaf(x.x, w1[j].x, v1))));
                            v2 = fmaf(x.w, w2[j].w, fmaf(x.z, w2[j].z, fmaf(x.y, w2[j].y, fmaf(x.x, w2[j].x, v2))));
                        }
                    }
                    v1 += __shfl_xor(v1, 1); v2 += __shfl_xor(v2, 1);
                    v1 += __shfl_xor(v1, 2); v2 += __shfl_xor(v2, 2);
                    if (part == 0 && lr < b.rows) {
                        sg_st4(l.X[3] + (size_t)lr * SG_TW + 32, make_float4(v1 + st.b1[32 * CH_NQ], 0.f, 0.f, 0.f));
                        sg_st4(l.X[0] + (size_t)lr * SG_TW + 32, make_float4(v2, 0.f, 0.f, 0.f));
                    }
                }
            }
            __syncthreads();                     // c1: P in X3, Q in X0; X1, X2 retired
        }
        {
            const int tid = ch_tid(), lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
            if (s + 1 < a.nstage) {              // the next stage's W2 quarter
                seg_copy_b(l.X[1], a.st[s + 1].w2_img, b.q, CH_K8, wave, lane);
                if (b.rem && tid < CH_REM / 4) sg_st4(l.rem[0] + tid * 4, sg_ld4(a.st[s + 1].w2_img + CH_ROFF + tid * 4));
            }
            // ---- P, Q out (the backward pass recomputes the pre-activation from them), and the walk
            const float* tP = l.X[3];
            const float* tQ = l.X[0];
            const float2* s_ea = reinterpret_cast<const float2*>(l.U);
            const float* s_we = l.U + 2 * CH_CAP;
            const bool poison = sy.failed && s + 1 == a.nstage;
            const int nitems = b.rows * b.cw;
            for (int it = tid; it < nitems; it += SG_THREADS) {
                const int lr = it / b.cw, lc = it - lr * b.cw;
                const int tc = ch_tcol(b, lc), gc = ch_gcol(b, lc);
                const float4 p4 = sg_ld4(tP + (size_t)lr * SG_TW + tc);
                const size_t o = (size_t)(b.r0 + lr) * CH_LD + gc;
                if (a.store_pq) {
                    sg_st4_wt(st.P + o, p4);
                    sg_st4_wt(st.Q + o, sg_ld4(tQ + (size_t)lr * SG_TW + tc));
                }
                const float4 w0 = sg_ld4(s_we + tc), w1 = sg_ld4(s_we + SG_TW + tc);
                float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
                const int beg = l.rp[lr], end = l.rp[lr + 1];
                if (b.nb_in_lds) {   // four slots per trip (slots past the row's end re-read its last edge and are not added)
                    const int last = end - 1;
                    for (int p = beg; p < end; p += 4) {
                        int s_[4];
                        float2 a_[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int qq = min(p + u, last);
                            s_[u] = l.nb[qq];
                            a_[u] = s_ea[qq];
                        }
                        float4 q_[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) q_[u] = sg_ld4(tQ + (size_t)s_[u] * SG_TW + tc);
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
                        const int ls = a.nbr[b.e0 + p] - b.r0;
                        const float2 a2 = reinterpret_cast<const float2*>(a.ea_in)[b.e0 + p];
                        float4 v = sg_add4(p4, sg_ld4(tQ + (size_t)ls * SG_TW + tc));
                        v = sg_fma4(a2.x, w0, v);
                        v = sg_fma4(a2.y, w1, v);
                        acc = sg_add4(acc, sg_relu4(v));
                    }
                }
                if (poison) acc = make_float4(__builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""));
                sg_st4_wt(st.S_out + o, acc);
            }
            if (s + 1 < a.nstage) {
                ch_drain();
                __syncthreads();                 // c2: X0, X3, U retire; the next W2 quarter landed
                ch_arrive(sy);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ host
struct ChainPlan { int rows_pb, nblocks, ny, grid; };
static bool chain_plan(int seg, int n, int ld, ChainPlan& p) {
    if (seg <= 0 || seg > CH_MAX_ROWS || n <= 0 || n % seg != 0) return false;
    const int gpb = std::max(1, CH_MAX_ROWS / seg);
    p.rows_pb = gpb * seg;
    p.nblocks = (n + p.rows_pb - 1) / p.rows_pb;
    int remv, nq;
    col_plan(ld, remv, nq);
    p.ny = nq;
    p.grid = 8 * p.ny * ((p.nblocks + 7) / 8);
    return p.ny >= 1 && p.rows_pb * 4 <= CH_CAP;
}
// two workgroups of this kernel per CU?  (asked once per device and kernel; the hand-offs need every workgroup resident)
static bool chain_two_per_cu(const void* kernel, std::atomic<uint64_t>& raised, std::atomic<uint64_t>& asked, std::atomic<uint64_t>& okmask) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
    const uint64_t bit = 1ull << dev;
    if (!(asked.load() & bit)) {
        if (ensure_dynamic_lds(kernel, CH_LDS_BYTES, raised) != PFN_OK) return false;
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, SG_THREADS, CH_LDS_BYTES) == hipSuccess && nb >= 2) okmask.fetch_or(bit);
        asked.fetch_or(bit);
    }
    return (okmask.load() & bit) != 0;
}
static std::atomic<uint64_t> chf_raised{0}, chf_asked{0}, chf_ok{0};

// The chain kernels' regime and shape: the latency regime of the graph-resident kernels (every workgroup of the launch
// co-resident: <= 2 per CU), hidden 129-shaped (four MFMA quarters + ONE trailing column, one 136-k piece with a one-step tail),
// K = 3 (four TAGConv terms: the four LDS slots), Fe = 2
bool seg_chain_fit(int seg, int n, int fe, int h, int K) {
    // OPT-IN (PFN_SEG_CHAIN=1): measured on MI355X the chain launch is SLOWER than the nine launches it replaces (case118v2 x 128:
    // 248 vs 200 us, profiles/r05_chain_phase_timestamps.txt) -- every phase costs inside the launch what it cost as its own, a
    // per-graph hand-off costs what a kernel boundary did, and the two workgroups of a CU stay in lockstep
    static const bool on = diag_env("PFN_SEG_CHAIN") != nullptr;
    if (!on) return false;
    const int ld = ld_of(h);
    ChainPlan p;
    int remv, nq;
    col_plan(ld, remv, nq);
    if (!(fe == 2 && K == CH_HOPS && h == CH_H && ld == CH_LD && nq == CH_NQ && remv == 4)) return false;
    if (!chain_plan(seg, n, ld, p) || (long)p.grid > 2L * device_cus()) return false;
    return chain_two_per_cu(reinterpret_cast<const void*>(seg_chain_fwd_kernel), chf_raised, chf_asked, chf_ok);
}
int seg_chain_blocks(int seg, int n, int ld) {
    ChainPlan p;
    return chain_plan(seg, n, ld, p) ? p.nblocks : 0;
}

int launch_seg_chain_fwd(const GraphView& g, const SegChainFwd& c, int seg, hipStream_t s) {
    ChainPlan p;
    if (!chain_plan(seg, g.n, c.ld, p) || c.nstage < 1 || c.nstage > CH_MAX_STAGES || c.nhops != 3) {
        set_error("seg_chain_fwd: shape outside the chain kernel's (seg %d, %d stages, K %d)", seg, c.nstage, c.nhops);
        return PFN_EINVAL;
    }
    ChainFwdArgs a;
    memset(&a, 0, sizeof(a));
    a.n = g.n; a.rows_pb = p.rows_pb; a.nblocks = p.nblocks; a.ny = p.ny; a.ld = c.ld; a.h = c.h; a.nhops = c.nhops;
    a.nstage = c.nstage; a.act = c.act; a.store_pq = c.store_pq; a.p_drop = c.p_drop; a.rng = c.rng;
    a.rowptr = g.rowptr_in; a.nbr = g.in_src; a.dinv = g.dinv; a.deg = g.deg; a.ea_in = c.ea_in;
    a.cnt = c.cnt; a.zero_words = c.zero_words; a.status = c.status; a.xk_stride = (size_t)g.n * c.ld;
    double flops = 0.0;
    for (int i = 0; i < c.nstage; ++i) {
        const SegChainFwdStage& h = c.st[i];
        ChainFwdStage& d = a.st[i];
        d.S_in = h.S_in; d.w2_img = h.w2_img; d.b2 = h.b2; d.y = h.y; d.xk = h.xk;
        for (int k = 0; k < 4; ++k) d.tag_img[k] = h.tag_img[k];
        d.tag_bias = h.tag_bias; d.h = h.h; d.w1i_img = h.w1i_img; d.w1j_img = h.w1j_img; d.b1 = h.b1; d.w1 = h.w1;
        d.P = h.P; d.Q = h.Q; d.S_out = h.S_out; d.stream_y = h.stream_y; d.stream_h = h.stream_h;
        flops += 2.0 * g.n * (double)c.h * c.h * 7.0;   // S W2^T, four TAGConv terms, P | Q
    }
    if (!chain_two_per_cu(reinterpret_cast<const void*>(seg_chain_fwd_kernel), chf_raised, chf_asked, chf_ok)) {
        set_error("seg_chain_fwd: two workgroups per CU are not resident");
        return PFN_EINVAL;
    }
    ProfScope ps("seg_chain_fwd", 0.0, flops, s);
    seg_chain_fwd_kernel<<<p.grid, SG_THREADS, CH_LDS_BYTES, s>>>(a);
    PFN_CHECK_LAUNCH();
    return PFN_OK;
}

}  // namespace pfn

