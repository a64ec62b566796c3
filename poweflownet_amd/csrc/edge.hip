// Segmented gather / scatter-add kernels over the CSR adjacency (gfx950, HBM/L2-bound).
//
// These replace what PyG's propagate() does around the reference's message functions:
//   * index_select lifting of x_i / x_j            (EdgeAggregation.forward, networks/MPN.py:53)
//   * the materialised cat[x_i, x_j, e]            (EdgeAggregation.message, networks/MPN.py:28)
//   * scatter_add of messages onto edge_index[1]   (aggr='add', networks/MPN.py:11)
//   * TAGConv's K weighted gather-scatter hops     (PyG TAGConv.propagate; call site :545)
// Work item = (node row, 16-byte column chunk): consecutive lanes read consecutive float4s of the same
// neighbour row (528 B contiguous for H = 129), every row is reduced by one lane sequentially in edge-id
// order -> no atomics, deterministic, same summation order as the reference's sequential scatter.
#include <stdlib.h>

#include <type_traits>

#include "pfn_internal.hpp"

namespace pfn {

// Barrier that publishes LDS traffic only: __syncthreads() also waits (vmcnt(0)) until every global STORE of the wave is
// acknowledged -- a round trip of 1-2 us under load at every barrier that follows output stores (seg_tile.hpp seg_lds_barrier).
// Only where no thread reads another thread's GLOBAL writes behind the barrier: the hop kernels' tiles live in LDS.
__device__ __forceinline__ void bh_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ float4 fma4(float a, float4 x, float4 acc) {
    acc.x = fmaf(a, x.x, acc.x);
    acc.y = fmaf(a, x.y, acc.y);
    acc.z = fmaf(a, x.z, acc.z);
    acc.w = fmaf(a, x.w, acc.w);
    return acc;
}
__device__ __forceinline__ float4 mul4(float a, float4 x) { return make_float4(a * x.x, a * x.y, a * x.z, a * x.w); }
__device__ __forceinline__ float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ unsigned char relu_bits(float4 v) {   // bit i = component i > 0 (the edge stage's ReLU mask)
    return (unsigned char)((v.x > 0.f ? 1 : 0) | (v.y > 0.f ? 2 : 0) | (v.z > 0.f ? 4 : 0) | (v.w > 0.f ? 8 : 0));
}
__device__ __forceinline__ float4 mask4(unsigned m, float4 g) {   // g where the mask bit is set, else 0
    return make_float4((m & 1) ? g.x : 0.f, (m & 2) ? g.y : 0.f, (m & 4) ? g.z : 0.f, (m & 8) ? g.w : 0.f);
}
__device__ __forceinline__ float4 sel4(bool k, float4 a, float4 b) { return make_float4(k ? a.x : b.x, k ? a.y : b.y, k ? a.z : b.z, k ? a.w : b.w); }
__device__ __forceinline__ float4 relu4(float4 v) { return make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f)); }

__device__ __forceinline__ int exp_block(int b, int) { return b; }
// ------------------------------------------------------------------------------------------- hop
template <bool NORM>
__global__ __launch_bounds__(256) void hop_kernel(int n, int nchunk, const int* __restrict__ rowptr,
                                                  const int* __restrict__ nbr, const float* __restrict__ dinv,
                                                  const float* __restrict__ x, const float* __restrict__ add,
                                                  float* __restrict__ y, const float* __restrict__ gate,
                                                  float gate_scale, int ld) {
    const long item = (long)exp_block(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x;
    const int row = (int)(item / nchunk);
    if (row >= n) return;
    const int col = (int)(item - (long)row * nchunk) * 4;
    const int beg = rowptr[row], end = rowptr[row + 1];
    const float di = NORM ? dinv[row] : 1.0f;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    // Four edge slots per trip, ALWAYS four gathers in flight: slots past the row's end re-read its last edge (the same
    // cache line again, no extra HBM traffic) and are dropped by a select, so rows of degree 1..3 -- most of a power grid --
    // no longer walk a chain of dependent index -> row loads one edge at a time.  Sums still run in edge-id order.
    for (int p = beg; p < end; p += 4) {
        const int last = end - 1;
        const int s0 = nbr[p], s1 = nbr[min(p + 1, last)], s2 = nbr[min(p + 2, last)], s3 = nbr[min(p + 3, last)];
        const float4 v0 = ld4(x + (size_t)s0 * ld + col), v1 = ld4(x + (size_t)s1 * ld + col);
        const float4 v2 = ld4(x + (size_t)s2 * ld + col), v3 = ld4(x + (size_t)s3 * ld + col);
        const bool k1 = p + 1 < end, k2 = p + 2 < end, k3 = p + 3 < end;
        if (NORM) {
            const float w0 = dinv[s0] * di, w1 = dinv[s1] * di, w2 = dinv[s2] * di, w3 = dinv[s3] * di;
            acc = fma4(w0, v0, acc);
            acc = sel4(k1, fma4(w1, v1, acc), acc);
            acc = sel4(k2, fma4(w2, v2, acc), acc);
            acc = sel4(k3, fma4(w3, v3, acc), acc);
        } else {
            acc = add4(acc, v0);
            acc = sel4(k1, add4(acc, v1), acc);
            acc = sel4(k2, add4(acc, v2), acc);
            acc = sel4(k3, add4(acc, v3), acc);
        }
    }
    const size_t o = (size_t)row * ld + col;
    if (add) acc = add4(acc, ld4(add + o));
    if (gate) {
        const float4 g = ld4(gate + o);
        acc.x = g.x > 0.f ? acc.x * gate_scale : 0.f;
        acc.y = g.y > 0.f ? acc.y * gate_scale : 0.f;
        acc.z = g.z > 0.f ? acc.z * gate_scale : 0.f;
        acc.w = g.w > 0.f ? acc.w * gate_scale : 0.f;
    }
    st4(y + o, acc);
}

int launch_hop(const GraphView& g, const HopArgs& a, hipStream_t s) {
    if (g.n == 0) return PFN_OK;
    const int nchunk = a.ld / 4;
    const long items = (long)g.n * nchunk;
    const int blocks = (int)((items + 255) / 256);
    const int* rp = a.transpose ? g.rowptr_out : g.rowptr_in;
    const int* nb = a.transpose ? g.out_dst : g.in_src;
    // algorithmic bytes B_sa(F) need the effective edge count, which lives on the device: bench.py applies
    // SURVEY 8(d)'s formula to the measured duration itself.
    ProfScope ps(a.normalize ? "hop_norm" : "scatter_add", 0.0, 0.0, s);
    if (a.normalize)
        hop_kernel<true><<<blocks, 256, 0, s>>>(g.n, nchunk, rp, nb, g.dinv, a.x, a.add, a.y, a.gate, a.gate_scale, a.ld);
    else
        hop_kernel<false><<<blocks, 256, 0, s>>>(g.n, nchunk, rp, nb, g.dinv, a.x, a.add, a.y, a.gate, a.gate_scale, a.ld);
    PFN_CHECK_LAUNCH();
    return PFN_OK;
}

// ------------------------------------------------------------------------------------ LDS-resident hops
// PyG batches are block diagonal: when every graph's node rows fit in LDS (118 buses x 528 B = 62 KB) the K hops of a
// TAGConv need no trip through L2/HBM between hops: a workgroup stages the rows of its graph(s) once, then ping-pongs
// between two LDS tiles, gathering neighbour rows with ds_read_b128.  Forward writes every hop result out (they are
// GEMM operands and saved for the weight gradients).  The TAGConv backward uses the same data flow over the transposed
// adjacency (it hops the incoming gradient, model.hip: tag_backward); the Horner mode (iterates kept on chip, only the end
// written) remains for layers whose output is wider than their input.
constexpr int FH_THREADS = 512;
constexpr int FH_LDS_BYTES = 156 * 1024;

bool fused_hops_fit(int seg, int ld, int n) {
    // A hop acts on every column independently, so a block may take a COLUMN SLICE of its graph(s): small batches still
    // fill the chip (case118 x 128: 128 graphs x 7 slices of 5 float4 columns = 896 blocks, 21 KB of LDS each) and the K
    // hops cost one launch instead of K (measured 47 -> ~11 us per TAGConv at 128 graphs).  Needs two
    // tiles of at least one float4 column of a whole graph in LDS.
    (void)ld;
    (void)n;
    if (seg <= 0 || (size_t)2 * seg * 4 * sizeof(float) + (size_t)(2 * seg + 1) * sizeof(int) > (size_t)FH_LDS_BYTES / 2) return false;
    return true;
}

__global__ __launch_bounds__(FH_THREADS) void fused_hops_kernel(int n, int rows_pb, int cw, int nbr_cap,
                                                                const int* __restrict__ rowptr, const int* __restrict__ nbr,
                                                                const float* __restrict__ dinv, const FusedHopsArgs a) {
    extern __shared__ __attribute__((aligned(16))) float tiles[];   // 2 x rows_pb x 4cw | dinv[rows_pb] | rp[rows_pb+1] | nbr[cap]
    const int r0 = blockIdx.x * rows_pb;
    const int rows = min(rows_pb, n - r0);
    const int c0 = blockIdx.y * cw;                       // first float4 column of this block's slice
    const int cwh = min(cw, (a.ld >> 2) - c0);            // float4 columns in the slice
    const int tld = 4 * cw;                               // tile row stride (floats)
    const int items = rows * cwh;
    float* cur = tiles;
    float* nxt = tiles + (size_t)rows_pb * tld;
    float* s_dinv = tiles + (size_t)2 * rows_pb * tld;
    int* s_rp = reinterpret_cast<int*>(s_dinv + rows_pb);
    int* s_nb = s_rp + rows_pb + 1;
    // the block's slice of the adjacency goes to LDS once: with several work items per thread and K hops, index loads
    // from global memory (three dependent latencies per item) would dominate everything else
    const int e0 = rowptr[r0], e1 = rowptr[r0 + rows];
    const bool nb_in_lds = e1 - e0 <= nbr_cap;
    const float* first = a.transpose ? a.G + (size_t)a.K * a.stride : a.x0;
    if (rows < FH_THREADS && items <= 4 * FH_THREADS && e1 - e0 <= FH_THREADS) {
        // Small blocks (the metric configuration: one graph, ~1,300 items): EVERY global load of the prologue is requested before
        // the first LDS store.  As load-store loops each store waited for all loads before it (VMEM returns in order) and the next
        // loop's loads went out only then: four serial round trips before the first hop (cf. ea_seg.hip).
        const int t = threadIdx.x;
        const int rpv = t <= rows ? rowptr[r0 + t] : 0;
        const float dv = t < rows ? dinv[r0 + t] : 0.f;
        float4 tv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = t + j * FH_THREADS;
            tv[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < items) {
                const int lr = i / cwh, lc = i - lr * cwh;
                tv[j] = ld4(first + (size_t)(r0 + lr) * a.ld + 4 * (c0 + lc));
            }
        }
        const int nbv = (nb_in_lds && t < e1 - e0) ? nbr[e0 + t] : 0;   // (second level: needs e0)
        if (t <= rows) s_rp[t] = rpv - e0;
        if (t < rows) s_dinv[t] = dv;
        if (nb_in_lds && t < e1 - e0) s_nb[t] = nbv - r0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = t + j * FH_THREADS;
            if (i < items) {
                const int lr = i / cwh, lc = i - lr * cwh;
                st4(cur + (size_t)lr * tld + 4 * lc, tv[j]);
            }
        }
    } else {
        for (int i = threadIdx.x; i <= rows; i += FH_THREADS) s_rp[i] = rowptr[r0 + i] - e0;
        for (int i = threadIdx.x; i < rows; i += FH_THREADS) s_dinv[i] = dinv[r0 + i];
        if (nb_in_lds)
            for (int i = threadIdx.x; i < e1 - e0; i += FH_THREADS) s_nb[i] = nbr[e0 + i] - r0;
        for (int i = threadIdx.x; i < items; i += FH_THREADS) {
            const int lr = i / cwh, lc = i - lr * cwh;
            st4(cur + (size_t)lr * tld + 4 * lc, ld4(first + (size_t)(r0 + lr) * a.ld + 4 * (c0 + lc)));
        }
    }
    __syncthreads();
    for (int k = 1; k <= a.K; ++k) {
        const bool last = k == a.K;
        const float* addp = a.transpose ? a.G + (size_t)(a.K - k) * a.stride : nullptr;
        float* gout = a.transpose ? (last ? a.out : nullptr) : a.xk + (size_t)(k - 1) * a.stride;
        for (int i = threadIdx.x; i < items; i += FH_THREADS) {
            const int lr = i / cwh, lc = i - lr * cwh;
            const int beg = s_rp[lr], end = s_rp[lr + 1];
            const float di = s_dinv[lr];
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            if (nb_in_lds) {   // four slots per trip (see hop_kernel): index -> (dinv, row) is a chain of dependent LDS reads
                const int last = end - 1;
                for (int p = beg; p < end; p += 4) {
                    int s_[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) s_[u] = s_nb[min(p + u, last)];
                    float w_[4];
                    float4 v_[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        w_[u] = s_dinv[s_[u]] * di;
                        v_[u] = ld4(cur + (size_t)s_[u] * tld + 4 * lc);
                    }
                    acc = fma4(w_[0], v_[0], acc);
                    acc = sel4(p + 1 < end, fma4(w_[1], v_[1], acc), acc);
                    acc = sel4(p + 2 < end, fma4(w_[2], v_[2], acc), acc);
                    acc = sel4(p + 3 < end, fma4(w_[3], v_[3], acc), acc);
                }
            } else {
                for (int p = beg; p < end; ++p) {
                    const int ls = nbr[e0 + p] - r0;
                    acc = fma4(s_dinv[ls] * di, ld4(cur + (size_t)ls * tld + 4 * lc), acc);
                }
            }
            const size_t o = (size_t)(r0 + lr) * a.ld + 4 * (c0 + lc);
            if (addp) acc = add4(acc, ld4(addp + o));
            if (last && a.gate) {
                const float4 g4 = ld4(a.gate + o);
                acc.x = g4.x > 0.f ? acc.x * a.gate_scale : 0.f;
                acc.y = g4.y > 0.f ? acc.y * a.gate_scale : 0.f;
                acc.z = g4.z > 0.f ? acc.z * a.gate_scale : 0.f;
                acc.w = g4.w > 0.f ? acc.w * a.gate_scale : 0.f;
            }
            if (!last) st4(nxt + (size_t)lr * tld + 4 * lc, acc);
            if (gout) st4(gout + o, acc);
        }
        __syncthreads();
        float* t = cur;
        cur = nxt;
        nxt = t;
    }
}

// ---- the same one-tile-plus-registers scheme for BIG BATCHES OF SMALL graphs (case118 x 2048): with the second ping-pong tile
// gone a block affords WHOLE ROWS (all 33 chunks) of its graph in half of the LDS -- two blocks per CU as before, but every global
// access is a full 528-byte row (fused_hops_kernel's two 17-column slices leave cache lines partly used: 600 MB of fabric traffic
// per launch for 503 MB of minimum traffic) and inputs / outputs stay row-major.  Item = (row, chunk), chunk fastest; <= 4 items
// per thread.
constexpr int RH_THREADS = 512;                  // (1024 threads x 4 items would need <= 64 VGPRs for two blocks per CU: spills)
constexpr int RH_IPT = 8;                       // items per thread at most: rows x chunks <= 4,096 per block
constexpr int RH_NBPT = 4;                      // staged neighbour ids per thread at most: 2,048 directed edges per block
__global__ __launch_bounds__(RH_THREADS, 4) void row_hops_kernel(int n, int rows_pb, int nchunk, int nb_cap,
                                                                const int* __restrict__ rowptr, const int* __restrict__ nbr,
                                                                const float* __restrict__ dinv, const float* __restrict__ x0,
                                                                float* __restrict__ xk, size_t stride, int ld, int K, int reverse) {
    extern __shared__ __attribute__((aligned(16))) float4 rh_tile[];          // [rows_pb * nchunk] | rp u16 [rows_pb + 2] | nb u16 [nb_cap]
    unsigned short* s_rp = reinterpret_cast<unsigned short*>(rh_tile + (size_t)rows_pb * nchunk);
    unsigned short* s_nb = s_rp + ((rows_pb + 2 + 7) & ~7);
    const int r0 = (reverse ? (int)(gridDim.x - 1 - blockIdx.x) : (int)blockIdx.x) * rows_pb, rows = min(rows_pb, n - r0), t = threadIdx.x;
    const int items = rows * nchunk;
    const int e0 = rowptr[r0], ne = rowptr[r0 + rows] - e0;
    const bool nb_in_lds = ne <= nb_cap && ne < 65536 && ne <= RH_NBPT * RH_THREADS;
    float di[RH_IPT];
    float4 z[RH_IPT];
    int nbv[RH_NBPT];
#pragma unroll
    for (int r = 0; r < RH_IPT; ++r) {          // every global load of the prologue is requested before the first LDS store
        const int i = t + r * RH_THREADS;
        const int row = i / nchunk, lc = i - row * nchunk;   // (recomputed per hop below: kept in registers they spilled)
        di[r] = 0.f;
        z[r] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < items) {
            di[r] = dinv[r0 + row];
            z[r] = ld4(x0 + (size_t)(r0 + row) * ld + 4 * lc);
        }
    }
    int rpv[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) rpv[q] = t + q * RH_THREADS <= rows ? rowptr[r0 + t + q * RH_THREADS] : 0;    // (rows_pb <= 1023)
#pragma unroll
    for (int jn = 0; jn < RH_NBPT; ++jn) {
        const int i = t + jn * RH_THREADS;
        nbv[jn] = (nb_in_lds && i < ne) ? nbr[e0 + i] : 0;
    }
#pragma unroll
    for (int q = 0; q < 2; ++q)
        if (t + q * RH_THREADS <= rows) s_rp[t + q * RH_THREADS] = (unsigned short)(rpv[q] - e0);
#pragma unroll
    for (int r = 0; r < RH_IPT; ++r)
        if (t + r * RH_THREADS < items) rh_tile[t + r * RH_THREADS] = mul4(di[r], z[r]);
    if (nb_in_lds) {
#pragma unroll
        for (int jn = 0; jn < RH_NBPT; ++jn) {
            const int i = t + jn * RH_THREADS;
            if (i < ne) s_nb[i] = (unsigned short)(nbv[jn] - r0);
        }
    }
    __syncthreads();
    for (int k = 1; k <= K; ++k) {
        float* outk = xk + (size_t)(k - 1) * stride;
#pragma unroll
        for (int r = 0; r < RH_IPT; ++r) {
            const int i = t + r * RH_THREADS;
            if (i >= items) continue;
            const int row = i / nchunk, lc = i - row * nchunk;
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            if (nb_in_lds) {
                const int beg = s_rp[row], end = s_rp[row + 1], last = end - 1;
                for (int p = beg; p < end; p += 4) {
                    int s_[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) s_[u] = s_nb[min(p + u, last)];
                    float4 v_[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) v_[u] = rh_tile[s_[u] * nchunk + lc];
                    acc = add4(acc, v_[0]);
                    acc = sel4(p + 1 < end, add4(acc, v_[1]), acc);
                    acc = sel4(p + 2 < end, add4(acc, v_[2]), acc);
                    acc = sel4(p + 3 < end, add4(acc, v_[3]), acc);
                }
            } else {
                for (int p = rowptr[r0 + row]; p < rowptr[r0 + row + 1]; ++p) acc = add4(acc, rh_tile[(nbr[p] - r0) * nchunk + lc]);
            }
            const float4 y = mul4(di[r], acc);
            st4(outk + (size_t)(r0 + row) * ld + 4 * lc, y);
            z[r] = mul4(di[r], y);
            __builtin_amdgcn_sched_barrier(0);   // (left alone the scheduler interleaves all eight items' gathers and spills)
        }
        if (k == K) break;
        bh_lds_barrier();          // (not __syncthreads(): the hop's output stores drain under the next hop)
#pragma unroll
        for (int r = 0; r < RH_IPT; ++r)
            if (t + r * RH_THREADS < items) rh_tile[t + r * RH_THREADS] = z[r];
        bh_lds_barrier();
    }
}
// whole graphs per block: rows x chunks <= 4,096 items, rows <= 1,024, tile + offsets in HALF of the LDS (two blocks per CU)
static int row_hops_graphs_per_block(int seg, int nchunk) {
    if (seg <= 0 || seg > 1023 || (long)seg * nchunk > (long)RH_IPT * RH_THREADS) return 0;
    int gpb = std::min((RH_IPT * RH_THREADS) / (seg * nchunk), 1023 / seg);
    while (gpb > 0 && (size_t)gpb * seg * nchunk * 16 + (size_t)((gpb * seg + 2 + 7) & ~7) * 2 + 4096 > (size_t)78 * 1024) --gpb;
    return gpb;
}

int launch_fused_hops(const GraphView& g, const FusedHopsArgs& a, hipStream_t s) {
    if (g.n == 0 || a.K == 0) return PFN_OK;
    const int nchunk = a.ld / 4, ngraphs = g.n / a.seg;
    {   // big batches of small graphs, forward data flow: whole rows per block, one LDS tile + registers (row_hops_kernel)
        static const bool off = diag_env("PFN_NO_ROW_HOPS") != nullptr;   // A/B switch: the two-tile column-slice kernel below
        const int gpb = row_hops_graphs_per_block(a.seg, nchunk);
        if (!off && !a.transpose && gpb > 0 && (long)(ngraphs + gpb - 1) / gpb >= 4L * device_cus()) {
            const int rows_pb = gpb * a.seg;
            const size_t fixed = (size_t)rows_pb * nchunk * 16 + (size_t)((rows_pb + 2 + 7) & ~7) * 2;
            const size_t want_nb = (size_t)(2 * (int64_t)g.e_stored / std::max(1, ngraphs) * gpb + 64) * 2;
            const size_t lds_total = std::min((size_t)80 * 1024, fixed + want_nb);
            const int nb_cap = (int)((lds_total - fixed) / 2);
            static std::atomic<uint64_t> lds_raised_rh{0};
            PFN_TRY(ensure_dynamic_lds(reinterpret_cast<const void*>(row_hops_kernel), 160 * 1024, lds_raised_rh));
            const bool adjt_rh = a.adjt < 0 ? false : a.adjt != 0;
            ProfScope ps(adjt_rh ? "fused_hops_bwd" : "fused_hops_fwd", 0.0, 0.0, s);
            row_hops_kernel<<<(g.n + rows_pb - 1) / rows_pb, RH_THREADS, lds_total, s>>>(
                g.n, rows_pb, nchunk, nb_cap, adjt_rh ? g.rowptr_out : g.rowptr_in, adjt_rh ? g.out_dst : g.in_src, g.dinv, a.x0, a.xk,
                a.stride, a.ld, a.K, next_sweep_direction());   // (serpentine sweeps, pfn_internal.hpp)
            PFN_CHECK_LAUNCH();
            return PFN_OK;
        }
    }
    // column slices: aim for >= 512 blocks; a slice must leave room for two tiles of a whole graph
    const size_t per_chunk_graph = (size_t)2 * a.seg * 4 * sizeof(float);
    const int max_cw = (int)std::min<size_t>(nchunk, ((size_t)FH_LDS_BYTES - (size_t)(2 * a.seg + 1) * sizeof(int)) / per_chunk_graph);
    if (max_cw < 1) {
        set_error("fused hops: a %d-row graph does not fit in LDS", a.seg);
        return PFN_EINVAL;
    }
    // (1024: case118 x 128 = 7 slices of 5 float4 columns, 896 blocks, 11.0-11.6 us per launch; 512 blocks of 9 columns: 12.1-12.8 us;
    //  11 slices of 3 columns: 13.4 us -- 48-byte row segments waste most of every cache line, hence the floor of 4 columns)
    const int want = 1024;
    int cs = std::max(1, std::min(nchunk, (want + ngraphs - 1) / std::max(1, ngraphs)));   // slices wanted
    int cw = std::max((nchunk + cs - 1) / cs, std::min(nchunk, 4));
    cw = std::min(cw, max_cw);
    cs = (nchunk + cw - 1) / cw;
    // Big batches (several rounds of workgroups anyway): a block's tiles stay under HALF of the LDS, so that two workgroups
    // share a CU and the loads / stores of one overlap the LDS hops of the other (case118 x 2048: 197 -> 144 us per launch
    // with 2 slices of 17 float4 columns; narrower slices lose it again to partly used cache lines: 165 / 181 us at 3 / 4).
    size_t budget = (size_t)FH_LDS_BYTES, lds_cap = (size_t)160 * 1024;
    {
        const size_t half = (size_t)FH_LDS_BYTES / 2;
        const int half_cw = (int)((half - (size_t)(2 * a.seg + 1) * sizeof(int)) / per_chunk_graph);
        if ((long)ngraphs * cs >= 2L * device_cus() && half_cw >= 8) {
            budget = half;
            lds_cap /= 2;
            if (cw > half_cw) {
                cs = (nchunk + half_cw - 1) / half_cw;
                cw = (nchunk + cs - 1) / cs;
                cs = (nchunk + cw - 1) / cw;
            }
        }
    }
    // whole graphs per block: as many as fit the two LDS tiles, but keep >= ~2 blocks per CU worth of parallelism
    int gpb = (int)(budget / ((size_t)2 * a.seg * cw * 4 * sizeof(float) + (size_t)2 * a.seg * sizeof(int)));
    gpb = std::max(1, gpb);
    while (gpb > 1 && (long)((ngraphs + gpb - 1) / gpb) * cs < want) --gpb;
    const int rows_pb = gpb * a.seg;
    const size_t tile_bytes = (size_t)2 * rows_pb * cw * 4 * sizeof(float);
    const size_t fixed = tile_bytes + (size_t)(2 * rows_pb + 1) * sizeof(int);
    // neighbour list: what the slice's rows need (average degree is small), capped by what is left of 160 KiB / blocks per CU
    const size_t want_nb = (size_t)rows_pb * 16 * sizeof(int);
    const size_t lds_total = std::max(fixed, std::min(lds_cap, fixed + want_nb));
    const int nbr_cap = (int)((lds_total - fixed) / sizeof(int));
    static std::atomic<uint64_t> lds_raised{0};
    PFN_TRY(ensure_dynamic_lds(reinterpret_cast<const void*>(fused_hops_kernel), 160 * 1024, lds_raised));
    const bool adjt = a.adjt < 0 ? a.transpose != 0 : a.adjt != 0;
    ProfScope ps((a.transpose || adjt) ? "fused_hops_bwd" : "fused_hops_fwd", 0.0, 0.0, s);
    fused_hops_kernel<<<dim3((g.n + rows_pb - 1) / rows_pb, cs), FH_THREADS, lds_total, s>>>(
        g.n, rows_pb, cw, nbr_cap, adjt ? g.rowptr_out : g.rowptr_in, adjt ? g.out_dst : g.in_src, g.dinv, a);
    PFN_CHECK_LAUNCH();
    return PFN_OK;
}

// ------------------------------------------------------------------------- K hops, large graphs
// The K hops of one TAGConv direction for graphs too large for fused_hops_kernel's two ping-pong tiles (6470 buses: one float4
// column of a whole graph is 103 KB).  ONE tile is enough: a hop reads its neighbours' values from the LDS tile and keeps its
// rows' results in REGISTERS (a thread owns rows tid, tid + 1024, ...: <= 8 float4), a barrier, the registers go back into the
// tile.  Work item = one graph x ONE float4 column chunk, 1024 threads; the graph's adjacency is staged as 16-bit local ids and
// offsets (6470 nodes / 18,010 directed edges: 36 + 13 KB beside the 103 KB tile).  The tile holds z = D^-1/2 x, so a hop is
// y_i = d_i sum_j z_j and the next tile is d_i y_i: no per-edge weight, no D^-1/2 table in LDS.  Global traffic per direction: x
// read once, K outputs written, the adjacency once per workgroup (L2) -- instead of K full gather passes (3 x 852 MB at 6470rte x 64).
// Inputs and outputs are chunk-major, so a chunk's rows are one contiguous run; the workgroups of a graph sit on ONE XCD
// (workgroup b -> XCD b mod 8) and share its adjacency in that L2.
constexpr int BH_THREADS = 1024;
constexpr int BH_RPT = 8;                       // rows per thread at most: graphs up to 8,192 nodes
// High-degree buses (a "hub" substation: 100+ lines) are NOT walked by their owner thread -- a 170-edge row was one thread's
// 43 serial trips per hop while the block's other 1,023 threads waited at the barrier (hub grid: 343 vs 268 us per launch).
// Rows above BH_HUB_DEG edges are listed once, and in every hop a WAVE sums such a row: lane l adds the edges l, l + 64, ... in
// that order, the 64 partial sums are combined by a fixed xor tree (deterministic; not the sequential order of a scatter_add:
// these rows match the oracle to fp32 tolerance, not bit for bit), the result goes back to the owner through LDS.
constexpr int BH_HUB_DEG = 32;
constexpr int BH_HUB_CAP = 128;                 // listed hub rows per graph (more: the rest stay with their owners)
__host__ __device__ constexpr size_t bh_hub_bytes() { return (size_t)BH_HUB_CAP * 16 + (size_t)BH_HUB_CAP * 2 + 16; }
// A workgroup is PERSISTENT over column chunks of ONE graph (chunk w, w + wpg, ...; the launcher picks wpg = workgroups per graph
// so that the grid is one round of the chip): the adjacency is staged ONCE per workgroup instead of once per (graph, chunk) --
// it is as many bytes as the tile itself -- and every row is PLANNED once for all chunks and hops: the tile rows of its first four
// slots (all of most rows of a power grid: mean degree 2.8) as 16-bit ids in two registers, a slot past the row's end pointing
// at a ZERO row behind the tile, plus first slot | degree | hub slot in a third for the rare longer rows.  A hop's first trip
// is then four independent tile reads and three adds: no row-pointer / id reads, no loop, no clamps, no select chain (adding the
// zero row is exact -- the running sum starts as +0 + v0 and is never -0 -- so the bits are those of `slot exists ? acc + v : acc`).
// The next chunk's rows are requested at the start of the LAST hop of the current one (the registers that hold them are dead
// from that hop's tile write on) and arrive under its gathers and stores.  Barriers publish LDS only (lgkmcnt): no thread reads
// another thread's global writes, so nothing waits for the output stores to be acknowledged.
// Same sums in the same order as the one-(graph, chunk)-per-workgroup form of rounds 3-4: bit-identical outputs (checked on the
// GPU against that kernel before it was removed, and build against build for every later step); 6470rte x 64: 284 / 273 ->
// 221 / 218 us per launch (profiles/r05_big_graph_hops_persistent.txt).
// What the compiler needed (1024 threads = 128 registers): the last hop PEELED out of the hop loop (z written in one branch and
// loaded in another doubled its 32 registers: 114-214 spills), and the plan / row offsets made opaque per hop (their unpacked
// fields, 64-bit offsets and lane masks are loop-invariant and were hoisted into 70-180 SGPRs and as many spills).
// A graph with more edges than the staged neighbour list holds (the list is sized for an equal share of the batch's edges plus
// slack): indices from global memory, nothing planned, no register arrays -- a thread re-reads its own hop output to refill the tile.
__device__ __noinline__ void bh_unstaged_graph(float4* bh_tile, int seg, int nchunk, int w, int wpg, int r0, const int* __restrict__ rowptr,
                                               const int* __restrict__ nbr, const float* __restrict__ dinv, const float* __restrict__ xb,
                                               size_t xrow, size_t xchunk, float* __restrict__ xk, size_t stride, int K, int n_total) {
    const int t = threadIdx.x;
    for (int c = w; c < nchunk; c += wpg) {
        __syncthreads();
        for (int row = t; row < seg; row += BH_THREADS) bh_tile[row] = mul4(dinv[r0 + row], ld4(xb + (size_t)c * xchunk + (size_t)row * xrow));
        __syncthreads();
        for (int k = 1; k <= K; ++k) {
            float* outc = xk + (size_t)(k - 1) * stride + ((size_t)c * n_total + r0) * 4;
            for (int row = t; row < seg; row += BH_THREADS) {
                float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
                for (int p = rowptr[r0 + row]; p < rowptr[r0 + row + 1]; ++p) acc = add4(acc, bh_tile[nbr[p] - r0]);
                st4(outc + (size_t)row * 4, mul4(dinv[r0 + row], acc));
            }
            __syncthreads();                    // every read of the tile is done (and this thread's stores are acknowledged)
            if (k == K) break;
            for (int row = t; row < seg; row += BH_THREADS) bh_tile[row] = mul4(dinv[r0 + row], ld4(outc + (size_t)row * 4));
            __syncthreads();
        }
    }
}
__global__ __launch_bounds__(BH_THREADS) void big_graph_hops_kernel(int seg, int nchunk, int ngraphs, int wpg, int nb_cap,
                                                                    const int* __restrict__ rowptr, const int* __restrict__ nbr,
                                                                    const float* __restrict__ dinv, const float* __restrict__ x0,
                                                                    float* __restrict__ xk, size_t stride, int ld, int K, int n_total,
                                                                    int x0_cm) {
    extern __shared__ __attribute__((aligned(16))) float4 bh_tile[];          // [seg + 1] | hub sums [BH_HUB_CAP] | hub rows u16 | count | rp u16 [seg + 2] | nb u16 [nb_cap]
    float4* s_hub_y = bh_tile + seg + 1;       // (tile row `seg` is the ZERO row: what a slot past a row's end reads)
    unsigned short* s_hub_row = reinterpret_cast<unsigned short*>(s_hub_y + BH_HUB_CAP);
    int* s_hub_n = reinterpret_cast<int*>(s_hub_row + BH_HUB_CAP);
    unsigned short* s_rp = reinterpret_cast<unsigned short*>(s_hub_n + 4);
    unsigned short* s_nb = s_rp + ((seg + 2 + 7) & ~7);
    const int b = blockIdx.x, xcd = b & 7, j = b >> 3;
    const int w = j % wpg, gi = (j / wpg) * 8 + xcd;      // the wpg workgroups of a graph sit on ONE XCD (they share its adjacency in L2)
    if (gi >= ngraphs || w >= nchunk) return;
    const int r0 = gi * seg, t = threadIdx.x;
    const int e0 = rowptr[r0], ne = rowptr[r0 + seg] - e0;
    // the layer input: row `row` of chunk c at xb + c * xchunk + row * xrow floats.  (The producing GEMM writes it chunk-major when
    // this kernel will read it: one contiguous run per chunk instead of 16 bytes per 528-byte row.)
    const float* xb = x0_cm ? x0 + (size_t)r0 * 4 : x0 + (size_t)r0 * ld;
    const size_t xrow = x0_cm ? 4 : (size_t)ld, xchunk = x0_cm ? (size_t)n_total * 4 : 4;
    if (!(ne + 4 <= nb_cap && ne < 65536)) {   // (a row's first trip reads four ids from its first slot on, unclamped)
        bh_unstaged_graph(bh_tile, seg, nchunk, w, wpg, r0, rowptr, nbr, dinv, xb, xrow, xchunk, xk, stride, K, n_total);
        return;
    }
    // ---- staging, amortised over the workgroup's chunks.  Rows past the graph's end are clamped to its last row (loaded, planned
    // and summed like it, never stored): straight-line code, no per-row branches around the register arrays.
    float di[BH_RPT];
    float4 z[BH_RPT];
    if (t == 0) {
        s_hub_n[0] = 0;
        bh_tile[seg] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int r = 0; r < BH_RPT; ++r) {
        const int row = min(t + r * BH_THREADS, seg - 1);
        di[r] = dinv[r0 + row];
        z[r] = ld4(xb + (size_t)w * xchunk + (size_t)row * xrow);
    }
    for (int row = t; row <= seg; row += BH_THREADS) s_rp[row] = (unsigned short)(rowptr[r0 + row] - e0);
    for (int i = t; i < ne; i += BH_THREADS) s_nb[i] = (unsigned short)(nbr[e0 + i] - r0);
    __syncthreads();                            // (also publishes the zeroed hub counter)
    // ---- per-row plan, once for every chunk and hop: plan = first slot of the row in the staged list | min(degree, 255) << 16 |
    // (hub slot + 1) << 24 (what the longer rows' later trips and the hub rows need), id01 / id23 = the first trip's tile rows.
    // Hub rows are listed once (the slot order is arrival order -- it decides which wave sums a row, not what the sum is).
    uint32_t plan[BH_RPT], id01[BH_RPT], id23[BH_RPT];
#pragma unroll
    for (int r = 0; r < BH_RPT; ++r) {
        const int rowu = t + r * BH_THREADS, row = min(rowu, seg - 1);
        const int beg = s_rp[row], cnt = (int)s_rp[row + 1] - beg;
        uint32_t hub = 0u;
        if (cnt > BH_HUB_DEG && rowu < seg) {
            const int sl = atomicAdd(s_hub_n, 1);
            if (sl < BH_HUB_CAP) {
                hub = (uint32_t)sl + 1u;
                s_hub_row[sl] = (unsigned short)row;
            }
        }
        plan[r] = (uint32_t)beg | ((uint32_t)min(cnt, 255) << 16) | (hub << 24);
        // ... and the first trip's four tile rows themselves, 16 bits each; a slot past the row's end points at the ZERO row
        const unsigned short* nbp = s_nb + beg;
        const uint32_t i0 = cnt > 0 ? nbp[0] : seg, i1 = cnt > 1 ? nbp[1] : seg, i2 = cnt > 2 ? nbp[2] : seg, i3 = cnt > 3 ? nbp[3] : seg;
        id01[r] = i0 | (i1 << 16);
        id23[r] = i2 | (i3 << 16);
    }
    __syncthreads();
    const int nhub = min(s_hub_n[0], BH_HUB_CAP);
    const int wave_ = t >> 6, lane_ = t & 63;
    // one hop over the tile: hub rows by waves, then every thread its rows; KEEP: the rows' next tile values d_i y_i go to z
    auto hop = [&](auto keep_c, float* outc) {
        constexpr bool KEEP = decltype(keep_c)::value;
        // opaque per hop: what is derived from them -- the rows' global offsets (64-bit), their lane masks (degree > 0..4, hub,
        // row < seg: SGPR pairs) and the unpacked plan fields -- is loop-invariant, gets hoisted out of both loops and spilled
        int to = t;
        asm volatile("" : "+v"(to));
#pragma unroll
        for (int r = 0; r < BH_RPT; ++r) asm volatile("" : "+v"(plan[r]), "+v"(id01[r]), "+v"(id23[r]));
        // ---- hub rows first: one wave per row, lanes stride over its edges, fixed xor tree
        for (int hs = wave_; hs < nhub; hs += BH_THREADS / 64) {
            const int row = s_hub_row[hs];
            const int beg = s_rp[row], end = s_rp[row + 1];
            float4 part = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int p = beg + lane_; p < end; p += 64) part = add4(part, bh_tile[s_nb[p]]);
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) {
                part.x += __shfl_xor(part.x, off);
                part.y += __shfl_xor(part.y, off);
                part.z += __shfl_xor(part.z, off);
                part.w += __shfl_xor(part.w, off);
            }
            if (lane_ == 0) s_hub_y[hs] = part;
        }
        if (nhub > 0) bh_lds_barrier();
#pragma unroll
        for (int r = 0; r < BH_RPT; ++r) {
            const int row = to + r * BH_THREADS;
            const int beg = (int)(plan[r] & 0xffffu), cnt = (int)((plan[r] >> 16) & 255u), hub = (int)(plan[r] >> 24);
            // the first trip: the four tile rows come out of the plan (a slot past the row's end points at the tile's ZERO row): four
            // independent tile reads, three adds.  Adding the zero row is exact -- the running sum starts as +0 + v0 and is never
            // -0 -- so the sum carries the bits of the select chain it replaces; that chain (16 v_cndmask per row and hop, a
            // clamp per slot) had made the hop vector-ALU bound: ~100 vector instructions per row and hop
            const uint32_t i0 = id01[r] & 0xffffu, i1 = id01[r] >> 16, i2 = id23[r] & 0xffffu, i3 = id23[r] >> 16;
            const float4 v0 = bh_tile[i0], v1 = bh_tile[i1], v2 = bh_tile[i2], v3 = bh_tile[i3];
            float4 acc = add4(add4(add4(add4(make_float4(0.f, 0.f, 0.f, 0.f), v0), v1), v2), v3);
            if (hub) {
                acc = s_hub_y[hub - 1];
            } else if (cnt > 4) {   // the rest of a longer row: four slots per trip, dependent index -> tile row reads
                const int rc = min(row, seg - 1);
                const int end = s_rp[rc + 1], last = end - 1;
                for (int p = beg + 4; p < end; p += 4) {
                    int s_[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) s_[u] = s_nb[min(p + u, last)];
                    float4 x_[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) x_[u] = bh_tile[s_[u]];
                    acc = add4(acc, x_[0]);
                    acc = sel4(p + 1 < end, add4(acc, x_[1]), acc);
                    acc = sel4(p + 2 < end, add4(acc, x_[2]), acc);
                    acc = sel4(p + 3 < end, add4(acc, x_[3]), acc);
                }
            }
            const float4 y = mul4(di[r], acc);
            // the hop outputs are written CHUNK-MAJOR ([chunk][row][float4]): this chunk's 6470 x 16 bytes are one contiguous
            // run.  Row-major they were 16 bytes per 528-byte row -- 41 M partial-line writes per direction at 6470rte x 64.
            // The consumers read the layout through GemmTerm::cm_rows (gemm_nt A operand) and TnPair::b_cm_rows (gemm_tn).
            if (row < seg) st4(outc + (size_t)row * 4, y);
            if (KEEP) z[r] = mul4(di[r], y);
        }
        bh_lds_barrier();                       // every read of the tile is done
    };
    for (int c = w; c < nchunk; c += wpg) {
        // ---- the tile of chunk c: z = D^-1/2 x  (every read of the previous chunk's tile is behind the barrier of its last hop)
#pragma unroll
        for (int r = 0; r < BH_RPT; ++r) {
            const int row = t + r * BH_THREADS;
            if (row < seg) bh_tile[row] = mul4(di[r], z[r]);
        }
        bh_lds_barrier();
        float* outc = xk + ((size_t)c * n_total + r0) * 4;
        for (int k = 1; k < K; ++k) {
            hop(std::true_type{}, outc);
#pragma unroll
            for (int r = 0; r < BH_RPT; ++r) {
                const int row = t + r * BH_THREADS;
                if (row < seg) bh_tile[row] = z[r];
            }
            bh_lds_barrier();
            outc += stride;
        }
        // the next chunk's rows are requested now (z is dead until that chunk's tile write) and arrive under the last hop; after
        // the workgroup's last chunk the same load re-reads the current one (unused)
        {
            const int cn = c + wpg < nchunk ? c + wpg : c;
            int to = t;
            asm volatile("" : "+v"(to));
#pragma unroll
            for (int r = 0; r < BH_RPT; ++r) z[r] = ld4(xb + (size_t)cn * xchunk + (size_t)min(to + r * BH_THREADS, seg - 1) * xrow);
        }
        hop(std::false_type{}, outc);
    }
}

// fits: graphs of <= 8,192 nodes whose one-column tile + 16-bit offsets leave room for (most of) the 16-bit neighbour list
bool big_hops_fit(int seg, int n, int64_t e_stored) {
    static const bool off = diag_env("PFN_NO_BIG_HOPS") != nullptr;   // A/B switch: K generic hop launches instead
    if (off || seg <= 0 || n <= 0 || n % seg != 0 || seg > BH_RPT * BH_THREADS || seg >= 65536) return false;
    (void)e_stored;
    return (size_t)(seg + 1) * 16 + bh_hub_bytes() + (size_t)((seg + 2 + 7) & ~7) * 2 + 1024 <= (size_t)160 * 1024;
}

bool tag_uses_big_hops(int seg, int ld, int n, int64_t e_stored, int K) {
    return K > 0 && !fused_hops_fit(seg, ld, n) && big_hops_fit(seg, n, e_stored);
}
bool tag_input_cm(int seg, int ld, int n, int64_t e_stored, int K) {
    return tag_uses_big_hops(seg, ld, n, e_stored, K);
}

int launch_big_graph_hops(const GraphView& g, const FusedHopsArgs& a, hipStream_t s) {
    if (g.n == 0 || a.K == 0) return PFN_OK;
    if (a.transpose) {
        set_error("big-graph hops: only the forward data flow (x -> A x -> A^2 x ...) is built");
        return PFN_EINVAL;
    }
    const int nchunk = a.ld / 4, ngraphs = g.n / a.seg;
    const size_t fixed = (size_t)(a.seg + 1) * 16 + bh_hub_bytes() + (size_t)((a.seg + 2 + 7) & ~7) * 2;
    // neighbour list: an equal share of the (undirected) edges per graph, with slack; the kernel reads indices from global memory
    // for a graph that has more
    const size_t want_nb = (size_t)(2 * (int64_t)g.e_stored / std::max(1, ngraphs) + 64) * 2;
    const size_t lds_total = std::min((size_t)160 * 1024, fixed + want_nb);
    const int nb_cap = (int)((lds_total - fixed) / 2);
    static std::atomic<uint64_t> lds_raised{0};
    PFN_TRY(ensure_dynamic_lds(reinterpret_cast<const void*>(big_graph_hops_kernel), 160 * 1024, lds_raised));
    const bool adjt = a.adjt < 0 ? false : a.adjt != 0;
    ProfScope ps(adjt ? "fused_hops_bwd" : "fused_hops_fwd", 0.0, 0.0, s);
    // workgroups per graph: one round of the chip (one 1024-thread workgroup per CU; four at 6470rte x 64 -- 3 and 8 measured:
    // 284 and 281 us per launch against 245), at least one, at most one per chunk
    const int g8 = ((ngraphs + 7) / 8) * 8;
    const int wpg = std::max(1, std::min(nchunk, device_cus() / std::max(1, g8)));
    const int blocks = g8 * wpg;
    big_graph_hops_kernel<<<blocks, BH_THREADS, lds_total, s>>>(a.seg, nchunk, ngraphs, wpg, nb_cap, adjt ? g.rowptr_out : g.rowptr_in,
                                                               adjt ? g.out_dst : g.in_src, g.dinv, a.x0, a.xk, a.stride, a.ld, a.K, g.n, a.x0_cm);
    PFN_CHECK_LAUNCH();
    return PFN_OK;
}

// ------------------------------------------------------------------------------ EdgeAggregation fwd
// The per-edge Linear(2Fi+Fe -> H) splits into per-node terms P = x W1[:, :Fi]^T + b1, Q = x W1[:, Fi:2Fi]^T
// (node GEMMs) and a per-edge residue sum_f a_e[f] W1[:, 2Fi+f]; the second Linear commutes with the
// segment sum, so only S[i] = sum_e relu(P[i] + Q[src] + residue) is formed per edge.
// LDS holds the Fe residue columns of W1 (strided in the nn.Linear layout) for the whole block.
// The LAST layer's second Linear W2 [fo][h] (Fo <= 4) staged in LDS as w2s[4][ld], rows >= Fo and units >= H zero: the fused
// last-layer kernels (edge_fwd_out_kernel, ds_row) read a thread's column chunk from there when they need it (held in 16
// registers across the walks it cost two to three waves per SIMD of occupancy and made the fusion a wash)
__device__ __forceinline__ void stage_w2(float* w2s, const float* __restrict__ w2, int h, int fo, int ld) {
    for (int i = threadIdx.x; i < 4 * ld; i += blockDim.x) {
        const int o = i / ld, k = i - o * ld;
        w2s[i] = (o < fo && k < h) ? w2[(size_t)o * h + k] : 0.f;
    }
}
__device__ __forceinline__ float dot4(float4 a, float4 b) { return fmaf(a.w, b.w, fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x))); }

// S[row][col..col+3] for one (row, column chunk): the walk over the row's incoming edges
// What a (row, chunk) walk needs before its first edge: requested BEFORE the barrier that publishes the block's residue weights
// (behind it these loads were one more serial round trip; a load cannot move across __syncthreads by itself)
struct EdgeRowHead { float4 p4; int beg, end, b4, b4n; };
// FLY (EdgeFwdArgs::x0): the first layer behind the 4-wide front.  `P` and `Q` are both the N x 4 array x0; the head carries the
// row's x0 and the walk forms P[row] = b1 + W1i x0[row], Q[src] = W1j x0[src] per chunk with the fma chains of front.hip (same
// operands, same order: the bits the front would have stored) from weights staged behind the residue weights in LDS.
template <bool MASK, bool FLY = false>
__device__ __forceinline__ EdgeRowHead edge_row_head(int row, int col, const int* __restrict__ rowptr, const float* __restrict__ P,
                                                     int ld, const int* __restrict__ rp4) {
    EdgeRowHead hd;
    hd.p4 = FLY ? ld4(P + (size_t)row * 4) : ld4(P + (size_t)row * ld + col);
    hd.beg = rowptr[row];
    hd.end = rowptr[row + 1];
    hd.b4 = MASK ? rp4[row] : 0;
    hd.b4n = MASK ? rp4[row + 1] : 0;
    return hd;
}
// the front's chains (front.hip): p = b1 + sum_f W1[u][f] x[f], q = sum_f W1[u][4 + f] x[f], f ascending, one fma per term
__device__ __forceinline__ float4 fly_chain(float4 x, float4 w0, float4 w1, float4 w2, float4 w3, float4 acc) {
    acc = fma4(x.x, w0, acc);
    acc = fma4(x.y, w1, acc);
    acc = fma4(x.z, w2, acc);
    return fma4(x.w, w3, acc);
}
// stages wi [4][ld] | wj [4][ld] | b1 [ld] (zero past H) for the FLY walks
__device__ __forceinline__ void stage_fly_weights(float* wf, const float* __restrict__ w1, const float* __restrict__ b1, int ld, int h,
                                                  int ldw) {
    for (int i = threadIdx.x; i < 9 * ld; i += blockDim.x) {
        const int f = i / ld, k = i - f * ld;
        wf[i] = k < h ? (f < 8 ? w1[(size_t)k * ldw + f] : b1[k]) : 0.f;
    }
}
template <int FE, bool MASK, bool FLY = false>   // MASK: also save the ReLU masks (EdgeFwdArgs::mask) -- a backward pass will follow
__device__ __forceinline__ float4 edge_sum_chunk(int row, int col, int e_stored, const int* __restrict__ rowptr,
                                                 const int* __restrict__ nbr, const int* __restrict__ eid,
                                                 const float* __restrict__ P, const float* __restrict__ Q,
                                                 const float* __restrict__ ea, const float* we, int ld, int fe,
                                                 unsigned* __restrict__ mask, const EdgeRowHead& hd) {
    float4 p4 = hd.p4;
    float4 wj0, wj1, wj2, wj3;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (FLY) {
        const float* wf = we + fe * ld;          // wi [4][ld] | wj [4][ld] | b1 [ld]
        p4 = fly_chain(hd.p4, ld4(wf + col), ld4(wf + ld + col), ld4(wf + 2 * ld + col), ld4(wf + 3 * ld + col), ld4(wf + 8 * ld + col));
        wj0 = ld4(wf + 4 * ld + col); wj1 = ld4(wf + 5 * ld + col); wj2 = ld4(wf + 6 * ld + col); wj3 = ld4(wf + 7 * ld + col);
    }
    // this (row, chunk)'s run of mask dwords (EdgeFwdArgs::mask): one dword per trip of four slots
    unsigned* mrun = nullptr;
    if (MASK) mrun = mask + (size_t)hd.b4 * (ld >> 2) + (size_t)(col >> 2) * (hd.b4n - hd.b4);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const int beg = hd.beg, end = hd.end;
    if (FE == 2) {
        // four edge slots per trip (see hop_kernel): indices first, then all gathers, then the sums in edge-id order
        const float4 w0 = ld4(we + col), w1 = ld4(we + ld + col);
        for (int p = beg; p < end; p += 4) {
            const int last = end - 1;
            int s_[4], id_[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int q = min(p + u, last);
                s_[u] = nbr[q];
                const int id = eid[q];
                id_[u] = id >= e_stored ? id - e_stored : id;
            }
            float4 q_[4];
            float2 a_[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                q_[u] = FLY ? ld4(Q + (size_t)s_[u] * 4) : ld4(Q + (size_t)s_[u] * ld + col);
                a_[u] = *reinterpret_cast<const float2*>(ea + (size_t)id_[u] * 2);
            }
            if (FLY) {
#pragma unroll
                for (int u = 0; u < 4; ++u) q_[u] = fly_chain(q_[u], wj0, wj1, wj2, wj3, zero4);
            }
            unsigned mword = 0u;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float4 v = add4(p4, q_[u]);
                v = fma4(a_[u].x, w0, v);
                v = fma4(a_[u].y, w1, v);
                acc = sel4(p + u < end, add4(acc, relu4(v)), acc);
                if (MASK) mword |= (p + u < end ? (unsigned)relu_bits(v) : 0u) << (8 * u);
            }
            if (MASK) mrun[(p - beg) >> 2] = mword;
        }
    } else {
        for (int p = beg; p < end; ++p) {
            const int s = nbr[p];
            int id = eid[p];
            id = id >= e_stored ? id - e_stored : id;
            float4 v = add4(p4, ld4(Q + (size_t)s * ld + col));
            for (int f = 0; f < fe; ++f) v = fma4(ea[(size_t)id * fe + f], ld4(we + f * ld + col), v);
            acc = add4(acc, relu4(v));
            if (MASK) reinterpret_cast<unsigned char*>(mrun)[p - beg] = relu_bits(v);   // (generic Fe: byte by byte)
        }
    }
    return acc;
}

template <int FE, bool MASK, bool FLY = false>
__global__ __launch_bounds__(256) void edge_fwd_kernel(int n, int nchunk, int e_stored, const int* __restrict__ rowptr,
                                                       const int* __restrict__ nbr, const int* __restrict__ eid,
                                                       const float* __restrict__ P, const float* __restrict__ Q,
                                                       const float* __restrict__ ea, const float* __restrict__ w1,
                                                       float* __restrict__ S, int ld, int h, int fi, int fe_rt,
                                                       unsigned* __restrict__ mask, const int* __restrict__ rp4,
                                                       const float* __restrict__ b1, int reverse) {
    extern __shared__ __attribute__((aligned(16))) float we[];   // [fe][ld] (| FLY: wi [4][ld] | wj [4][ld] | b1 [ld])
    const int fe = FE > 0 ? FE : fe_rt;
    const int ldw = 2 * fi + fe;
    // PERSISTENT over the (row, chunk) items (grid = the workgroups the chip holds at once, launch_edge_fwd): the residue weights
    // are staged -- a round of loads and a barrier -- once per workgroup instead of once per 256 items, and an item's head (row
    // pointers, its P chunk) is requested while the item before it is walked: one level less in every item's chain of dependent loads
    const long stride = (long)gridDim.x * blockDim.x, total = (long)n * nchunk;
    long item = (long)blockIdx.x * blockDim.x + threadIdx.x;
    // (serpentine sweeps, pfn_internal.hpp: `reverse` walks the items from the last to the first -- item i stands for total - 1 - i)
    const long flip = reverse ? total - 1 : 0, sgn = reverse ? -1 : 1;
    int row = (int)((flip + sgn * item) / nchunk);
    int col = (int)((flip + sgn * item) - (long)row * nchunk) * 4;
    EdgeRowHead hd;
    if (item < total) hd = edge_row_head<MASK, FLY>(row, col, rowptr, P, ld, rp4);
    for (int i = threadIdx.x; i < fe * ld; i += blockDim.x) {
        const int f = i / ld, k = i - f * ld;
        we[i] = k < h ? w1[(size_t)k * ldw + 2 * fi + f] : 0.f;
    }
    if (FLY) stage_fly_weights(we + fe * ld, w1, b1, ld, h, ldw);
    __syncthreads();
    while (item < total) {
        const long nitem = item + stride, nphys = nitem < total ? flip + sgn * nitem : 0;
        const int nrow = (int)(nphys / nchunk), ncol = (int)(nphys - (long)nrow * nchunk) * 4;
        EdgeRowHead hn = hd;
        if (nitem < total) hn = edge_row_head<MASK, FLY>(nrow, ncol, rowptr, P, ld, rp4);
        st4_wt(S + (size_t)row * ld + col, edge_sum_chunk<FE, MASK, FLY>(row, col, e_stored, rowptr, nbr, eid, P, Q, ea, we, ld, fe, mask, hd));
        hd = hn;
        item = nitem;
        row = nrow;
        col = ncol;
    }
}

// The network's LAST EdgeAggregation layer (Fo <= 4, no activation): the second Linear rides in the same launch.  Block =
// rows_pb rows x nchunk chunk-lanes (the front's mapping, front.hip): a lane forms its S chunk, multiplies it with its W2
// column chunk (registers), and the row's nchunk float4 partials are added in a FIXED two-level order:
//   out[row][o] = sum_u S[row][u] W2[o][u] + deg[row] * b2[o]
// S is still written (the backward pairs it with gout for dW2).  LDS: we[FE][ld] | part[rows_pb][nchunk] float4.
template <int FE, bool MASK>
__global__ __launch_bounds__(256) void edge_fwd_out_kernel(int n, int nchunk, int rows_pb, int e_stored,
                                                           const int* __restrict__ rowptr, const int* __restrict__ nbr,
                                                           const int* __restrict__ eid, const float* __restrict__ P,
                                                           const float* __restrict__ Q, const float* __restrict__ ea,
                                                           const float* __restrict__ w1, float* __restrict__ S,
                                                           const float* __restrict__ w2, const float* __restrict__ b2,
                                                           const float* __restrict__ deg, float* __restrict__ out, int ld,
                                                           int h, int fi, int fo, unsigned* __restrict__ mask, const int* __restrict__ rp4,
                                                           int reverse) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* we = smem;
    float* w2s = smem + FE * ld;
    float4* part = reinterpret_cast<float4*>(smem + (FE + 4) * ld);
    const int ldw = 2 * fi + FE;
    for (int i = threadIdx.x; i < FE * ld; i += blockDim.x) {
        const int f = i / ld, k = i - f * ld;
        we[i] = k < h ? w1[(size_t)k * ldw + 2 * fi + f] : 0.f;
    }
    stage_w2(w2s, w2, h, fo, ld);
    const int r = threadIdx.x / nchunk, c = threadIdx.x - r * nchunk;
    const int row = (reverse ? (int)(gridDim.x - 1 - blockIdx.x) : (int)blockIdx.x) * rows_pb + r, col = 4 * c;   // (serpentine sweeps)
    const bool on = r < rows_pb && row < n;
    EdgeRowHead hd;
    if (on) hd = edge_row_head<MASK>(row, col, rowptr, P, ld, rp4);
    __syncthreads();
    if (on) {
        const float4 s4 = edge_sum_chunk<FE, MASK>(row, col, e_stored, rowptr, nbr, eid, P, Q, ea, we, ld, FE, mask, hd);
        st4(S + (size_t)row * ld + col, s4);
        float4 o;
        o.x = dot4(s4, ld4(w2s + col));
        o.y = dot4(s4, ld4(w2s + ld + col));
        o.z = dot4(s4, ld4(w2s + 2 * ld + col));
        o.w = dot4(s4, ld4(w2s + 3 * ld + col));
        part[r * nchunk + c] = o;
    }
    __syncthreads();
    row_sum(part, r, c, nchunk, on);   // (barriers inside: every thread calls it)
    if (on && c == 0) {
        const float4 t = part[r * nchunk];
        const float d = deg[row];
        float4 o;
        o.x = fmaf(d, b2[0], t.x);
        o.y = fo > 1 ? fmaf(d, b2[1], t.y) : 0.f;
        o.z = fo > 2 ? fmaf(d, b2[2], t.z) : 0.f;
        o.w = fo > 3 ? fmaf(d, b2[3], t.w) : 0.f;
        st4(out + (size_t)row * 4, o);
    }
}

// ---- the edge stage of big batches of SMALL graphs (case118 x 2048, inference): block = whole graphs x all column chunks, the
// graphs' Q rows LDS-resident (one tile, as row_hops_kernel), adjacency and edge attributes staged in slot order; P and S move as
// full 528-byte rows.  The generic kernel gathers every Q row from L2 / HBM once per edge (691 MB of fabric traffic per launch
// for 383 MB of P + Q + S); here Q is read once.  Same arithmetic, same edge order as edge_sum_chunk.
constexpr int ER_THREADS = 512;
constexpr int ER_IPT = 8;                       // (row, chunk) items per thread at most: rows x chunks <= 4,096 per block
constexpr int ER_NBPT = 4;                      // staged edge slots per thread at most: 2,048 per block
template <bool FLY>   // FLY (EdgeFwdArgs::x0): P = Q = x0 (N x 4); the tile holds the block's x0 rows, wf the front's weights
__global__ __launch_bounds__(ER_THREADS, 4) void edge_rows_fwd_kernel(int n, int rows_pb, int nchunk, int nb_cap, int e_stored,
                                                                     const int* __restrict__ rowptr, const int* __restrict__ nbr,
                                                                     const int* __restrict__ eid, const float* __restrict__ P,
                                                                     const float* __restrict__ Q, const float* __restrict__ ea,
                                                                     const float* __restrict__ w1, float* __restrict__ S, int ld,
                                                                     int h, int fi, const float* __restrict__ w2,
                                                                     const float* __restrict__ b2, const float* __restrict__ deg,
                                                                     float* __restrict__ out, int fo, const float* __restrict__ b1, int reverse) {
    // `out` != null: the network's LAST layer (Fo <= 4): out[row] = S[row] W2^T + deg[row] b2 is formed here and S itself (which
    // only a backward pass reads) is not written
    extern __shared__ __attribute__((aligned(16))) float4 er_tile[];   // Q [rows_pb * nchunk] | we [2 * nchunk] | w2 [4 * nchunk] | ea float2 [nb_cap] | rp u16 | nb u16
    // (FLY: x0 [rows_pb] | we [2 * nchunk] | wf [9 * nchunk] | ea | rp | nb)
    float4* s_we = er_tile + (FLY ? (size_t)rows_pb : (size_t)rows_pb * nchunk);
    float4* s_w2 = s_we + 2 * nchunk;            // (FLY: wi [4] | wj [4] | b1, nchunk float4 each)
    float2* s_ea = reinterpret_cast<float2*>(s_w2 + (FLY ? 9 : 4) * nchunk);
    unsigned short* s_rp = reinterpret_cast<unsigned short*>(s_ea + nb_cap);
    unsigned short* s_nb = s_rp + ((rows_pb + 2 + 7) & ~7);
    const int r0 = (reverse ? (int)(gridDim.x - 1 - blockIdx.x) : (int)blockIdx.x) * rows_pb, rows = min(rows_pb, n - r0), t = threadIdx.x;
    const int items = rows * nchunk;
    const int e0 = rowptr[r0], ne = rowptr[r0 + rows] - e0;
    const bool in_lds = ne <= nb_cap && ne < 65536 && ne <= ER_NBPT * ER_THREADS;
    const int ldw = 2 * fi + 2;
    // every global load of the prologue is requested before the first LDS store (the attributes need their edge id first: two trips)
    float4 q[ER_IPT];
    if (!FLY) {
#pragma unroll
        for (int r = 0; r < ER_IPT; ++r) {
            const int i = t + r * ER_THREADS;
            const int row = i / nchunk, lc = i - row * nchunk;
            q[r] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < items) q[r] = ld4(Q + (size_t)(r0 + row) * ld + 4 * lc);
        }
    } else {
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) q[k2] = t + k2 * ER_THREADS < rows ? ld4(Q + (size_t)(r0 + t + k2 * ER_THREADS) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    int rpv[2], nbv[ER_NBPT], idv[ER_NBPT];
#pragma unroll
    for (int k2 = 0; k2 < 2; ++k2) rpv[k2] = t + k2 * ER_THREADS <= rows ? rowptr[r0 + t + k2 * ER_THREADS] : 0;
#pragma unroll
    for (int jn = 0; jn < ER_NBPT; ++jn) {
        const int i = t + jn * ER_THREADS;
        const bool on = in_lds && i < ne;
        nbv[jn] = on ? nbr[e0 + i] : 0;
        idv[jn] = on ? eid[e0 + i] : 0;
    }
    float wv[2] = {0.f, 0.f};                    // the residue weights W1[:, 2 Fi + f], f = 0, 1, as two [ld] rows
    if (t < 2 * 4 * nchunk) {
        const int f = t / (4 * nchunk), k = t - f * 4 * nchunk;
        wv[0] = k < h ? w1[(size_t)k * ldw + 2 * fi + f] : 0.f;
    }
    float2 eav[ER_NBPT];
#pragma unroll
    for (int jn = 0; jn < ER_NBPT; ++jn) {
        const int id = idv[jn] >= e_stored ? idv[jn] - e_stored : idv[jn];
        eav[jn] = (in_lds && t + jn * ER_THREADS < ne) ? *reinterpret_cast<const float2*>(ea + (size_t)id * 2) : make_float2(0.f, 0.f);
    }
    if (!FLY) {
#pragma unroll
        for (int r = 0; r < ER_IPT; ++r)
            if (t + r * ER_THREADS < items) er_tile[t + r * ER_THREADS] = q[r];
    } else {
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2)
            if (t + k2 * ER_THREADS < rows) er_tile[t + k2 * ER_THREADS] = q[k2];
        stage_fly_weights(reinterpret_cast<float*>(s_w2), w1, b1, ld, h, ldw);
    }
#pragma unroll
    for (int k2 = 0; k2 < 2; ++k2)
        if (t + k2 * ER_THREADS <= rows) s_rp[t + k2 * ER_THREADS] = (unsigned short)(rpv[k2] - e0);
    if (t < 2 * 4 * nchunk) reinterpret_cast<float*>(s_we)[t] = wv[0];
    if (!FLY && out) stage_w2(reinterpret_cast<float*>(s_w2), w2, h, fo, ld);
    if (in_lds) {
#pragma unroll
        for (int jn = 0; jn < ER_NBPT; ++jn) {
            const int i = t + jn * ER_THREADS;
            if (i < ne) {
                s_nb[i] = (unsigned short)(nbv[jn] - r0);
                s_ea[i] = eav[jn];
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < ER_IPT; ++r) {
        const int i = t + r * ER_THREADS;
        if (i >= items) continue;
        const int row = i / nchunk, lc = i - row * nchunk;
        float4 p4;
        float4 wj0, wj1, wj2, wj3;
        const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (FLY) {
            p4 = fly_chain(er_tile[row], s_w2[lc], s_w2[nchunk + lc], s_w2[2 * nchunk + lc], s_w2[3 * nchunk + lc], s_w2[8 * nchunk + lc]);
            wj0 = s_w2[4 * nchunk + lc]; wj1 = s_w2[5 * nchunk + lc]; wj2 = s_w2[6 * nchunk + lc]; wj3 = s_w2[7 * nchunk + lc];
        } else {
            p4 = ld4(P + (size_t)(r0 + row) * ld + 4 * lc);
        }
        const float4 w0 = s_we[lc], w1v = s_we[nchunk + lc];
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (in_lds) {
            const int beg = s_rp[row], end = s_rp[row + 1], last = end - 1;
            for (int p = beg; p < end; p += 4) {
                int s_[4];
                float2 a_[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int qq = min(p + u, last);
                    s_[u] = s_nb[qq];
                    a_[u] = s_ea[qq];
                }
                float4 q_[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) q_[u] = FLY ? fly_chain(er_tile[s_[u]], wj0, wj1, wj2, wj3, zero4) : er_tile[s_[u] * nchunk + lc];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    float4 v = add4(p4, q_[u]);
                    v = fma4(a_[u].x, w0, v);
                    v = fma4(a_[u].y, w1v, v);
                    acc = sel4(p + u < end, add4(acc, relu4(v)), acc);
                }
            }
        } else {            // a block with more edges than the staged lists hold: indices and attributes from global memory
            for (int p = rowptr[r0 + row]; p < rowptr[r0 + row + 1]; ++p) {
                int id = eid[p];
                id = id >= e_stored ? id - e_stored : id;
                const float2 a2 = *reinterpret_cast<const float2*>(ea + (size_t)id * 2);
                float4 v = add4(p4, FLY ? fly_chain(er_tile[nbr[p] - r0], wj0, wj1, wj2, wj3, zero4) : er_tile[(nbr[p] - r0) * nchunk + lc]);
                v = fma4(a2.x, w0, v);
                v = fma4(a2.y, w1v, v);
                acc = add4(acc, relu4(v));
            }
        }
        if (FLY || !out) {
            st4(S + (size_t)(r0 + row) * ld + 4 * lc, acc);
        } else {   // this chunk's share of the four output dot products; q[] (dead since the prologue) keeps it until the tile is free
            q[r].x = dot4(acc, s_w2[lc]);
            q[r].y = dot4(acc, s_w2[nchunk + lc]);
            q[r].z = dot4(acc, s_w2[2 * nchunk + lc]);
            q[r].w = dot4(acc, s_w2[3 * nchunk + lc]);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    if (FLY || !out) return;
    __syncthreads();                             // every walk is done with the Q tile: it now holds the partials
#pragma unroll
    for (int r = 0; r < ER_IPT; ++r)
        if (t + r * ER_THREADS < items) er_tile[t + r * ER_THREADS] = q[r];
    __syncthreads();
    for (int row = t; row < rows; row += ER_THREADS) {   // a row's chunks added in chunk order (fixed)
        float4 sum = er_tile[row * nchunk];
        for (int k2 = 1; k2 < nchunk; ++k2) sum = add4(sum, er_tile[row * nchunk + k2]);
        const float d = deg[r0 + row];
        float4 o;
        o.x = fmaf(d, b2[0], sum.x);
        o.y = fo > 1 ? fmaf(d, b2[1], sum.y) : 0.f;
        o.z = fo > 2 ? fmaf(d, b2[2], sum.z) : 0.f;
        o.w = fo > 3 ? fmaf(d, b2[3], sum.w) : 0.f;
        st4(out + (size_t)(r0 + row) * 4, o);
    }
}
static int edge_rows_graphs_per_block(int seg, int nchunk) {
    if (seg <= 0 || seg > 1023 || (long)seg * nchunk > (long)ER_IPT * ER_THREADS || 8 * nchunk > ER_THREADS) return 0;
    int gpb = std::min((ER_IPT * ER_THREADS) / (seg * nchunk), 1023 / seg);
    while (gpb > 0 && (size_t)gpb * seg * nchunk * 16 + 4096 > (size_t)66 * 1024) --gpb;   // (+ attributes, adjacency: <= 80 KB in all)
    return gpb;
}

bool edge_fwd_out_ok(int fe, int h, int fo, int ldo) { return fe == 2 && fo >= 1 && fo <= 4 && ldo == 4 && ld_of(h) / 4 <= 256; }
int launch_edge_fwd(const GraphView& g, const EdgeFwdArgs& a, hipStream_t s) {
    const int rev = g.n > 0 ? next_sweep_direction() : 0;   // every kernel of this launcher walks its rows either way (serpentine sweeps)
    if (g.n == 0) return PFN_OK;
    const int nchunk = a.ld / 4;
    const bool fly = a.x0 != nullptr;
    if (fly && (a.fe != 2 || a.fi != 4 || !a.b1 || a.out || a.P || a.Q)) {
        set_error("edge stage: P | Q from x0 needs Fi = 4, Fe = 2, a bias and a layer that is not the last (internal)");
        return PFN_EINVAL;
    }
    if (a.seg > 0 && a.fe == 2 && !a.mask && g.n % a.seg == 0) {   // inference on a big batch of small graphs
        static const bool off = diag_env("PFN_NO_EDGE_ROWS") != nullptr;      // A/B switch: the generic gather kernel
        const int ngraphs = g.n / a.seg, gpb = edge_rows_graphs_per_block(a.seg, nchunk);
        if (!off && gpb > 0 && (long)(ngraphs + gpb - 1) / gpb >= 4L * device_cus()) {
            const int rows_pb = gpb * a.seg;
            const size_t fixed = (fly ? (size_t)rows_pb * 16 + (size_t)11 * nchunk * 16 : (size_t)rows_pb * nchunk * 16 + (size_t)6 * nchunk * 16) +
                                 (size_t)((rows_pb + 2 + 7) & ~7) * 2;
            const size_t per_slot = 8 + 2;
            const size_t want = (size_t)(2 * (int64_t)g.e_stored / std::max(1, ngraphs) * gpb + 64);
            const size_t lds_total = std::min((size_t)80 * 1024, fixed + want * per_slot + 16);
            const int nb_cap = (int)((lds_total - fixed - 16) / per_slot) & ~3;
            static std::atomic<uint64_t> lds_raised_er{0}, lds_raised_erf{0};
            ProfScope ps("edge_rows_fwd", 0.0, 0.0, s);
            if (fly) {
                PFN_TRY(ensure_dynamic_lds(reinterpret_cast<const void*>(edge_rows_fwd_kernel<true>), 160 * 1024, lds_raised_erf));
                edge_rows_fwd_kernel<true><<<(g.n + rows_pb - 1) / rows_pb, ER_THREADS, lds_total, s>>>(
                    g.n, rows_pb, nchunk, nb_cap, g.e_stored, g.rowptr_in, g.in_src, g.in_eid, a.x0, a.x0, a.edge_attr, a.w1, a.S, a.ld, a.h,
                    a.fi, nullptr, nullptr, g.deg, nullptr, 0, a.b1, rev);
            } else {
                PFN_TRY(ensure_dynamic_lds(reinterpret_cast<const void*>(edge_rows_fwd_kernel<false>), 160 * 1024, lds_raised_er));
                edge_rows_fwd_kernel<false><<<(g.n + rows_pb - 1) / rows_pb, ER_THREADS, lds_total, s>>>(
                    g.n, rows_pb, nchunk, nb_cap, g.e_stored, g.rowptr_in, g.in_src, g.in_eid, a.P, a.Q, a.edge_attr, a.w1, a.S, a.ld, a.h, a.fi,
                    a.w2, a.b2, g.deg, a.out, a.fo, nullptr, rev);
            }
            PFN_CHECK_LAUNCH();
            return PFN_OK;
        }
    }
    const long items = (long)g.n * nchunk;
    const int all_blocks = (int)((items + 255) / 256);
    const size_t lds = (size_t)a.fe * a.ld * sizeof(float);
    // the generic walk is persistent: as many workgroups as the chip holds at once (PFN_EDGE_FWD_BPC tunes)
    static const int bpc_env = diag_env("PFN_EDGE_FWD_BPC") ? std::max(1, atoi(diag_env("PFN_EDGE_FWD_BPC"))) : 0;
    const int bpc = bpc_env > 0 ? bpc_env : (fly ? 5 : 7);   // (what the kernels' 86 / 72 VGPRs let a CU hold)
    const int blocks = (int)std::min<long>(all_blocks, (long)bpc * device_cus());
    ProfScope ps("edge_fwd", 0.0, 0.0, s);
    if (a.out) {   // last layer: S and out = S W2^T + deg b2 in one launch (edge_fwd_out_ok)
        const int rows_pb = 256 / nchunk;
        const size_t lds_out = lds + (size_t)4 * a.ld * sizeof(float) + (size_t)rows_pb * nchunk * sizeof(float4);
        const dim3 grid_out((g.n + rows_pb - 1) / rows_pb);
        if (a.mask)
            edge_fwd_out_kernel<2, true><<<grid_out, 256, lds_out, s>>>(g.n, nchunk, rows_pb, g.e_stored, g.rowptr_in, g.in_src, g.in_eid,
                                                                      a.P, a.Q, a.edge_attr, a.w1, a.S, a.w2, a.b2, g.deg, a.out, a.ld,
                                                                      a.h, a.fi, a.fo, a.mask, g.rp4, rev);
        else
            edge_fwd_out_kernel<2, false><<<grid_out, 256, lds_out, s>>>(g.n, nchunk, rows_pb, g.e_stored, g.rowptr_in, g.in_src, g.in_eid,
                                                                       a.P, a.Q, a.edge_attr, a.w1, a.S, a.w2, a.b2, g.deg, a.out, a.ld,
                                                                       a.h, a.fi, a.fo, nullptr, nullptr, rev);
        PFN_CHECK_LAUNCH();
        return PFN_OK;
    }
#define PFN_EDGE_FWD(FE_, M_)                                                                                                  \
    edge_fwd_kernel<FE_, M_><<<blocks, 256, lds, s>>>(g.n, nchunk, g.e_stored, g.rowptr_in, g.in_src, g.in_eid, a.P, a.Q,          \
                                                      a.edge_attr, a.w1, a.S, a.ld, a.h, a.fi, a.fe, a.mask, g.rp4, nullptr, rev)
#define PFN_EDGE_FWD_FLY(M_)                                                                                                   \
    edge_fwd_kernel<2, M_, true><<<blocks, 256, lds + (size_t)9 * a.ld * sizeof(float), s>>>(                                  \
        g.n, nchunk, g.e_stored, g.rowptr_in, g.in_src, g.in_eid, a.x0, a.x0, a.edge_attr, a.w1, a.S, a.ld, a.h, a.fi, a.fe, a.mask, g.rp4, a.b1, rev)
    if (fly && a.mask) PFN_EDGE_FWD_FLY(true);
    else if (fly) PFN_EDGE_FWD_FLY(false);
    else if (a.fe == 2 && a.mask) PFN_EDGE_FWD(2, true);
    else if (a.fe == 2) PFN_EDGE_FWD(2, false);
    else if (a.mask) PFN_EDGE_FWD(0, true);
    else PFN_EDGE_FWD(0, false);
#undef PFN_EDGE_FWD
#undef PFN_EDGE_FWD_FLY
    PFN_CHECK_LAUNCH();
    return PFN_OK;
}

// ------------------------------------------------------------------------------ EdgeAggregation bwd
// dh_e = dS[dst(e)] where the recomputed pre-activation P[dst] + Q[src] + residue is > 0.
//   dst walk (by-destination CSR):  dP[i] = sum_{e -> i} dh_e ;  dWe[f] += a_e[f] * dh_e
//   src walk (by-source CSR):       dQ[j] = sum_{e: src(e) = j} dh_e
// dWe is reduced deterministically: per-block ordered partial -> launch_dwe_reduce.
#define PFN_MAX_FE 8
// edge slots per trip of the backward walks: four (as in the forward kernels) cost 86 VGPRs -> 5 waves per SIMD and made the
// kernel SLOWER (492 -> 566 us at 6470rte x 64); two keep 8 waves per SIMD
constexpr int BW_SLOTS = 2;

// dS row chunk for the LAST EdgeAggregation layer (Fo <= 4): dS[row][col..col+3] = sum_o gout[row][o] * W2[o][col..col+3], formed
// from the 16-byte gout row and the W2 column chunk in LDS instead of reading an N x H dS that a K = 4 GEMM wrote
template <bool DSG>
__device__ __forceinline__ float4 ds_row(const float* __restrict__ dS, int row, int ld, int col, const float* w2s) {
    if (!DSG) return ld4(dS + (size_t)row * ld + col);
    const float4 g = ld4(dS + (size_t)row * 4);          // gout row, ld 4 (columns >= Fo are zero-weighted)
    int co = col;
    asm volatile("" : "+v"(co));                         // re-read the chunk here: hoisted out of the walk it is 16 live VGPRs
    float4 r = mul4(g.x, ld4(w2s + co));
    r = fma4(g.y, ld4(w2s + ld + co), r);
    r = fma4(g.z, ld4(w2s + 2 * ld + co), r);
    r = fma4(g.w, ld4(w2s + 3 * ld + co), r);
    return r;
}

template <int FE, bool DSG>
__device__ __forceinline__ void edge_bwd_dst_body(int bid, int nblk, int n, int nchunk, int bdx, int bdy, int e_stored,
                                                           const int* __restrict__ rowptr, const int* __restrict__ nbr,
                                                           const int* __restrict__ eid, const float* __restrict__ P,
                                                           const float* __restrict__ Q, const float* __restrict__ dS,
                                                           const float* __restrict__ ea, const float* __restrict__ w1,
                                                           float* __restrict__ dP, float* __restrict__ dWe_partial,
                                                           int ld, int h, int fi, const float* __restrict__ w2, int fo) {
    // Block = bdx column chunks x bdy rows (bdx * bdy <= 256, lanes run along the columns of one row, then the
    // next row); a thread keeps ONE column chunk for every row it visits, so the dWe partial sums stay in
    // registers across the block's whole row range and each block emits a single ordered partial.
    extern __shared__ __attribute__((aligned(16))) float smem[];   // we[FE][ld] | part[256][FE] float4
    float* we = smem;
    float4* part = reinterpret_cast<float4*>(smem + FE * ld);
    float* w2s = smem + FE * ld + 256 * FE * 4;                    // (DSG only) W2 [4][ld]
    const int ldw = 2 * fi + FE;
    for (int i = threadIdx.x; i < FE * ld; i += blockDim.x) {
        const int f = i / ld, k = i - f * ld;
        we[i] = k < h ? w1[(size_t)k * ldw + 2 * fi + f] : 0.f;
    }
    if (DSG) stage_w2(w2s, w2, h, fo, ld);
    __syncthreads();
    const int ty = threadIdx.x / bdx, tx = threadIdx.x - ty * bdx;
    const bool active = ty < bdy;
    for (int c0 = 0; c0 < nchunk; c0 += bdx) {
        const int c = c0 + tx, col = 4 * c;
        float4 dwe[FE];
#pragma unroll
        for (int f = 0; f < FE; ++f) dwe[f] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (active && c < nchunk) {
            float4 w4[FE];
#pragma unroll
            for (int f = 0; f < FE; ++f) w4[f] = ld4(we + f * ld + col);
            for (int row = bid * bdy + ty; row < n; row += nblk * bdy) {
                const float4 p4 = ld4(P + (size_t)row * ld + col);
                const float4 g4 = ds_row<DSG>(dS, row, ld, col, w2s);
                float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
                const int beg = rowptr[row], end = rowptr[row + 1];
                // BW_SLOTS edge slots per trip (see hop_kernel): a slot past the row's end re-reads the last edge and
                // contributes dh = 0 (an exact no-op on acc and dWe), sums stay in edge-id order
                for (int p = beg; p < end; p += BW_SLOTS) {
                    const int last = end - 1;
                    int s_[BW_SLOTS], id_[BW_SLOTS];
#pragma unroll
                    for (int u = 0; u < BW_SLOTS; ++u) {
                        const int q = min(p + u, last);
                        s_[u] = nbr[q];
                        const int id = eid[q];
                        id_[u] = id >= e_stored ? id - e_stored : id;
                    }
                    float4 q_[BW_SLOTS];
                    float a_[BW_SLOTS][FE];
#pragma unroll
                    for (int u = 0; u < BW_SLOTS; ++u) {
                        q_[u] = ld4(Q + (size_t)s_[u] * ld + col);
#pragma unroll
                        for (int f = 0; f < FE; ++f) a_[u][f] = ea[(size_t)id_[u] * FE + f];
                    }
#pragma unroll
                    for (int u = 0; u < BW_SLOTS; ++u) {
                        float4 v = add4(p4, q_[u]);
#pragma unroll
                        for (int f = 0; f < FE; ++f) v = fma4(a_[u][f], w4[f], v);
                        const bool k = p + u < end;
                        float4 dh;
                        dh.x = (k && v.x > 0.f) ? g4.x : 0.f;
                        dh.y = (k && v.y > 0.f) ? g4.y : 0.f;
                        dh.z = (k && v.z > 0.f) ? g4.z : 0.f;
                        dh.w = (k && v.w > 0.f) ? g4.w : 0.f;
                        acc = add4(acc, dh);
#pragma unroll
                        for (int f = 0; f < FE; ++f) dwe[f] = fma4(a_[u][f], dh, dwe[f]);
                    }
                }
                st4(dP + (size_t)row * ld + col, acc);
            }
        }
        // ordered in-block reduction over the bdy row lanes of each column chunk
        if (active) {
#pragma unroll
            for (int f = 0; f < FE; ++f) part[threadIdx.x * FE + f] = dwe[f];
        }
        __syncthreads();
        if (ty == 0 && c < nchunk) {
            float4 sum[FE];
#pragma unroll
            for (int f = 0; f < FE; ++f) sum[f] = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int y = 0; y < bdy; ++y) {
#pragma unroll
                for (int f = 0; f < FE; ++f) sum[f] = add4(sum[f], part[(y * bdx + tx) * FE + f]);
            }
#pragma unroll
            for (int f = 0; f < FE; ++f) st4(dWe_partial + ((size_t)bid * FE + f) * ld + col, sum[f]);
        }
        __syncthreads();
    }
}

template <int FE, bool DSG>
__device__ __forceinline__ void edge_bwd_src_body(int bid, int n, int nchunk, int e_stored,
                                                           const int* __restrict__ rowptr, const int* __restrict__ nbr,
                                                           const int* __restrict__ eid, const float* __restrict__ P,
                                                           const float* __restrict__ Q, const float* __restrict__ dS,
                                                           const float* __restrict__ ea, const float* __restrict__ w1,
                                                           float* __restrict__ dQ, int ld, int h, int fi,
                                                           const float* __restrict__ w2, int fo) {
    extern __shared__ __attribute__((aligned(16))) float we[];
    const int ldw = 2 * fi + FE;
    for (int i = threadIdx.x; i < FE * ld; i += blockDim.x) {
        const int f = i / ld, k = i - f * ld;
        we[i] = k < h ? w1[(size_t)k * ldw + 2 * fi + f] : 0.f;
    }
    float* w2s = we + FE * ld;                                     // (DSG only) W2 [4][ld]
    if (DSG) stage_w2(w2s, w2, h, fo, ld);
    __syncthreads();
    const long item = (long)bid * blockDim.x + threadIdx.x;
    const int row = (int)(item / nchunk);
    if (row >= n) return;
    const int col = (int)(item - (long)row * nchunk) * 4;
    const float4 q4 = ld4(Q + (size_t)row * ld + col);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const int beg = rowptr[row], end = rowptr[row + 1];
    float4 w4[FE];
#pragma unroll
    for (int f = 0; f < FE; ++f) w4[f] = ld4(we + f * ld + col);
    for (int p = beg; p < end; p += BW_SLOTS) {      // BW_SLOTS edge slots per trip (see hop_kernel)
        const int last = end - 1;
        int d_[BW_SLOTS], id_[BW_SLOTS];
#pragma unroll
        for (int u = 0; u < BW_SLOTS; ++u) {
            const int q = min(p + u, last);
            d_[u] = nbr[q];
            const int id = eid[q];
            id_[u] = id >= e_stored ? id - e_stored : id;
        }
        float4 p_[BW_SLOTS], g_[BW_SLOTS];
        float a_[BW_SLOTS][FE];
#pragma unroll
        for (int u = 0; u < BW_SLOTS; ++u) {
            p_[u] = ld4(P + (size_t)d_[u] * ld + col);
            g_[u] = ds_row<DSG>(dS, d_[u], ld, col, w2s);
#pragma unroll
            for (int f = 0; f < FE; ++f) a_[u][f] = ea[(size_t)id_[u] * FE + f];
        }
#pragma unroll
        for (int u = 0; u < BW_SLOTS; ++u) {
            float4 v = add4(p_[u], q4);
#pragma unroll
            for (int f = 0; f < FE; ++f) v = fma4(a_[u][f], w4[f], v);
            const bool k = p + u < end;
            acc.x += (k && v.x > 0.f) ? g_[u].x : 0.f;
            acc.y += (k && v.y > 0.f) ? g_[u].y : 0.f;
            acc.z += (k && v.z > 0.f) ? g_[u].z : 0.f;
            acc.w += (k && v.w > 0.f) ? g_[u].w : 0.f;
        }
    }
    st4(dQ + (size_t)row * ld + col, acc);
}

// ---- the same two walks reading the forward pass's ReLU masks (Fe = 2): dh_e = dS[dst e] where the saved mask bit is set.
// Nothing is recomputed, so P, Q, the residue weights and (in the by-source half) the edge attributes are never touched: the
// kernel gathers dS rows only.  Four slots per trip again (no Q rows to hold).
template <bool DSG>
__device__ __forceinline__ void edge_bwd_dst_body_mask(int bid, int nblk, int n, int nchunk, int bdx, int bdy, int e_stored,
                                                       const int* __restrict__ rowptr, const int* __restrict__ eid,
                                                       const float* __restrict__ dS, const float* __restrict__ ea,
                                                       const unsigned* __restrict__ mask, const int* __restrict__ rp4,
                                                       float* __restrict__ dP,
                                                       float* __restrict__ dWe_partial, int ld, int h,
                                                       const float* __restrict__ w2, int fo) {
    constexpr int FE = 2;
    extern __shared__ __attribute__((aligned(16))) float smem[];   // (we[FE][ld] unused) | part[256][FE] float4 | w2s
    float4* part = reinterpret_cast<float4*>(smem + FE * ld);
    float* w2s = smem + FE * ld + 256 * FE * 4;
    if (DSG) stage_w2(w2s, w2, h, fo, ld);
    __syncthreads();
    const int ty = threadIdx.x / bdx, tx = threadIdx.x - ty * bdx;
    const bool active = ty < bdy;
    for (int c0 = 0; c0 < nchunk; c0 += bdx) {
        const int c = c0 + tx, col = 4 * c;
        float4 dwe[FE];
#pragma unroll
        for (int f = 0; f < FE; ++f) dwe[f] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (active && c < nchunk) {
            for (int row = bid * bdy + ty; row < n; row += nblk * bdy) {
                const float4 g4 = ds_row<DSG>(dS, row, ld, col, w2s);
                float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
                const int beg = rowptr[row], end = rowptr[row + 1];
                const int b4 = rp4[row];
                const unsigned* mrun = mask + (size_t)b4 * nchunk + (size_t)c * (rp4[row + 1] - b4);
                for (int p = beg; p < end; p += 4) {
                    const int last = end - 1;
                    const unsigned mword = mrun[(p - beg) >> 2];     // (bytes of slots past the row's end are zero)
                    unsigned m_[4];
                    float2 a_[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int q = min(p + u, last);
                        m_[u] = (mword >> (8 * u)) & 0xffu;
                        const int id = eid[q];
                        a_[u] = *reinterpret_cast<const float2*>(ea + (size_t)(id >= e_stored ? id - e_stored : id) * 2);
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const float4 dh = mask4(m_[u], g4);
                        acc = add4(acc, dh);
                        dwe[0] = fma4(a_[u].x, dh, dwe[0]);
                        dwe[1] = fma4(a_[u].y, dh, dwe[1]);
                    }
                }
                st4(dP + (size_t)row * ld + col, acc);
            }
        }
        if (active) {
#pragma unroll
            for (int f = 0; f < FE; ++f) part[threadIdx.x * FE + f] = dwe[f];
        }
        __syncthreads();
        if (ty == 0 && c < nchunk) {
            float4 sum[FE];
#pragma unroll
            for (int f = 0; f < FE; ++f) sum[f] = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int y = 0; y < bdy; ++y) {
#pragma unroll
                for (int f = 0; f < FE; ++f) sum[f] = add4(sum[f], part[(y * bdx + tx) * FE + f]);
            }
#pragma unroll
            for (int f = 0; f < FE; ++f) st4(dWe_partial + ((size_t)bid * FE + f) * ld + col, sum[f]);
        }
        __syncthreads();
    }
}

template <bool DSG>
__device__ __forceinline__ void edge_bwd_src_body_mask(int bid, int n, int nchunk, const int* __restrict__ rowptr,
                                                       const int* __restrict__ nbr, const int* __restrict__ out_mbase,
                                                       const int2* __restrict__ out_ml4k, const float* __restrict__ dS,
                                                       const unsigned* __restrict__ mask,
                                                       float* __restrict__ dQ, int ld, int h, const float* __restrict__ w2,
                                                       int fo) {
    extern __shared__ __attribute__((aligned(16))) float we[];
    float* w2s = we + 2 * ld;
    if (DSG) stage_w2(w2s, w2, h, fo, ld);
    __syncthreads();
    const long item = (long)bid * blockDim.x + threadIdx.x;
    const int row = (int)(item / nchunk);
    if (row >= n) return;
    const int c = (int)(item - (long)row * nchunk), col = 4 * c;
    const unsigned char* mbytes = reinterpret_cast<const unsigned char*>(mask);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const int beg = rowptr[row], end = rowptr[row + 1];
    for (int p = beg; p < end; p += 4) {
        const int last = end - 1;
        int d_[4];
        unsigned m_[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int q = min(p + u, last);
            d_[u] = nbr[q];
            const int2 l4k = out_ml4k[q];                            // {ceil(deg_dst / 4), position among dst's incoming edges}
            const size_t at = ((size_t)out_mbase[q] * nchunk + (size_t)c * l4k.x) * 4 + l4k.y;
            m_[u] = p + u < end ? mbytes[at] : 0u;
        }
        float4 g_[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) g_[u] = ds_row<DSG>(dS, d_[u], ld, col, w2s);
#pragma unroll
        for (int u = 0; u < 4; ++u) acc = add4(acc, mask4(m_[u], g_[u]));
    }
    st4(dQ + (size_t)row * ld + col, acc);
}

template <bool DSG>
__global__ __launch_bounds__(256) void edge_bwd_mask_kernel(int nb_dst, int n, int nchunk, int bdx, int bdy, int e_stored,
                                                            const int* __restrict__ rp_in, const int* __restrict__ in_eid,
                                                            const int* __restrict__ rp_out, const int* __restrict__ out_dst,
                                                            const int* __restrict__ out_mbase, const int2* __restrict__ out_ml4k,
                                                            const int* __restrict__ rp4, const float* __restrict__ dS,
                                                            const float* __restrict__ ea, const unsigned* __restrict__ mask,
                                                            float* __restrict__ dP, float* __restrict__ dQ,
                                                            float* __restrict__ dWe_partial, int ld, int h,
                                                            const float* __restrict__ w2, int fo) {
    if ((int)blockIdx.x < nb_dst)
        edge_bwd_dst_body_mask<DSG>(blockIdx.x, nb_dst, n, nchunk, bdx, bdy, e_stored, rp_in, in_eid, dS, ea, mask, rp4, dP,
                                    dWe_partial, ld, h, w2, fo);
    else
        edge_bwd_src_body_mask<DSG>(blockIdx.x - nb_dst, n, nchunk, rp_out, out_dst, out_mbase, out_ml4k, dS, mask, dQ, ld, h, w2,
                                    fo);
}

// One launch for both halves of the EdgeAggregation backward: blocks [0, nb_dst) walk the by-destination CSR (dP, dWe
// partials; block-persistent), the rest walk the by-source CSR (dQ).  Both halves are latency-bound at small batches; in
// one grid they overlap instead of paying two launch floors.
template <int FE, bool DSG>
__global__ __launch_bounds__(256) void edge_bwd_kernel(int nb_dst, int n, int nchunk, int bdx, int bdy, int e_stored,
                                                       const int* __restrict__ rp_in, const int* __restrict__ in_src,
                                                       const int* __restrict__ in_eid, const int* __restrict__ rp_out,
                                                       const int* __restrict__ out_dst, const int* __restrict__ out_eid,
                                                       const float* __restrict__ P, const float* __restrict__ Q,
                                                       const float* __restrict__ dS, const float* __restrict__ ea,
                                                       const float* __restrict__ w1, float* __restrict__ dP,
                                                       float* __restrict__ dQ, float* __restrict__ dWe_partial, int ld, int h,
                                                       int fi, const float* __restrict__ w2, int fo) {
    // DSG: `dS` is the layer's N x 4 output gradient and dS rows are formed on the fly from it and W2 (ds_row)
    if ((int)blockIdx.x < nb_dst)
        edge_bwd_dst_body<FE, DSG>(blockIdx.x, nb_dst, n, nchunk, bdx, bdy, e_stored, rp_in, in_src, in_eid, P, Q, dS, ea, w1,
                                   dP, dWe_partial, ld, h, fi, w2, fo);
    else
        edge_bwd_src_body<FE, DSG>(blockIdx.x - nb_dst, n, nchunk, e_stored, rp_out, out_dst, out_eid, P, Q, dS, ea, w1, dQ, ld,
                                   h, fi, w2, fo);
}

static void dst_block_shape(int ld, int& bdx, int& bdy) {
    const int nchunk = ld / 4;
    bdx = nchunk < 256 ? nchunk : 256;
    bdy = 256 / bdx;
}
int edge_bwd_dst_blocks(const GraphView& g, int ld) {
    int bdx, bdy;
    dst_block_shape(ld, bdx, bdy);
    const int want = (g.n + bdy - 1) / bdy;
    return want < 1024 ? (want > 0 ? want : 1) : 1024;
}

template <int FE>
static int launch_edge_bwd_fe(const GraphView& g, const EdgeBwdArgs& a, hipStream_t s) {
    const int nchunk = a.ld / 4;
    int bdx, bdy;
    dst_block_shape(a.ld, bdx, bdy);
    const size_t lds_dst = (size_t)FE * a.ld * sizeof(float) + (size_t)256 * FE * sizeof(float4) +
                           (a.gout ? (size_t)4 * a.ld * sizeof(float) : 0);
    const int nb_dst = edge_bwd_dst_blocks(g, a.ld);
    const long items = (long)g.n * nchunk;
    const int nb_src = (int)((items + 255) / 256);
    ProfScope ps("edge_bwd", 0.0, 0.0, s);
    if (FE == 2 && a.mask) {
        if (a.gout)
            edge_bwd_mask_kernel<true><<<nb_dst + nb_src, 256, lds_dst, s>>>(nb_dst, g.n, nchunk, bdx, bdy, g.e_stored, g.rowptr_in,
                                                                           g.in_eid, g.rowptr_out, g.out_dst, g.out_mbase, g.out_ml4k,
                                                                           g.rp4, a.gout, a.edge_attr, a.mask, a.dP, a.dQ,
                                                                           a.dWe_partial, a.ld, a.h, a.w2, a.fo);
        else
            edge_bwd_mask_kernel<false><<<nb_dst + nb_src, 256, lds_dst, s>>>(nb_dst, g.n, nchunk, bdx, bdy, g.e_stored, g.rowptr_in,
                                                                            g.in_eid, g.rowptr_out, g.out_dst, g.out_mbase, g.out_ml4k,
                                                                            g.rp4, a.dS, a.edge_attr, a.mask, a.dP, a.dQ,
                                                                            a.dWe_partial, a.ld, a.h, nullptr, 0);
        PFN_CHECK_LAUNCH();
        return PFN_OK;
    }
    if (a.gout)
        edge_bwd_kernel<FE, true><<<nb_dst + nb_src, 256, lds_dst, s>>>(
            nb_dst, g.n, nchunk, bdx, bdy, g.e_stored, g.rowptr_in, g.in_src, g.in_eid, g.rowptr_out, g.out_dst, g.out_eid, a.P,
            a.Q, a.gout, a.edge_attr, a.w1, a.dP, a.dQ, a.dWe_partial, a.ld, a.h, a.fi, a.w2, a.fo);
    else
        edge_bwd_kernel<FE, false><<<nb_dst + nb_src, 256, lds_dst, s>>>(
            nb_dst, g.n, nchunk, bdx, bdy, g.e_stored, g.rowptr_in, g.in_src, g.in_eid, g.rowptr_out, g.out_dst, g.out_eid, a.P,
            a.Q, a.dS, a.edge_attr, a.w1, a.dP, a.dQ, a.dWe_partial, a.ld, a.h, a.fi, nullptr, 0);
    PFN_CHECK_LAUNCH();
    return PFN_OK;
}

int launch_edge_bwd(const GraphView& g, const EdgeBwdArgs& a, const int64_t*, hipStream_t s) {
    if (g.n == 0) return PFN_OK;
    switch (a.fe) {
        case 1: return launch_edge_bwd_fe<1>(g, a, s);
        case 2: return launch_edge_bwd_fe<2>(g, a, s);
        case 3: return launch_edge_bwd_fe<3>(g, a, s);
        case 4: return launch_edge_bwd_fe<4>(g, a, s);
        case 5: return launch_edge_bwd_fe<5>(g, a, s);
        case 6: return launch_edge_bwd_fe<6>(g, a, s);
        default: set_error("edge feature width %d unsupported in backward (1..6)", a.fe); return PFN_EINVAL;
    }
}

// dWe partial [nblocks][fe][ld]  ->  grad_w1[k][col0 + f]  (ordered: 4 interleaved lanes, then a fixed tree), for up to
// DWE_MAX_JOBS EdgeAggregation layers in one launch (blockIdx.y = layer)
// stamp / stamp_want: the workspace guard of pfn_mpn_backward (model.hip WS_STAMP_TRAIN): on a mismatch the gradients are NaN
__global__ __launch_bounds__(1024) void dwe_reduce_kernel(const DweJobs jobs, int fe, int ld, int h, const int* __restrict__ stamp,
                                                          int stamp_want) {
    __shared__ float red[64][17];
    dwe_reduce_body<64>(jobs.job[blockIdx.y], blockIdx.x, fe, ld, h, red, stamp != nullptr && *stamp != stamp_want);
}

int launch_dwe_reduce_multi(const DweJob* jobs, int njobs, int fe, int ld, int h, hipStream_t s, const int* stamp, int stamp_want) {
    for (int j0 = 0; j0 < njobs; j0 += DWE_MAX_JOBS) {
        DweJobs a;
        const int nj = njobs - j0 < DWE_MAX_JOBS ? njobs - j0 : DWE_MAX_JOBS;
        for (int j = 0; j < nj; ++j) a.job[j] = jobs[j0 + j];
        ProfScope ps("dwe_reduce", 0.0, 0.0, s);
        dwe_reduce_kernel<<<dim3((fe * h + 15) / 16, nj), 1024, 0, s>>>(a, fe, ld, h, stamp, stamp_want);
        PFN_CHECK_LAUNCH();
    }
    return PFN_OK;
}

int launch_dwe_reduce(const float* partial, int nblocks, int fe, int ld, int h, float* gw1, int ldw, int col0,
                      hipStream_t s) {
    const DweJob jb{partial, gw1, nblocks, ldw, col0, 0};
    return launch_dwe_reduce_multi(&jb, 1, fe, ld, h, s);
}

// d edge_attr[e][f] = sum over the (one or two) effective copies of stored edge e of  We_f . dh_copy.
// One wave per stored edge walks H; only launched when the caller asks for grad_edge_attr.
__global__ __launch_bounds__(256) void edge_attr_grad_kernel(int e_stored, const int* __restrict__ rowptr_in,
                                                             const int* __restrict__ in_src,
                                                             const int* __restrict__ in_eid, int n,
                                                             const float* __restrict__ P, const float* __restrict__ Q,
                                                             const float* __restrict__ dS, const float* __restrict__ ea,
                                                             const float* __restrict__ w1, float* __restrict__ gea,
                                                             int ld, int h, int fi, int fe, float* __restrict__ tmp) {
    // work item: one (destination row, slot) pair = one effective edge; a wave per effective edge
    const int wave = (int)(((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    const int lane = threadIdx.x & 63;
    const int e_eff = rowptr_in[n];
    if (wave >= e_eff) return;
    // binary search the destination row owning slot `wave`
    int lo = 0, hi = n;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (rowptr_in[mid] <= wave) lo = mid; else hi = mid;
    }
    const int dst = lo, src = in_src[wave];
    const int id_copy = in_eid[wave];            // [0, 2 e_stored): the stored edge, or e_stored + the stored edge for its mirror copy
    const int id = id_copy >= e_stored ? id_copy - e_stored : id_copy;
    const int ldw = 2 * fi + fe;
    float part[PFN_MAX_FE];
    for (int f = 0; f < fe; ++f) part[f] = 0.f;
    for (int k = lane; k < h; k += 64) {
        float v = P[(size_t)dst * ld + k] + Q[(size_t)src * ld + k];
        for (int f = 0; f < fe; ++f) v = fmaf(ea[(size_t)id * fe + f], w1[(size_t)k * ldw + 2 * fi + f], v);
        const float dh = v > 0.f ? dS[(size_t)dst * ld + k] : 0.f;
        for (int f = 0; f < fe; ++f) part[f] = fmaf(w1[(size_t)k * ldw + 2 * fi + f], dh, part[f]);
    }
    for (int f = 0; f < fe; ++f) {
        float v = part[f];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
        if (lane != 0) continue;
        if (tmp) tmp[(size_t)id_copy * fe + f] = v;   // one slot per copy; edge_attr_grad_fold_kernel adds them in a fixed order
        else atomicAdd(&gea[(size_t)id * fe + f], v);   // single layer, gea zeroed by the caller: at most two terms, order-free
    }
}
// gea[e][f] += tmp[e][f] + tmp[e_stored + e][f]: a network's layers all add into one gea, and with atomics three or more terms met
// in an order that changed from run to run (last-bit differences); tmp is zeroed per layer (a copy that does not exist adds 0)
__global__ __launch_bounds__(256) void edge_attr_grad_fold_kernel(long count, long mirror, const float* __restrict__ tmp,
                                                                  float* __restrict__ gea) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) gea[i] += tmp[i] + tmp[mirror + i];
}

int launch_edge_attr_grad(const GraphView& g, const EdgeBwdArgs& a, hipStream_t s) {
    if (g.e_stored == 0) return PFN_OK;
    if (a.fe > PFN_MAX_FE) {
        set_error("edge feature width %d > %d", a.fe, PFN_MAX_FE);
        return PFN_EINVAL;
    }
    // ACCUMULATES into grad_edge_attr (every EdgeAggregation layer of a model adds its share): the caller zeroes it once per
    // backward pass (pfn_mpn_backward, pfn_edge_aggr_backward).  A memset here made the model-level gradient the LAST layer's alone.
    const long waves = 2l * g.e_stored;   // upper bound on effective edges
    const int blocks = (int)((waves * 64 + 255) / 256);
    const long count = (long)g.e_stored * a.fe;
    if (a.gea_tmp) PFN_CHECK_HIP(hipMemsetAsync(a.gea_tmp, 0, (size_t)2 * count * sizeof(float), s));
    edge_attr_grad_kernel<<<blocks, 256, 0, s>>>(g.e_stored, g.rowptr_in, g.in_src, g.in_eid, g.n, a.P, a.Q, a.dS,
                                                 a.edge_attr, a.w1, a.grad_edge_attr, a.ld, a.h, a.fi, a.fe, a.gea_tmp);
    PFN_CHECK_LAUNCH();
    if (a.gea_tmp) {
        edge_attr_grad_fold_kernel<<<(int)((count + 255) / 256), 256, 0, s>>>(count, count, a.gea_tmp, a.grad_edge_attr);
        PFN_CHECK_LAUNCH();
    }
    return PFN_OK;
}

// ------------------------------------------------------------------------------------ small kernels
__global__ void pad_rows_kernel(const float* __restrict__ src, int64_t lds_, float* __restrict__ dst, int64_t ldd,
                                int64_t rows, int64_t f) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * ldd) return;
    const int64_t r = i / ldd, c = i - r * ldd;
    dst[i] = c < f ? src[r * lds_ + c] : 0.f;
}
int launch_pad_rows(const float* src, int64_t ld_src, float* dst, int64_t ld_dst, int64_t rows, int64_t f,
                    hipStream_t s) {
    const int64_t total = rows * ld_dst;
    if (total == 0) return PFN_OK;
    pad_rows_kernel<<<(int)((total + 255) / 256), 256, 0, s>>>(src, ld_src, dst, ld_dst, rows, f);
    PFN_CHECK_LAUNCH();
    return PFN_OK;
}


}  // namespace pfn
