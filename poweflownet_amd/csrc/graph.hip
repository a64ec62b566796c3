// Graph preparation for the message-passing hot path (gfx950).
//
// Replaces, per forward call: MaskEmbdMultiMPN.is_directed / undirect_graph (networks/MPN.py:498-523),
// the index_select lifting and degree scatter PyG performs under propagate()/gcn_norm, and the dead
// degree computation of EdgeAggregation.forward (networks/MPN.py:43-47, not reproduced: it does not
// reach the output).  One histogram pass, a two-launch scan, one fill, one placement pass (rows in edge-id order); no host sync:
// the "directed" decision of the reference's first-edge heuristic is taken on device and consumed by
// the later kernels through flags[].
#include <atomic>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>

#include "pfn_internal.hpp"

namespace pfn {

static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
const char* last_error() { return g_err; }

const char* diag_env(const char* name) { return getenv(name); }
int next_sweep_direction() {   // (pfn_internal.hpp: serpentine sweeps)
    static const bool off = diag_env("PFN_NO_SERPENTINE") != nullptr;   // A/B switch: every launch walks its rows first to last
    static thread_local unsigned n = 0;
    return off ? 0 : (int)(n++ & 1u);
}

int device_cus() {
    static std::atomic<int> cus[64];          // zero-initialised; 0 = not asked yet
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    std::atomic<int>& slot = cus[dev & 63];
    int v = slot.load(std::memory_order_relaxed);
    if (v == 0) {
        hipDeviceProp_t prop;
        v = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
        slot.store(v, std::memory_order_relaxed);
    }
    return v;
}

int ensure_dynamic_lds(const void* kernel, int bytes, std::atomic<uint64_t>& done) {
    int dev = 0;
    PFN_CHECK_HIP(hipGetDevice(&dev));
    const uint64_t bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return PFN_OK;
    PFN_CHECK_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    done.fetch_or(bit, std::memory_order_release);
    return PFN_OK;
}

GraphView graph_view(void* ws, int64_t n, int64_t e) {
    Carver c(ws);
    GraphView g;
    g.n = (int)n;
    g.e_stored = (int)e;
    g.flags = c.take<int>(64);
    g.scan_sums = c.take<int>(3 * GRAPH_SCAN_BLOCKS);
    g.rowptr_in = c.take<int>(n + 1);
    g.rowptr_out = c.take<int>(n + 1);
    g.in_src = c.take<int>(2 * e + 1);
    g.in_eid = c.take<int>(2 * e + 1);
    g.out_dst = c.take<int>(2 * e + 1);
    g.out_eid = c.take<int>(2 * e + 1);
    g.rp4 = c.take<int>(n + 1);
    g.out_mbase = c.take<int>(2 * e + 1);
    g.out_ml4k = c.take<int2>(2 * e + 1);
    g.slot_of_eid = c.take<int>(2 * e + 1);
    g.cur_in = c.take<int>(n + 1);
    g.cur_out = c.take<int>(n + 1);
    g.deg = c.take<float>(n + 1);
    g.dinv = c.take<float>(n + 1);
    g.bytes = c.off;
    return g;
}

// ---------------------------------------------------------------------------------------- kernels
// Pass 1: histogram of destinations / sources, range check, and the first-edge heuristic:
// "directed" iff no stored edge (v0 -> u0) exists for the first stored edge (u0 -> v0).
__global__ __launch_bounds__(256) void graph_hist_kernel(const int64_t* __restrict__ ei, int e, int n, int* cnt_dst,
                                                         int* cnt_src, int* flags) {
    const int64_t u0 = ei[0], v0 = ei[e];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < e; i += gridDim.x * blockDim.x) {
        const int64_t s = ei[i], d = ei[(size_t)e + i];
        if (s < 0 || s >= n || d < 0 || d >= n) {
            atomicOr(&flags[2], 1);
            continue;
        }
        atomicAdd(&cnt_dst[d], 1);
        atomicAdd(&cnt_src[s], 1);
        if (s == v0 && d == u0) atomicOr(&flags[3], 1);
    }
}

// Pass 2 (two launches over the same tiling: block b owns rows [b * tile, (b + 1) * tile), tile a multiple of 1024, at most
// GRAPH_SCAN_BLOCKS blocks): decide `directed`, form in/out degrees, exclusive-scan them into the two rowptr arrays (and
// rp4 = prefix of ceil(in-degree / 4), the row offsets in dwords per column chunk of the edge stage's ReLU masks), emit
// deg / deg^-1/2, and clear the histograms so pass 3 can reuse them as cursors.  First the per-block totals, then every
// block adds up the totals before it and scans its own rows.  (One 1024-thread block walking all rows was 0.8 ms at
// 414 k nodes -- most of a cold-topology build.)
__device__ __forceinline__ void degrees_of(int directed, int cd, int cs, int& di, int& dout) {
    di = directed ? cd + cs : cd;
    dout = directed ? cd + cs : cs;
}
__device__ __forceinline__ int scan_directed(int mode, int e, const int* flags) {
    return (mode == 1) ? 1 : (mode == 0 ? 0 : (e > 0 && flags[3] == 0));
}
// sum over the block of three ints per thread -> every thread gets the totals (two barriers)
__device__ __forceinline__ void block_sum3(int& a, int& b, int& c, int* red /* [3][16] */) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        a += __shfl_xor(a, off);
        b += __shfl_xor(b, off);
        c += __shfl_xor(c, off);
    }
    __syncthreads();   // (red may still be read from a previous call)
    if (lane == 0) {
        red[wave] = a;
        red[16 + wave] = b;
        red[32 + wave] = c;
    }
    __syncthreads();
    a = b = c = 0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) {
        a += red[w];
        b += red[16 + w];
        c += red[32 + w];
    }
}
__global__ __launch_bounds__(1024) void graph_scan_sums_kernel(int n, int e, int mode, int tile, const int* __restrict__ cnt_dst,
                                                               const int* __restrict__ cnt_src, const int* __restrict__ flags,
                                                               int* __restrict__ bsum) {
    __shared__ int red[48];
    const int directed = scan_directed(mode, e, flags);
    const int r0 = blockIdx.x * tile, r1 = min(n, r0 + tile);
    int a = 0, b = 0, c = 0;
    for (int i = r0 + threadIdx.x; i < r1; i += 1024) {
        int di, dout;
        degrees_of(directed, cnt_dst[i], cnt_src[i], di, dout);
        a += di;
        b += dout;
        c += (di + 3) >> 2;
    }
    block_sum3(a, b, c, red);
    if (threadIdx.x == 0) {
        bsum[blockIdx.x] = a;
        bsum[GRAPH_SCAN_BLOCKS + blockIdx.x] = b;
        bsum[2 * GRAPH_SCAN_BLOCKS + blockIdx.x] = c;
    }
}
__global__ __launch_bounds__(1024) void graph_scan_kernel(int n, int e, int mode, int tile, int* cnt_dst, int* cnt_src,
                                                          int* rowptr_in, int* rowptr_out, int* rp4, float* deg, float* dinv,
                                                          int* flags, const int* __restrict__ bsum) {
    __shared__ int wsum_in[16], wsum_out[16], wsum_4[16], red[48];
    __shared__ int carry_in, carry_out, carry_4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int directed = scan_directed(mode, e, flags);
    {
        // totals of the blocks before this one (at most GRAPH_SCAN_BLOCKS = blockDim.x of them: one per thread)
        int a = 0, b = 0, c = 0;
        if (tid < (int)blockIdx.x) {
            a = bsum[tid];
            b = bsum[GRAPH_SCAN_BLOCKS + tid];
            c = bsum[2 * GRAPH_SCAN_BLOCKS + tid];
        }
        block_sum3(a, b, c, red);
        if (tid == 0) {
            carry_in = a;
            carry_out = b;
            carry_4 = c;
            if (blockIdx.x == 0) {
                flags[0] = directed;
                flags[1] = directed ? 2 * e : e;
            }
        }
    }
    __syncthreads();
    const int r0 = blockIdx.x * tile, r1 = min(n, r0 + tile);
    for (int base = r0; base < r1; base += 1024) {
        const int i = base + tid;
        int di = 0, dout = 0;
        if (i < r1) {
            degrees_of(directed, cnt_dst[i], cnt_src[i], di, dout);
            cnt_dst[i] = 0;
            cnt_src[i] = 0;
            deg[i] = (float)di;
            dinv[i] = di > 0 ? 1.0f / sqrtf((float)di) : 0.0f;
        }
        const int d4 = (di + 3) >> 2;
        int si = di, so = dout, s4 = d4;   // inclusive wave scan
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int ti = __shfl_up(si, off), to = __shfl_up(so, off), t4 = __shfl_up(s4, off);
            if (lane >= off) {
                si += ti;
                so += to;
                s4 += t4;
            }
        }
        if (lane == 63) {
            wsum_in[wave] = si;
            wsum_out[wave] = so;
            wsum_4[wave] = s4;
        }
        __syncthreads();
        int pre_in = carry_in, pre_out = carry_out, pre_4 = carry_4;
        for (int w = 0; w < wave; ++w) {
            pre_in += wsum_in[w];
            pre_out += wsum_out[w];
            pre_4 += wsum_4[w];
        }
        if (i < r1) {
            rowptr_in[i] = pre_in + si - di;
            rowptr_out[i] = pre_out + so - dout;
            rp4[i] = pre_4 + s4 - d4;
        }
        __syncthreads();
        if (tid == 1023) {
            carry_in = pre_in + si;
            carry_out = pre_out + so;
            carry_4 = pre_4 + s4;
        }
        __syncthreads();
    }
    if (tid == 0 && blockIdx.x == gridDim.x - 1) {
        rowptr_in[n] = carry_in;
        rowptr_out[n] = carry_out;
        rp4[n] = carry_4;
    }
}

// Pass 3: every effective edge drops its edge id into its destination's by-destination row and its source's by-source row
// (scratch arrays; slot order inside a row is whatever the atomics gave).
__global__ __launch_bounds__(256) void graph_fill_kernel(const int64_t* __restrict__ ei, int e, int n,
                                                         const int* __restrict__ rowptr_in,
                                                         const int* __restrict__ rowptr_out, int* cur_in, int* cur_out,
                                                         int* __restrict__ tmp_in, int* __restrict__ tmp_out, const int* flags) {
    const int directed = flags[0];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < e; i += gridDim.x * blockDim.x) {
        const int64_t s64 = ei[i], d64 = ei[(size_t)e + i];
        if (s64 < 0 || s64 >= n || d64 < 0 || d64 >= n) continue;
        const int s = (int)s64, d = (int)d64;
        tmp_in[rowptr_in[d] + atomicAdd(&cur_in[d], 1)] = i;
        tmp_out[rowptr_out[s] + atomicAdd(&cur_out[s], 1)] = i;
        if (directed) {   // reversed copy (d -> s), edge id e + i: "originals first, reverses second"
            tmp_in[rowptr_in[s] + atomicAdd(&cur_in[s], 1)] = e + i;
            tmp_out[rowptr_out[d] + atomicAdd(&cur_out[d], 1)] = e + i;
        }
    }
}

// Pass 4: order each row by edge id -> segment sums run in the stored edge order (the order the reference's sequential
// scatter_add visits them) and results are run-to-run deterministic.  Edge ids are unique, so an edge's slot in a row is the
// number of smaller ids in that row: every edge counts them itself (a row of degree d costs d reads on d threads -- a
// per-row insertion sort was d^2 steps on ONE thread, 0.8 ms for the 170-edge hub rows of the 6470rte x 64 hub case).
__device__ __forceinline__ int rank_in_row(const int* __restrict__ tmp, int beg, int end, int id) {
    int r = 0;
    for (int q = beg; q < end; ++q) r += tmp[q] < id ? 1 : 0;
    return r;
}
__global__ __launch_bounds__(256) void graph_place_kernel(const int64_t* __restrict__ ei, int e, int n,
                                                          const int* __restrict__ rowptr_in, const int* __restrict__ rowptr_out,
                                                          const int* __restrict__ tmp_in, const int* __restrict__ tmp_out,
                                                          int* __restrict__ in_src, int* __restrict__ in_eid,
                                                          int* __restrict__ out_dst, int* __restrict__ out_eid, const int* flags) {
    const int directed = flags[0];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < e; i += gridDim.x * blockDim.x) {
        const int64_t s64 = ei[i], d64 = ei[(size_t)e + i];
        if (s64 < 0 || s64 >= n || d64 < 0 || d64 >= n) continue;
        const int s = (int)s64, d = (int)d64;
        const int ib = rowptr_in[d], ie = rowptr_in[d + 1], ob = rowptr_out[s], oe = rowptr_out[s + 1];
        int p = ib + rank_in_row(tmp_in, ib, ie, i);
        in_src[p] = s;
        in_eid[p] = i;
        p = ob + rank_in_row(tmp_out, ob, oe, i);
        out_dst[p] = d;
        out_eid[p] = i;
        if (directed) {
            const int rb = rowptr_in[s], re = rowptr_in[s + 1], qb = rowptr_out[d], qe = rowptr_out[d + 1];
            p = rb + rank_in_row(tmp_in, rb, re, e + i);
            in_src[p] = d;
            in_eid[p] = e + i;
            p = qb + rank_in_row(tmp_out, qb, qe, e + i);
            out_dst[p] = s;
            out_eid[p] = e + i;
        }
    }
}

// Segment check for the LDS-resident multi-hop path: ok iff no effective edge crosses a multiple of `seg` (the batch
// is then a disjoint union of index-contiguous blocks of `seg` nodes, e.g. the graphs of a PyG batch of one case).
__global__ __launch_bounds__(256) void graph_segcheck_kernel(int n, int seg, const int* __restrict__ rowptr_in,
                                                             const int* __restrict__ in_src, int* flags) {
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= n) return;
    const int sid = row / seg;
    for (int p = rowptr_in[row]; p < rowptr_in[row + 1]; ++p)
        if (in_src[p] / seg != sid) atomicOr(&flags[4], 1);
}

__global__ void graph_export_kernel(int n, const int* __restrict__ rowptr_in, const int* __restrict__ in_src,
                                    const int* __restrict__ in_eid, const int* flags, int cap, int64_t* out) {
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= n) return;
    for (int p = rowptr_in[row]; p < rowptr_in[row + 1]; ++p) {
        out[in_eid[p]] = in_src[p];
        out[(size_t)cap + in_eid[p]] = row;
    }
}

}  // namespace pfn

using namespace pfn;

extern "C" {

int pfn_abi_version(void) { return PFN_ABI_VERSION; }
const char* pfn_last_error(void) { return pfn::last_error(); }
int64_t pfn_padded_ld(int64_t f) { return round_up(f, 4); }

size_t pfn_graph_workspace_bytes(int64_t n, int64_t e) {
    if (n < 0 || e < 0) return 0;
    return graph_view(nullptr, n, e).bytes;
}

// where the backward by-source walk finds an edge's ReLU mask (see EdgeFwdArgs::mask): by-destination slot per edge id first (edge
// ids are unique per directed edge), then per by-source slot the destination row's mask offset and the edge's position in it
__global__ __launch_bounds__(256) void graph_slot_of_eid_kernel(int n, const int* __restrict__ rowptr_in,
                                                                const int* __restrict__ in_eid, int* __restrict__ slot_of_eid) {
    const int nslot = rowptr_in[n];
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < nslot; p += gridDim.x * blockDim.x) slot_of_eid[in_eid[p]] = p;
}
__global__ __launch_bounds__(256) void graph_mask_index_kernel(int n, const int* __restrict__ rowptr_in,
                                                               const int* __restrict__ rowptr_out, const int* __restrict__ out_dst,
                                                               const int* __restrict__ out_eid, const int* __restrict__ slot_of_eid,
                                                               const int* __restrict__ rp4, int* __restrict__ out_mbase,
                                                               int2* __restrict__ out_ml4k) {
    const int nslot = rowptr_out[n];
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < nslot; p += gridDim.x * blockDim.x) {
        const int d = out_dst[p];
        const int k = slot_of_eid[out_eid[p]] - rowptr_in[d];       // position of the edge among d's incoming edges
        out_mbase[p] = rp4[d];
        out_ml4k[p] = make_int2(rp4[d + 1] - rp4[d], k);   // (two full ints: any in-degree)
    }
}

int pfn_graph_build(const int64_t* edge_index, int64_t e, int64_t n, int mode, void* ws, size_t ws_bytes,
                    void* stream) {
    PFN_CHECK_ARG(n >= 0 && e >= 0, "pfn_graph_build: negative sizes");
    PFN_CHECK_ARG(n < (1ll << 30) && e < (1ll << 29), "pfn_graph_build: graph too large for int32 adjacency");
    PFN_CHECK_ARG(ws != nullptr, "pfn_graph_build: null workspace");
    PFN_CHECK_ARG(mode >= -1 && mode <= 1, "pfn_graph_build: mode must be -1, 0 or 1");
    PFN_CHECK_ARG(e == 0 || edge_index != nullptr, "pfn_graph_build: null edge_index");
    GraphView g = graph_view(ws, n, e);
    if (ws_bytes < g.bytes) {
        set_error("pfn_graph_build: workspace %zu < %zu bytes", ws_bytes, g.bytes);
        return PFN_ENOSPACE;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    PFN_CHECK_HIP(hipMemsetAsync(g.flags, 0, 64 * sizeof(int), s));
    PFN_CHECK_HIP(hipMemsetAsync(g.cur_in, 0, (size_t)(n + 1) * sizeof(int), s));
    PFN_CHECK_HIP(hipMemsetAsync(g.cur_out, 0, (size_t)(n + 1) * sizeof(int), s));
    const int ie = (int)e, in = (int)n;
    if (ie > 0) {
        const int blocks = (int)std::min<int64_t>((e + 255) / 256, 2048);
        graph_hist_kernel<<<blocks, 256, 0, s>>>(edge_index, ie, in, g.cur_in, g.cur_out, g.flags);
        PFN_CHECK_LAUNCH();
    }
    {
        const int tile = 1024 * std::max(1, (in + 1024 * GRAPH_SCAN_BLOCKS - 1) / (1024 * GRAPH_SCAN_BLOCKS));
        const int nb = std::max(1, (in + tile - 1) / tile);
        graph_scan_sums_kernel<<<nb, 1024, 0, s>>>(in, ie, mode, tile, g.cur_in, g.cur_out, g.flags, g.scan_sums);
        PFN_CHECK_LAUNCH();
        graph_scan_kernel<<<nb, 1024, 0, s>>>(in, ie, mode, tile, g.cur_in, g.cur_out, g.rowptr_in, g.rowptr_out, g.rp4, g.deg,
                                              g.dinv, g.flags, g.scan_sums);
        PFN_CHECK_LAUNCH();
    }
    if (ie > 0) {
        const int blocks = (int)std::min<int64_t>((e + 255) / 256, 2048);
        // (scratch: the two mask-index arrays, written for good by graph_mask_index_kernel below)
        graph_fill_kernel<<<blocks, 256, 0, s>>>(edge_index, ie, in, g.rowptr_in, g.rowptr_out, g.cur_in, g.cur_out,
                                                 g.out_mbase, reinterpret_cast<int*>(g.out_ml4k), g.flags);
        PFN_CHECK_LAUNCH();
        graph_place_kernel<<<blocks, 256, 0, s>>>(edge_index, ie, in, g.rowptr_in, g.rowptr_out, g.out_mbase, reinterpret_cast<int*>(g.out_ml4k),
                                                  g.in_src, g.in_eid, g.out_dst, g.out_eid, g.flags);
        PFN_CHECK_LAUNCH();
        const int sblocks = (int)std::min<int64_t>((2 * e + 255) / 256, 2048);
        graph_slot_of_eid_kernel<<<sblocks, 256, 0, s>>>(in, g.rowptr_in, g.in_eid, g.slot_of_eid);
        PFN_CHECK_LAUNCH();
        graph_mask_index_kernel<<<sblocks, 256, 0, s>>>(in, g.rowptr_in, g.rowptr_out, g.out_dst, g.out_eid, g.slot_of_eid, g.rp4,
                                                        g.out_mbase, g.out_ml4k);
        PFN_CHECK_LAUNCH();
    }
    return PFN_OK;
}

int pfn_graph_info(const void* ws, int64_t n, int64_t e, int32_t* directed, int64_t* e_eff, void* stream) {
    PFN_CHECK_ARG(ws != nullptr, "pfn_graph_info: null workspace");
    GraphView g = graph_view(const_cast<void*>(ws), n, e);
    int h[4] = {0, 0, 0, 0};
    hipStream_t s = static_cast<hipStream_t>(stream);
    PFN_CHECK_HIP(hipMemcpyAsync(h, g.flags, sizeof(h), hipMemcpyDeviceToHost, s));
    PFN_CHECK_HIP(hipStreamSynchronize(s));
    if (directed) *directed = h[0];
    if (e_eff) *e_eff = h[1];
    if (h[2]) {
        set_error("edge_index holds a node id outside [0, %lld)", (long long)n);
        return PFN_EINDEX;
    }
    return PFN_OK;
}

int pfn_graph_segments(void* ws, int64_t n, int64_t e, int64_t seg_nodes, int32_t* ok, void* stream) {
    PFN_CHECK_ARG(ws != nullptr && ok != nullptr, "pfn_graph_segments: null pointer");
    *ok = 0;
    if (seg_nodes <= 0 || n <= 0 || n % seg_nodes != 0 || seg_nodes > (1 << 20)) return PFN_OK;
    GraphView g = graph_view(ws, n, e);
    hipStream_t s = static_cast<hipStream_t>(stream);
    PFN_CHECK_HIP(hipMemsetAsync(g.flags + 4, 0, sizeof(int), s));
    graph_segcheck_kernel<<<((int)n + 255) / 256, 256, 0, s>>>((int)n, (int)seg_nodes, g.rowptr_in, g.in_src, g.flags);
    PFN_CHECK_LAUNCH();
    int bad = 1;
    PFN_CHECK_HIP(hipMemcpyAsync(&bad, g.flags + 4, sizeof(int), hipMemcpyDeviceToHost, s));
    PFN_CHECK_HIP(hipStreamSynchronize(s));
    *ok = bad ? 0 : 1;
    return PFN_OK;
}

// fill `out` with NaN when the adjacency's error flags are set (one block; the fill only ever runs in the error case)
__global__ __launch_bounds__(256) void graph_poison_kernel(const int* __restrict__ flags, float* __restrict__ out, int64_t count) {
    if ((flags[2] | flags[4]) == 0) return;
    const float nan = __builtin_nanf("");
    for (int64_t i = threadIdx.x; i < count; i += blockDim.x) out[i] = nan;
}

int pfn_graph_segments_async(void* ws, int64_t n, int64_t e, int64_t seg_nodes, void* stream) {
    PFN_CHECK_ARG(ws != nullptr, "pfn_graph_segments_async: null workspace");
    PFN_CHECK_ARG(seg_nodes == 0 || (seg_nodes > 0 && seg_nodes <= (1 << 20) && n > 0 && n % seg_nodes == 0),
                  "pfn_graph_segments_async: seg_nodes must be 0, or positive and divide n_nodes");
    GraphView g = graph_view(ws, n, e);
    hipStream_t s = static_cast<hipStream_t>(stream);
    PFN_CHECK_HIP(hipMemsetAsync(g.flags + 4, 0, sizeof(int), s));
    if (seg_nodes == 0) return PFN_OK;   // no segment promise: the verdict of an earlier check of this workspace is withdrawn
    graph_segcheck_kernel<<<((int)n + 255) / 256, 256, 0, s>>>((int)n, (int)seg_nodes, g.rowptr_in, g.in_src, g.flags);
    PFN_CHECK_LAUNCH();
    return PFN_OK;
}

int pfn_graph_poison_if_bad(const void* ws, int64_t n, int64_t e, float* out, int64_t count, void* stream) {
    PFN_CHECK_ARG(ws != nullptr && (count == 0 || out != nullptr), "pfn_graph_poison_if_bad: null pointer");
    if (count == 0) return PFN_OK;
    GraphView g = graph_view(const_cast<void*>(ws), n, e);
    graph_poison_kernel<<<1, 256, 0, static_cast<hipStream_t>(stream)>>>(g.flags, out, count);
    PFN_CHECK_LAUNCH();
    return PFN_OK;
}

int pfn_graph_export_edges(const void* ws, int64_t n, int64_t e, int64_t* out, void* stream) {
    PFN_CHECK_ARG(ws != nullptr && out != nullptr, "pfn_graph_export_edges: null pointer");
    GraphView g = graph_view(const_cast<void*>(ws), n, e);
    hipStream_t s = static_cast<hipStream_t>(stream);
    PFN_CHECK_HIP(hipMemsetAsync(out, 0xff, (size_t)4 * e * sizeof(int64_t), s));
    if (n > 0) {
        graph_export_kernel<<<((int)n + 255) / 256, 256, 0, s>>>((int)n, g.rowptr_in, g.in_src, g.in_eid, g.flags,
                                                                 (int)(2 * e), out);
        PFN_CHECK_LAUNCH();
    }
    return PFN_OK;
}

}  // extern "C"
