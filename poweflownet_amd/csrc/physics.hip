// PowerImbalance, the reference's physics loss (utils/custom_loss_functions.py:99-286), as two CSR walks (gfx950).
//
// The reference de-normalises the prediction and the branch parameters, undirects the stored-once branch list, runs a
// PyG MessagePassing with flow='target_to_source' whose message (:159-228) is, for the branch i -> j seen from i,
//     e = Vm cos(Va pi/180), f = Vm sin(Va pi/180), g = r / (r^2 + x^2), b = -x / (r^2 + x^2)
//     Pji = g (e_i e_j - e_i^2 + f_i f_j - f_i^2) + b (f_i e_j - e_i f_j)
//     Qji = g (f_i e_j - e_i f_j) + b (-e_i e_j + e_i^2 - f_i f_j + f_i^2)
// sums the messages onto i = edge_index[0], forms dP_i = P_i - sum Pji, dQ_i = Q_i - sum Qji (update, :229-246) and
// returns mean_i (dP_i^2 + dQ_i^2) (:277-281).  Same CSR skeleton as the EdgeAggregation kernels, a different per-edge
// function: the forward walks the by-SOURCE rows (the aggregation index is edge_index[0]); the gradient of node k
// collects from the rows where k is the source (its own dP, dQ) and from the rows where k is the destination (the
// source's dP, dQ).  No atomics, edge-id order, an ordered last-arriver sum for the scalar.
#include "pfn_internal.hpp"

namespace pfn {

struct PiStats { float xm[4], xs[4], em[2], es[2]; };
struct PiHeader { float partial[256]; int counter; int pad_[63]; };

struct Bus { float e, f, c, s, vm; };   // rectangular voltage, cos / sin of the angle, magnitude
__device__ __forceinline__ Bus bus_of(const float* __restrict__ x, int i, const PiStats& st) {
    const float vm = fmaf(x[4 * i], st.xs[0], st.xm[0]);
    const float va = fmaf(x[4 * i + 1], st.xs[1], st.xm[1]) * (3.14159265358979323846f / 180.0f);
    Bus b;
    b.c = cosf(va);
    b.s = sinf(va);
    b.vm = vm;
    b.e = vm * b.c;
    b.f = vm * b.s;
    return b;
}
__device__ __forceinline__ void admittance(const float* __restrict__ ea, int eid, int e_stored, const PiStats& st, float& g,
                                           float& b) {
    const int id = eid >= e_stored ? eid - e_stored : eid;   // the reversed copy shares the branch parameters
    const float r = fmaf(ea[2 * id], st.es[0], st.em[0]), xx = fmaf(ea[2 * id + 1], st.es[1], st.em[1]);
    const float d = r * r + xx * xx;
    g = r / d;
    b = -xx / d;
}

__global__ __launch_bounds__(256) void power_imbalance_fwd_kernel(int n, int e_stored, const int* __restrict__ rp_out,
                                                                  const int* __restrict__ out_dst,
                                                                  const int* __restrict__ out_eid, const float* __restrict__ x,
                                                                  const float* __restrict__ ea, const PiStats st,
                                                                  float* __restrict__ dpq, PiHeader* __restrict__ hd,
                                                                  float* __restrict__ loss) {
    __shared__ float red[256];
    __shared__ int s_last;
    float acc = 0.f;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const Bus bi = bus_of(x, i, st);
        float ap = 0.f, aq = 0.f;
        for (int p = rp_out[i]; p < rp_out[i + 1]; ++p) {
            const Bus bj = bus_of(x, out_dst[p], st);
            float g, b;
            admittance(ea, out_eid[p], e_stored, st, g, b);
            const float t1 = bi.e * bj.e - bi.e * bi.e + bi.f * bj.f - bi.f * bi.f, t2 = bi.f * bj.e - bi.e * bj.f;
            ap += g * t1 + b * t2;
            aq += g * t2 - b * t1;
        }
        const float dp = fmaf(x[4 * i + 2], st.xs[2], st.xm[2]) - ap, dq = fmaf(x[4 * i + 3], st.xs[3], st.xm[3]) - aq;
        dpq[2 * i] = dp;
        dpq[2 * i + 1] = dq;
        acc += dp * dp + dq * dq;
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        hd->partial[blockIdx.x] = red[0];
        __threadfence();
        const int t = __hip_atomic_fetch_add(&hd->counter, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        s_last = t == (int)gridDim.x - 1;
    }
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    red[threadIdx.x] = threadIdx.x < gridDim.x ? __hip_atomic_load(&hd->partial[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.f;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        loss[0] = red[0] / (float)n;    // n == 0: NaN, torch's mean of nothing
        hd->counter = 0;
    }
}

// d loss / d x[k][:]:  loss = (1/N) sum_i (dP_i^2 + dQ_i^2)
__global__ __launch_bounds__(256) void power_imbalance_bwd_kernel(int n, int e_stored, const int* __restrict__ rp_out,
                                                                  const int* __restrict__ out_dst,
                                                                  const int* __restrict__ out_eid, const int* __restrict__ rp_in,
                                                                  const int* __restrict__ in_src, const int* __restrict__ in_eid,
                                                                  const float* __restrict__ x, const float* __restrict__ ea,
                                                                  const PiStats st, const float* __restrict__ dpq,
                                                                  float* __restrict__ grad) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const float w = 2.0f / (float)n;
    const Bus bk = bus_of(x, k, st);
    const float gp = w * dpq[2 * k], gq = w * dpq[2 * k + 1];
    float de = 0.f, df = 0.f;
    // rows where k is the source i: its own imbalance, derivative w.r.t. (e_i, f_i)
    for (int p = rp_out[k]; p < rp_out[k + 1]; ++p) {
        const Bus bj = bus_of(x, out_dst[p], st);
        float g, b;
        admittance(ea, out_eid[p], e_stored, st, g, b);
        const float dPde = g * (bj.e - 2.f * bk.e) - b * bj.f, dPdf = g * (bj.f - 2.f * bk.f) + b * bj.e;
        const float dQde = -g * bj.f - b * (bj.e - 2.f * bk.e), dQdf = g * bj.e - b * (bj.f - 2.f * bk.f);
        de -= gp * dPde + gq * dQde;
        df -= gp * dPdf + gq * dQdf;
    }
    // rows where k is the destination j: the source's imbalance, derivative w.r.t. (e_j, f_j)
    for (int p = rp_in[k]; p < rp_in[k + 1]; ++p) {
        const int s = in_src[p];
        const Bus bi = bus_of(x, s, st);
        float g, b;
        admittance(ea, in_eid[p], e_stored, st, g, b);
        const float sp = w * dpq[2 * s], sq = w * dpq[2 * s + 1];
        const float dPde = g * bi.e + b * bi.f, dPdf = g * bi.f - b * bi.e;
        const float dQde = g * bi.f - b * bi.e, dQdf = -g * bi.e - b * bi.f;
        de -= sp * dPde + sq * dQde;
        df -= sp * dPdf + sq * dQdf;
    }
    float4 out;
    out.x = (de * bk.c + df * bk.s) * st.xs[0];                                               // Vm
    out.y = (-de * bk.f + df * bk.e) * (3.14159265358979323846f / 180.0f) * st.xs[1];         // Va (degrees)
    out.z = gp * st.xs[2];                                                                    // P
    out.w = gq * st.xs[3];                                                                    // Q
    *reinterpret_cast<float4*>(grad + 4 * (size_t)k) = out;
}

}  // namespace pfn

using namespace pfn;

extern "C" int pfn_power_imbalance(const void* graph_ws, int64_t n_nodes, int64_t e_stored, const float* x, const float* edge_attr,
                                   const float* stats, float* loss, float* grad_x, float* dpq, void* ws, size_t ws_bytes,
                                   void* stream) {
    PFN_CHECK_ARG(graph_ws && x && stats && loss && dpq && ws, "pfn_power_imbalance: null pointer");
    PFN_CHECK_ARG(edge_attr || e_stored == 0, "pfn_power_imbalance: null edge_attr");
    if (ws_bytes < sizeof(PiHeader)) {
        set_error("pfn_power_imbalance: workspace too small (need %zu bytes)", sizeof(PiHeader));
        return PFN_ENOSPACE;
    }
    const GraphView g = graph_view(const_cast<void*>(graph_ws), n_nodes, e_stored);
    PiStats st;
    for (int i = 0; i < 4; ++i) { st.xm[i] = stats[i]; st.xs[i] = stats[4 + i]; }
    for (int i = 0; i < 2; ++i) { st.em[i] = stats[8 + i]; st.es[i] = stats[10 + i]; }
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int n = (int)n_nodes;
    const int nb = std::max(1, std::min((n + 255) / 256, 256));
    power_imbalance_fwd_kernel<<<nb, 256, 0, s>>>(n, (int)e_stored, g.rowptr_out, g.out_dst, g.out_eid, x, edge_attr, st, dpq,
                                                  static_cast<PiHeader*>(ws), loss);
    PFN_CHECK_LAUNCH();
    if (grad_x && n > 0) {
        power_imbalance_bwd_kernel<<<(n + 255) / 256, 256, 0, s>>>(n, (int)e_stored, g.rowptr_out, g.out_dst, g.out_eid, g.rowptr_in,
                                                                   g.in_src, g.in_eid, x, edge_attr, st, dpq, grad_x);
        PFN_CHECK_LAUNCH();
    }
    return PFN_OK;
}
