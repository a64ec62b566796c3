// Internal declarations shared by the HIP translation units of libpfn_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <atomic>

#include "../../include/pfn_hip.h"

namespace pfn {

// ------------------------------------------------------------------------------------------ errors
void set_error(const char* fmt, ...);
#define PFN_CHECK_ARG(cond, ...)                 \
    do {                                         \
        if (!(cond)) {                           \
            ::pfn::set_error(__VA_ARGS__);       \
            return PFN_EINVAL;                   \
        }                                        \
    } while (0)
#define PFN_CHECK_HIP(expr)                                                              \
    do {                                                                                 \
        hipError_t e__ = (expr);                                                         \
        if (e__ != hipSuccess) {                                                         \
            ::pfn::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
            return PFN_EHIP;                                                             \
        }                                                                                \
    } while (0)
#define PFN_CHECK_LAUNCH() PFN_CHECK_HIP(hipGetLastError())
#define PFN_TRY(expr)              \
    do {                           \
        int rc__ = (expr);         \
        if (rc__ != PFN_OK) return rc__; \
    } while (0)

// RAII event bracket around one kernel launch (prof.hip); `bytes`/`flops` are the ALGORITHMIC figures of
// SURVEY.md 8(d) for that launch, so bench.py can turn measured durations into roofline fractions.
struct ProfScope {
    ProfScope(const char* name, double bytes, double flops, hipStream_t s);
    ~ProfScope();
    long idx_;
    hipStream_t s_;
};

// ---- per-device facts, safe from any host thread and with several devices in one process (no process-global "first
// caller wins" state): compute-unit count of the CURRENT device, and "raise this kernel's dynamic-LDS limit" done once
// per device (hipFuncSetAttribute applies to the current device only; repeating it is harmless, so no lock is needed --
// `done` is a bit mask over device ordinals).
int device_cus();
// The ONE place the library reads the environment: the diagnostic switches listed under "Diagnostic environment switches" in
// include/pfn_hip.h (A/B aids for tests and tuning; unset = the product's behaviour).  Call sites keep the value in a
// function-local static, i.e. a switch is read once per process.
const char* diag_env(const char* name);
// SERPENTINE SWEEPS (round 6).  The kernels that stream whole activation tensors (gemm_nt, the LDS-resident walks and hops, the
// generic forward walk) can visit their rows first-to-last or last-to-first with identical results.  Consecutive launches alternate:
// a consumer then starts with the rows its producer wrote LAST -- still in L2 / Infinity Cache -- instead of with the rows written
// first, which a 128-255 MB tensor has pushed out by the time it is complete (case118v2 x 2048: 2.29 -> 2.23 ms per forward; with
// `nt` stores, i.e. nothing left behind for the consumer, the same step takes 2.55).  next_sweep_direction() = the direction of the
// next such launch: the parity of a thread-local count of them (PFN_NO_SERPENTINE=1: always 0).  It orders memory traffic only --
// no result depends on it.
int next_sweep_direction();
int ensure_dynamic_lds(const void* kernel, int bytes, std::atomic<uint64_t>& done);

static inline int64_t round_up(int64_t v, int64_t m) { return (v + m - 1) / m * m; }
static inline int ld_of(int f) { return (int)round_up(f, 4); }

// bump allocator over a caller-provided workspace (256-byte aligned sub-buffers)
struct Carver {
    char* base;
    size_t off = 0;
    explicit Carver(void* p) : base(static_cast<char*>(p)) {}
    template <typename T>
    T* take(size_t count) {
        T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
        off += (size_t)round_up((int64_t)(count * sizeof(T)), 256);
        return p;
    }
};

// ------------------------------------------------------------------------------------------- graph
// Device-side view of the adjacency built by pfn_graph_build.  "in" = CSR by destination (row i lists
// the edges arriving at i: what the forward aggregation walks); "out" = CSR by source (what the
// gradient w.r.t. the gathered operand walks).  eid >= e_stored marks the reversed copy of eid - e_stored.
constexpr int GRAPH_SCAN_BLOCKS = 1024;   // blocks of the degree scan (= its block size: one total per thread)
struct GraphView {
    int n;          // nodes
    int e_stored;   // stored edges
    int* flags;     // [0] directed, [1] effective edge count, [2] index error, [3] reverse-found scratch
    int* scan_sums; // [3][GRAPH_SCAN_BLOCKS] per-block totals of the degree scan (graph.hip pass 2)
    int* rowptr_in;   // [n+1]
    int* rowptr_out;  // [n+1]
    int* in_src;      // [2*e_stored]
    int* in_eid;
    int* out_dst;
    int* out_eid;
    int* rp4;         // [n+1] prefix sum of ceil(in-degree / 4): row offsets of the edge stage's ReLU masks (EdgeFwdArgs::mask)
    int* out_mbase;   // [2*e_stored] per by-source slot: rp4[dst], and {ceil(deg_dst / 4), position of the edge among dst's incoming
    int2* out_ml4k;   //              edges} -- where the by-source half of the backward walk finds the edge's mask byte
    int* slot_of_eid; // [2*e_stored] build scratch: by-destination slot of edge id
    int* cur_in;      // [n] scratch: histogram, then fill cursor
    int* cur_out;     // [n]
    float* deg;       // [n] in-degree (multiplicity kept)
    float* dinv;      // [n] deg^-1/2, 0 where deg == 0
    size_t bytes;
};
GraphView graph_view(void* ws, int64_t n, int64_t e_stored);

// ------------------------------------------------------------------------------------------- GEMMs
// Weights are re-laid out once per forward into zero-padded LDS images (gemm_nt.hip: pack), K8 = roundup(K, 8):
//   image[q][g][c][i] = B[4g + i][32q + c]   for the nq 32-column MFMA quarters of an output of ld_out columns, then
//   rem[g][c][i]      = B[4g + i][32nq + c]  for the up to 4 trailing (VALU) columns,
// with B[k][n] = trans ? W[(wn0 + n) * ldw + wk0 + k] : W[(wk0 + k) * ldw + wn0 + n], zero outside [0,K) x [0,ncols).
struct PackJob {
    const float* src;
    float* dst;
    int ldw, wk0, wn0, trans, K, ncols, ld_out;
};
constexpr int PACK_MAX_JOBS = 64;
// optional rider of the pack launches: the edge attributes gathered ONCE per forward into CSR slot order (by destination and by
// source), Fe = 2 -- the graph-resident EdgeAggregation kernels (ea_seg.hip) then stage a block's attributes with one
// coalesced load instead of an index load followed by a dependent gather in each of their eight launches per step
struct SlotEa {
    const int* rowptr_in = nullptr;   // [n + 1]: rowptr_in[n] = number of slots
    const int* in_eid = nullptr;
    const int* out_eid = nullptr;
    const float* ea = nullptr;        // [e_stored][2]
    float* ea_in = nullptr;           // [slots][2], null: no rider
    float* ea_out = nullptr;
    int n = 0, e_stored = 0;
};
__device__ inline void slot_ea_body(const SlotEa& se, int64_t gtid, int64_t nthr) {
    if (!se.ea_in) return;
    const int nslot = se.rowptr_in[se.n];
    for (int64_t i = gtid; i < nslot; i += nthr) {
        int a = se.in_eid[i], b = se.out_eid[i];
        a = a >= se.e_stored ? a - se.e_stored : a;
        b = b >= se.e_stored ? b - se.e_stored : b;
        reinterpret_cast<float2*>(se.ea_in)[i] = *reinterpret_cast<const float2*>(se.ea + (size_t)a * 2);
        reinterpret_cast<float2*>(se.ea_out)[i] = *reinterpret_cast<const float2*>(se.ea + (size_t)b * 2);
    }
}
struct PackArgs {
    PackJob job[PACK_MAX_JOBS];
    SlotEa slot_ea;
    int njobs;
    uint64_t* rng_advance;   // device {seed, offset}: offset += 1 (dropout stream), or null
    int* stamp;              // optional: *stamp = stamp_value (the forward pass marks its workspace: training / inference)
    int stamp_value;
    // optional rider (one launch floor less per forward): pred_mask -> float32, spread over all blocks of the launch
    const void* mask;
    float* maskf;
    int64_t mask_count;
    int mask_dtype;          // 0: int64, 1: float32
};
size_t packed_floats(int K, int ld_out);
// k rows of a packed image: K rounded up to a multiple of 8 -- or, beyond one 136-k piece, to whole PIECES (hidden_dim 512 ->
// 544): every piece of a wide product is then the straight-line 17-chunk kind (the zero rows multiply clamped A reads), which is
// what lets the weight-streaming kernel take it
__host__ __device__ inline int k8_of(int K) { return K <= 136 ? ((K + 7) & ~7) : (K + 135) / 136 * 136; }
__host__ __device__ inline size_t packed_fp32_floats(int K, int ld_out) {
    int remv, nq;
    const int m = ld_out & 31;
    remv = (m != 0 && m <= 4) ? m : 0;
    nq = (ld_out - remv + 31) / 32;
    const int G = k8_of(K) >> 2;
    return (size_t)(((int64_t)nq * G * 128 + (int64_t)G * 16 + 255) / 256 * 256);
}
// column plan of an output of `ld` (padded) columns: `remv` trailing columns (0 or 4) go to the VALU path when that
// saves a whole MFMA quarter; `nq` 32-column MFMA quarters.
__host__ __device__ inline void col_plan(int ld, int& remv, int& nq) {
    const int m = ld & 31;
    remv = (m != 0 && m <= 4) ? m : 0;
    nq = (ld - remv + 31) / 32;
}
// one block's share (block bx of nbx) of one weight re-layout job
__device__ inline void pack_job_body(const PackJob& jb, int bx, int nbx) {
    int remv, nq;
    col_plan(jb.ld_out, remv, nq);
    const int G = k8_of(jb.K) >> 2;             // groups of four k's
    const long main_floats = (long)nq * G * 128, total = main_floats + (long)G * 16;
    // four elements per trip, all four (strided, transposing) loads requested before the first store: one element per trip was a
    // chain of dependent gather -> store round trips (a workgroup with 36 trips: 30 us)
    const long step = (long)nbx * blockDim.x;
    for (long i0 = (long)bx * blockDim.x + threadIdx.x; i0 < total; i0 += 4 * step) {
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long i = i0 + u * step < total ? i0 + u * step : i0;
            int k, n;
            if (i < main_floats) {
                const int q = (int)(i / ((long)G * 128));
                const int r = (int)(i - (long)q * G * 128);
                k = 4 * (r >> 7) + (r & 3);
                n = 32 * q + ((r & 127) >> 2);
            } else {
                const int r = (int)(i - main_floats);
                k = 4 * (r >> 4) + (r & 3);
                n = 32 * nq + ((r & 15) >> 2);
            }
            const bool in = k < jb.K && n < jb.ncols;
            const int kc = in ? k : 0, nc = in ? n : 0;
            const float x = jb.trans ? jb.src[(size_t)(jb.wn0 + nc) * jb.ldw + jb.wk0 + kc] : jb.src[(size_t)(jb.wk0 + kc) * jb.ldw + jb.wn0 + nc];
            v[u] = in ? x : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (i0 + u * step < total) jb.dst[i0 + u * step] = v[u];
    }
}
int launch_pack(const PackJob* jobs, int njobs, uint64_t* rng_advance, hipStream_t s, const void* mask = nullptr,
                int mask_dtype = 0, float* maskf = nullptr, int64_t mask_count = 0, const SlotEa* slot_ea = nullptr,
                int* stamp = nullptr, int stamp_value = 0);

// C[g] (M x ldc) = sum over terms t with t.group == g of  A_t (M x K_t) * B_t (K_t x ncols)  + epilogue,
// B_t given as a packed image (Bp).
struct GemmTerm {
    const float* A;
    const float* Bp;
    int lda, K, group;
    int cm_rows;   // 0: A is row-major [M][lda].  > 0: A is CHUNK-MAJOR, [lda / 4 planes][cm_rows rows][float4] (element (r, k) at
                   // ((k / 4) * cm_rows + r) * 4 + k % 4): what the big-graph hop kernel writes; cm_rows = rows of the whole buffer
};
enum { ACT_NONE = 0, ACT_RELU = 1, ACT_DROPOUT_RELU = 2 };
struct GemmArgs {
    int M, ncols, ldc, nterm, ngroup, pad_;
    float* C[8];            // per group
    GemmTerm term[8];
    const float* bias;      // [ncols] or null
    int bias_group;         // group that receives the bias (-1: all)
    const float* rowscale;  // [M] or null  : + rowscale[m] * rowbias[n]
    const float* rowbias;   // [ncols]
    const float* resid;     // [M x ldr] or null
    int ldr;
    int act;
    float p_drop;
    const uint64_t* rng;    // device {seed, offset}
    uint32_t rng_stream;
    const float* gate;      // [M x ldg] or null : out *= gate > 0 ? gate_scale : 0
    int ldg;
    float gate_scale;
    int c_cm_rows;          // > 0: every C is written CHUNK-major ([ldc / 4 planes][c_cm_rows][float4], see GemmTerm::cm_rows)
    int aux_cm_rows;        // > 0: `gate` / `resid` is chunk-major
    int row0;               // global index of this call's first row (the dropout stream is keyed by the GLOBAL row): a launch over
                            // a row range of a larger product passes its offset here; 0 otherwise
};
int launch_gemm_nt(const GemmArgs& a, hipStream_t s);

// dW[(gn0 + i) * ldg + gk0 + j] = sum_m A[m][i] * B[m][j],  i < na, j < nb  (A = grad of the layer output,
// B = the layer input: the nn.Linear weight gradient), reduced deterministically in two stages.  When bias_out
// is set, bias_out[i] = sum_m bias_rowscale[m] * A[m][i] rides along as a virtual extra column of B.
struct TnPair {
    const float* A;
    const float* B;
    float* G;
    float* bias_out;
    const float* bias_rowscale;
    int lda, ldb, na, nb, ldg, gn0, gk0;
    int b_cm_rows;   // 0: B is row-major [M][ldb].  > 0: B is chunk-major (see GemmTerm::cm_rows), that many rows per plane
    int a_cm_rows;   // the same for A
};
struct ReduceWs {
    float* partial;   // scratch for split partials
    size_t floats;
};
size_t reduce_ws_floats(int64_t M, int max_na, int max_nb, int max_pairs);
// stamp / stamp_want: optional device guard word -- when *stamp != stamp_want every gradient of the launch is written as NaN
// (pfn_mpn_backward on a workspace whose last forward did not save what the backward reads: model.hip WS_STAMP_TRAIN)
int launch_weight_grads(const TnPair* pairs, int npairs, int64_t M, ReduceWs ws, hipStream_t s, const struct DweRide* ride = nullptr,
                        const int* stamp = nullptr, int stamp_want = 0);

// ------------------------------------------------------------------------------------- edge kernels
// y[i] = (add ? add[i] : 0) + dinv[i] * sum_{e in row i} dinv[nbr(e)] * x[nbr(e)]   (normalize)
// y[i] = sum_{e in row i} x[nbr(e)]                                                  (!normalize)
// transpose = walk the by-source CSR (A_hat^T).  gate: y *= gate > 0 ? gate_scale : 0.
struct HopArgs {
    const float* x;
    const float* add;
    float* y;
    const float* gate;
    float gate_scale;
    int ld, normalize, transpose;
};
int launch_hop(const GraphView& g, const HopArgs& a, hipStream_t s);

// K normalised hops with the node rows of `seg_per_block` whole graphs resident in LDS (two ping-pong tiles):
//   forward  (transpose = 0):  xs[k] = A_hat xs[k-1], xs[0] = x0;  every xs[k] (k = 1..K) is written to `xk + (k-1)*stride`
//   backward (transpose = 1):  z = G[K]; z = G[k] + A_hat^T z for k = K-1..0; only the final z is written (to `out`), gated
// Requires the segment property checked by pfn_graph_segments.  Returns false from fused_hops_fit() when a graph's rows
// do not fit in LDS (the caller then runs K launch_hop passes).
struct FusedHopsArgs {
    const float* x0;      // forward: layer input;          backward: unused
    float* xk;            // forward: K output buffers;     backward: unused
    const float* G;       // backward: K+1 buffers G[0..K]; forward: unused
    float* out;           // backward: final gradient
    const float* gate;
    float gate_scale;
    size_t stride;        // floats between consecutive k buffers
    int ld, K, transpose, seg;   // transpose = backward (Horner) data flow
    int adjt = -1;               // which adjacency: 1 = by-source rows (A_hat^T), 0 = by-destination; -1 = same as `transpose`
    int x0_cm = 0;               // big-graph hops only: x0 is chunk-major ([ld / 4 planes][n][float4])
};
// Does a TAGConv over this batch take big_graph_hops_kernel (and hence chunk-major hop buffers)?  One predicate for the layer that
// PRODUCES the TAGConv's input (it then writes it chunk-major), the TAGConv itself, and their backward passes.
bool tag_uses_big_hops(int seg, int ld, int n, int64_t e_stored, int K);
bool tag_input_cm(int seg, int ld, int n, int64_t e_stored, int K);   // ... and is that TAGConv's INPUT written chunk-major too?
bool fused_hops_fit(int seg, int ld, int n);
int launch_fused_hops(const GraphView& g, const FusedHopsArgs& a, hipStream_t s);
// the same K hops (forward data flow only) for graphs too large for two LDS tiles: one float4 column of one graph per block,
// one tile + registers (edge.hip big_graph_hops_kernel)
bool big_hops_fit(int seg, int n, int64_t e_stored);
int launch_big_graph_hops(const GraphView& g, const FusedHopsArgs& a, hipStream_t s);

// S[i] = sum_{e -> i} relu(P[i] + Q[src(e)] + sum_f a_e[f] * W1[:, 2Fi + f])
// Sum of a row's nchunk float4 partials in a FIXED order, in two levels (a single thread walking all 33 was a chain of 33
// dependent LDS reads, ~1 us per row group): lanes c < 8 each add the partials c, c + 8, c + 16, ... in that order, then lane 0
// adds the eight sub-sums in lane order.  Called by every thread of the block (two barriers inside); result in part[r * nchunk].
__device__ __forceinline__ void row_sum(float4* part, int r, int c, int nchunk, bool on) {
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (on && c < 8) {
        for (int k = c; k < nchunk; k += 8) {
            const float4 p = part[r * nchunk + k];
            s.x += p.x; s.y += p.y; s.z += p.z; s.w += p.w;
        }
    }
    __syncthreads();
    if (on && c < 8) part[r * nchunk + c] = s;
    __syncthreads();
    if (on && c == 0) {
        float4 t = part[r * nchunk];
        const int m = nchunk < 8 ? nchunk : 8;
        for (int k = 1; k < m; ++k) {
            const float4 p = part[r * nchunk + k];
            t.x += p.x; t.y += p.y; t.z += p.z; t.w += p.w;
        }
        part[r * nchunk] = t;
    }
}

// ReLU masks of the edge stage: one byte per (incoming edge, float4 column chunk), bit i = pre-activation of column 4c + i > 0.
// Written by the forward walk when a backward pass will follow, read by the backward walks INSTEAD of recomputing
// P[dst] + Q[src] + a_e We: they then gather dS rows only (a third of the gathers).  Layout: destination row i owns
// nchunk * L4_i dwords at dword offset rp4[i] * nchunk, L4_i = ceil(deg_i / 4); chunk c's run of L4_i dwords holds the bytes of
// the row's incoming edges in slot order, so the thread (i, c) of a walk writes / reads ONE whole dword per trip of four
// slots (byte stores into slot-major records made the forward walk 23 % slower).  Size: mask_dwords(n, e, ld).
__host__ __device__ inline size_t mask_dwords(size_t n, size_t e_stored, int ld) { return (n + e_stored / 2 + 2) * (size_t)(ld / 4); }
struct EdgeFwdArgs {
    const float* P;
    const float* Q;
    const float* edge_attr;
    const float* w1;
    float* S;
    int ld, h, fi, fe;
    unsigned* mask = nullptr;        // optional, mask_dwords(n, e, ld) dwords
    // last layer only (edge_fwd_out_ok): out[N][4] = S W2^T + deg b2 is formed in the same launch
    float* out = nullptr;
    const float* w2 = nullptr;
    const float* b2 = nullptr;
    int fo = 0;
    int seg = 0;                     // > 0: the batch is a union of graphs of this many nodes (pfn_graph_segments): big batches of
                                     // small graphs take edge_rows_fwd_kernel (the graph's Q rows LDS-resident)
    // first layer behind the 4-wide front (Fi = 4, Fe = 2): P and Q null, the walk forms P[i] = b1 + W1i x0[i] and Q[j] = W1j x0[j]
    // from the 16-byte x0 rows itself, with the front's fma chains (front.hip) -- 2 N H floats neither written nor read back
    const float* x0 = nullptr;
    const float* b1 = nullptr;
};
bool edge_fwd_out_ok(int fe, int h, int fo, int ldo);
int launch_edge_fwd(const GraphView& g, const EdgeFwdArgs& a, hipStream_t s);

// EdgeAggregation for batches of small graphs, graph-resident in LDS (ea_seg.hip): the node GEMM of one 32-column quarter and
// the edge walk over it in ONE launch per direction.  `seg` = nodes per graph (pfn_graph_segments); ea_seg_fit() says whether
// the rows of whole graphs fit (else: gemm_nt + edge_fwd / edge_bwd).  Fe = 2 only.
// `loss = MSELoss()(out, y); loss.backward()` riding in the last layer's graph-resident backward launch (pfn_mpn_backward_mse):
// the out Linear (lin_out4_wave_kernel's summation tree, bit for bit), the loss and its gradient need the S rows of a graph only, so
// every (graph, quarter) workgroup recomputes its graph's 16-byte out rows; ONE quarter per graph stores out / grad_out and
// contributes the graph's loss partial; the last of those to finish sums the partials in block order.
struct MseTail {
    const float* S = nullptr;      // N x ld, the forward edge stage's sums
    const float* b2 = nullptr;
    const float* deg = nullptr;
    const float* y = nullptr;      // N x 4 targets; null: no tail
    float* out = nullptr;          // N x 4
    float* gout = nullptr;         // N x 4: grad_out, read by the weight-gradient launch
    float* partial = nullptr;      // [row blocks] loss partials
    int* counter = nullptr;        // arrival counter, zero between launches
    float* loss = nullptr;
    float inv_n = 0.f;             // 1 / (4 N)
    // Masked_L2_loss instead of MSELoss (utils/custom_loss_functions.py:10-46) when `maskf` is set: the float mask rows the
    // forward's front kept (N x 4) and the per-row-block counts {#(m != 0), #(1 - m != 0)} its front_seg_fwd_kernel left behind
    // (the two means' denominators are sums over the whole batch: every block adds up the `count_blocks` pairs itself);
    // partial then holds TWO floats per block (the two squared-error sums)
    const float* maskf = nullptr;
    const int* counts = nullptr;   // [count_blocks][2]
    int count_blocks = 0;
    int regularize = 0;
    float regcoeff = 0.f;
};
struct EaSegFwdArgs {
    const float* x;        // layer input, N x ldx, K = Fi real columns
    const float* Bi;       // packed images (ld_out = ld) of W1[:, :Fi]^T and W1[:, Fi:2Fi]^T
    const float* Bj;
    const float* b1;
    const float* w1;       // raw W1 [H][2Fi + 2] (residue columns)
    const float* ea_in;    // edge attributes in by-destination CSR slot order (SlotEa)
    float* P;
    float* Q;
    float* S;
    int ldx, K, ld, h, fi;
};
struct EaSegBwdArgs {
    const float* gout;     // N x ldgo gradient of the layer output, Fo real columns
    const float* Bd;       // packed image (ld_out = ld) of W2 (Fo -> H), or null: last layer (Fo <= 4, ldgo == 4), dS from `w2` on the fly
    const float* w2;       // raw W2 [Fo][H]
    const float* P;
    const float* Q;
    const float* ea_in;    // edge attributes in CSR slot order (SlotEa), by destination / by source
    const float* ea_out;
    const float* w1;
    float* dP;
    float* dQ;
    float* dWe_partial;    // [ea_seg_blocks][2][ld]
    int ldgo, fo, ld, h, fi;
    // MSELoss tail (last layer only; `mse.y` set): the launch forms out = S W2^T + deg b2 (lin_out4's bits), the loss partials and
    // grad_out = 2 (out - y) / (4 N) itself instead of reading `gout` -- see MseTail
    struct MseTail mse;
};
bool ea_seg_fit(int seg, int n, int fe, int ld, bool bwd);
// mask_embd + residual AND the first EdgeAggregation's edge stage in one graph-resident launch (ea_seg.hip front_seg_fwd_kernel),
// carrying the forward pass's weight re-layout jobs like launch_front_fwd_pack; writes maskf, me_h (when f.me_h), x0, P, Q, S
bool front_seg_fit(int seg, int n, int h, int fe);
int launch_front_seg_fwd(const GraphView& g, const struct FrontFwdArgs& f, const PackJob* jobs, int njobs, uint64_t* rng_advance,
                         const SlotEa* slot_ea, int* stamp, int stamp_value, const float* ea, float* S, int seg, hipStream_t s);
int ea_seg_blocks(int seg, int n, int ld);
int launch_ea_seg_fwd(const GraphView& g, const EaSegFwdArgs& a, int seg, hipStream_t s);
int launch_ea_seg_bwd(const GraphView& g, const EaSegBwdArgs& a, int seg, hipStream_t s);

// A per-node Linear (one or two terms, K = H = 129-shaped) and the K TAGConv hops over its output in ONE launch, graph-resident
// in LDS (seg_lin_hops.hip): forward  y = act(S W2^T + deg b2), x^(k) = A_hat x^(k-1);  backward  g = (dP W1i + dQ W1j)[gate > 0],
// g^(k) = A_hat^T g^(k-1).  Bit-identical to gemm_nt + fused_hops (two launches), which remain the path for every other shape.
struct SegLinHopsArgs {
    const float* A0;        // operand of term 0, N x lda, K real columns
    const float* A1;        // operand of term 1 (same lda / K), or null: one term
    const float* B0;        // packed images (K -> ncols, ld_out = ld)
    const float* B1;
    const float* rowscale;  // [N] or null:  + rowscale[row] * rowbias[col]
    const float* rowbias;   // [ncols]
    const float* gate;      // [N x ldg] or null:  out *= gate > 0 ? gate_scale : 0
    const uint64_t* rng;    // device {seed, offset} (ACT_DROPOUT_RELU)
    float* y;               // N x ld: the Linear's output after the epilogue (= hop 0)
    float* xk;              // hop k = 1..nhops -> xk + (k - 1) * stride
    size_t stride;
    float gate_scale, p_drop;
    uint32_t rng_stream;
    int act, lda, K, ld, ncols, ldg, nhops;
    int adjt;               // 0: by-destination rows (A_hat), 1: by-source rows (A_hat^T)
};
bool seg_lin_hops_fit(int seg, int n, int ld, int K, int ncols, int nhops, int nterm);
int launch_seg_lin_hops(const GraphView& g, const SegLinHopsArgs& a, int seg, hipStream_t s);


struct EdgeBwdArgs {
    const float* P;
    const float* Q;
    const float* dS;
    const float* edge_attr;
    const float* w1;
    float* dP;
    float* dQ;
    float* dWe_partial;   // [edge_bwd_dst_blocks][fe][ld] partial sums of a_e[f] * dh_e
    float* grad_edge_attr;  // optional [e_stored][fe]
    int ld, h, fi, fe;
    // last layer (Fo <= 4, N x 4 output gradient): dS rows are formed inside the walks from gout and W2 [fo][h]; dS is unused
    const float* gout = nullptr;
    const float* w2 = nullptr;
    int fo = 0;
    const unsigned* mask = nullptr;        // the forward walk's ReLU masks (Fe = 2): P, Q and the residue weights are not read
    float* gea_tmp = nullptr;              // optional [2 e_stored][fe] scratch: grad_edge_attr accumulates in a fixed order (a network's
                                           // layers all add into it); null: atomics, order-free only for a zeroed grad_edge_attr
};
int edge_bwd_dst_blocks(const GraphView& g, int ld);   // number of dWe partials the dst walk emits (<= 1024)
int launch_edge_bwd(const GraphView& g, const EdgeBwdArgs& a, const int64_t* edge_index_unused, hipStream_t s);
int launch_edge_attr_grad(const GraphView& g, const EdgeBwdArgs& a, hipStream_t s);
// the same for several EdgeAggregation layers in one launch (all with the same fe / ld / h: the layers of one model)
struct DweJob {
    const float* partial;
    float* gw1;
    int nblocks, ldw, col0, pad_;
};
constexpr int DWE_MAX_JOBS = 16;
struct DweJobs { DweJob job[DWE_MAX_JOBS]; };
// dWe partial [nblocks][fe][ld]  ->  grad_w1[k][col0 + f]  (ordered: interleaved chains per lane, then a fixed tree): one block's
// share, block = 16 output elements x LANES partial lanes (LANES = blockDim.x / 16: 64 in dwe_reduce_kernel, 32 when the work
// rides in the gemm_tn launch).  `red`: LANES x 17 floats of LDS.  Every thread of the block calls it (barriers inside).
template <int LANES>
__device__ inline void dwe_reduce_body(const DweJob& jb, int bx, int fe, int ld, int h, float (*red)[17], bool poison = false) {
    const int nblocks = jb.nblocks;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int i = bx * 16 + tx;
    const bool ok = i < fe * h;
    const int f = ok ? i / h : 0, k = ok ? i - f * h : 0;
    const float* p = jb.partial + (size_t)f * ld + k;
    const size_t stride = (size_t)fe * ld;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (ok) {
        // up to 8 partials of the lane's stride-LANES subset are requested at once, then added in a fixed order: the kernel is
        // a chain of dependent loads, not bandwidth
        int b = ty;
        for (; b + 7 * LANES < nblocks; b += 8 * LANES) {
            const float a0 = p[(size_t)b * stride], a1 = p[(size_t)(b + LANES) * stride], a2 = p[(size_t)(b + 2 * LANES) * stride],
                        a3 = p[(size_t)(b + 3 * LANES) * stride], a4 = p[(size_t)(b + 4 * LANES) * stride],
                        a5 = p[(size_t)(b + 5 * LANES) * stride], a6 = p[(size_t)(b + 6 * LANES) * stride],
                        a7 = p[(size_t)(b + 7 * LANES) * stride];
            s0 += a0; s1 += a1; s2 += a2; s3 += a3;
            s0 += a4; s1 += a5; s2 += a6; s3 += a7;
        }
        for (; b + 3 * LANES < nblocks; b += 4 * LANES) {
            const float a0 = p[(size_t)b * stride], a1 = p[(size_t)(b + LANES) * stride], a2 = p[(size_t)(b + 2 * LANES) * stride],
                        a3 = p[(size_t)(b + 3 * LANES) * stride];
            s0 += a0; s1 += a1; s2 += a2; s3 += a3;
        }
        for (; b < nblocks; b += LANES) s0 += p[(size_t)b * stride];
    }
    red[ty][tx] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    for (int off = LANES / 2; off > 0; off >>= 1) {
        if (ty < off) red[ty][tx] += red[ty + off][tx];
        __syncthreads();
    }
    if (ty == 0 && ok) jb.gw1[(size_t)k * jb.ldw + jb.col0 + f] = poison ? __builtin_nanf("") : red[0][tx];   // (poison: T3Args::stamp)
}
// the dWe reductions of a backward pass riding in the gemm_tn launch (independent work, one launch floor less per step)
struct DweRide {
    DweJobs jobs;
    int njobs = 0, fe = 0, ld = 0, h = 0;
};
int launch_dwe_reduce_multi(const DweJob* jobs, int njobs, int fe, int ld, int h, hipStream_t s, const int* stamp = nullptr,
                            int stamp_want = 0);
// sums dWe partials [nblocks][fe][ld] into grad_w1[:, 2Fi + f]
int launch_dwe_reduce(const float* partial, int nblocks, int fe, int ld, int h, float* grad_w1, int ldw, int col0,
                      hipStream_t s);

// ------------------------------------------------------------------------------------- 4-wide front
// front.hip: mask_embd + residual + the first EdgeAggregation's P | Q in one launch (forward), and the gradient w.r.t. x0 +
// mask_embd's hidden-layer gradient in one launch (backward).  Only for nfeature_dim == 4 (what the reference asserts).
bool front_fused_ok(int f0, int h);
bool front_latency_regime(int h, int n);   // few rows: one row per wave, P | Q of the first layer stored (front.hip)
// the last layer's second Linear (Fo <= 4) as one row per wave, for small batches (front.hip)
bool lin_out4_ok(int h, int fo, int ldo, int n);
int launch_lin_out4(int n, int h, int fo, const float* S, const float* w2, const float* b2, const float* deg, float* out,
                    hipStream_t s);
struct FrontFwdArgs {
    int n, h, ldw1, mask_dtype;            // mask_dtype 0: int64, 1: float32 (pred_mask as the caller holds it)
    const float* x;
    const void* mask;
    const float *wa, *ba, *wb, *bb, *w1, *b1;
    float *maskf, *me_h, *x0, *P, *Q;      // maskf: pred_mask.float() (networks/MPN.py:533), kept for the weight gradients;
                                           // P, Q null: not written (the first edge stage forms them from x0, EdgeFwdArgs::x0)
    int* mask_counts = nullptr;            // front_seg_fwd_kernel only, optional: [row blocks][2] = #(m != 0), #(1 - m != 0) of the
                                           // block's mask entries (the denominators of Masked_L2_loss, MseTail::counts)
};
// the front AND the weight re-layout of a forward pass (independent of each other) in one launch; `rng_advance` as in launch_pack
int launch_front_fwd_pack(const FrontFwdArgs& f, const PackJob* jobs, int njobs, uint64_t* rng_advance, hipStream_t s,
                          const SlotEa* slot_ea = nullptr, int* stamp = nullptr, int stamp_value = 0);
int launch_front_bwd(int n, int h, int ldw1, const float* dP, const float* dQ, const float* me_h, const float* w1,
                     const float* wb, float* g0, float* dh, hipStream_t s);
// The backward front beyond the latency regime when the forward did not store mask_embd's hidden layer (FrontFwdArgs::me_h null
// in training): recomputes it, forms g0, and accumulates mask_embd's four weight gradients itself (neither me_h nor dh touches
// memory); `scratch` >= front_bwd_wg_scratch_floats(n, h) floats of per-workgroup partial sums
size_t front_bwd_wg_scratch_floats(int n, int h);
int launch_front_bwd_wg(int n, int h, int ldw1, const float* dP, const float* dQ, const float* maskf, const float* w1, const float* wa,
                        const float* ba, const float* wb, float* g0, float* scratch, float* gwa, float* gba, float* gwb, float* gbb,
                        hipStream_t s, const int* stamp = nullptr, int stamp_want = 0);
// mask_embd's hidden layer written from the float mask rows after the fact (the gate export), bit-identical to the front's
int launch_front_meh(int n, int h, const void* mask, int mask_dtype, const float* wa, const float* ba, float* me_h, hipStream_t s);
// P | Q of the first layer written from x0 after the fact (FrontFwdArgs::P null in the forward pass), bit-identical to the front's
int launch_front_pq(int n, int h, int ldw1, const float* x0, const float* w1, const float* b1, float* P, float* Q, hipStream_t s);

// ------------------------------------------------------------------------------------ small kernels
int launch_pad_rows(const float* src, int64_t ld_src, float* dst, int64_t ld_dst, int64_t rows, int64_t f,
                    hipStream_t s);

// 16-byte WRITE-THROUGH store of a kernel OUTPUT (global memory only; sc1).  A chain kernel's plain stores leave its output dirty
// in the XCD's L2, and the kernel boundary then waits for the write-back (MI355X_MICROARCH.md "boundary"); written through, the
// lines drain while the kernel still runs.  Inline asm (hipcc has no 128-bit scoped store); the s_nop is the ISA's "VMEM store
// wider than 64 bits -> VALU overwrites its data registers" hazard, invisible to hipcc inside the asm.
__device__ __forceinline__ void st4_wt(float* p, float4 v) {
    typedef float f4wt_ __attribute__((ext_vector_type(4)));
    const f4wt_ d = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(p), "v"(d) : "memory");
}

// ---------------------------------------------------------------------------------------- dropout RNG
// Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11): counter-based, so the mask of a
// forward pass is a pure function of (seed, offset, layer, element) -- nothing is stored, any kernel can re-derive it.
//   key     = { seed[31:0], seed[63:32] ^ offset[63:32] }
//   counter = { row, column / 4, layer stream id, offset[31:0] }         (no flattened index: rows up to 2^32)
// The four output words serve the four columns 4*(col/4) .. +3 of that row; word -> uniform in [0,1) with 24 bits.
struct DropKey { uint32_t k0, k1, stream, off; };
__host__ __device__ __forceinline__ DropKey drop_key(uint64_t seed, uint64_t offset, uint32_t stream) {
    DropKey k;
    k.k0 = (uint32_t)seed;
    k.k1 = (uint32_t)(seed >> 32) ^ (uint32_t)(offset >> 32);
    k.stream = stream;
    k.off = (uint32_t)offset;
    return k;
}
__device__ __forceinline__ void philox4x32_10(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        // (one 64-bit product per multiplier -- v_mad_u64_u32 -- instead of v_mul_hi_u32 + v_mul_lo_u32: 32-bit integer multiplies
        //  are quarter rate, and the ten rounds are most of a dropout epilogue's instruction time)
        const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        c[0] = hi1 ^ c[1] ^ k0;
        c[1] = lo1;
        c[2] = hi0 ^ c[3] ^ k1;
        c[3] = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
}
// four uniforms in [0,1) for columns 4*cg .. 4*cg+3 of `row`
__device__ __forceinline__ void dropout_uniform4(const DropKey& k, uint32_t row, uint32_t cg, float (&u)[4]) {
    uint32_t c[4] = {row, cg, k.stream, k.off};
    philox4x32_10(c, k.k0, k.k1);
#pragma unroll
    for (int i = 0; i < 4; ++i) u[i] = (float)(c[i] >> 8) * (1.0f / 16777216.0f);
}

}  // namespace pfn
