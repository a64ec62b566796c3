// Layer and whole-model drivers + the C ABI of libpfn_hip.so (gfx950).
//
// Mirrors, call for call, the dataflow of the reference's MaskEmbdMultiMPN.forward
// (networks/MPN.py:525-559), EdgeAggregation.forward/message (:23-56) and PyG TAGConv.forward
// (call sites :477-484,:545), plus their autograd, as sequences of the HIP kernels in graph.hip /
// edge.hip / gemm.hip.  Nothing here synchronises or allocates: every buffer is carved out of the
// caller's workspace, so a whole training step can be captured into one hipGraph.
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "pfn_internal.hpp"

namespace pfn {

static GemmArgs gemm_defaults(int M, int ncols, int ldc) {
    GemmArgs a;
    memset(&a, 0, sizeof(a));
    a.M = M;
    a.ncols = ncols;
    a.ldc = ldc;
    a.ngroup = 1;
    a.gate_scale = 1.f;
    a.bias_group = -1;
    return a;
}
static GemmTerm term(const float* A, int lda, int K, const float* Bp, int group) {
    GemmTerm t;
    t.A = A; t.Bp = Bp; t.lda = lda; t.K = K; t.group = group; t.cm_rows = 0;
    return t;
}
static TnPair tn_pair(const float* A, int lda, int na, const float* B, int ldb, int nb, float* G, int ldg, int gk0,
                      float* bias_out, const float* bias_rowscale) {
    TnPair p;
    p.A = A; p.B = B; p.G = G; p.bias_out = bias_out; p.bias_rowscale = bias_rowscale;
    p.lda = lda; p.ldb = ldb; p.na = na; p.nb = nb; p.ldg = ldg; p.gn0 = 0; p.gk0 = gk0; p.b_cm_rows = 0; p.a_cm_rows = 0;
    return p;
}

// Collects the weight re-layout jobs of a forward pass (gemm_nt.hip: pack) and hands out the packed addresses.
// With base == nullptr it only measures.
struct Packer {
    float* base;
    size_t off = 0;
    std::vector<PackJob> jobs;
    explicit Packer(float* b) : base(b) {}
    // B[k][n] = trans ? W[(wn0+n)*ldw + wk0+k] : W[(wk0+k)*ldw + wn0+n],  k < K, n < ncols, for an output of ld_out columns
    const float* add(const float* W, int ldw, int trans, int wk0, int wn0, int K, int ncols, int ld_out) {
        float* dst = base ? base + off : nullptr;
        off += packed_floats(K, ld_out);
        PackJob j;
        j.src = W; j.dst = dst; j.ldw = ldw; j.wk0 = wk0; j.wn0 = wn0; j.trans = trans; j.K = K; j.ncols = ncols;
        j.ld_out = ld_out;
        jobs.push_back(j);
        return dst ? dst : reinterpret_cast<const float*>(0x10);   // non-null sentinel while measuring
    }
    int flush(hipStream_t s, uint64_t* rng_advance = nullptr, const void* mask = nullptr, int mask_dtype = 0,
              float* maskf = nullptr, int64_t mask_count = 0, const SlotEa* slot_ea = nullptr, int* stamp = nullptr,
              int stamp_value = 0) {
        return launch_pack(jobs.data(), (int)jobs.size(), rng_advance, s, mask, mask_dtype, maskf, mask_count, slot_ea, stamp, stamp_value);
    }
};

struct Act {
    int act = ACT_NONE;
    float p = 0.f;
    const uint64_t* rng = nullptr;
    uint32_t stream = 0;
    int out_cm = 0;             // > 0: the layer output is written CHUNK-major with that many rows per plane (its only reader chain
                                // is a TAGConv on big_graph_hops_kernel: tag_uses_big_hops)
};
struct Gate {
    const float* y = nullptr;   // post-activation output of the producing layer
    int ld = 0;
    float scale = 1.f;
    int cm = 0;                 // > 0: y is chunk-major (see Act::out_cm)
};

// ------------------------------------------------------------------------------------ deferred weight gradients
// Weight gradients only feed the optimizer, so no layer's backward waits for them: every layer appends its (dY, X) pairs to
// a list and ONE launch at the end of the backward pass computes all of them (gemm.hip) -- the chip is filled by the whole
// network's dW work at once instead of by 8 small launches competing with the input-gradient chain.  (Round 1 ran them
// per layer on a second stream: measured 1-2 % over no overlap at all, and the sharing slowed the HBM-bound hops 2x.)
// The price is memory: every layer's incoming gradient and dP / dQ stay alive until the end (sized for 288 GB).
struct Deferred {
    std::vector<TnPair> pairs;
    std::vector<DweJob> dwe;     // the dWe partial reductions of the EdgeAggregation layers, one launch for all
};
typedef Deferred PairList;

// ---------------------------------------------------------------------------------- EdgeAggregation
struct EaSaved { float *P, *Q, *S; };
struct EaScratch { float *dS, *dP, *dQ, *dWe; ReduceWs red; float* gea_tmp = nullptr; };
struct EaPack { const float *w1i_t, *w1j_t, *w2_t, *w2_d, *w1i_d, *w1j_d; };

static EaPack ea_pack(Packer& pk, int fi, int fe, int h, int fo, const float* w1, const float* w2) {
    const int ldw1 = 2 * fi + fe, ld = ld_of(h);
    EaPack p;
    p.w1i_t = pk.add(w1, ldw1, 1, 0, 0, fi, h, ld);           // x (N x Fi) -> P
    p.w1j_t = pk.add(w1, ldw1, 1, fi, 0, fi, h, ld);          // x -> Q
    p.w2_t = pk.add(w2, h, 1, 0, 0, h, fo, ld_of(fo));        // S (N x H) -> out
    p.w2_d = pk.add(w2, h, 0, 0, 0, fo, h, ld);               // gout (N x Fo) -> dS
    p.w1i_d = pk.add(w1, ldw1, 0, 0, 0, h, fi, ld_of(fi));    // dP (N x H) -> dx
    p.w1j_d = pk.add(w1, ldw1, 0, 0, fi, h, fi, ld_of(fi));   // dQ -> dx
    return p;
}

static bool back_fused_ok() {
    static const bool off = diag_env("PFN_NO_FUSED_BACK") != nullptr;   // A/B switch: the last layer's fused kernels
    return !off;
}

static int ea_forward(const GraphView& g, int fi, int fe, int h, int fo, const float* x, int ldx, const float* ea,
                      const float* w1, const float* b1, const float* w2, const float* b2, const EaPack& pw, float* out, int ldo,
                      const Act& act, const EaSaved& sv, hipStream_t s, bool pq_ready = false, int seg = 0,
                      const float* ea_in = nullptr, unsigned* relu_mask = nullptr, bool pq_fly = false, float* hop_xk = nullptr,
                      int hop_K = 0, bool walk_done = false) {
    // walk_done: layer 0 behind front_seg_fwd_kernel -- P, Q AND S are already written (ea_seg.hip)
    // hop_xk / hop_K: the TAGConv behind this layer takes its K hops from here (seg_lin_hops.hip: the S W2^T Linear and the hops
    // in one launch; the caller has asked seg_lin_hops_fit)
    const int ld = ld_of(h);
    // batches of small graphs: the P | Q GEMM and the edge walk in one launch, graph-resident in LDS (ea_seg.hip)
    const bool seg_walk = !pq_ready && ea_in && ea_seg_fit(seg, g.n, fe, ld, false);
    if (seg_walk) {
        EaSegFwdArgs e{x, pw.w1i_t, pw.w1j_t, b1, w1, ea_in, sv.P, sv.Q, sv.S, ldx, fi, ld, h, fi};
        PFN_TRY(launch_ea_seg_fwd(g, e, seg, s));
    } else if (!pq_ready) {   // P = x W1[:, :Fi]^T + b1 ; Q = x W1[:, Fi:2Fi]^T   (layer 0: already written by the fused front)
        GemmArgs a = gemm_defaults(g.n, h, ld);
        a.ngroup = 2;
        a.C[0] = sv.P;
        a.C[1] = sv.Q;
        a.nterm = 2;
        a.term[0] = term(x, ldx, fi, pw.w1i_t, 0);
        a.term[1] = term(x, ldx, fi, pw.w1j_t, 1);
        a.bias = b1;
        a.bias_group = 0;
        PFN_TRY(launch_gemm_nt(a, s));
    }
    // the network's last layer (Fo <= 4, no activation): the second Linear rides in the edge walk's launch (edge_fwd_out_kernel)
    const bool out_in_walk = !seg_walk && !walk_done && w2 && act.act == ACT_NONE && edge_fwd_out_ok(fe, h, fo, ldo) && back_fused_ok();
    if (!seg_walk && !walk_done) {
        EdgeFwdArgs e{sv.P, sv.Q, ea, w1, sv.S, ld, h, fi, fe};
        e.mask = relu_mask;   // (a backward pass will follow: it reads the masks instead of recomputing the pre-activations)
        e.seg = seg;
        if (pq_fly) {         // layer 0 behind the front: P | Q were not written, the walk forms them from x0 (first_layer_fly)
            e.P = e.Q = nullptr;
            e.x0 = x;
            e.b1 = b1;
        }
        if (out_in_walk) {
            e.out = out;
            e.w2 = w2;
            e.b2 = b2;
            e.fo = fo;
        }
        PFN_TRY(launch_edge_fwd(g, e, s));
    }
    if (!out_in_walk && w2 && act.act == ACT_NONE && lin_out4_ok(h, fo, ldo, g.n)) {
        // the last layer at small batches: one row per wave.  out == nullptr: deferred -- pfn_mpn_backward_mse forms the rows in its
        // first launch (mse_tail_ok was asked by the caller)
        if (out) PFN_TRY(launch_lin_out4(g.n, h, fo, sv.S, w2, b2, g.deg, out, s));
    } else if (!out_in_walk && hop_xk) {   // out = act(S W2^T + deg * b2) and the next TAGConv's K hops over it, one launch
        SegLinHopsArgs f;
        memset(&f, 0, sizeof(f));
        f.A0 = sv.S; f.B0 = pw.w2_t; f.rowscale = g.deg; f.rowbias = b2; f.rng = act.rng; f.rng_stream = act.stream;
        f.y = out; f.xk = hop_xk; f.stride = (size_t)g.n * ldo; f.gate_scale = 1.f; f.p_drop = act.p; f.act = act.act;
        f.lda = ld; f.K = h; f.ld = ldo; f.ncols = fo; f.nhops = hop_K; f.adjt = 0;
        PFN_TRY(launch_seg_lin_hops(g, f, seg, s));
    } else if (!out_in_walk) {   // out = S W2^T + deg * b2   (the second Linear commutes with the segment sum)
        GemmArgs a = gemm_defaults(g.n, fo, ldo);
        a.C[0] = out;
        a.nterm = 1;
        a.term[0] = term(sv.S, ld, h, pw.w2_t, 0);
        a.rowscale = g.deg;
        a.rowbias = b2;
        a.act = act.act;
        a.p_drop = act.p;
        a.rng = act.rng;
        a.rng_stream = act.stream;
        a.c_cm_rows = act.out_cm;
        PFN_TRY(launch_gemm_nt(a, s));
    } else if (act.out_cm) {
        set_error("EdgeAggregation: a chunk-major output needs the S W2^T GEMM (internal)");
        return PFN_EINVAL;
    }
    return PFN_OK;
}

static int ea_backward(const GraphView& g, int fi, int fe, int h, int fo, const float* x, int ldx, const float* ea,
                       const float* w1, const float* w2, const EaPack& pw, const float* gout, int ldgo, const Gate& gate, float* gx,
                       int ldgx, float* gw1, float* gb1, float* gw2, float* gb2, float* gea, const EaSaved& sv,
                       const EaScratch& sc, hipStream_t s, PairList* defer, int seg = 0, const float* ea_in = nullptr,
                       const float* ea_out = nullptr, const unsigned* relu_mask = nullptr, int gx_cm = 0, float* hop_out = nullptr,
                       int hop_K = 0, const MseTail* mse = nullptr) {
    // mse: `gout` is not written yet -- the graph-resident backward launch forms out, the loss and gout itself (MseTail)
    // hop_out / hop_K: gx is the output gradient of a TAGConv whose backward hops it K times over A_hat^T first: the dx Linear and
    // those hops in one launch (seg_lin_hops.hip; the caller has asked seg_lin_hops_fit)
    const int ld = ld_of(h), ldw1 = 2 * fi + fe;
    // batches of small graphs: the dS GEMM and both backward walks in one launch, graph-resident in LDS (ea_seg.hip)
    const bool seg_walk = !gea && ea_in && ea_out && ea_seg_fit(seg, g.n, fe, ld, true) && (fo > 4 || ldgo == 4);
    if (seg_walk) {
        const bool last = fo <= 4 && ldgo == 4;
        EaSegBwdArgs e{gout, last ? nullptr : pw.w2_d, w2, sv.P, sv.Q, ea_in, ea_out, w1, sc.dP, sc.dQ, sc.dWe, ldgo, fo, ld, h, fi};
        if (mse) e.mse = *mse;
        PFN_TRY(launch_ea_seg_bwd(g, e, seg, s));
    } else if (mse) {
        set_error("EdgeAggregation backward: the MSELoss tail without the graph-resident launch (internal)");
        return PFN_EINVAL;
    }
    // the network's last layer (Fo <= 4): the walks form dS rows from the 16-byte gout rows themselves (edge.hip ds_row), so
    // the K = 4 GEMM that would write N x H (and the walks' re-read of it) goes away
    const bool ds_in_walk = !seg_walk && w2 && fo <= 4 && ldgo == 4 && !gea && back_fused_ok();
    if (!seg_walk && !ds_in_walk) {   // dS = gout W2
        GemmArgs a = gemm_defaults(g.n, h, ld);
        a.C[0] = sc.dS;
        a.nterm = 1;
        a.term[0] = term(gout, ldgo, fo, pw.w2_d, 0);
        PFN_TRY(launch_gemm_nt(a, s));
    }
    EdgeBwdArgs e{sv.P, sv.Q, sc.dS, ea, w1, sc.dP, sc.dQ, sc.dWe, gea, ld, h, fi, fe};
    if (!gea && fe == 2) e.mask = relu_mask;   // written by this layer's generic forward walk (ea_saves_mask)
    e.gea_tmp = sc.gea_tmp;
    if (ds_in_walk) {
        e.gout = gout;
        e.w2 = w2;
        e.fo = fo;
    }
    if (!seg_walk) PFN_TRY(launch_edge_bwd(g, e, nullptr, s));
    if (gea) PFN_TRY(launch_edge_attr_grad(g, e, s));
    if (gx && hop_out) {
        SegLinHopsArgs f;
        memset(&f, 0, sizeof(f));
        f.A0 = sc.dP; f.A1 = sc.dQ; f.B0 = pw.w1i_d; f.B1 = pw.w1j_d; f.gate = gate.y; f.ldg = gate.ld; f.gate_scale = gate.scale;
        f.y = gx; f.xk = hop_out; f.stride = (size_t)g.n * ldgx; f.act = ACT_NONE;
        f.lda = ld; f.K = h; f.ld = ldgx; f.ncols = fi; f.nhops = hop_K; f.adjt = 1;
        PFN_TRY(launch_seg_lin_hops(g, f, seg, s));
    } else if (gx) {   // dx = dP W1[:, :Fi] + dQ W1[:, Fi:2Fi], gated by the producing layer's activation
        GemmArgs a = gemm_defaults(g.n, fi, ldgx);
        a.C[0] = gx;
        a.nterm = 2;
        a.term[0] = term(sc.dP, ld, h, pw.w1i_d, 0);
        a.term[1] = term(sc.dQ, ld, h, pw.w1j_d, 0);
        a.gate = gate.y;
        a.ldg = gate.ld;
        a.gate_scale = gate.scale;
        a.aux_cm_rows = gate.cm;
        a.c_cm_rows = gx_cm;     // (the gradient handed to a TAGConv on the big-graph hop kernel: chunk-major, like its input)
        PFN_TRY(launch_gemm_nt(a, s));
    }
    // weight gradients: dWe partials -> W1[:, 2Fi:], and three (dY, X) pairs
    const int dwe_blocks = g.n > 0 ? (seg_walk ? ea_seg_blocks(seg, g.n, ld) : edge_bwd_dst_blocks(g, ld)) : 0;
    if (defer) defer->dwe.push_back(DweJob{sc.dWe, gw1, dwe_blocks, ldw1, 2 * fi, 0});
    else PFN_TRY(launch_dwe_reduce(sc.dWe, dwe_blocks, fe, ld, h, gw1, ldw1, 2 * fi, s));
    const TnPair pairs[3] = {
        tn_pair(gout, ldgo, fo, sv.S, ld, h, gw2, h, 0, gb2, g.deg),      // dW2 ; db2 = sum_i deg_i gout_i
        tn_pair(sc.dP, ld, h, x, ldx, fi, gw1, ldw1, 0, gb1, nullptr),    // dW1[:, :Fi] ; db1 = sum_i dP_i
        tn_pair(sc.dQ, ld, h, x, ldx, fi, gw1, ldw1, fi, nullptr, nullptr),
    };
    if (defer) {
        defer->pairs.insert(defer->pairs.end(), pairs, pairs + 3);
        return PFN_OK;
    }
    return launch_weight_grads(pairs, 3, g.n, sc.red, s);
}

// ------------------------------------------------------------------------------------------ TAGConv
struct TagPack { const float* wt[8]; const float* wd[8]; };
static TagPack tag_pack(Packer& pk, int cin, int cout, int K, const float* const* w) {
    TagPack p;
    for (int k = 0; k <= K; ++k) {
        p.wt[k] = pk.add(w[k], cin, 1, 0, 0, cin, cout, ld_of(cout));   // x^(k) (N x cin) -> out
        p.wd[k] = pk.add(w[k], cin, 0, 0, 0, cout, cin, ld_of(cin));    // gout (N x cout) -> G_k
    }
    return p;
}

static int tag_forward(const GraphView& g, int cin, int cout, int K, const float* x, int ldx, const TagPack& pw,
                       const float* bias, float* out, int ldo, const Act& act, float* xk, hipStream_t s, int seg = 0,
                       int x_cm = 0, bool hops_done = false) {
    // hops_done: the layer in front already wrote the K hop buffers (seg_lin_hops.hip)
    // xk: K buffers of n * ldx floats holding A_hat^k x, k = 1..K.  x_cm > 0: x itself is chunk-major (the producing layer wrote
    // it so because this TAGConv takes the big-graph hop kernel)
    const size_t stride = (size_t)g.n * ldx;
    int xk_cm = 0;   // > 0: the hop outputs are chunk-major with that many rows per plane (big-graph hops)
    if (x_cm && !tag_uses_big_hops(seg, ldx, g.n, g.e_stored, K)) {
        set_error("TAGConv: chunk-major input without the big-graph hop kernel (internal)");
        return PFN_EINVAL;
    }
    if (hops_done) {
    } else if (K > 0 && fused_hops_fit(seg, ldx, g.n)) {
        FusedHopsArgs fh{x, xk, nullptr, nullptr, nullptr, 1.f, stride, ldx, K, 0, seg};
        PFN_TRY(launch_fused_hops(g, fh, s));
    } else if (K > 0 && big_hops_fit(seg, g.n, g.e_stored)) {
        FusedHopsArgs fh{x, xk, nullptr, nullptr, nullptr, 1.f, stride, ldx, K, 0, seg};
        fh.x0_cm = x_cm;
        PFN_TRY(launch_big_graph_hops(g, fh, s));   // (writes the hop outputs chunk-major)
        xk_cm = g.n;
    } else {
        const float* prev = x;
        for (int k = 1; k <= K; ++k) {
            HopArgs hp{prev, nullptr, xk + (size_t)(k - 1) * stride, nullptr, 1.f, ldx, 1, 0};
            PFN_TRY(launch_hop(g, hp, s));
            prev = hp.y;
        }
    }
    GemmArgs a = gemm_defaults(g.n, cout, ldo);
    a.C[0] = out;
    a.nterm = K + 1;
    for (int k = 0; k <= K; ++k) {
        a.term[k] = term(k == 0 ? x : xk + (size_t)(k - 1) * stride, ldx, cin, pw.wt[k], 0);
        a.term[k].cm_rows = k > 0 ? xk_cm : x_cm;
    }
    a.bias = bias;
    a.act = act.act;
    a.p_drop = act.p;
    a.rng = act.rng;
    a.rng_stream = act.stream;
    return launch_gemm_nt(a, s);
}

struct TagScratch { float* G; float *z0, *z1; ReduceWs red; };

static int tag_backward(const GraphView& g, int cin, int cout, int K, const float* x, int ldx, const TagPack& pw,
                        const float* gout, int ldgo, const Gate& gate, float* gx, int ldgx, float* const* gw,
                        float* gbias, const float* xk, const TagScratch& sc, hipStream_t s, PairList* defer, int seg = 0,
                        int x_cm = 0, int gout_cm = 0, bool hops_done = false) {
    // hops_done: the layer behind (whose dx Linear produced gout) already hopped it into sc.G (seg_lin_hops.hip)
    const size_t stride = (size_t)g.n * ldx;
    float* hk = sc.G;
    if (gout_cm && !(gx && K > 0 && ldgo <= ldx && tag_uses_big_hops(seg, ldgo, g.n, g.e_stored, K))) {
        set_error("TAGConv backward: chunk-major output gradient without the big-graph hop kernel (internal)");
        return PFN_EINVAL;
    }
    if (gx) {
        if (ldgx != ldx) {
            set_error("TAGConv backward: grad_x stride %d != x stride %d", ldgx, ldx);
            return PFN_EINVAL;
        }
        if (K > 0 && ldgo <= ldx) {
            // dx = sum_k (A^T)^k (gout W_k) = sum_k ((A^T)^k gout) W_k: hop the incoming gradient first (K hops over the
            // transposed adjacency, LDS-resident when the graphs fit), then ONE multi-term GEMM with the gate in its epilogue --
            // one output instead of K + 1 (measured 607 vs 754 us for the GEMM at 414 k nodes) and no Horner pass.
            const size_t gstride = (size_t)g.n * ldgo;
            int hk_cm = 0;
            if (hops_done) {
            } else if (fused_hops_fit(seg, ldgo, g.n)) {
                FusedHopsArgs fh{gout, hk, nullptr, nullptr, nullptr, 1.f, gstride, ldgo, K, 0, seg, 1};
                PFN_TRY(launch_fused_hops(g, fh, s));
            } else if (big_hops_fit(seg, g.n, g.e_stored)) {
                FusedHopsArgs fh{gout, hk, nullptr, nullptr, nullptr, 1.f, gstride, ldgo, K, 0, seg, 1};
                fh.x0_cm = gout_cm;
                PFN_TRY(launch_big_graph_hops(g, fh, s));
                hk_cm = g.n;
            } else {
                const float* prev = gout;
                for (int k = 1; k <= K; ++k) {
                    HopArgs hp{prev, nullptr, hk + (size_t)(k - 1) * gstride, nullptr, 1.f, ldgo, 1, 1};
                    PFN_TRY(launch_hop(g, hp, s));
                    prev = hp.y;
                }
            }
            GemmArgs a = gemm_defaults(g.n, cin, ldx);
            a.C[0] = gx;
            a.nterm = K + 1;
            for (int k = 0; k <= K; ++k) {
                a.term[k] = term(k == 0 ? gout : hk + (size_t)(k - 1) * gstride, ldgo, cout, pw.wd[k], 0);
                a.term[k].cm_rows = k > 0 ? hk_cm : gout_cm;
            }
            a.gate = gate.y;
            a.ldg = gate.ld;
            a.gate_scale = gate.scale;
            a.aux_cm_rows = gate.cm;
            PFN_TRY(launch_gemm_nt(a, s));
        } else {
            if (gate.cm || x_cm || gout_cm || hops_done) {
                set_error("TAGConv backward: chunk-major tensors on the Horner path (internal)");
                return PFN_EINVAL;
            }
            // G_k = gout W_k ; dx = G_0 + A^T (G_1 + A^T (G_2 + ...))   (Horner over the transposed adjacency)
            GemmArgs a = gemm_defaults(g.n, cin, ldx);
            a.ngroup = K + 1;
            a.nterm = K + 1;
            for (int k = 0; k <= K; ++k) {
                a.C[k] = (K == 0) ? gx : sc.G + (size_t)k * stride;
                a.term[k] = term(gout, ldgo, cout, pw.wd[k], k);
            }
            if (K == 0) {
                a.gate = gate.y;
                a.ldg = gate.ld;
                a.gate_scale = gate.scale;
            }
            PFN_TRY(launch_gemm_nt(a, s));
            if (K > 0 && fused_hops_fit(seg, ldx, g.n)) {
                FusedHopsArgs fh{nullptr, nullptr, sc.G, gx, gate.y, gate.scale, stride, ldx, K, 1, seg};
                PFN_TRY(launch_fused_hops(g, fh, s));
            } else {
                const float* z = sc.G + (size_t)K * stride;
                for (int k = K - 1; k >= 0; --k) {
                    float* dst = (k == 0) ? gx : ((k & 1) ? sc.z1 : sc.z0);
                    HopArgs hp{z, sc.G + (size_t)k * stride, dst, k == 0 ? gate.y : nullptr, k == 0 ? gate.scale : 1.f, ldx, 1, 1};
                    PFN_TRY(launch_hop(g, hp, s));
                    z = dst;
                }
            }
        }
    }
    // weight gradients need only gout and the saved hops
    std::vector<TnPair> local;
    std::vector<TnPair>& pairs = defer ? defer->pairs : local;
    // (the forward's hop outputs are chunk-major when it took the big-graph hop kernel: the same predicate as tag_forward)
    const int xk_cm = (K > 0 && !fused_hops_fit(seg, ldx, g.n) && big_hops_fit(seg, g.n, g.e_stored)) ? g.n : 0;
    for (int k = 0; k <= K; ++k) {
        pairs.push_back(tn_pair(gout, ldgo, cout, k == 0 ? x : xk + (size_t)(k - 1) * stride, ldx, cin, gw[k], cin, 0,
                                k == 0 ? gbias : nullptr, nullptr));
        pairs.back().b_cm_rows = k > 0 ? xk_cm : x_cm;
        pairs.back().a_cm_rows = gout_cm;
    }
    if (defer) return PFN_OK;
    return launch_weight_grads(local.data(), (int)local.size(), g.n, sc.red, s);
}

// -------------------------------------------------------------------------------------- whole model
struct Layout {
    // dims
    int n, e, f0, fe, fo, h, L, K, ld, ld0, ldo, nlayers;
    int* stamp;                  // guard word: WS_STAMP_TRAIN after a forward that saved what the backward pass reads
    int* mask_counts;            // [1024 row blocks][2]: front_seg_fwd_kernel's mask census (FrontFwdArgs::mask_counts; training only)
    // forward-saved
    float *maskf, *me_h, *x0, *packed;
    float *ea_in, *ea_out;       // edge attributes in CSR slot order (Fe = 2; SlotEa), filled once per forward
    std::vector<unsigned*> relu_mask;   // per EA layer: the edge stage's ReLU masks (EdgeFwdArgs::mask; need_backward)
    size_t packed_floats;
    std::vector<float*> y;       // per layer output (post-activation); last = nullptr (caller's out)
    std::vector<EaSaved> ea;     // per EA layer
    std::vector<float*> xk;      // per TAG layer: K * n * ld
    // backward: per-layer buffers that stay alive until the deferred weight-gradient launch ...
    std::vector<float*> gin;     // gradient w.r.t. the INPUT of layer i (= the incoming gradient of layer i - 1)
    std::vector<float*> dP, dQ, dWe;   // per EA layer
    // ... and scratch shared by all layers
    float* dh;
    EaScratch eas;               // dS, reduction workspace (dP / dQ / dWe are taken from the per-layer buffers)
    TagScratch tags;
    size_t bytes;
};
static bool is_ea(int i) { return (i & 1) == 0; }
// The forward pass stamps its workspace (a rider of its first launch): TRAIN when it saved what a backward pass reads
// (need_backward), INFER otherwise.  pfn_mpn_backward hands the word to its weight-gradient launch, which writes every
// parameter gradient as NaN unless it reads TRAIN -- a caller that ran the forward with need_backward = 0 on a training-sized
// workspace (nothing on the host can tell) gets NaN gradients instead of plausible garbage.  Device-side, no host sync.
enum { WS_STAMP_INFER = 0x1f0e4e00, WS_STAMP_TRAIN = 0x7a11e7a1 };

// packed-weight plan of the whole network (identical walk in forward, which fills it, and backward, which reads it)
struct ModelPack {
    std::vector<EaPack> ea;
    std::vector<TagPack> tag;
    const float *wa_t, *wb_t, *wb_d;
};
static void plan_pack(Packer& pk, int f0, int fe, int fo, int h, int L, int K, const float* const* params,
                      ModelPack& mp) {
    const int nlayers = 2 * L - 1;
    mp.ea.assign(nlayers, EaPack{});
    mp.tag.assign(nlayers, TagPack{});
    int pi = 0;
    for (int i = 0; i < nlayers; ++i) {
        if (is_ea(i)) {
            const int fi = i == 0 ? f0 : h, fo_ = (i + 1 == nlayers) ? fo : h;
            mp.ea[i] = ea_pack(pk, fi, fe, h, fo_, params ? params[pi] : nullptr, params ? params[pi + 2] : nullptr);
            pi += 4;
        } else {
            const float* none[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
            mp.tag[i] = tag_pack(pk, h, h, K, params ? params + pi : none);
            pi += K + 2;
        }
    }
    const float* wa = params ? params[pi] : nullptr;
    const float* wb = params ? params[pi + 2] : nullptr;
    mp.wa_t = pk.add(wa, f0, 1, 0, 0, f0, h, ld_of(h));      // maskf (N x F0) -> me_h
    mp.wb_t = pk.add(wb, h, 1, 0, 0, h, f0, ld_of(f0));      // me_h (N x H) -> x0
    mp.wb_d = pk.add(wb, h, 0, 0, 0, f0, h, ld_of(h));       // g (N x F0) -> dh
}

static int make_layout(const pfn_mpn_config& c, int64_t n, int64_t e, void* ws, Layout& lo) {
    PFN_CHECK_ARG(c.n_gnn_layers >= 2, "n_gnn_layers must be >= 2 (L == 1 is shape-broken in the reference)");
    PFN_CHECK_ARG(c.K >= 0 && c.K <= 7, "K must be in [0, 7]");
    PFN_CHECK_ARG(c.nfeature_dim > 0 && c.efeature_dim > 0 && c.efeature_dim <= 6 && c.output_dim > 0 && c.hidden_dim > 0,
                  "bad feature dimensions");
    PFN_CHECK_ARG(c.dropout_rate >= 0.f && c.dropout_rate < 1.f, "dropout_rate must be in [0, 1)");
    lo.n = (int)n; lo.e = (int)e;
    lo.f0 = c.nfeature_dim; lo.fe = c.efeature_dim; lo.fo = c.output_dim; lo.h = c.hidden_dim;
    lo.L = c.n_gnn_layers; lo.K = c.K;
    lo.ld = ld_of(lo.h); lo.ld0 = ld_of(lo.f0); lo.ldo = ld_of(lo.fo);
    lo.nlayers = 2 * lo.L - 1;   // E T E T ... E
    Carver cv(ws);
    const size_t nld = (size_t)n * lo.ld;
    {
        Packer measure(nullptr);
        ModelPack mp;
        plan_pack(measure, lo.f0, lo.fe, lo.fo, lo.h, lo.L, lo.K, nullptr, mp);
        lo.packed_floats = measure.off;
    }
    lo.stamp = cv.take<int>(4);
    lo.mask_counts = cv.take<int>(2048);
    lo.packed = cv.take<float>(lo.packed_floats);
    lo.maskf = cv.take<float>((size_t)n * lo.ld0);
    lo.ea_in = cv.take<float>((size_t)4 * e + 4);
    lo.ea_out = cv.take<float>((size_t)4 * e + 4);
    lo.me_h = cv.take<float>(nld);
    lo.x0 = cv.take<float>((size_t)n * lo.ld0);
    lo.y.assign(lo.nlayers, nullptr);
    lo.ea.assign(lo.nlayers, EaSaved{nullptr, nullptr, nullptr});
    lo.xk.assign(lo.nlayers, nullptr);
    for (int i = 0; i < lo.nlayers; ++i) {
        if (i + 1 < lo.nlayers) lo.y[i] = cv.take<float>(nld);
        if (is_ea(i)) {
            lo.ea[i].P = cv.take<float>(nld);
            lo.ea[i].Q = cv.take<float>(nld);
            lo.ea[i].S = cv.take<float>(nld);
        } else {
            lo.xk[i] = cv.take<float>(nld * std::max(1, lo.K));
        }
    }
    lo.gin.assign(lo.nlayers, nullptr);
    lo.dP.assign(lo.nlayers, nullptr);
    lo.dQ.assign(lo.nlayers, nullptr);
    lo.dWe.assign(lo.nlayers, nullptr);
    lo.relu_mask.assign(lo.nlayers, nullptr);
    lo.dh = nullptr;
    lo.eas = EaScratch{nullptr, nullptr, nullptr, nullptr, ReduceWs{nullptr, 0}};
    lo.tags = TagScratch{nullptr, nullptr, nullptr, ReduceWs{nullptr, 0}};
    if (!c.need_backward) {   // inference: none of the backward pass's buffers exist (they were ~2/3 of the footprint)
        lo.bytes = cv.off;
        return PFN_OK;
    }
    for (int i = 0; i < lo.nlayers; ++i) {
        lo.gin[i] = cv.take<float>(i == 0 ? (size_t)n * lo.ld0 : nld);
        if (is_ea(i)) {
            lo.dP[i] = cv.take<float>(nld);
            lo.dQ[i] = cv.take<float>(nld);
            lo.dWe[i] = cv.take<float>((size_t)1025 * lo.fe * lo.ld);
            lo.relu_mask[i] = cv.take<unsigned>(mask_dwords((size_t)n, (size_t)e, lo.ld));
        }
    }
    lo.dh = cv.take<float>(nld);
    lo.eas.dS = cv.take<float>(nld);
    lo.eas.dP = lo.eas.dQ = lo.eas.dWe = nullptr;
    lo.eas.gea_tmp = cv.take<float>((size_t)2 * e * lo.fe);   // (edge-attribute gradients: one slot per edge copy, edge.hip)
    lo.tags.G = cv.take<float>(nld * (lo.K + 1));
    lo.tags.z0 = cv.take<float>(nld);
    lo.tags.z1 = cv.take<float>(nld);
    const int maxf = std::max(std::max(lo.h, lo.f0), lo.fo);
    const size_t red = reduce_ws_floats(n, maxf, maxf, 8);
    lo.eas.red.partial = cv.take<float>(red);
    lo.eas.red.floats = red;
    lo.tags.red = lo.eas.red;
    lo.bytes = cv.off;
    return PFN_OK;
}

// Does layer i's forward edge walk save its ReLU masks?  When a backward pass was announced, Fe = 2, the layer's forward walk is
// the generic one (the graph-resident forward does not write masks), and the backward pass will run the generic walks too (the
// graph-resident backward keeps its tiles in LDS and recomputes: masks bought nothing there).  Forward and backward evaluate
// the same predicate.
static bool ea_saves_mask(const pfn_mpn_config& c, const Layout& lo, int seg, bool fused_front, int i) {
    const bool generic_fwd = !ea_seg_fit(seg, lo.n, lo.fe, lo.ld, false) || (fused_front && i == 0);
    return c.need_backward != 0 && lo.fe == 2 && generic_fwd && !ea_seg_fit(seg, lo.n, lo.fe, lo.ld, true);
}

// Layer 0 behind the 4-wide front: are its P | Q rows left unwritten (the edge walk forms them from the 16-byte x0 rows with
// the front's own fma chains, edge.hip FLY)?  Yes when nothing later reads them from memory: inference, or a training pass whose
// backward walks read the saved ReLU masks.  (A backward pass that was asked for edge-attribute gradients and the gate export
// write the rows then, launch_front_pq.)  Forward, backward and the export evaluate the same predicate.
static bool first_layer_fly(const pfn_mpn_config& c, const Layout& lo, int seg, bool fused_front) {
    static const bool off = diag_env("PFN_NO_L0_FLY") != nullptr;   // A/B switch: the front writes P | Q, the walk gathers them
    return !off && fused_front && lo.nlayers > 1 && lo.fe == 2 && lo.f0 == 4 && !front_latency_regime(lo.h, lo.n) &&
           (c.need_backward == 0 || ea_saves_mask(c, lo, seg, fused_front, 0));
}

// Training beyond the latency regime with layer 0 on the fly: mask_embd's hidden layer is not stored either -- the backward front
// recomputes it from the 16-byte mask rows and forms mask_embd's weight gradients itself (front.hip front_bwd_wg_kernel), so
// neither me_h nor dh touches memory and two N x H pairs leave the weight-gradient launch.  (The gate export writes me_h first.)
static bool front_recomputes_meh(const pfn_mpn_config& c, const Layout& lo, int seg, bool fused_front) {
    static const bool off = diag_env("PFN_FRONT_STORE_MEH") != nullptr;   // A/B switch: me_h stored, the (dY, X) pairs in gemm_tn
    return !off && c.need_backward != 0 && first_layer_fly(c, lo, seg, fused_front) &&
           front_bwd_wg_scratch_floats(lo.n, lo.h) <= (size_t)lo.n * lo.ld;   // (the partial sums live in the unused me_h buffer)
}

// batches of small graphs: the front AND layer 0's edge stage in one graph-resident launch (ea_seg.hip front_seg_fwd_kernel)
// (front_seg_fwd_kernel writes no ReLU masks: never where the backward pass of layer 0 would read them -- today the two fit
//  predicates exclude that by a grid bound only)
static bool uses_seg_front(const pfn_mpn_config& c, const Layout& lo, int seg) {
    const bool fused_front = front_fused_ok(lo.f0, lo.h);
    return fused_front && ea_seg_fit(seg, lo.n, lo.fe, lo.ld, false) && !first_layer_fly(c, lo, seg, fused_front) && lo.nlayers > 1 &&
           front_seg_fit(seg, lo.n, lo.h, lo.fe) && !(c.need_backward && lo.fe == 2 && !ea_seg_fit(seg, lo.n, lo.fe, lo.ld, true));
}

static int model_forward(const pfn_mpn_config& c, const GraphView& g, const Layout& lo, const float* const* params,
                         const float* x, const void* pred_mask, int mask_dtype, const float* edge_attr, float* out,
                         uint64_t* rng, int seg, hipStream_t s) {
    const bool drop = c.training && c.dropout_rate > 0.f;
    PFN_CHECK_ARG(!drop || rng != nullptr, "training with dropout needs rng_state");
    const int nparams = pfn_mpn_num_params(&c);
    const float* const* me = params + (nparams - 4);   // Wa, ba, Wb, bb
    // every weight -> its packed LDS images (both orientations; backward reuses them)
    Packer pk(lo.packed);
    ModelPack mp;
    plan_pack(pk, lo.f0, lo.fe, lo.fo, lo.h, lo.L, lo.K, params, mp);
    // batches of small graphs (ea_seg.hip): the edge attributes go to CSR slot order once, riding in the pack launch
    SlotEa se;
    const bool seg_ea = ea_seg_fit(seg, lo.n, lo.fe, lo.ld, false);
    if (seg_ea) {
        se.rowptr_in = g.rowptr_in; se.in_eid = g.in_eid; se.out_eid = g.out_eid; se.ea = edge_attr;
        se.ea_in = lo.ea_in; se.ea_out = lo.ea_out; se.n = g.n; se.e_stored = g.e_stored;
    }
    // mask_embd(mask) + x   (networks/MPN.py:533,:537)
    const bool fused_front = front_fused_ok(lo.f0, lo.h);
    const bool l0_fly = first_layer_fly(c, lo, seg, fused_front);
    const int ws_stamp = c.need_backward ? WS_STAMP_TRAIN : WS_STAMP_INFER;
    const bool seg_front = uses_seg_front(c, lo, seg);
    if (fused_front) {
        // ONE launch: the weight re-layout (which also advances the dropout stream for this forward) next to the front --
        // pred_mask.float(), mask_embd, the residual add and the first EdgeAggregation's P | Q (front.hip)
        FrontFwdArgs f;
        f.n = lo.n; f.h = lo.h; f.ldw1 = 2 * lo.f0 + lo.fe; f.mask_dtype = mask_dtype;
        f.x = x; f.mask = pred_mask;
        f.wa = me[0]; f.ba = me[1]; f.wb = me[2]; f.bb = me[3]; f.w1 = params[0]; f.b1 = params[1];
        f.maskf = lo.maskf; f.me_h = (c.need_backward && !front_recomputes_meh(c, lo, seg, fused_front)) ? lo.me_h : nullptr; f.x0 = lo.x0;
        f.P = l0_fly ? nullptr : lo.ea[0].P;
        f.Q = l0_fly ? nullptr : lo.ea[0].Q;
        f.mask_counts = (seg_front && c.need_backward) ? lo.mask_counts : nullptr;   // (read by pfn_mpn_backward_masked_l2)
        if (seg_front)
            PFN_TRY(launch_front_seg_fwd(g, f, pk.jobs.data(), (int)pk.jobs.size(), drop ? rng : nullptr, &se, lo.stamp, ws_stamp, edge_attr,
                                         lo.ea[0].S, seg, s));
        else
            PFN_TRY(launch_front_fwd_pack(f, pk.jobs.data(), (int)pk.jobs.size(), drop ? rng : nullptr, s, seg_ea ? &se : nullptr, lo.stamp,
                                          ws_stamp));
    } else {
        // ... the pack launch also advances the dropout stream for this forward and converts pred_mask to float32
        PFN_TRY(pk.flush(s, drop ? rng : nullptr, pred_mask, mask_dtype, lo.maskf, (int64_t)lo.n * lo.ld0, seg_ea ? &se : nullptr, lo.stamp,
                         ws_stamp));
        {
            GemmArgs a = gemm_defaults(lo.n, lo.h, lo.ld);
            a.C[0] = lo.me_h;
            a.nterm = 1;
            a.term[0] = term(lo.maskf, lo.ld0, lo.f0, mp.wa_t, 0);
            a.bias = me[1];
            a.act = ACT_RELU;
            PFN_TRY(launch_gemm_nt(a, s));
        }
        {
            GemmArgs a = gemm_defaults(lo.n, lo.f0, lo.ld0);
            a.C[0] = lo.x0;
            a.nterm = 1;
            a.term[0] = term(lo.me_h, lo.ld, lo.h, mp.wb_t, 0);
            a.bias = me[3];
            a.resid = x;
            a.ldr = lo.ld0;
            PFN_TRY(launch_gemm_nt(a, s));
        }
    }
    const float* cur = lo.x0;
    int ldc = lo.ld0, fcur = lo.f0, pi = 0;
    bool hops_fused = false;
    for (int i = 0; i < lo.nlayers; ++i) {
        const bool last = i + 1 == lo.nlayers;
        Act act;
        if (!last) {
            act.act = drop ? ACT_DROPOUT_RELU : ACT_RELU;   // dropout then ReLU (:546-547)
            act.p = c.dropout_rate;
            act.rng = rng;
            act.stream = (uint32_t)i;
        }
        float* y = last ? out : lo.y[i];
        const int ldy = last ? lo.ldo : lo.ld;
        // an EdgeAggregation output that feeds a TAGConv on the big-graph hop kernel is written chunk-major (layer_out_cm)
        const int big_cm = tag_input_cm(seg, lo.ld, lo.n, g.e_stored, lo.K) ? lo.n : 0;
        if (is_ea(i) && !last) act.out_cm = big_cm;
        if (is_ea(i)) {
            const int fo = last ? lo.fo : lo.h;
            // batches of small graphs: this layer's second Linear also runs the K hops of the TAGConv behind it (seg_lin_hops.hip)
            hops_fused = !last && !big_cm && seg_lin_hops_fit(seg, lo.n, lo.ld, lo.h, lo.h, lo.K, 1);
            PFN_TRY(ea_forward(g, fcur, lo.fe, lo.h, fo, cur, ldc, edge_attr, params[pi], params[pi + 1], params[pi + 2],
                               params[pi + 3], mp.ea[i], y, ldy, act, lo.ea[i], s, fused_front && i == 0, seg, seg_ea ? lo.ea_in : nullptr,
                               ea_saves_mask(c, lo, seg, fused_front, i) ? lo.relu_mask[i] : nullptr, l0_fly && i == 0,
                               hops_fused ? lo.xk[i + 1] : nullptr, lo.K, seg_front && i == 0));
            pi += 4;
            fcur = fo;
        } else {
            PFN_TRY(tag_forward(g, lo.h, lo.h, lo.K, cur, ldc, mp.tag[i], params[pi + lo.K + 1], y, ldy, act, lo.xk[i], s, seg, big_cm,
                                hops_fused));
            hops_fused = false;
            pi += lo.K + 2;
        }
        cur = y;
        ldc = ldy;
    }
    return PFN_OK;
}

static int model_backward(const pfn_mpn_config& c, const GraphView& g, const Layout& lo, const float* const* params,
                          float* const* grads, const float* x, const float* edge_attr, const float* gout, float* gx,
                          float* gea, int seg, hipStream_t s, const MseTail* mse = nullptr) {
    (void)x;
    const bool drop = c.training && c.dropout_rate > 0.f;
    const float gscale = drop ? 1.f / (1.f - c.dropout_rate) : 1.f;
    const int nparams = pfn_mpn_num_params(&c);
    Packer pk(lo.packed);                      // same walk as forward: addresses only, the images are already filled
    ModelPack mp;
    plan_pack(pk, lo.f0, lo.fe, lo.fo, lo.h, lo.L, lo.K, params, mp);
    std::vector<int> poff(lo.nlayers);
    int pi = 0;
    for (int i = 0; i < lo.nlayers; ++i) {
        poff[i] = pi;
        pi += is_ea(i) ? 4 : lo.K + 2;
    }
    if (gea) PFN_CHECK_HIP(hipMemsetAsync(gea, 0, (size_t)lo.e * lo.fe * sizeof(float), s));
    PairList pairs;                            // every weight-gradient pair of the network, launched once at the end
    const bool fused_front = front_fused_ok(lo.f0, lo.h);
    const float* gcur = gout;
    int ldg = lo.ldo;
    bool hops_fused = false;
    for (int i = lo.nlayers - 1; i >= 0; --i) {
        const bool last = i + 1 == lo.nlayers;
        const float* inp = i == 0 ? lo.x0 : lo.y[i - 1];
        const int ldi = i == 0 ? lo.ld0 : lo.ld;
        Gate gate;
        if (i > 0) {
            gate.y = inp;
            gate.ld = ldi;
            gate.scale = gscale;
        }
        float* gnext = lo.gin[i];
        const int p0 = poff[i];
        const int big_cm = tag_input_cm(seg, lo.ld, lo.n, g.e_stored, lo.K) ? lo.n : 0;
        if (!is_ea(i)) gate.cm = big_cm;     // a TAGConv's input is the EdgeAggregation output before it (model_forward)
        // ... and the gradient an EdgeAggregation hands DOWN to a TAGConv (layers 2, 4, ...: their input is a TAGConv's output) is
        // written chunk-major too: the TAGConv's backward hops, its GEMM and the weight-gradient pairs read it through the flags
        const int gx_cm = (is_ea(i) && i >= 2) ? big_cm : 0, gout_cm = !is_ea(i) ? big_cm : 0;
        if (is_ea(i)) {
            const int fi = i == 0 ? lo.f0 : lo.h, fo = last ? lo.fo : lo.h;
            EaScratch sc = lo.eas;
            sc.dP = lo.dP[i];
            sc.dQ = lo.dQ[i];
            sc.dWe = lo.dWe[i];
            // the forward pass left layer 0's P | Q unwritten and the edge-attribute gradient recomputes the pre-activations
            if (i == 0 && gea && first_layer_fly(c, lo, seg, fused_front))
                PFN_TRY(launch_front_pq(lo.n, lo.h, 2 * lo.f0 + lo.fe, lo.x0, params[0], params[1], lo.ea[0].P, lo.ea[0].Q, s));
            // (layer 0 with the fused front: its input gradient is formed together with mask_embd's, below)
            // batches of small graphs: the dx Linear also runs the backward hops of the TAGConv in front (seg_lin_hops.hip)
            hops_fused = i >= 2 && !gx_cm && !gate.cm && seg_lin_hops_fit(seg, lo.n, lo.ld, lo.h, lo.h, lo.K, 2);
            PFN_TRY(ea_backward(g, fi, lo.fe, lo.h, fo, inp, ldi, edge_attr, params[p0], params[p0 + 2], mp.ea[i], gcur, ldg, gate,
                                (fused_front && i == 0) ? nullptr : gnext, ldi, grads[p0], grads[p0 + 1], grads[p0 + 2],
                                grads[p0 + 3], gea, lo.ea[i], sc, s, &pairs, seg, lo.ea_in, lo.ea_out,
                                ea_saves_mask(c, lo, seg, fused_front, i) ? lo.relu_mask[i] : nullptr, gx_cm,
                                hops_fused ? lo.tags.G : nullptr, lo.K, last ? mse : nullptr));
        } else {
            PFN_TRY(tag_backward(g, lo.h, lo.h, lo.K, inp, ldi, mp.tag[i], gcur, ldg, gate, gnext, ldi, grads + p0,
                                 grads[p0 + lo.K + 1], lo.xk[i], lo.tags, s, &pairs, seg, big_cm, gout_cm, hops_fused));
            hops_fused = false;
        }
        gcur = gnext;
        ldg = ldi;
    }
    // mask_embd backward: x0 = me_h Wb^T + bb + x ; me_h = relu(maskf Wa^T + ba)
    float* const* gme = grads + (nparams - 4);
    const bool meh_rc = front_recomputes_meh(c, lo, seg, fused_front);
    if (meh_rc) {        // ... and mask_embd's weight gradients too, me_h recomputed (front_bwd_wg_kernel; partials in the me_h buffer)
        PFN_TRY(launch_front_bwd_wg(lo.n, lo.h, 2 * lo.f0 + lo.fe, lo.dP[0], lo.dQ[0], lo.maskf, params[0], params[nparams - 4],
                                    params[nparams - 3], params[nparams - 2], lo.gin[0], lo.me_h, gme[0], gme[1], gme[2], gme[3], s, lo.stamp,
                                    WS_STAMP_TRAIN));
    } else if (fused_front) {   // g0 = dP0 W1i + dQ0 W1j and dh = (g0 Wb) [me_h > 0] in one launch (front.hip)
        PFN_TRY(launch_front_bwd(lo.n, lo.h, 2 * lo.f0 + lo.fe, lo.dP[0], lo.dQ[0], lo.me_h, params[0], params[nparams - 2],
                                 lo.gin[0], lo.dh, s));
    } else {
        GemmArgs a = gemm_defaults(lo.n, lo.h, lo.ld);
        a.C[0] = lo.dh;
        a.nterm = 1;
        a.term[0] = term(gcur, lo.ld0, lo.f0, mp.wb_d, 0);
        a.gate = lo.me_h;
        a.ldg = lo.ld;
        PFN_TRY(launch_gemm_nt(a, s));
    }
    if (!meh_rc) {
        pairs.pairs.push_back(tn_pair(gcur, lo.ld0, lo.f0, lo.me_h, lo.ld, lo.h, gme[2], lo.h, 0, gme[3], nullptr));     // dWb, dbb
        pairs.pairs.push_back(tn_pair(lo.dh, lo.ld, lo.h, lo.maskf, lo.ld0, lo.f0, gme[0], lo.f0, 0, gme[1], nullptr));  // dWa, dba
    }
    // the dWe partial reductions ride in the weight-gradient launch (independent work, one launch floor less)
    if ((int)pairs.dwe.size() <= DWE_MAX_JOBS && lo.n > 0 && !pairs.pairs.empty()) {
        DweRide ride;
        ride.njobs = (int)pairs.dwe.size();
        for (int j = 0; j < ride.njobs; ++j) ride.jobs.job[j] = pairs.dwe[j];
        ride.fe = lo.fe; ride.ld = lo.ld; ride.h = lo.h;
        PFN_TRY(launch_weight_grads(pairs.pairs.data(), (int)pairs.pairs.size(), lo.n, lo.eas.red, s, &ride, lo.stamp, WS_STAMP_TRAIN));
    } else {
        PFN_TRY(launch_dwe_reduce_multi(pairs.dwe.data(), (int)pairs.dwe.size(), lo.fe, lo.ld, lo.h, s, lo.stamp, WS_STAMP_TRAIN));
        PFN_TRY(launch_weight_grads(pairs.pairs.data(), (int)pairs.pairs.size(), lo.n, lo.eas.red, s, nullptr, lo.stamp, WS_STAMP_TRAIN));
    }
    if (gx) PFN_CHECK_HIP(hipMemcpyAsync(gx, gcur, (size_t)lo.n * lo.ld0 * sizeof(float), hipMemcpyDeviceToDevice, s));
    return PFN_OK;
}

// ---------------------------------------------------------------------------------------- utilities
// ---- Masked_L2_loss (utils/custom_loss_functions.py:10-46): two masked means of (out - y)^2.  Kernel 1 reduces
// (sum, count) of both sets with an ordered last-arriver combine and writes the loss and the totals; kernel 2 turns the
// totals into the two gradient scales.
struct MaskedL2Ws {
    float s1[256], s0[256];
    int c1[256], c0[256];
    float tot_s1, tot_s0;
    int tot_c1, tot_c0;
    int pad_[3];
    int counter;   // byte 4124
};
__device__ __forceinline__ float mask_value(const void* m, int dtype, int64_t i) {
    return dtype == 0 ? (float)static_cast<const int64_t*>(m)[i] : static_cast<const float*>(m)[i];
}
__global__ __launch_bounds__(256) void masked_l2_reduce_kernel(const float* __restrict__ o, const float* __restrict__ y,
                                                               const void* __restrict__ mask, int mask_dtype, int64_t n,
                                                               int regularize, float regcoeff, MaskedL2Ws* __restrict__ w,
                                                               float* __restrict__ loss) {
    __shared__ float rs1[256], rs0[256];
    __shared__ int rc1[256], rc0[256];
    __shared__ int s_last;
    float a1 = 0.f, a0 = 0.f;
    int k1 = 0, k0 = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float d = o[i] - y[i], m = mask_value(mask, mask_dtype, i);
        if (m != 0.f) { a1 = fmaf(d, d, a1); ++k1; }                 // mask.type(bool)
        if (1.f - m != 0.f) { a0 = fmaf(d, d, a0); ++k0; }           // (1 - mask).type(bool)
    }
    const int t = threadIdx.x;
    rs1[t] = a1; rs0[t] = a0; rc1[t] = k1; rc0[t] = k0;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (t < off) { rs1[t] += rs1[t + off]; rs0[t] += rs0[t + off]; rc1[t] += rc1[t + off]; rc0[t] += rc0[t + off]; }
        __syncthreads();
    }
    if (t == 0) {
        w->s1[blockIdx.x] = rs1[0]; w->s0[blockIdx.x] = rs0[0]; w->c1[blockIdx.x] = rc1[0]; w->c0[blockIdx.x] = rc0[0];
        __threadfence();
        const int tk = __hip_atomic_fetch_add(&w->counter, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        s_last = tk == (int)gridDim.x - 1;
    }
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    const bool in = t < (int)gridDim.x;
    rs1[t] = in ? __hip_atomic_load(&w->s1[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.f;
    rs0[t] = in ? __hip_atomic_load(&w->s0[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.f;
    rc1[t] = in ? __hip_atomic_load(&w->c1[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
    rc0[t] = in ? __hip_atomic_load(&w->c0[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (t < off) { rs1[t] += rs1[t + off]; rs0[t] += rs0[t + off]; rc1[t] += rc1[t + off]; rc0[t] += rc0[t + off]; }
        __syncthreads();
    }
    if (t == 0) {
        w->tot_s1 = rs1[0]; w->tot_s0 = rs0[0]; w->tot_c1 = rc1[0]; w->tot_c0 = rc0[0];
        float l = rs1[0] / (float)rc1[0];                            // 0/0 = NaN: torch's mean of an empty selection
        if (regularize) l += regcoeff * (rs0[0] / (float)rc0[0]);
        loss[0] = l;
        w->counter = 0;
    }
}
__global__ __launch_bounds__(256) void masked_l2_grad_kernel(const float* __restrict__ o, const float* __restrict__ y,
                                                             const void* __restrict__ mask, int mask_dtype, int64_t n,
                                                             int regularize, float regcoeff, const MaskedL2Ws* __restrict__ w,
                                                             float* __restrict__ grad) {
    const float g1 = 2.f / (float)w->tot_c1, g0 = regularize ? 2.f * regcoeff / (float)w->tot_c0 : 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float d = o[i] - y[i], m = mask_value(mask, mask_dtype, i);
        float g = 0.f;
        if (m != 0.f) g += g1 * d;
        if (regularize && 1.f - m != 0.f) g += g0 * d;
        grad[i] = g;
    }
}

// One launch: every block reduces its slice to a partial and takes a ticket; the last arriver sums the partials in
// block order (not arrival order: deterministic) and re-arms the counter for the next call.
__global__ __launch_bounds__(256) void mse_kernel(const float* __restrict__ o, const float* __restrict__ y, int64_t n,
                                                  float inv_n, float* __restrict__ grad, float* __restrict__ partial,
                                                  int* __restrict__ counter, float* __restrict__ loss) {
    __shared__ float red[256];
    __shared__ int s_last;
    float acc = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float d = o[i] - y[i];
        acc = fmaf(d, d, acc);
        if (grad) grad[i] = 2.f * d * inv_n;
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        // hand-off to the last arriver without __threadfence() (an L2 write-back + L1 invalidate, ~3.5 us each on this
        // multi-XCD part, and the kernel had two): the partial is stored WRITE-THROUGH (agent-scope atomic store = sc1), the
        // store is drained, then the ticket is taken; the last arriver reads the partials with agent-scope atomic loads (sc1:
        // served by L2 / memory, never by its L1).  This is the "sc1 payload -> asm vmcnt(0) -> flag, sc1 loads on the consumer"
        // form MI355X_MICROARCH.md lists as valid ON gfx950 (vmcnt covers stores there; the language memory model does not
        // promise it) -- hence the target guard below, and tests/test_gpu_parity.py::test_mse_loss_handoff_stress.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "mse_kernel's last-arriver hand-off relies on gfx950 semantics (sc1 write-through stores drained by s_waitcnt vmcnt(0))"
#endif
        __hip_atomic_store(partial + blockIdx.x, red[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const int t = __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = t == (int)gridDim.x - 1;
    }
    __syncthreads();
    if (!s_last) return;
    red[threadIdx.x] = threadIdx.x < gridDim.x ? __hip_atomic_load(partial + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.f;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        loss[0] = red[0] * inv_n;
        *counter = 0;
    }
}

// one block per CU at most: every block ends with one atomic on the arrival counter (64 blocks of 1024 threads were tried for
// that reason: 8.9 against 8.4 us)
static int adamw_blocks(int64_t count) { return (int)std::max<int64_t>(1, std::min<int64_t>((count + 255) / 256, 256)); }

// hp (optional): device {lr, beta1, beta2, eps, weight_decay} read instead of the by-value arguments, so that a captured
// launch follows a learning-rate schedule without being captured again
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                    float* __restrict__ m, float* __restrict__ v, int64_t n, float lr,
                                                    float b1, float b2, float eps, float wd, int64_t* step,
                                                    const float* __restrict__ hp, const float* __restrict__ guard) {
    // guard (optional): a device scalar -- normally the step's loss; not finite = the batch was flagged bad on the device
    // (pfn_graph_poison_if_bad): every block sees the same value and leaves, nothing is updated, the step is not counted
    if (guard) {
        const float gv = *guard;
        if (!(fabsf(gv) <= 3.402823466e+38f)) {
            if (blockIdx.x == 0 && threadIdx.x == 0) step[2] += 1;   // skipped updates: visible to the host loop (train_epoch warns)
            return;
        }
    }
    // 16 bytes per lane and array when the four flat buffers allow it (they are whole allocations: 256-byte aligned); the
    // update is a chain of dependent loads per element otherwise (10 us for 355 k parameters, 2x its memory time).  A thread's
    // FIRST four-element group is requested before the hyper-parameters are even read: their load -> powf chain and this load
    // were two serial round trips.
    const bool vec = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
                       reinterpret_cast<uintptr_t>(v)) & 15) == 0;
    const int64_t n4 = vec ? n >> 2 : 0;
    const int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (int64_t)gridDim.x * blockDim.x;
    float4 p4 = make_float4(0.f, 0.f, 0.f, 0.f), m4 = p4, v4 = p4, g4 = p4;
    if (i0 < n4) {
        p4 = reinterpret_cast<float4*>(p)[i0];
        m4 = reinterpret_cast<float4*>(m)[i0];
        v4 = reinterpret_cast<float4*>(v)[i0];
        g4 = reinterpret_cast<const float4*>(g)[i0];
    }
    if (hp) {
        lr = hp[0]; b1 = hp[1]; b2 = hp[2]; eps = hp[3]; wd = hp[4];
    }
    // step[0] = completed steps, step[1] = arrival counter: the last block to ARRIVE bumps the step and re-arms the counter, so the
    // whole update is ONE launch and stays hipGraph-replayable.  The ticket is taken as soon as every wave of the block has READ
    // step[0] (the barrier below waits for that scalar load only, not for the element loads in flight) -- the bump has to come after
    // all blocks' reads, not after their updates; taken behind the update, the launch ended with stores drained -> atomic round
    // trip -> store, ~1 us of nothing.
    const int64_t step_now = step[0];
    const float t = (float)(step_now + 1);
    asm volatile("s_barrier" ::"s"((int)step_now) : "memory");   // (the operand: this wave's read of step[0] has returned)
    if (threadIdx.x == 0) {
        const unsigned long long prev = atomicAdd(reinterpret_cast<unsigned long long*>(step + 1), 1ull);
        if (prev == (unsigned long long)gridDim.x - 1) {
            step[1] = 0;
            step[0] = step_now + 1;
        }
    }
    const float bc1 = 1.f - powf(b1, t), bc2 = 1.f - powf(b2, t);
    const float step_size = lr / bc1, inv_sqrt_bc2 = 1.f / sqrtf(bc2);
    auto upd = [&](float& pi, float gi, float& mi, float& vi) {
        pi *= (1.f - lr * wd);                           // decoupled weight decay
        mi = b1 * mi + (1.f - b1) * gi;
        vi = b2 * vi + (1.f - b2) * gi * gi;
        const float denom = sqrtf(vi) * inv_sqrt_bc2 + eps;
        pi -= step_size * (mi / denom);
    };
    for (int64_t i = i0; i < n4; i += stride) {
        if (i != i0) {
            p4 = reinterpret_cast<float4*>(p)[i];
            m4 = reinterpret_cast<float4*>(m)[i];
            v4 = reinterpret_cast<float4*>(v)[i];
            g4 = reinterpret_cast<const float4*>(g)[i];
        }
        upd(p4.x, g4.x, m4.x, v4.x);
        upd(p4.y, g4.y, m4.y, v4.y);
        upd(p4.z, g4.z, m4.z, v4.z);
        upd(p4.w, g4.w, m4.w, v4.w);
        st4_wt(p + 4 * i, p4);      // (write-through: the next step's first kernel does not wait for 4 MB of dirty lines)
        st4_wt(m + 4 * i, m4);
        st4_wt(v + 4 * i, v4);
    }
    for (int64_t i = 4 * n4 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float pi = p[i], mi = m[i], vi = v[i];
        upd(pi, g[i], mi, vi);
        p[i] = pi;
        m[i] = mi;
        v[i] = vi;
    }
}

// The keep mask (1 = kept, 0 = dropped) the dropout epilogue of layer `stream` applies for the CURRENT {seed, offset} of
// rng -- the same dropout_uniform4 call, element for element -- so a test can replay a train-mode pass on the CPU oracle.
__global__ __launch_bounds__(256) void dropout_mask_kernel(const uint64_t* __restrict__ rng, uint32_t stream, int64_t rows,
                                                           int ncols, float p, float* __restrict__ out) {
    const int ncg = (ncols + 3) >> 2;
    const DropKey dk = drop_key(rng[0], rng[1], stream);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < rows * ncg; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / ncg;
        const int cg = (int)(i - row * ncg);
        float u[4];
        dropout_uniform4(dk, (uint32_t)row, (uint32_t)cg, u);
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (4 * cg + e < ncols) out[row * ncols + 4 * cg + e] = u[e] >= p ? 1.f : 0.f;
    }
}

// ---- ReLU gate export (verification aid, pfn_mpn_export_gates): the decisions of the three kinds of ReLU a forward pass
// took, as bytes, so that a float64 run of the oracle can be held to the SAME piecewise-linear branch and the gradients compared
// at north_star's tolerance (a pre-activation within the fp32 forward error of zero otherwise flips its gate in one of the two
// runs and moves a weight gradient by 1e-5..2e-4 of its largest entry).
// Edge stage of an EdgeAggregation layer: out[eid][k] = (P[dst][k] + Q[src][k] + sum_f a_e[f] We[k][f] > 0) with EXACTLY the
// expression the walks evaluate (edge.hip edge_sum_chunk / edge_bwd_*_body, ea_seg.hip: add, then one fmaf per attribute, in
// attribute order) on the P | Q the forward saved -- which is also what the saved mask bytes hold where the forward saved them.
__global__ __launch_bounds__(256) void export_edge_gates_kernel(int n, int e_stored, const int* __restrict__ rowptr,
                                                                const int* __restrict__ nbr, const int* __restrict__ eid,
                                                                const float* __restrict__ P, const float* __restrict__ Q,
                                                                const float* __restrict__ ea, const float* __restrict__ w1,
                                                                int ld, int h, int fi, int fe, uint8_t* __restrict__ out) {
    const int ldw = 2 * fi + fe;
    for (int64_t it = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; it < (int64_t)n * h; it += (int64_t)gridDim.x * blockDim.x) {
        const int row = (int)(it / h), k = (int)(it - (int64_t)row * h);
        const float p = P[(size_t)row * ld + k];
        for (int q = rowptr[row]; q < rowptr[row + 1]; ++q) {
            const int id = eid[q], idm = id >= e_stored ? id - e_stored : id;
            float v = p + Q[(size_t)nbr[q] * ld + k];
            for (int f = 0; f < fe; ++f) v = fmaf(ea[(size_t)idm * fe + f], w1[(size_t)k * ldw + 2 * fi + f], v);
            out[(size_t)id * h + k] = v > 0.f ? 1 : 0;
        }
    }
}
// Layer outputs (and mask_embd's hidden layer): out[row][k] = y[row][k] > 0, the test the backward pass applies (GemmArgs::gate).
__global__ __launch_bounds__(256) void export_row_gates_kernel(int64_t n, int h, int ld, const float* __restrict__ y,
                                                               uint8_t* __restrict__ out, int cm) {
    for (int64_t it = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; it < n * h; it += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = it / h, col = it - row * h;
        const float v = cm ? y[((col >> 2) * n + row) * 4 + (col & 3)] : y[row * ld + col];   // (chunk-major: Act::out_cm)
        out[it] = v > 0.f ? 1 : 0;
    }
}

}  // namespace pfn

using namespace pfn;

// =============================================================================================== C ABI
extern "C" {

int pfn_mpn_num_params(const pfn_mpn_config* c) {
    if (!c || c->n_gnn_layers < 2) return -1;
    return c->n_gnn_layers * 4 + (c->n_gnn_layers - 1) * (c->K + 2) + 4;
}

size_t pfn_mpn_workspace_bytes(const pfn_mpn_config* c, int64_t n, int64_t e) {
    if (!c) return 0;
    Layout lo;
    if (make_layout(*c, n, e, nullptr, lo) != PFN_OK) return 0;
    return lo.bytes;
}

static int check_common(const pfn_mpn_config* c, const void* gws, int64_t n, int64_t e, const void* ws) {
    PFN_CHECK_ARG(c != nullptr, "null config");
    PFN_CHECK_ARG(gws != nullptr, "null graph workspace");
    PFN_CHECK_ARG(ws != nullptr, "null model workspace");
    PFN_CHECK_ARG(n >= 0 && e >= 0 && n < (1ll << 30) && e < (1ll << 29), "bad graph size");
    return PFN_OK;
}

// The MSELoss tail (pfn_mpn_backward_mse): the last layer's backward must be the graph-resident launch in its last-layer form,
// and the out rows those of lin_out4_wave_kernel
static bool mse_tail_ok(const pfn_mpn_config& c, const Layout& lo, int seg) {
    static const bool off = diag_env("PFN_NO_MSE_TAIL") != nullptr;   // A/B switch: lin_out4 + mse_kernel + the plain backward
    return !off && c.need_backward != 0 && lo.n > 0 && lo.nlayers > 1 && lo.fe == 2 && lo.fo == 4 && lo.ldo == 4 &&
           lin_out4_ok(lo.h, lo.fo, lo.ldo, lo.n) && back_fused_ok() && ea_seg_fit(seg, lo.n, lo.fe, lo.ld, false) &&
           ea_seg_fit(seg, lo.n, lo.fe, lo.ld, true) && lo.ld / 4 <= 34;
}

int pfn_mpn_forward(const pfn_mpn_config* c, const void* gws, int64_t n, int64_t e, const float* const* params,
                    const float* x, const void* pred_mask, int mask_dtype, const float* edge_attr, float* out, void* ws,
                    size_t ws_bytes, uint64_t* rng, int64_t seg_nodes, void* stream) {
    PFN_TRY(check_common(c, gws, n, e, ws));
    PFN_CHECK_ARG(params && (n == 0 || (x && pred_mask)) && (e == 0 || edge_attr), "pfn_mpn_forward: null tensor");
    Layout lo;
    PFN_TRY(make_layout(*c, n, e, ws, lo));
    if (ws_bytes < lo.bytes) {
        set_error("pfn_mpn_forward: workspace %zu < %zu bytes", ws_bytes, lo.bytes);
        return PFN_ENOSPACE;
    }
    // out == NULL: the output rows are left to pfn_mpn_backward_mse -- only where that entry point is available
    PFN_CHECK_ARG(n == 0 || out || (seg_nodes > 0 && n % seg_nodes == 0 && mse_tail_ok(*c, lo, (int)seg_nodes)),
                  "pfn_mpn_forward: out may be NULL only where pfn_mpn_mse_tail_ok says so");
    GraphView g = graph_view(const_cast<void*>(gws), n, e);
    PFN_CHECK_ARG(c->nfeature_dim % 4 == 0, "nfeature_dim must be a multiple of 4 (the reference asserts 4, networks/MPN.py:528)");
    PFN_CHECK_ARG(seg_nodes >= 0 && (seg_nodes == 0 || n % seg_nodes == 0), "seg_nodes must be 0 or divide n_nodes");
    return model_forward(*c, g, lo, params, x, pred_mask, mask_dtype, edge_attr, out, rng, (int)seg_nodes,
                         static_cast<hipStream_t>(stream));
}

int pfn_mpn_mse_tail_ok(const pfn_mpn_config* c, int64_t n, int64_t e, int64_t seg_nodes) {
    if (!c || n <= 0 || e < 0 || n >= (1ll << 30) || e >= (1ll << 29) || seg_nodes <= 0 || n % seg_nodes != 0) return 0;
    Layout lo;
    if (make_layout(*c, n, e, nullptr, lo) != PFN_OK) return 0;
    // (one answer for both losses: the masked one also needs the front launch that leaves the mask census behind)
    return (mse_tail_ok(*c, lo, (int)seg_nodes) && uses_seg_front(*c, lo, (int)seg_nodes)) ? 1 : 0;
}

static int backward_with_loss_tail(const pfn_mpn_config* c, const void* gws, int64_t n, int64_t e, const float* const* params,
                                   float* const* grads, const float* x, const float* edge_attr, const float* y, float* out, float* loss,
                                   float* grad_out, float* gx, void* ws, size_t ws_bytes, void* loss_ws, size_t loss_ws_bytes,
                                   int64_t seg_nodes, void* stream, bool masked, int regularize, float regcoeff) {
    PFN_TRY(check_common(c, gws, n, e, ws));
    PFN_CHECK_ARG(params && grads && x && y && out && loss && grad_out && loss_ws, "pfn_mpn_backward_mse / _masked_l2: null tensor");
    PFN_CHECK_ARG(c->need_backward != 0, "pfn_mpn_backward_mse / _masked_l2: the forward ran with need_backward = 0");
    PFN_CHECK_ARG(loss_ws_bytes >= (masked ? 8196u : 4100u), "pfn_mpn_backward_mse: loss workspace must hold 4100 bytes (_masked_l2: 8196)");
    Layout lo;
    PFN_TRY(make_layout(*c, n, e, ws, lo));
    if (ws_bytes < lo.bytes) {
        set_error("pfn_mpn_backward_mse: workspace %zu < %zu bytes", ws_bytes, lo.bytes);
        return PFN_ENOSPACE;
    }
    PFN_CHECK_ARG(seg_nodes > 0 && n % seg_nodes == 0, "seg_nodes must divide n_nodes");
    if (!mse_tail_ok(*c, lo, (int)seg_nodes) || (masked && !uses_seg_front(*c, lo, (int)seg_nodes))) {
        set_error("pfn_mpn_backward_mse / _masked_l2: not available for this model / batch (ask pfn_mpn_mse_tail_ok)");
        return PFN_EINVAL;
    }
    GraphView g = graph_view(const_cast<void*>(gws), n, e);
    const int nparams = pfn_mpn_num_params(c);
    MseTail t;
    t.S = lo.ea[lo.nlayers - 1].S;
    t.b2 = params[nparams - 4 - 1];   // the last EdgeAggregation's b2 (its four tensors end in front of mask_embd's)
    t.deg = g.deg;
    t.y = y; t.out = out; t.gout = grad_out;
    t.partial = static_cast<float*>(loss_ws);
    t.counter = reinterpret_cast<int*>(static_cast<char*>(loss_ws) + (masked ? 8192 : 4096));
    t.loss = loss;
    t.inv_n = 1.f / (float)(n * 4);
    if (masked) {
        t.maskf = lo.maskf;
        t.counts = lo.mask_counts;
        t.count_blocks = ea_seg_blocks((int)seg_nodes, lo.n, lo.ld);
        t.regularize = regularize ? 1 : 0;
        t.regcoeff = regcoeff;
    }
    return model_backward(*c, g, lo, params, grads, x, edge_attr, grad_out, gx, nullptr, (int)seg_nodes, static_cast<hipStream_t>(stream), &t);
}

int pfn_mpn_backward_mse(const pfn_mpn_config* c, const void* gws, int64_t n, int64_t e, const float* const* params,
                         float* const* grads, const float* x, const float* edge_attr, const float* y, float* out, float* loss,
                         float* grad_out, float* gx, void* ws, size_t ws_bytes, void* loss_ws, size_t loss_ws_bytes,
                         int64_t seg_nodes, void* stream) {
    return backward_with_loss_tail(c, gws, n, e, params, grads, x, edge_attr, y, out, loss, grad_out, gx, ws, ws_bytes, loss_ws,
                                   loss_ws_bytes, seg_nodes, stream, false, 0, 0.f);
}

int pfn_mpn_backward_masked_l2(const pfn_mpn_config* c, const void* gws, int64_t n, int64_t e, const float* const* params,
                               float* const* grads, const float* x, const float* edge_attr, const float* y, int regularize,
                               float regcoeff, float* out, float* loss, float* grad_out, float* gx, void* ws, size_t ws_bytes,
                               void* loss_ws, size_t loss_ws_bytes, int64_t seg_nodes, void* stream) {
    return backward_with_loss_tail(c, gws, n, e, params, grads, x, edge_attr, y, out, loss, grad_out, gx, ws, ws_bytes, loss_ws,
                                   loss_ws_bytes, seg_nodes, stream, true, regularize, regcoeff);
}

int pfn_mpn_backward(const pfn_mpn_config* c, const void* gws, int64_t n, int64_t e, const float* const* params,
                     float* const* grads, const float* x, const void* pred_mask, int mask_dtype, const float* edge_attr,
                     const float* gout, float* gx, float* gea, void* ws, size_t ws_bytes, int64_t seg_nodes, void* stream) {
    (void)pred_mask; (void)mask_dtype;
    PFN_TRY(check_common(c, gws, n, e, ws));
    PFN_CHECK_ARG(params && grads && (n == 0 || (x && gout)), "pfn_mpn_backward: null tensor");
    PFN_CHECK_ARG(c->need_backward != 0, "pfn_mpn_backward: the forward ran with need_backward = 0 (inference: what only the backward reads was not saved)");
    Layout lo;
    PFN_TRY(make_layout(*c, n, e, ws, lo));
    if (ws_bytes < lo.bytes) {
        set_error("pfn_mpn_backward: workspace %zu < %zu bytes", ws_bytes, lo.bytes);
        return PFN_ENOSPACE;
    }
    GraphView g = graph_view(const_cast<void*>(gws), n, e);
    PFN_CHECK_ARG(seg_nodes >= 0 && (seg_nodes == 0 || n % seg_nodes == 0), "seg_nodes must be 0 or divide n_nodes");
    return model_backward(*c, g, lo, params, grads, x, edge_attr, gout, gx, gea, (int)seg_nodes, static_cast<hipStream_t>(stream));
}

int pfn_mpn_export_gates(const pfn_mpn_config* c, const void* gws, int64_t n, int64_t e, const float* const* params,
                         const float* edge_attr, void* ws, size_t ws_bytes, int64_t seg_nodes, int32_t kind, int32_t layer,
                         uint8_t* out, void* stream) {
    PFN_TRY(check_common(c, gws, n, e, ws));
    PFN_CHECK_ARG(params && out, "pfn_mpn_export_gates: null pointer");
    PFN_CHECK_ARG(c->need_backward != 0, "pfn_mpn_export_gates: the forward ran with need_backward = 0 (mask_embd's hidden layer was not saved)");
    Layout lo;
    PFN_TRY(make_layout(*c, n, e, ws, lo));
    if (ws_bytes < lo.bytes) {
        set_error("pfn_mpn_export_gates: workspace %zu < %zu bytes", ws_bytes, lo.bytes);
        return PFN_ENOSPACE;
    }
    if (n == 0) return PFN_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int64_t items = n * lo.h;
    const int blocks = (int)std::min<int64_t>((items + 255) / 256, 16384);
    if (kind == 0) {
        PFN_CHECK_ARG(layer >= 0 && layer < lo.nlayers && is_ea(layer), "pfn_mpn_export_gates: kind 0 needs an EdgeAggregation layer index");
        PFN_CHECK_ARG(e == 0 || edge_attr, "pfn_mpn_export_gates: null edge_attr");
        GraphView g = graph_view(const_cast<void*>(gws), n, e);
        int pi = 0;
        for (int i = 0; i < layer; ++i) pi += is_ea(i) ? 4 : lo.K + 2;
        const int fi = layer == 0 ? lo.f0 : lo.h;
        if (layer == 0 && first_layer_fly(*c, lo, (int)seg_nodes, front_fused_ok(lo.f0, lo.h)))   // (the forward left them unwritten)
            PFN_TRY(launch_front_pq(lo.n, lo.h, 2 * lo.f0 + lo.fe, lo.x0, params[0], params[1], lo.ea[0].P, lo.ea[0].Q, s));
        export_edge_gates_kernel<<<blocks, 256, 0, s>>>(g.n, g.e_stored, g.rowptr_in, g.in_src, g.in_eid, lo.ea[layer].P, lo.ea[layer].Q,
                                                       edge_attr, params[pi], lo.ld, lo.h, fi, lo.fe, out);
    } else if (kind == 1) {
        PFN_CHECK_ARG(layer >= 0 && layer + 1 < lo.nlayers, "pfn_mpn_export_gates: kind 1 needs a hidden layer index");
        const int cm = is_ea(layer) && tag_input_cm((int)seg_nodes, lo.ld, lo.n, e, lo.K);
        export_row_gates_kernel<<<blocks, 256, 0, s>>>(n, lo.h, lo.ld, lo.y[layer], out, cm);
    } else if (kind == 2) {
        if (front_recomputes_meh(*c, lo, (int)seg_nodes, front_fused_ok(lo.f0, lo.h))) {   // (the forward did not store it)
            const int np = pfn_mpn_num_params(c);
            PFN_TRY(launch_front_meh(lo.n, lo.h, lo.maskf, 1, params[np - 4], params[np - 3], lo.me_h, s));
        }
        export_row_gates_kernel<<<blocks, 256, 0, s>>>(n, lo.h, lo.ld, lo.me_h, out, 0);
    } else {
        set_error("pfn_mpn_export_gates: kind must be 0 (edge stage), 1 (layer output) or 2 (mask_embd hidden)");
        return PFN_EINVAL;
    }
    PFN_CHECK_LAUNCH();
    return PFN_OK;
}

// ------------------------------------------------------------------------------------- single layers
struct EaLayerWs { EaSaved sv; EaScratch sc; float* packed; size_t bytes; };
static EaLayerWs ea_layer_ws(void* ws, int64_t n, int fi, int fe, int h, int fo) {
    Carver cv(ws);
    EaLayerWs w;
    const int ld = ld_of(h);
    const size_t nld = (size_t)n * ld;
    {
        Packer measure(nullptr);
        ea_pack(measure, fi, fe, h, fo, nullptr, nullptr);
        w.packed = cv.take<float>(measure.off);
    }
    w.sv.P = cv.take<float>(nld);
    w.sv.Q = cv.take<float>(nld);
    w.sv.S = cv.take<float>(nld);
    w.sc.dS = cv.take<float>(nld);
    w.sc.dP = cv.take<float>(nld);
    w.sc.dQ = cv.take<float>(nld);
    w.sc.dWe = cv.take<float>((size_t)1025 * fe * ld);
    const int maxf = std::max(std::max(h, fi), fo);
    w.sc.red.floats = reduce_ws_floats(n, maxf, maxf, 3);
    w.sc.red.partial = cv.take<float>(w.sc.red.floats);
    w.bytes = cv.off;
    return w;
}

size_t pfn_edge_aggr_workspace_bytes(int64_t n, int64_t e, int fi, int fe, int h, int fo) {
    (void)e;
    return ea_layer_ws(nullptr, n, fi, fe, h, fo).bytes;
}

int pfn_edge_aggr_forward(const void* gws, int64_t n, int64_t e, int fi, int fe, int h, int fo, const float* x,
                          int64_t ldx, const float* ea, const float* w1, const float* b1, const float* w2,
                          const float* b2, float* out, int64_t ldo, void* ws, size_t ws_bytes, void* stream) {
    PFN_CHECK_ARG(gws && ws && w1 && b1 && w2 && b2, "pfn_edge_aggr_forward: null pointer");
    PFN_CHECK_ARG(ldx == ld_of(fi) && ldo == ld_of(fo), "pfn_edge_aggr_forward: row strides must be pfn_padded_ld(F)");
    PFN_CHECK_ARG(fe >= 1 && fe <= 6, "efeature_dim must be in [1, 6]");
    EaLayerWs w = ea_layer_ws(ws, n, fi, fe, h, fo);
    if (ws_bytes < w.bytes) {
        set_error("pfn_edge_aggr_forward: workspace %zu < %zu bytes", ws_bytes, w.bytes);
        return PFN_ENOSPACE;
    }
    GraphView g = graph_view(const_cast<void*>(gws), n, e);
    hipStream_t s = static_cast<hipStream_t>(stream);
    Packer pk(w.packed);
    const EaPack pw = ea_pack(pk, fi, fe, h, fo, w1, w2);
    PFN_TRY(pk.flush(s));
    return ea_forward(g, fi, fe, h, fo, x, (int)ldx, ea, w1, b1, w2, b2, pw, out, (int)ldo, Act{}, w.sv, s);
}

int pfn_edge_aggr_backward(const void* gws, int64_t n, int64_t e, int fi, int fe, int h, int fo, const float* x,
                           int64_t ldx, const float* ea, const float* w1, const float* b1, const float* w2,
                           const float* b2, const float* gout, int64_t ldgo, float* gx, int64_t ldgx, float* gea,
                           float* gw1, float* gb1, float* gw2, float* gb2, void* ws, size_t ws_bytes, void* stream) {
    (void)b1; (void)b2;
    PFN_CHECK_ARG(gws && ws && w1 && w2 && gout && gw1 && gb1 && gw2 && gb2, "pfn_edge_aggr_backward: null pointer");
    PFN_CHECK_ARG(ldx == ld_of(fi) && ldgo == ld_of(fo) && (!gx || ldgx == ld_of(fi)),
                  "pfn_edge_aggr_backward: row strides must be pfn_padded_ld(F)");
    PFN_CHECK_ARG(fe >= 1 && fe <= 6, "efeature_dim must be in [1, 6]");
    EaLayerWs w = ea_layer_ws(ws, n, fi, fe, h, fo);
    if (ws_bytes < w.bytes) {
        set_error("pfn_edge_aggr_backward: workspace %zu < %zu bytes", ws_bytes, w.bytes);
        return PFN_ENOSPACE;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    GraphView g = graph_view(const_cast<void*>(gws), n, e);
    if (gea) PFN_CHECK_HIP(hipMemsetAsync(gea, 0, (size_t)e * fe * sizeof(float), s));
    Packer pk(w.packed);                       // images were filled by the forward call on the same workspace
    const EaPack pw = ea_pack(pk, fi, fe, h, fo, w1, w2);
    return ea_backward(g, fi, fe, h, fo, x, (int)ldx, ea, w1, w2, pw, gout, (int)ldgo, Gate{}, gx, (int)ldgx, gw1, gb1, gw2,
                       gb2, gea, w.sv, w.sc, s, nullptr);
}

struct TagLayerWs { float* xk; TagScratch sc; float* packed; size_t bytes; };
static TagLayerWs tag_layer_ws(void* ws, int64_t n, int cin, int cout, int K) {
    Carver cv(ws);
    TagLayerWs w;
    const size_t nld = (size_t)n * ld_of(cin);
    {
        Packer measure(nullptr);
        const float* none[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
        tag_pack(measure, cin, cout, K, none);
        w.packed = cv.take<float>(measure.off);
    }
    w.xk = cv.take<float>(nld * std::max(1, K));
    w.sc.G = cv.take<float>(nld * (K + 1));
    w.sc.z0 = cv.take<float>(nld);
    w.sc.z1 = cv.take<float>(nld);
    const int maxf = std::max(cin, cout);
    w.sc.red.floats = reduce_ws_floats(n, maxf, maxf, K + 1);
    w.sc.red.partial = cv.take<float>(w.sc.red.floats);
    w.bytes = cv.off;
    return w;
}

size_t pfn_tag_conv_workspace_bytes(int64_t n, int64_t e, int cin, int cout, int K) {
    (void)e;
    return tag_layer_ws(nullptr, n, cin, cout, K).bytes;
}

int pfn_tag_conv_forward(const void* gws, int64_t n, int64_t e, int cin, int cout, int K, const float* x, int64_t ldx,
                         const float* const* weights, const float* bias, float* out, int64_t ldo, void* ws,
                         size_t ws_bytes, int64_t seg_nodes, void* stream) {
    PFN_CHECK_ARG(gws && ws && weights, "pfn_tag_conv_forward: null pointer");
    PFN_CHECK_ARG(ldx == ld_of(cin) && ldo == ld_of(cout), "pfn_tag_conv_forward: row strides must be pfn_padded_ld(F)");
    PFN_CHECK_ARG(K >= 0 && K <= 7, "K must be in [0, 7]");
    TagLayerWs w = tag_layer_ws(ws, n, cin, cout, K);
    if (ws_bytes < w.bytes) {
        set_error("pfn_tag_conv_forward: workspace %zu < %zu bytes", ws_bytes, w.bytes);
        return PFN_ENOSPACE;
    }
    GraphView g = graph_view(const_cast<void*>(gws), n, e);
    hipStream_t s = static_cast<hipStream_t>(stream);
    Packer pk(w.packed);
    const TagPack pw = tag_pack(pk, cin, cout, K, weights);
    PFN_TRY(pk.flush(s));
    PFN_CHECK_ARG(seg_nodes >= 0 && (seg_nodes == 0 || n % seg_nodes == 0), "seg_nodes must be 0 or divide n_nodes");
    return tag_forward(g, cin, cout, K, x, (int)ldx, pw, bias, out, (int)ldo, Act{}, w.xk, s, (int)seg_nodes);
}

int pfn_tag_conv_backward(const void* gws, int64_t n, int64_t e, int cin, int cout, int K, const float* x, int64_t ldx,
                          const float* const* weights, const float* gout, int64_t ldgo, float* gx, int64_t ldgx,
                          float* const* gweights, float* gbias, void* ws, size_t ws_bytes, int64_t seg_nodes, void* stream) {
    PFN_CHECK_ARG(gws && ws && weights && gout && gweights, "pfn_tag_conv_backward: null pointer");
    PFN_CHECK_ARG(ldx == ld_of(cin) && ldgo == ld_of(cout), "pfn_tag_conv_backward: row strides must be pfn_padded_ld(F)");
    PFN_CHECK_ARG(K >= 0 && K <= 7, "K must be in [0, 7]");
    TagLayerWs w = tag_layer_ws(ws, n, cin, cout, K);
    if (ws_bytes < w.bytes) {
        set_error("pfn_tag_conv_backward: workspace %zu < %zu bytes", ws_bytes, w.bytes);
        return PFN_ENOSPACE;
    }
    GraphView g = graph_view(const_cast<void*>(gws), n, e);
    Packer pk(w.packed);                       // images were filled by the forward call on the same workspace
    const TagPack pw = tag_pack(pk, cin, cout, K, weights);
    PFN_CHECK_ARG(seg_nodes >= 0 && (seg_nodes == 0 || n % seg_nodes == 0), "seg_nodes must be 0 or divide n_nodes");
    return tag_backward(g, cin, cout, K, x, (int)ldx, pw, gout, (int)ldgo, Gate{}, gx, (int)ldgx, gweights, gbias, w.xk,
                        w.sc, static_cast<hipStream_t>(stream), nullptr, (int)seg_nodes);
}

// ----------------------------------------------------------------------------------------- utilities
int pfn_scatter_add(const void* gws, int64_t n, int64_t e, const float* x, float* out, int64_t f, void* stream) {
    PFN_CHECK_ARG(gws && (n == 0 || (x && out)), "pfn_scatter_add: null pointer");
    GraphView g = graph_view(const_cast<void*>(gws), n, e);
    HopArgs hp{x, nullptr, out, nullptr, 1.f, ld_of((int)f), 0, 0};
    return launch_hop(g, hp, static_cast<hipStream_t>(stream));
}

int pfn_pad_rows(const float* src, int64_t ld_src, float* dst, int64_t ld_dst, int64_t rows, int64_t f, void* stream) {
    PFN_CHECK_ARG(rows == 0 || (src && dst), "pfn_pad_rows: null pointer");
    PFN_CHECK_ARG(f <= ld_src && ld_dst >= 0, "pfn_pad_rows: bad strides");
    return launch_pad_rows(src, ld_src, dst, ld_dst, rows, std::min(f, ld_dst), static_cast<hipStream_t>(stream));
}

int pfn_mse_loss(const float* out, const float* y, int64_t count, float* loss, float* grad, void* ws, size_t ws_bytes,
                 void* stream) {
    PFN_CHECK_ARG(out && y && loss && ws, "pfn_mse_loss: null pointer");
    const int nb = (int)std::max<int64_t>(1, std::min<int64_t>((count + 1023) / 1024, 256));
    if (ws_bytes < 257 * sizeof(float)) {
        set_error("pfn_mse_loss: workspace too small (need %zu bytes)", 257 * sizeof(float));
        return PFN_ENOSPACE;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    const float inv_n = count > 0 ? 1.0f / (float)count : 0.f;
    mse_kernel<<<nb, 256, 0, s>>>(out, y, count, inv_n, grad, static_cast<float*>(ws),
                                  reinterpret_cast<int*>(static_cast<float*>(ws) + 256), loss);
    PFN_CHECK_LAUNCH();
    return PFN_OK;
}

int pfn_masked_l2_loss(const float* out, const float* y, const void* mask, int mask_dtype, int64_t count, int regularize,
                       float regcoeff, float* loss, float* grad, void* ws, size_t ws_bytes, void* stream) {
    PFN_CHECK_ARG(out && y && mask && loss && ws, "pfn_masked_l2_loss: null pointer");
    PFN_CHECK_ARG(mask_dtype == 0 || mask_dtype == 1, "pfn_masked_l2_loss: mask_dtype must be 0 (int64) or 1 (float32)");
    if (ws_bytes < sizeof(MaskedL2Ws)) {
        set_error("pfn_masked_l2_loss: workspace too small (need %zu bytes)", sizeof(MaskedL2Ws));
        return PFN_ENOSPACE;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    MaskedL2Ws* w = static_cast<MaskedL2Ws*>(ws);
    const int nb = (int)std::max<int64_t>(1, std::min<int64_t>((count + 1023) / 1024, 256));
    masked_l2_reduce_kernel<<<nb, 256, 0, s>>>(out, y, mask, mask_dtype, count, regularize, regcoeff, w, loss);
    PFN_CHECK_LAUNCH();
    if (grad && count > 0) {
        masked_l2_grad_kernel<<<(int)std::min<int64_t>((count + 255) / 256, 1024), 256, 0, s>>>(out, y, mask, mask_dtype, count,
                                                                                              regularize, regcoeff, w, grad);
        PFN_CHECK_LAUNCH();
    }
    return PFN_OK;
}

int pfn_dropout_mask(const uint64_t* rng_state, int32_t layer, int64_t rows, int64_t ncols, float p, float* keep,
                     void* stream) {
    PFN_CHECK_ARG(rng_state && (rows == 0 || keep), "pfn_dropout_mask: null pointer");
    PFN_CHECK_ARG(layer >= 0 && rows >= 0 && rows < (1ll << 32) && ncols > 0 && ncols < (1ll << 30), "pfn_dropout_mask: bad sizes");
    if (rows == 0) return PFN_OK;
    const int64_t items = rows * ((ncols + 3) / 4);
    dropout_mask_kernel<<<(int)std::min<int64_t>((items + 255) / 256, 4096), 256, 0, static_cast<hipStream_t>(stream)>>>(
        rng_state, (uint32_t)layer, rows, (int)ncols, p, keep);
    PFN_CHECK_LAUNCH();
    return PFN_OK;
}

int pfn_adamw_step(float* p, const float* g, float* m, float* v, int64_t count, float lr, float b1, float b2,
                   float eps, float wd, int64_t* step, void* stream) {
    PFN_CHECK_ARG(p && g && m && v && step, "pfn_adamw_step: null pointer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    adamw_kernel<<<adamw_blocks(count), 256, 0, s>>>(p, g, m, v, count, lr, b1, b2, eps, wd, step, nullptr, nullptr);
    PFN_CHECK_LAUNCH();
    return PFN_OK;
}

int pfn_adamw_step_dev(float* p, const float* g, float* m, float* v, int64_t count, const float* hyper, int64_t* step,
                       void* stream) {
    PFN_CHECK_ARG(p && g && m && v && step && hyper, "pfn_adamw_step_dev: null pointer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    adamw_kernel<<<adamw_blocks(count), 256, 0, s>>>(p, g, m, v, count, 0.f, 0.f, 0.f, 0.f, 0.f, step, hyper, nullptr);
    PFN_CHECK_LAUNCH();
    return PFN_OK;
}

int pfn_adamw_step_guarded(float* p, const float* g, float* m, float* v, int64_t count, const float* hyper, int64_t* step,
                           const float* guard, void* stream) {
    PFN_CHECK_ARG(p && g && m && v && step && hyper && guard, "pfn_adamw_step_guarded: null pointer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    adamw_kernel<<<adamw_blocks(count), 256, 0, s>>>(p, g, m, v, count, 0.f, 0.f, 0.f, 0.f, 0.f, step, hyper, guard);
    PFN_CHECK_LAUNCH();
    return PFN_OK;
}

}  // extern "C"
