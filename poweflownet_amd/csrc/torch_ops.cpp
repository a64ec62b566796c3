// TORCH_LIBRARY(pfn, ...): the hot path as PyTorch operators above the C ABI (last table row; BASELINE north_star
// "exposed through a PyTorch-ROCm C++/HIP extension").
//
// Every operator here is plumbing and nothing else: it validates dtype / device / contiguity / shapes with TORCH_CHECK (a bad
// input surfaces as RuntimeError, as in the reference -- never an abort, never a silently clamped index), allocates outputs and
// workspaces through the ATen caching allocator on the inputs' device (hipGraph-capturable: no hipMalloc in a step), takes
// torch's CURRENT HIP stream, and calls exactly one entry point of include/pfn_hip.h with raw pointers.  No kernel, no
// arithmetic and no state lives in this file; libpfn_hip.so stays free of torch types and is what the tests grade.
//
//   callers this serves                                     operator
//   utils/training.py:58 / utils/evaluation.py:79  model(data)              pfn::mpn (differentiable: one C++ autograd node),
//                                                                          pfn::mpn_forward / pfn::mpn_backward (its two halves)
//   networks/MPN.py:498-523 is_directed + undirect_graph                   pfn::graph_build
//   networks/MPN.py:30-56   EdgeAggregation.forward (+ autograd)           pfn::edge_aggr (differentiable), pfn::edge_aggr_forward / _backward
//   networks/MPN.py:477-484 PyG TAGConv.forward (+ autograd)               pfn::tag_conv (differentiable), pfn::tag_conv_forward / _backward
//   PyG propagate(aggr='add') in isolation (the roofline run)              pfn::scatter_add
//   train.py:103 MSELoss, utils/training.py:72-74                          pfn::mse_loss
//   train.py:123 AdamW.step                                                pfn::adamw_step_
//
// The Python package binds the same C ABI with ctypes (poweflownet_amd/_lib.py); tests/test_torch_ops.py holds the two bindings
// bit-identical on the GPU.  Built by poweflownet_amd/csrc/Makefile (g++, torch headers; no device code).
#include <ATen/ATen.h>
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>   // (torch-ROCm tensors carry DeviceType::CUDA: the plain HIPGuard refuses them)
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <torch/csrc/autograd/custom_function.h>
#include <torch/library.h>

#include <tuple>
#include <vector>

#include "../../include/pfn_hip.h"

namespace {

using at::Tensor;

void* cur_stream(const Tensor& on) { return static_cast<void*>(c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(on.device().index()).stream()); }

void pfn_ok(int rc, const char* what) { TORCH_CHECK(rc == 0, what, " failed (code ", rc, "): ", pfn_last_error()); }

void want(const Tensor& t, const char* name, at::ScalarType dt, const Tensor& like) {
    TORCH_CHECK(t.defined(), name, " is undefined");
    TORCH_CHECK(t.is_cuda(), "poweflownet_amd: ", name, " must live on a HIP device (got ", t.device(), "); there is no CPU fallback");
    TORCH_CHECK(t.device() == like.device(), name, " is on ", t.device(), ", expected ", like.device());
    TORCH_CHECK(t.scalar_type() == dt, name, " must be ", dt, " (got ", t.scalar_type(), ")");
    TORCH_CHECK(t.is_contiguous(), name, " must be contiguous");
}
void want_f32(const Tensor& t, const char* name, const Tensor& like) { want(t, name, at::kFloat, like); }
// a graph workspace is only ever what graph_build returned for THIS (num_nodes, e_stored): the kernels index it by those two
// numbers, so a mismatch would read out of bounds on the device
void want_graph(const Tensor& g, const Tensor& like, int64_t n, int64_t e_stored) {
    want(g, "graph_ws", at::kByte, like);
    TORCH_CHECK(n >= 0 && e_stored >= 0, "num_nodes and e_stored must be non-negative");
    TORCH_CHECK((size_t)g.numel() == pfn_graph_workspace_bytes(n, e_stored), "graph_ws holds ", g.numel(), " bytes, but graph_build(edge_index [2, ",
                e_stored, "], num_nodes = ", n, ") returns ", pfn_graph_workspace_bytes(n, e_stored), ": it was built for another batch");
}
// What keeps a bad graph from producing plausible numbers through torch.ops (the Python module raises, or poisons, in the same
// cases): a caller that has NOT run graph_check / graph_segments (`validated` false) gets (i) the segment check of its
// `seg_nodes` run on the device in front of the forward and (ii) `out` overwritten with NaN behind it when the workspace records
// a node id outside [0, num_nodes) or an edge that crosses a segment boundary (pfn_graph_poison_if_bad).  No host sync.
void graph_precheck(const Tensor& graph_ws, int64_t n, int64_t e_stored, int64_t seg_nodes, void* stream) {
    // (seg_nodes = 0: no promise to check -- the call withdraws the verdict an earlier check may have left in this workspace)
    TORCH_CHECK(seg_nodes >= 0 && (seg_nodes == 0 || (n > 0 && n % seg_nodes == 0 && seg_nodes <= (1 << 20))), "seg_nodes = ", seg_nodes,
                " must be 0 or divide num_nodes = ", n);
    pfn_ok(pfn_graph_segments_async(graph_ws.data_ptr(), n, e_stored, seg_nodes, stream), "pfn_graph_segments_async");
}
void graph_poison(const Tensor& graph_ws, int64_t n, int64_t e_stored, Tensor& out, void* stream) {
    pfn_ok(pfn_graph_poison_if_bad(graph_ws.data_ptr(), n, e_stored, out.data_ptr<float>(), out.numel(), stream), "pfn_graph_poison_if_bad");
}

// rows padded to the library's ld = roundup(F, 4) with zero pad columns (pfn_padded_ld); a no-op when F % 4 == 0
Tensor pad_rows(const Tensor& x, int64_t f) {
    const int64_t ld = pfn_padded_ld(f);
    if (ld == f) return x;
    Tensor p = at::empty({x.size(0), ld}, x.options());
    pfn_ok(pfn_pad_rows(x.data_ptr<float>(), f, p.data_ptr<float>(), ld, x.size(0), f, cur_stream(x)), "pfn_pad_rows");
    return p;
}
Tensor unpad_rows(const Tensor& p, int64_t f) { return p.size(1) == f ? p : p.narrow(1, 0, f).contiguous(); }

std::vector<const float*> ptrs(at::TensorList ts, const char* name, const Tensor& like) {
    std::vector<const float*> v;
    v.reserve(ts.size());
    for (const Tensor& t : ts) {
        want_f32(t, name, like);
        v.push_back(t.data_ptr<float>());
    }
    return v;
}

// dims = {nfeature_dim, efeature_dim, output_dim, hidden_dim, n_gnn_layers, K}: the constructor arguments of MaskEmbdMultiMPN
// (networks/MPN.py:462) in order
pfn_mpn_config config_of(at::IntArrayRef dims, double dropout, bool training, bool need_backward) {
    TORCH_CHECK(dims.size() == 6, "dims must be {nfeature_dim, efeature_dim, output_dim, hidden_dim, n_gnn_layers, K}");
    pfn_mpn_config c;
    c.nfeature_dim = (int32_t)dims[0];
    c.efeature_dim = (int32_t)dims[1];
    c.output_dim = (int32_t)dims[2];
    c.hidden_dim = (int32_t)dims[3];
    c.n_gnn_layers = (int32_t)dims[4];
    c.K = (int32_t)dims[5];
    c.dropout_rate = (float)dropout;
    c.training = training ? 1 : 0;
    c.need_backward = need_backward ? 1 : 0;
    return c;
}
int mask_dtype_of(const Tensor& m) {
    TORCH_CHECK(m.scalar_type() == at::kLong || m.scalar_type() == at::kFloat, "pred_mask must be int64 or float32 (got ", m.scalar_type(), ")");
    return m.scalar_type() == at::kLong ? 0 : 1;
}

// ---- networks/MPN.py:498-523.  mode -1: the reference's first-edge heuristic on the device; 0: the list as given; 1: always undirect
Tensor graph_build(const Tensor& edge_index, int64_t num_nodes, int64_t mode) {
    TORCH_CHECK(edge_index.defined() && edge_index.is_cuda(), "edge_index must live on a HIP device; there is no CPU fallback");
    TORCH_CHECK(edge_index.scalar_type() == at::kLong && edge_index.dim() == 2 && edge_index.size(0) == 2 && edge_index.is_contiguous(),
                "edge_index must be a contiguous int64 [2, E] tensor");
    TORCH_CHECK(num_nodes >= 0 && mode >= -1 && mode <= 1, "num_nodes >= 0 and mode in {-1, 0, 1} expected");
    const c10::hip::HIPGuardMasqueradingAsCUDA guard(edge_index.device());
    const int64_t e = edge_index.size(1);
    const size_t bytes = pfn_graph_workspace_bytes(num_nodes, e);
    Tensor ws = at::empty({(int64_t)bytes}, edge_index.options().dtype(at::kByte));
    pfn_ok(pfn_graph_build(edge_index.data_ptr<int64_t>(), e, num_nodes, (int)mode, ws.data_ptr(), bytes, cur_stream(edge_index)),
           "pfn_graph_build");
    return ws;
}
// ---- validation with a read-back (once per topology): raises RuntimeError when edge_index held a node id outside
// [0, num_nodes) (the reference's index_select raises in its forward, networks/MPN.py:53); returns (the is_directed verdict of
// networks/MPN.py:498-504, the effective edge count after undirect_graph)
std::tuple<bool, int64_t> graph_check(const Tensor& graph_ws, int64_t num_nodes, int64_t e_stored) {
    want_graph(graph_ws, graph_ws, num_nodes, e_stored);
    const c10::hip::HIPGuardMasqueradingAsCUDA guard(graph_ws.device());
    int32_t directed = 0;
    int64_t e_eff = 0;
    pfn_ok(pfn_graph_info(graph_ws.data_ptr(), num_nodes, e_stored, &directed, &e_eff, cur_stream(graph_ws)), "pfn_graph_info");
    return {directed != 0, e_eff};
}
// ---- is the batch a disjoint union of index-contiguous graphs of seg_nodes nodes (PyG's Batch of one case)?  With a read-back;
// a caller that got True may pass seg_nodes and validated=True to the model operators
bool graph_segments(Tensor graph_ws, int64_t num_nodes, int64_t e_stored, int64_t seg_nodes) {
    want_graph(graph_ws, graph_ws, num_nodes, e_stored);
    const c10::hip::HIPGuardMasqueradingAsCUDA guard(graph_ws.device());
    int32_t ok = 0;
    pfn_ok(pfn_graph_segments(graph_ws.data_ptr(), num_nodes, e_stored, seg_nodes, &ok, cur_stream(graph_ws)), "pfn_graph_segments");
    return ok != 0;
}

// ---- MaskEmbdMultiMPN.forward, networks/MPN.py:525-559.  Returns (out [N, output_dim], ws: what mpn_backward needs)
std::tuple<Tensor, Tensor> mpn_forward(const Tensor& graph_ws, int64_t e_stored, int64_t seg_nodes, at::IntArrayRef dims, double dropout,
                                       bool training, bool need_backward, at::TensorList params, const Tensor& x, const Tensor& pred_mask,
                                       const Tensor& edge_attr, const c10::optional<Tensor>& rng_state, bool defer_out = false,
                                       bool validated = false) {
    // defer_out: the output rows are left to mpn_backward_mse (where mpn_mse_tail_ok says so); `out` comes back unwritten
    // validated: the caller ran graph_check (and graph_segments for its seg_nodes) on this workspace -- see graph_precheck
    TORCH_CHECK(validated || !defer_out, "defer_out needs a validated graph (graph_check / graph_segments): the NaN poison of an unvalidated one "
                                         "travels through `out`");
    TORCH_CHECK(x.defined() && x.dim() == 2, "x must be [N, nfeature_dim]");
    const pfn_mpn_config c = config_of(dims, dropout, training, need_backward);
    TORCH_CHECK(x.size(1) == c.nfeature_dim, "x must be [N, ", c.nfeature_dim, "], got [", x.size(0), ", ", x.size(1), "]");
    want_f32(x, "x", x);
    want_graph(graph_ws, x, x.size(0), e_stored);
    want(pred_mask, "pred_mask", pred_mask.scalar_type(), x);
    TORCH_CHECK(pred_mask.sizes() == x.sizes(), "pred_mask must have x's shape");
    want_f32(edge_attr, "edge_attr", x);
    TORCH_CHECK(edge_attr.dim() == 2 && edge_attr.size(0) == e_stored && edge_attr.size(1) == c.efeature_dim, "edge_attr must be [e_stored, efeature_dim]");
    TORCH_CHECK((int64_t)params.size() == pfn_mpn_num_params(&c), "expected ", pfn_mpn_num_params(&c), " parameter tensors, got ", params.size());
    const std::vector<const float*> pp = ptrs(params, "parameter", x);
    uint64_t* rng = nullptr;
    if (rng_state.has_value() && rng_state->defined()) {
        want(*rng_state, "rng_state", at::kLong, x);
        TORCH_CHECK(rng_state->numel() >= 2, "rng_state must hold {seed, offset}");
        rng = reinterpret_cast<uint64_t*>(rng_state->data_ptr<int64_t>());
    }
    TORCH_CHECK(!(training && dropout > 0.0) || rng != nullptr, "a training forward with dropout needs rng_state");
    const c10::hip::HIPGuardMasqueradingAsCUDA guard(x.device());
    const int64_t n = x.size(0);
    Tensor out = at::empty({n, pfn_padded_ld(c.output_dim)}, x.options());
    const size_t bytes = pfn_mpn_workspace_bytes(&c, n, e_stored);
    Tensor ws = at::empty({(int64_t)bytes}, x.options().dtype(at::kByte));
    if (!validated) graph_precheck(graph_ws, n, e_stored, seg_nodes, cur_stream(x));
    pfn_ok(pfn_mpn_forward(&c, graph_ws.data_ptr(), n, e_stored, pp.data(), x.data_ptr<float>(), pred_mask.data_ptr(), mask_dtype_of(pred_mask),
                           edge_attr.data_ptr<float>(), defer_out ? nullptr : out.data_ptr<float>(), ws.data_ptr(), bytes, rng, seg_nodes,
                           cur_stream(x)),
           "pfn_mpn_forward");
    if (!validated) graph_poison(graph_ws, n, e_stored, out, cur_stream(x));
    return {unpad_rows(out, c.output_dim), ws};
}

// ---- is the MSELoss tail (mpn_backward_mse) available for this model and batch?  (pfn_mpn_mse_tail_ok)
bool mpn_mse_tail_ok(int64_t num_nodes, int64_t e_stored, int64_t seg_nodes, at::IntArrayRef dims, double dropout, bool training) {
    const pfn_mpn_config c = config_of(dims, dropout, training, true);
    return pfn_mpn_mse_tail_ok(&c, num_nodes, e_stored, seg_nodes) == 1;
}

// ---- loss = MSELoss()(out, y); loss.backward() -- train.py:103, utils/training.py:70-74 -- riding in the backward pass's first
// launch (pfn_mpn_backward_mse).  `out`: the tensor mpn_forward returned (with defer_out it is written HERE).  `loss_ws`: float32
// [>= 1025], zero before the first call.  Returns (flat gradient of every parameter, loss, grad_out, grad_x?)
std::tuple<Tensor, Tensor, Tensor, Tensor> mpn_backward_mse(const Tensor& graph_ws, int64_t e_stored, int64_t seg_nodes, at::IntArrayRef dims,
                                                            double dropout, bool training, at::TensorList params, const Tensor& x,
                                                            const Tensor& edge_attr, const Tensor& y, Tensor out, const Tensor& ws,
                                                            Tensor loss_ws, bool need_grad_x) {
    const pfn_mpn_config c = config_of(dims, dropout, training, true);
    want_f32(x, "x", x);
    want_graph(graph_ws, x, x.size(0), e_stored);
    want(ws, "ws", at::kByte, x);
    want_f32(edge_attr, "edge_attr", x);
    want_f32(y, "y", x);
    want_f32(out, "out", x);
    want_f32(loss_ws, "loss_ws", x);
    const int64_t n = x.size(0);
    TORCH_CHECK(x.dim() == 2 && x.size(1) == c.nfeature_dim, "x must be [N, nfeature_dim]");
    TORCH_CHECK(c.output_dim == 4 && y.dim() == 2 && y.size(0) == n && y.size(1) == 4 && out.sizes() == y.sizes(), "y and out must be [N, 4]");
    TORCH_CHECK(loss_ws.numel() >= 1025, "loss_ws must hold 1025 floats");
    TORCH_CHECK((int64_t)params.size() == pfn_mpn_num_params(&c), "expected ", pfn_mpn_num_params(&c), " parameter tensors, got ", params.size());
    const std::vector<const float*> pp = ptrs(params, "parameter", x);
    const c10::hip::HIPGuardMasqueradingAsCUDA guard(x.device());
    TORCH_CHECK((size_t)ws.numel() == pfn_mpn_workspace_bytes(&c, n, e_stored), "ws is not the buffer mpn_forward(need_backward=True) returned");
    int64_t total = 0;
    for (const Tensor& p : params) total += p.numel();
    Tensor flat = at::empty({total}, x.options());
    std::vector<float*> gp;
    gp.reserve(params.size());
    int64_t off = 0;
    for (const Tensor& p : params) {
        gp.push_back(flat.data_ptr<float>() + off);
        off += p.numel();
    }
    Tensor loss = at::empty({}, x.options());
    Tensor gout = at::empty_like(y);
    Tensor gx = need_grad_x ? at::empty_like(x) : Tensor();
    pfn_ok(pfn_mpn_backward_mse(&c, graph_ws.data_ptr(), n, e_stored, pp.data(), gp.data(), x.data_ptr<float>(), edge_attr.data_ptr<float>(),
                                y.data_ptr<float>(), out.data_ptr<float>(), loss.data_ptr<float>(), gout.data_ptr<float>(),
                                need_grad_x ? gx.data_ptr<float>() : nullptr, ws.data_ptr(), (size_t)ws.numel(), loss_ws.data_ptr(),
                                (size_t)loss_ws.numel() * 4, seg_nodes, cur_stream(x)),
           "pfn_mpn_backward_mse");
    return {flat, loss, gout, gx};
}

// ---- what loss.backward() runs, utils/training.py:74.  Returns (flat gradient of every parameter in table order, grad_x?, grad_edge_attr?)
std::tuple<Tensor, Tensor, Tensor> mpn_backward(const Tensor& graph_ws, int64_t e_stored, int64_t seg_nodes, at::IntArrayRef dims, double dropout,
                                                bool training, at::TensorList params, const Tensor& x, const Tensor& pred_mask,
                                                const Tensor& edge_attr, const Tensor& grad_out, const Tensor& ws, bool need_grad_x,
                                                bool need_grad_edge_attr) {
    const pfn_mpn_config c = config_of(dims, dropout, training, true);
    want_f32(x, "x", x);
    want_graph(graph_ws, x, x.size(0), e_stored);
    want(ws, "ws", at::kByte, x);
    want_f32(edge_attr, "edge_attr", x);
    want_f32(grad_out, "grad_out", x);
    TORCH_CHECK(x.dim() == 2 && x.size(1) == c.nfeature_dim, "x must be [N, nfeature_dim]");
    TORCH_CHECK(grad_out.dim() == 2 && grad_out.size(0) == x.size(0) && grad_out.size(1) == c.output_dim, "grad_out must be [N, output_dim]");
    TORCH_CHECK((int64_t)params.size() == pfn_mpn_num_params(&c), "expected ", pfn_mpn_num_params(&c), " parameter tensors, got ", params.size());
    const std::vector<const float*> pp = ptrs(params, "parameter", x);
    const c10::hip::HIPGuardMasqueradingAsCUDA guard(x.device());
    const int64_t n = x.size(0);
    TORCH_CHECK((size_t)ws.numel() == pfn_mpn_workspace_bytes(&c, n, e_stored), "ws is not the buffer mpn_forward(need_backward=True) returned");
    int64_t total = 0;
    for (const Tensor& p : params) total += p.numel();
    Tensor flat = at::empty({total}, x.options());
    std::vector<float*> gp;
    gp.reserve(params.size());
    int64_t off = 0;
    for (const Tensor& p : params) {
        gp.push_back(flat.data_ptr<float>() + off);
        off += p.numel();
    }
    Tensor go = pad_rows(grad_out, c.output_dim);
    Tensor gx = need_grad_x ? at::empty_like(x) : Tensor();
    Tensor gea = need_grad_edge_attr ? at::empty_like(edge_attr) : Tensor();
    pfn_ok(pfn_mpn_backward(&c, graph_ws.data_ptr(), n, e_stored, pp.data(), gp.data(), x.data_ptr<float>(), pred_mask.data_ptr(),
                            mask_dtype_of(pred_mask), edge_attr.data_ptr<float>(), go.data_ptr<float>(), need_grad_x ? gx.data_ptr<float>() : nullptr,
                            need_grad_edge_attr ? gea.data_ptr<float>() : nullptr, ws.data_ptr(), (size_t)ws.numel(), seg_nodes, cur_stream(x)),
           "pfn_mpn_backward");
    return {flat, gx, gea};
}

// ---- EdgeAggregation.forward, networks/MPN.py:30-56.  Returns (out [N, Fo], ws)
std::tuple<Tensor, Tensor> edge_aggr_forward(const Tensor& graph_ws, int64_t e_stored, const Tensor& x, const Tensor& edge_attr, const Tensor& w1,
                                             const Tensor& b1, const Tensor& w2, const Tensor& b2) {
    want_f32(x, "x", x);
    want_graph(graph_ws, x, x.size(0), e_stored);
    want_f32(edge_attr, "edge_attr", x);
    want_f32(w1, "W1", x); want_f32(b1, "b1", x); want_f32(w2, "W2", x); want_f32(b2, "b2", x);
    TORCH_CHECK(x.dim() == 2 && edge_attr.dim() == 2 && w1.dim() == 2 && w2.dim() == 2, "x, edge_attr, W1, W2 must be matrices");
    const int64_t n = x.size(0), fi = x.size(1), fe = edge_attr.size(1), h = w1.size(0), fo = w2.size(0);
    TORCH_CHECK(edge_attr.size(0) == e_stored && w1.size(1) == 2 * fi + fe && b1.numel() == h && w2.size(1) == h && b2.numel() == fo,
                "EdgeAggregation shapes: W1 (H, 2 Fi + Fe), b1 (H), W2 (Fo, H), b2 (Fo), edge_attr [e_stored, Fe]");
    const c10::hip::HIPGuardMasqueradingAsCUDA guard(x.device());
    Tensor xp = pad_rows(x, fi);
    Tensor out = at::empty({n, pfn_padded_ld(fo)}, x.options());
    const size_t bytes = pfn_edge_aggr_workspace_bytes(n, e_stored, (int)fi, (int)fe, (int)h, (int)fo);
    Tensor ws = at::empty({(int64_t)bytes}, x.options().dtype(at::kByte));
    graph_precheck(graph_ws, n, e_stored, 0, cur_stream(x));   // (the layer makes no segment promise: only the id-range flag may poison)
    pfn_ok(pfn_edge_aggr_forward(graph_ws.data_ptr(), n, e_stored, (int)fi, (int)fe, (int)h, (int)fo, xp.data_ptr<float>(), xp.size(1),
                                 edge_attr.data_ptr<float>(), w1.data_ptr<float>(), b1.data_ptr<float>(), w2.data_ptr<float>(), b2.data_ptr<float>(),
                                 out.data_ptr<float>(), out.size(1), ws.data_ptr(), bytes, cur_stream(x)),
           "pfn_edge_aggr_forward");
    graph_poison(graph_ws, n, e_stored, out, cur_stream(x));
    return {unpad_rows(out, fo), ws};
}
// Returns (grad_x, grad_edge_attr, grad_W1, grad_b1, grad_W2, grad_b2)
std::tuple<Tensor, Tensor, Tensor, Tensor, Tensor, Tensor> edge_aggr_backward(const Tensor& graph_ws, int64_t e_stored, const Tensor& x,
                                                                              const Tensor& edge_attr, const Tensor& w1, const Tensor& b1,
                                                                              const Tensor& w2, const Tensor& b2, const Tensor& grad_out,
                                                                              const Tensor& ws) {
    want_f32(x, "x", x);
    want_graph(graph_ws, x, x.size(0), e_stored);
    want(ws, "ws", at::kByte, x);
    want_f32(edge_attr, "edge_attr", x); want_f32(grad_out, "grad_out", x);
    want_f32(w1, "W1", x); want_f32(b1, "b1", x); want_f32(w2, "W2", x); want_f32(b2, "b2", x);
    const int64_t n = x.size(0), fi = x.size(1), fe = edge_attr.size(1), h = w1.size(0), fo = w2.size(0);
    TORCH_CHECK(grad_out.dim() == 2 && grad_out.size(0) == n && grad_out.size(1) == fo, "grad_out must be [N, Fo]");
    TORCH_CHECK((size_t)ws.numel() == pfn_edge_aggr_workspace_bytes(n, e_stored, (int)fi, (int)fe, (int)h, (int)fo), "ws is not edge_aggr_forward's");
    const c10::hip::HIPGuardMasqueradingAsCUDA guard(x.device());
    Tensor xp = pad_rows(x, fi), go = pad_rows(grad_out, fo);
    Tensor gx = at::empty_like(xp), gea = at::empty_like(edge_attr);
    Tensor gw1 = at::empty_like(w1), gb1 = at::empty_like(b1), gw2 = at::empty_like(w2), gb2 = at::empty_like(b2);
    pfn_ok(pfn_edge_aggr_backward(graph_ws.data_ptr(), n, e_stored, (int)fi, (int)fe, (int)h, (int)fo, xp.data_ptr<float>(), xp.size(1),
                                  edge_attr.data_ptr<float>(), w1.data_ptr<float>(), b1.data_ptr<float>(), w2.data_ptr<float>(), b2.data_ptr<float>(),
                                  go.data_ptr<float>(), go.size(1), gx.data_ptr<float>(), gx.size(1), gea.data_ptr<float>(), gw1.data_ptr<float>(),
                                  gb1.data_ptr<float>(), gw2.data_ptr<float>(), gb2.data_ptr<float>(), ws.data_ptr(), (size_t)ws.numel(), cur_stream(x)),
           "pfn_edge_aggr_backward");
    return {unpad_rows(gx, fi), gea, gw1, gb1, gw2, gb2};
}

// ---- PyG TAGConv.forward (call sites networks/MPN.py:477-484,:545).  weights = lins.0.weight .. lins.K.weight.  Returns (out, ws)
std::tuple<Tensor, Tensor> tag_conv_forward(const Tensor& graph_ws, int64_t e_stored, int64_t seg_nodes, const Tensor& x, at::TensorList weights,
                                            const c10::optional<Tensor>& bias) {
    want_f32(x, "x", x);
    want_graph(graph_ws, x, x.size(0), e_stored);
    TORCH_CHECK(x.dim() == 2 && !weights.empty(), "x must be a matrix and weights non-empty");
    const int64_t n = x.size(0), cin = x.size(1), cout = weights[0].size(0), K = (int64_t)weights.size() - 1;
    for (const Tensor& w : weights) TORCH_CHECK(w.dim() == 2 && w.size(0) == cout && w.size(1) == cin, "every TAGConv weight must be (out, in)");
    const std::vector<const float*> wp = ptrs(weights, "weight", x);
    const float* bp = nullptr;
    if (bias.has_value() && bias->defined()) {
        want_f32(*bias, "bias", x);
        TORCH_CHECK(bias->numel() == cout, "bias must have `out` entries");
        bp = bias->data_ptr<float>();
    }
    const c10::hip::HIPGuardMasqueradingAsCUDA guard(x.device());
    Tensor xp = pad_rows(x, cin);
    Tensor out = at::empty({n, pfn_padded_ld(cout)}, x.options());
    const size_t bytes = pfn_tag_conv_workspace_bytes(n, e_stored, (int)cin, (int)cout, (int)K);
    Tensor ws = at::empty({(int64_t)bytes}, x.options().dtype(at::kByte));
    graph_precheck(graph_ws, n, e_stored, seg_nodes, cur_stream(x));
    pfn_ok(pfn_tag_conv_forward(graph_ws.data_ptr(), n, e_stored, (int)cin, (int)cout, (int)K, xp.data_ptr<float>(), xp.size(1), wp.data(), bp,
                                out.data_ptr<float>(), out.size(1), ws.data_ptr(), bytes, seg_nodes, cur_stream(x)),
           "pfn_tag_conv_forward");
    graph_poison(graph_ws, n, e_stored, out, cur_stream(x));
    return {unpad_rows(out, cout), ws};
}
// Returns (grad_x, grad_bias (empty when has_bias is false), grad_weights...)
std::tuple<Tensor, Tensor, std::vector<Tensor>> tag_conv_backward(const Tensor& graph_ws, int64_t e_stored, int64_t seg_nodes, const Tensor& x,
                                                                  at::TensorList weights, const Tensor& grad_out, const Tensor& ws, bool has_bias) {
    want_f32(x, "x", x);
    want_graph(graph_ws, x, x.size(0), e_stored);
    want(ws, "ws", at::kByte, x);
    want_f32(grad_out, "grad_out", x);
    TORCH_CHECK(x.dim() == 2 && !weights.empty(), "x must be a matrix and weights non-empty");
    const int64_t n = x.size(0), cin = x.size(1), cout = weights[0].size(0), K = (int64_t)weights.size() - 1;
    TORCH_CHECK(grad_out.dim() == 2 && grad_out.size(0) == n && grad_out.size(1) == cout, "grad_out must be [N, out]");
    TORCH_CHECK((size_t)ws.numel() == pfn_tag_conv_workspace_bytes(n, e_stored, (int)cin, (int)cout, (int)K), "ws is not tag_conv_forward's");
    const std::vector<const float*> wp = ptrs(weights, "weight", x);
    const c10::hip::HIPGuardMasqueradingAsCUDA guard(x.device());
    Tensor xp = pad_rows(x, cin), go = pad_rows(grad_out, cout);
    Tensor gx = at::empty_like(xp);
    std::vector<Tensor> gws;
    std::vector<float*> gwp;
    for (const Tensor& w : weights) {
        gws.push_back(at::empty_like(w));
        gwp.push_back(gws.back().data_ptr<float>());
    }
    Tensor gb = has_bias ? at::empty({cout}, x.options()) : Tensor();
    pfn_ok(pfn_tag_conv_backward(graph_ws.data_ptr(), n, e_stored, (int)cin, (int)cout, (int)K, xp.data_ptr<float>(), xp.size(1), wp.data(),
                                 go.data_ptr<float>(), go.size(1), gx.data_ptr<float>(), gx.size(1), gwp.data(), has_bias ? gb.data_ptr<float>() : nullptr,
                                 ws.data_ptr(), (size_t)ws.numel(), seg_nodes, cur_stream(x)),
           "pfn_tag_conv_backward");
    return {unpad_rows(gx, cin), gb, gws};
}

// ---- PyG propagate(aggr='add') alone: out[i] = sum_{e -> i} x[src(e)]
Tensor scatter_add(const Tensor& graph_ws, int64_t e_stored, const Tensor& x) {
    want_f32(x, "x", x);
    want_graph(graph_ws, x, x.size(0), e_stored);
    TORCH_CHECK(x.dim() == 2, "x must be [N, F]");
    const c10::hip::HIPGuardMasqueradingAsCUDA guard(x.device());
    const int64_t f = x.size(1);
    Tensor xp = pad_rows(x, f);
    Tensor out = at::empty_like(xp);
    graph_precheck(graph_ws, x.size(0), e_stored, 0, cur_stream(x));
    pfn_ok(pfn_scatter_add(graph_ws.data_ptr(), x.size(0), e_stored, xp.data_ptr<float>(), out.data_ptr<float>(), f, cur_stream(x)), "pfn_scatter_add");
    graph_poison(graph_ws, x.size(0), e_stored, out, cur_stream(x));
    return unpad_rows(out, f);
}

// ---- train.py:103 MSELoss + its gradient in one pass.  ws: float32[>= 264], zero before the first call (left zero by every call)
std::tuple<Tensor, Tensor> mse_loss(const Tensor& out, const Tensor& y, Tensor ws) {
    want_f32(out, "out", out);
    want_f32(y, "y", out);
    want_f32(ws, "ws", out);
    TORCH_CHECK(out.sizes() == y.sizes(), "MSELoss: shape mismatch");
    TORCH_CHECK(ws.numel() >= 264, "ws must hold at least 264 floats");
    const c10::hip::HIPGuardMasqueradingAsCUDA guard(out.device());
    Tensor loss = at::empty({}, out.options());
    Tensor grad = at::empty_like(out);
    pfn_ok(pfn_mse_loss(out.data_ptr<float>(), y.data_ptr<float>(), out.numel(), loss.data_ptr<float>(), grad.data_ptr<float>(), ws.data_ptr(),
                        (size_t)ws.numel() * 4, cur_stream(out)),
           "pfn_mse_loss");
    return {loss, grad};
}

// ---- train.py:123 AdamW.step on one flat buffer, in place; step: device int64[2] {completed steps, 0}
void adamw_step_(Tensor param, const Tensor& grad, Tensor exp_avg, Tensor exp_avg_sq, double lr, double beta1, double beta2, double eps,
                 double weight_decay, Tensor step) {
    want_f32(param, "param", param);
    want_f32(grad, "grad", param);
    want_f32(exp_avg, "exp_avg", param);
    want_f32(exp_avg_sq, "exp_avg_sq", param);
    want(step, "step", at::kLong, param);
    TORCH_CHECK(grad.numel() == param.numel() && exp_avg.numel() == param.numel() && exp_avg_sq.numel() == param.numel() && step.numel() >= 2,
                "AdamW buffers must have param's size; step must be int64[>= 2]");
    const c10::hip::HIPGuardMasqueradingAsCUDA guard(param.device());
    pfn_ok(pfn_adamw_step(param.data_ptr<float>(), grad.data_ptr<float>(), exp_avg.data_ptr<float>(), exp_avg_sq.data_ptr<float>(), param.numel(),
                          (float)lr, (float)beta1, (float)beta2, (float)eps, (float)weight_decay, step.data_ptr<int64_t>(), cur_stream(param)),
           "pfn_adamw_step");
}

// ---- the differentiable form: `pfn::mpn` = MaskEmbdMultiMPN.forward with its autograd edge (what utils/training.py:58 + :74 use
// as model(data) ... loss.backward()).  One autograd node for the whole network, like the Python mirror's `_MpnFn`: forward =
// mpn_forward (need_backward iff anything differentiable asks for it), backward = mpn_backward; the parameter gradients are views
// of ONE flat buffer (the data-parallel all-reduce unit).
struct MpnFunction : public torch::autograd::Function<MpnFunction> {
    // (a C++ autograd node differentiates its TENSOR arguments only -- not the members of a Tensor[] -- so the parameters arrive as
    //  ONE flat tensor, concatenated by the caller below with a differentiable cat, and are handed to the C ABI as views of it)
    static std::vector<Tensor> views_of(const Tensor& flat, const std::vector<int64_t>& numels) {
        std::vector<Tensor> v;
        int64_t off = 0;
        for (int64_t n : numels) {
            v.push_back(flat.narrow(0, off, n));
            off += n;
        }
        return v;
    }
    static Tensor forward(torch::autograd::AutogradContext* ctx, const Tensor& graph_ws, int64_t e_stored, int64_t seg_nodes,
                          std::vector<int64_t> dims, double dropout, bool training, const Tensor& flat_params, std::vector<int64_t> numels,
                          const Tensor& x, const Tensor& pred_mask, const Tensor& edge_attr, const c10::optional<Tensor>& rng_state,
                          bool need, bool validated) {
        // (need: decided by the caller -- inside `apply` grad mode is off and says nothing)
        at::AutoDispatchBelowADInplaceOrView guard;
        const std::vector<Tensor> params = views_of(flat_params, numels);
        auto res = mpn_forward(graph_ws, e_stored, seg_nodes, dims, dropout, training, need, params, x, pred_mask, edge_attr, rng_state, false, validated);
        if (need) {
            ctx->save_for_backward({graph_ws, x, pred_mask, edge_attr, std::get<1>(res), flat_params});
            ctx->saved_data["e_stored"] = e_stored;
            ctx->saved_data["seg_nodes"] = seg_nodes;
            ctx->saved_data["dims"] = dims;
            ctx->saved_data["numels"] = numels;
            ctx->saved_data["dropout"] = dropout;
            ctx->saved_data["training"] = training;
            ctx->saved_data["need_gx"] = x.requires_grad();
            ctx->saved_data["need_gea"] = edge_attr.requires_grad();
        }
        return std::get<0>(res);
    }
    static torch::autograd::variable_list backward(torch::autograd::AutogradContext* ctx, torch::autograd::variable_list grad_outputs) {
        const auto saved = ctx->get_saved_variables();
        TORCH_CHECK(saved.size() == 6, "pfn::mpn: backward through a forward that recorded nothing");
        const Tensor &graph_ws = saved[0], &x = saved[1], &pred_mask = saved[2], &edge_attr = saved[3], &ws = saved[4];
        const std::vector<Tensor> params = views_of(saved[5], ctx->saved_data["numels"].toIntVector());
        const bool need_gx = ctx->saved_data["need_gx"].toBool(), need_gea = ctx->saved_data["need_gea"].toBool();
        const std::vector<int64_t> dims = ctx->saved_data["dims"].toIntVector();
        auto res = mpn_backward(graph_ws, ctx->saved_data["e_stored"].toInt(), ctx->saved_data["seg_nodes"].toInt(), dims,
                                ctx->saved_data["dropout"].toDouble(), ctx->saved_data["training"].toBool(), params, x, pred_mask, edge_attr,
                                grad_outputs[0].contiguous(), ws, need_gx, need_gea);
        // inputs in order: graph_ws, e_stored, seg_nodes, dims, dropout, training, flat_params, numels, x, pred_mask, edge_attr,
        // rng_state, need, validated
        torch::autograd::variable_list grads(14, Tensor());
        grads[6] = std::get<0>(res);            // the flat gradient of every parameter (the cat's backward hands out its slices)
        grads[8] = std::get<1>(res);            // x
        grads[10] = std::get<2>(res);           // edge_attr
        return grads;
    }
};
Tensor mpn_autograd(const Tensor& graph_ws, int64_t e_stored, int64_t seg_nodes, at::IntArrayRef dims, double dropout, bool training,
                    at::TensorList params, const Tensor& x, const Tensor& pred_mask, const Tensor& edge_attr,
                    const c10::optional<Tensor>& rng_state, bool validated) {
    bool need = x.requires_grad() || edge_attr.requires_grad();
    std::vector<Tensor> flats;
    std::vector<int64_t> numels;
    for (const Tensor& p : params) {
        need = need || p.requires_grad();
        flats.push_back(p.reshape({-1}));
        numels.push_back(p.numel());
    }
    need = need && at::GradMode::is_enabled();
    return MpnFunction::apply(graph_ws, e_stored, seg_nodes, dims.vec(), dropout, training, at::cat(flats), numels, x, pred_mask, edge_attr,
                              rng_state, need, validated);
}

// ---- the two layers as differentiable operators (what the reference's other model classes compose, networks/MPN.py:143-453)
struct EdgeAggrFunction : public torch::autograd::Function<EdgeAggrFunction> {
    static Tensor forward(torch::autograd::AutogradContext* ctx, const Tensor& graph_ws, int64_t e_stored, const Tensor& x,
                          const Tensor& edge_attr, const Tensor& w1, const Tensor& b1, const Tensor& w2, const Tensor& b2) {
        at::AutoDispatchBelowADInplaceOrView guard;
        auto res = edge_aggr_forward(graph_ws, e_stored, x, edge_attr, w1, b1, w2, b2);
        ctx->save_for_backward({graph_ws, x, edge_attr, w1, b1, w2, b2, std::get<1>(res)});
        ctx->saved_data["e_stored"] = e_stored;
        return std::get<0>(res);
    }
    static torch::autograd::variable_list backward(torch::autograd::AutogradContext* ctx, torch::autograd::variable_list go) {
        const auto sv = ctx->get_saved_variables();
        auto r = edge_aggr_backward(sv[0], ctx->saved_data["e_stored"].toInt(), sv[1], sv[2], sv[3], sv[4], sv[5], sv[6], go[0].contiguous(), sv[7]);
        // inputs: graph_ws, e_stored, x, edge_attr, w1, b1, w2, b2
        return {Tensor(), Tensor(), std::get<0>(r), std::get<1>(r), std::get<2>(r), std::get<3>(r), std::get<4>(r), std::get<5>(r)};
    }
};
Tensor edge_aggr_autograd(const Tensor& graph_ws, int64_t e_stored, const Tensor& x, const Tensor& edge_attr, const Tensor& w1, const Tensor& b1,
                          const Tensor& w2, const Tensor& b2) {
    return EdgeAggrFunction::apply(graph_ws, e_stored, x, edge_attr, w1, b1, w2, b2);
}
struct TagConvFunction : public torch::autograd::Function<TagConvFunction> {
    // (the K + 1 weights as ONE flat tensor: see MpnFunction)
    static Tensor forward(torch::autograd::AutogradContext* ctx, const Tensor& graph_ws, int64_t e_stored, int64_t seg_nodes, const Tensor& x,
                          const Tensor& flat_w, int64_t nw, const c10::optional<Tensor>& bias) {
        at::AutoDispatchBelowADInplaceOrView guard;
        const int64_t cin = x.size(1), cout = flat_w.numel() / (nw * cin);
        std::vector<Tensor> ws_;
        for (int64_t k = 0; k < nw; ++k) ws_.push_back(flat_w.narrow(0, k * cout * cin, cout * cin).view({cout, cin}));
        auto res = tag_conv_forward(graph_ws, e_stored, seg_nodes, x, ws_, bias);
        const bool has_bias = bias.has_value() && bias->defined();
        ctx->save_for_backward({graph_ws, x, flat_w, std::get<1>(res)});
        ctx->saved_data["e_stored"] = e_stored;
        ctx->saved_data["seg_nodes"] = seg_nodes;
        ctx->saved_data["nw"] = nw;
        ctx->saved_data["has_bias"] = has_bias;
        return std::get<0>(res);
    }
    static torch::autograd::variable_list backward(torch::autograd::AutogradContext* ctx, torch::autograd::variable_list go) {
        const auto sv = ctx->get_saved_variables();
        const int64_t nw = ctx->saved_data["nw"].toInt(), cin = sv[1].size(1), cout = sv[2].numel() / (nw * cin);
        std::vector<Tensor> ws_;
        for (int64_t k = 0; k < nw; ++k) ws_.push_back(sv[2].narrow(0, k * cout * cin, cout * cin).view({cout, cin}));
        auto r = tag_conv_backward(sv[0], ctx->saved_data["e_stored"].toInt(), ctx->saved_data["seg_nodes"].toInt(), sv[1], ws_, go[0].contiguous(),
                                   sv[3], ctx->saved_data["has_bias"].toBool());
        std::vector<Tensor> gflat;
        for (const Tensor& g : std::get<2>(r)) gflat.push_back(g.reshape({-1}));
        // inputs: graph_ws, e_stored, seg_nodes, x, flat_w, nw, bias
        return {Tensor(), Tensor(), Tensor(), std::get<0>(r), at::cat(gflat), Tensor(), std::get<1>(r)};
    }
};
Tensor tag_conv_autograd(const Tensor& graph_ws, int64_t e_stored, int64_t seg_nodes, const Tensor& x, at::TensorList weights,
                         const c10::optional<Tensor>& bias) {
    TORCH_CHECK(!weights.empty(), "weights must be non-empty");
    std::vector<Tensor> flats;
    for (const Tensor& w : weights) flats.push_back(w.reshape({-1}));
    return TagConvFunction::apply(graph_ws, e_stored, seg_nodes, x, at::cat(flats), (int64_t)weights.size(), bias);
}

int64_t abi_version() { return pfn_abi_version(); }

}  // namespace

TORCH_LIBRARY(pfn, m) {
    m.def("abi_version() -> int", &abi_version);
    m.def("graph_build(Tensor edge_index, int num_nodes, int mode=-1) -> Tensor");
    m.def("graph_check(Tensor graph_ws, int num_nodes, int e_stored) -> (bool, int)");
    m.def("graph_segments(Tensor(a!) graph_ws, int num_nodes, int e_stored, int seg_nodes) -> bool");
    m.def("mpn_forward(Tensor graph_ws, int e_stored, int seg_nodes, int[] dims, float dropout, bool training, bool need_backward, "
          "Tensor[] params, Tensor x, Tensor pred_mask, Tensor edge_attr, Tensor? rng_state=None, bool defer_out=False, bool validated=False) -> (Tensor, Tensor)");
    m.def("mpn_mse_tail_ok(int num_nodes, int e_stored, int seg_nodes, int[] dims, float dropout, bool training) -> bool", &mpn_mse_tail_ok);
    m.def("mpn_backward_mse(Tensor graph_ws, int e_stored, int seg_nodes, int[] dims, float dropout, bool training, Tensor[] params, Tensor x, "
          "Tensor edge_attr, Tensor y, Tensor(a!) out, Tensor ws, Tensor(b!) loss_ws, bool need_grad_x=False) -> (Tensor, Tensor, Tensor, Tensor)");
    m.def("mpn_backward(Tensor graph_ws, int e_stored, int seg_nodes, int[] dims, float dropout, bool training, Tensor[] params, Tensor x, "
          "Tensor pred_mask, Tensor edge_attr, Tensor grad_out, Tensor ws, bool need_grad_x=False, bool need_grad_edge_attr=False) "
          "-> (Tensor, Tensor, Tensor)");
    m.def("edge_aggr_forward(Tensor graph_ws, int e_stored, Tensor x, Tensor edge_attr, Tensor w1, Tensor b1, Tensor w2, Tensor b2) -> (Tensor, Tensor)");
    m.def("edge_aggr_backward(Tensor graph_ws, int e_stored, Tensor x, Tensor edge_attr, Tensor w1, Tensor b1, Tensor w2, Tensor b2, "
          "Tensor grad_out, Tensor ws) -> (Tensor, Tensor, Tensor, Tensor, Tensor, Tensor)");
    m.def("tag_conv_forward(Tensor graph_ws, int e_stored, int seg_nodes, Tensor x, Tensor[] weights, Tensor? bias=None) -> (Tensor, Tensor)");
    m.def("tag_conv_backward(Tensor graph_ws, int e_stored, int seg_nodes, Tensor x, Tensor[] weights, Tensor grad_out, Tensor ws, "
          "bool has_bias=True) -> (Tensor, Tensor, Tensor[])");
    m.def("scatter_add(Tensor graph_ws, int e_stored, Tensor x) -> Tensor");
    m.def("edge_aggr(Tensor graph_ws, int e_stored, Tensor x, Tensor edge_attr, Tensor w1, Tensor b1, Tensor w2, Tensor b2) -> Tensor");
    m.def("tag_conv(Tensor graph_ws, int e_stored, int seg_nodes, Tensor x, Tensor[] weights, Tensor? bias=None) -> Tensor");
    m.def("mpn(Tensor graph_ws, int e_stored, int seg_nodes, int[] dims, float dropout, bool training, Tensor[] params, Tensor x, "
          "Tensor pred_mask, Tensor edge_attr, Tensor? rng_state=None, bool validated=False) -> Tensor");
    m.def("mse_loss(Tensor out, Tensor y, Tensor(a!) ws) -> (Tensor, Tensor)");
    m.def("adamw_step_(Tensor(a!) param, Tensor grad, Tensor(b!) exp_avg, Tensor(c!) exp_avg_sq, float lr, float beta1, float beta2, float eps, "
          "float weight_decay, Tensor(d!) step) -> ()");
}

// One backend key: these operators exist for HIP tensors only (a CPU tensor finds no kernel -> the dispatcher's own RuntimeError)
TORCH_LIBRARY_IMPL(pfn, CUDA, m) {
    m.impl("graph_build", &graph_build);
    m.impl("graph_check", &graph_check);
    m.impl("graph_segments", &graph_segments);
    m.impl("mpn_forward", &mpn_forward);
    m.impl("mpn_backward", &mpn_backward);
    m.impl("mpn_backward_mse", &mpn_backward_mse);
    m.impl("edge_aggr_forward", &edge_aggr_forward);
    m.impl("edge_aggr_backward", &edge_aggr_backward);
    m.impl("tag_conv_forward", &tag_conv_forward);
    m.impl("tag_conv_backward", &tag_conv_backward);
    m.impl("scatter_add", &scatter_add);
    m.impl("mse_loss", &mse_loss);
    m.impl("adamw_step_", &adamw_step_);
}

// `pfn::mpn` carries its own autograd node (MpnFunction); with nothing differentiable in sight it is a plain inference forward
TORCH_LIBRARY_IMPL(pfn, Autograd, m) {
    m.impl("mpn", &mpn_autograd);
    m.impl("edge_aggr", &edge_aggr_autograd);
    m.impl("tag_conv", &tag_conv_autograd);
}
TORCH_LIBRARY_IMPL(pfn, CUDA, m) {
    m.impl("mpn", [](const Tensor& graph_ws, int64_t e_stored, int64_t seg_nodes, at::IntArrayRef dims, double dropout, bool training,
                     at::TensorList params, const Tensor& x, const Tensor& pred_mask, const Tensor& edge_attr,
                     const c10::optional<Tensor>& rng_state, bool validated) {
        return std::get<0>(mpn_forward(graph_ws, e_stored, seg_nodes, dims, dropout, training, false, params, x, pred_mask, edge_attr, rng_state,
                                       false, validated));
    });
    m.impl("edge_aggr", [](const Tensor& graph_ws, int64_t e_stored, const Tensor& x, const Tensor& edge_attr, const Tensor& w1, const Tensor& b1,
                           const Tensor& w2, const Tensor& b2) { return std::get<0>(edge_aggr_forward(graph_ws, e_stored, x, edge_attr, w1, b1, w2, b2)); });
    m.impl("tag_conv", [](const Tensor& graph_ws, int64_t e_stored, int64_t seg_nodes, const Tensor& x, at::TensorList weights,
                          const c10::optional<Tensor>& bias) { return std::get<0>(tag_conv_forward(graph_ws, e_stored, seg_nodes, x, weights, bias)); });
}
