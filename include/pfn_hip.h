/*
 * pfn_hip.h -- C ABI of libpfn_hip.so: the MI355X (gfx950) implementation of PowerFlowNet's
 * message-passing hot path (MaskEmbdMultiMPN forward + backward).
 *
 * The reference has no FFI of its own: its boundary is the Python class surface of
 * networks/MPN.py.  Each entry point below names the reference interface it
 * replaces (file:line into /root/reference).  The host-side mirror of that class surface
 * (poweflownet_amd/networks/MPN.py) binds these symbols with ctypes; INTEGRATION.md shows the stub.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (except `params`/`grads` tables, which are
 *     HOST arrays of device pointers, and out-parameters documented as host);
 *   - all floating point is fp32; node/edge ids arrive as int64 (the PyG layout) and are narrowed to
 *     int32 inside pfn_graph_build;
 *   - work is enqueued on `stream` (a hipStream_t passed as void*); no entry point synchronises,
 *     allocates or frees device memory except where stated, so every call is hipGraph-capturable;
 *   - activations handed between layers use a padded row stride ld = pfn_padded_ld(F) = roundup(F, 4)
 *     floats whose pad columns are kept zero;
 *   - return value: 0 on success, a negative PFN_E* code otherwise; pfn_last_error() gives the text.
 */
#ifndef PFN_HIP_H
#define PFN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PFN_ABI_VERSION 8

enum {
    PFN_OK = 0,
    PFN_EINVAL = -1,      /* bad argument (null pointer, misaligned ld, unsupported dimension) */
    PFN_ENOSPACE = -2,    /* a caller-provided workspace is too small */
    PFN_EHIP = -3,        /* a HIP runtime call failed */
    PFN_EINDEX = -4       /* edge_index holds a node id outside [0, n_nodes) */
};

/* Model hyper-parameters: the constructor arguments of MaskEmbdMultiMPN (networks/MPN.py:462-470). */
typedef struct pfn_mpn_config {
    int32_t nfeature_dim;   /* node input width (4, asserted networks/MPN.py:528) */
    int32_t efeature_dim;   /* edge feature width (2) */
    int32_t output_dim;     /* node output width (4) */
    int32_t hidden_dim;     /* H */
    int32_t n_gnn_layers;   /* L >= 2 (L == 1 is shape-broken in the reference, networks/MPN.py:475-477) */
    int32_t K;              /* TAGConv hops */
    float dropout_rate;     /* p of nn.Dropout (networks/MPN.py:496) */
    int32_t training;       /* 1: dropout active (model.train()), 0: model.eval() */
    int32_t need_backward;  /* 1: pfn_mpn_backward will be called on this forward's workspace (autograd records the call): the
                             * forward edge walks also save their ReLU masks, which the backward walks then read instead of
                             * recomputing the edge pre-activations.  0: inference: nothing extra is written and tensors only
                             * the backward reads (mask_embd's hidden layer) are not stored; pfn_mpn_backward then returns
                             * PFN_EINVAL.  pfn_mpn_backward must be given the value the forward call had: the forward stamps
                             * its workspace on the device, and a backward pass on a workspace whose LAST forward ran with
                             * need_backward = 0 (the host cannot tell) writes every parameter gradient as NaN.          */
} pfn_mpn_config;

/* The library keeps NO process-global mutable state: no stream, event, cache or "first caller" device binding of its own
 * (per-device facts such as the CU count are looked up per call; kernel attributes are raised once per device, lock-free).
 * Every call works on the caller's stream and the caller's workspaces only, so two models on two devices or two host
 * threads may run concurrently as long as they do not share a workspace.                                              */
int pfn_abi_version(void);
const char* pfn_last_error(void);            /* thread-local, valid until the next failing call */
int64_t pfn_padded_ld(int64_t features);     /* roundup(features, 4) */

/* ------------------------------------------------------------------------------------------ graph
 * Replaces MaskEmbdMultiMPN.is_directed + undirect_graph (networks/MPN.py:498-523) and the per-call
 * PyG bookkeeping underneath propagate()/gcn_norm (degree scatter, index_select lifting):
 * one pass turns the stored edge list into destination-sorted and source-sorted CSR adjacency
 * (rows ordered by edge id, so segment sums run in the reference's edge order), in-degree and
 * D^-1/2.  `mode`: -1 = apply the reference's first-edge heuristic on device (no host sync),
 * 0 = use the list as given, 1 = always append the reversed copies.                              */
size_t pfn_graph_workspace_bytes(int64_t n_nodes, int64_t e_stored);
int pfn_graph_build(const int64_t* edge_index /* [2, e_stored] */, int64_t e_stored, int64_t n_nodes,
                    int mode, void* graph_ws, size_t graph_ws_bytes, void* stream);
/* Synchronising debug/validation read-back (host out-params): directed flag, effective edge count,
 * error flag (non-zero when an id was out of range -> PFN_EINDEX).                               */
int pfn_graph_info(const void* graph_ws, int64_t n_nodes, int64_t e_stored, int32_t* directed,
                   int64_t* e_effective, void* stream);
/* Synchronising check (once per topology): *ok = 1 iff n_nodes % seg_nodes == 0 and no effective edge crosses a multiple
 * of seg_nodes, i.e. the batch is a disjoint union of index-contiguous graphs of seg_nodes nodes (what PyG's Batch of one
 * reference case is).  A caller that got ok may pass seg_nodes to the model / TAGConv entry points, which then keep the K
 * propagation hops of a TAGConv resident in LDS per graph instead of running K gather kernels over HBM/L2.            */
int pfn_graph_segments(void* graph_ws, int64_t n_nodes, int64_t e_stored, int64_t seg_nodes, int32_t* ok, void* stream);
/* The same check WITHOUT the read-back, for callers that cannot synchronise (a topology that changes per batch inside a captured
 * hipGraph -- the reference's `perturbed` datasets, dataset_generator.py:250-253, utils/data_utils.py:12-59): the verdict stays in
 * the workspace.  The caller passes seg_nodes to the model on trust and ends its forward pass with pfn_graph_poison_if_bad, which
 * overwrites `out` (count floats) with NaN when the workspace records a node id outside [0, n_nodes) or an edge that crosses a
 * segment boundary -- so a bad batch surfaces as a NaN loss instead of a silently wrong one (pfn_graph_info still reports the
 * id error, with a sync, whenever the caller can afford one).  Both calls are hipGraph-capturable.  seg_nodes = 0 (ABI 8): no
 * segment promise is made -- the verdict a previous check left in this workspace is cleared, nothing is checked.               */
int pfn_graph_segments_async(void* graph_ws, int64_t n_nodes, int64_t e_stored, int64_t seg_nodes, void* stream);
int pfn_graph_poison_if_bad(const void* graph_ws, int64_t n_nodes, int64_t e_stored, float* out, int64_t count, void* stream);
/* Copies the effective (post-undirect) edge list back out as int64 [2, 2*e_stored] (tests). */
int pfn_graph_export_edges(const void* graph_ws, int64_t n_nodes, int64_t e_stored,
                           int64_t* edge_index_out, void* stream);

/* -------------------------------------------------------------------------------------- whole model
 * Parameter table order (host array of device pointers), mirroring the module tree of
 * networks/MPN.py:462-496:
 *   for each layer i of `layers`:  EdgeAggregation -> W1 (H, 2*Fi+Fe), b1 (H), W2 (Fo, H), b2 (Fo)
 *                                  TAGConv         -> W_0 .. W_K (H, H) each, bias (H)
 *   then mask_embd: Wa (H, F0), ba (H), Wb (F0, H), bb (F0).
 * All in the nn.Linear (out, in) row-major layout of the state_dict.  pfn_mpn_num_params gives the
 * table length; `grads` uses the same order.                                                      */
int pfn_mpn_num_params(const pfn_mpn_config* cfg);
size_t pfn_mpn_workspace_bytes(const pfn_mpn_config* cfg, int64_t n_nodes, int64_t e_stored);

/* MaskEmbdMultiMPN.forward (networks/MPN.py:525-559), including EdgeAggregation.forward/message
 * (:23-56) and TAGConv.forward.  x [N, F0] f32, pred_mask [N, F0] (mask_dtype 0: int64, the dataset
 * layout of datasets/PowerFlowData.py:193; 1: float32), edge_attr [e_stored, Fe] f32, out [N, output_dim] f32.  `ws` receives the activations backward needs.  `rng_state`: device
 * uint64[2] {seed, offset}; when training && dropout_rate > 0 the offset is advanced by one at the start of the call
 * and then read by the dropout epilogues.  `out` may be NULL where pfn_mpn_mse_tail_ok answers 1 and
 * pfn_mpn_backward_mse follows (which then writes the rows).                                        */
int pfn_mpn_forward(const pfn_mpn_config* cfg, const void* graph_ws, int64_t n_nodes, int64_t e_stored,
                    const float* const* params, const float* x, const void* pred_mask,
                    int mask_dtype, const float* edge_attr, float* out, void* ws, size_t ws_bytes,
                    uint64_t* rng_state, int64_t seg_nodes /* 0, or a value pfn_graph_segments accepted */, void* stream);

/* Autograd of the above (what loss.backward() runs, utils/training.py:74): grad_out [N, output_dim];
 * writes every entry of `grads` (overwrites, does not accumulate); grad_x [N, F0] and
 * grad_edge_attr [e_stored, Fe] are optional (NULL to skip).  `ws` is the buffer forward filled.   */
int pfn_mpn_backward(const pfn_mpn_config* cfg, const void* graph_ws, int64_t n_nodes, int64_t e_stored,
                     const float* const* params, float* const* grads, const float* x,
                     const void* pred_mask, int mask_dtype, const float* edge_attr, const float* grad_out,
                     float* grad_x, float* grad_edge_attr, void* ws, size_t ws_bytes, int64_t seg_nodes, void* stream);

/* `loss = torch.nn.MSELoss()(out, y); loss.backward()` (train.py:103; the else-branch of train_epoch, utils/training.py:70-74)
 * together with the autograd pass above, for batches of small graphs: the first backward launch -- the graph-resident
 * EdgeAggregation backward of the LAST layer -- forms the output rows `out = S W2^T + deg b2` (networks/MPN.py:559 hands them to
 * the loss), the loss and `grad_out = 2 (out - y) / (4 N)` itself, so the output Linear's launch and the loss launch leave the
 * step (three launches -> one).  Call pfn_mpn_forward on the same workspace first; its `out` may then be NULL (the rows are only
 * written here).  Results: `out` bit-identical to pfn_mpn_forward's, `grad_out` and every entry of `grads` bit-identical to
 * pfn_mpn_forward -> pfn_mse_loss -> pfn_mpn_backward; loss[0] = mean((out - y)^2) summed per row block, the blocks in block
 * order (deterministic; not pfn_mse_loss's partition, so equal to its value up to the rounding of a different summation order).
 * y, out, grad_out: [N, 4] f32, no padding (output_dim must be 4).  `loss_ws`: >= 4100 bytes -- 1024 float partials + one int32
 * arrival counter at byte 4096 that must be ZERO before the first call and is left zero by every call.  grad_x optional.
 * Available where pfn_mpn_mse_tail_ok returns 1 (seg_nodes from pfn_graph_segments, Fe = 2, output_dim 4, the batch in the
 * graph-resident regime); PFN_EINVAL elsewhere -- the caller then runs the three calls above.                              */
int pfn_mpn_mse_tail_ok(const pfn_mpn_config* cfg, int64_t n_nodes, int64_t e_stored, int64_t seg_nodes);
/* The same with `Masked_L2_loss(regularize, regcoeff)(out, y, pred_mask)` (utils/custom_loss_functions.py:10-46, the default
 * --train_loss_fn, dispatch utils/training.py:61-62) as the loss, `pred_mask` being THE mask the forward call was given (its
 * first launch kept the float mask rows and the per-block counts of the two index sets in `ws`): loss and grad as
 * pfn_masked_l2_loss defines them -- grad_out bit-identical to it, loss[0] to the rounding of another summation order.
 * `loss_ws`: >= 8196 bytes (2048 float partials + the int32 arrival counter at byte 8192, zero between calls).         */
int pfn_mpn_backward_masked_l2(const pfn_mpn_config* cfg, const void* graph_ws, int64_t n_nodes, int64_t e_stored,
                               const float* const* params, float* const* grads, const float* x, const float* edge_attr,
                               const float* y, int regularize, float regcoeff, float* out, float* loss, float* grad_out,
                               float* grad_x, void* ws, size_t ws_bytes, void* loss_ws, size_t loss_ws_bytes,
                               int64_t seg_nodes, void* stream);
int pfn_mpn_backward_mse(const pfn_mpn_config* cfg, const void* graph_ws, int64_t n_nodes, int64_t e_stored,
                         const float* const* params, float* const* grads, const float* x, const float* edge_attr,
                         const float* y, float* out, float* loss, float* grad_out, float* grad_x, void* ws, size_t ws_bytes,
                         void* loss_ws, size_t loss_ws_bytes, int64_t seg_nodes, void* stream);

/* Verification aid for the autograd above (what loss.backward() differentiates, utils/training.py:74; the ReLUs are
 * networks/MPN.py:19 (edge MLP), :547 (layer outputs) and :493 (mask_embd)): the ReLU gate decisions of the forward pass that
 * filled `ws`, as bytes (1 = the unit passed), so that a float64 run of the CPU oracle can be held to the same piecewise-linear
 * branch and every gradient compared at 1e-5.  Call after pfn_mpn_forward (need_backward = 1), before `ws` is reused.
 *   kind 0: edge stage of EdgeAggregation layer `layer` (index into `layers`): out [E_effective][H] in edge-id order
 *           (originals first, reversed copies second -- the order undirect_graph produces, networks/MPN.py:506-523);
 *           out must hold 2 * e_stored * H bytes;
 *   kind 1: output of hidden layer `layer` after dropout -> ReLU (:546-547): out [N][H];
 *   kind 2: mask_embd's hidden layer (:493): out [N][H].                                                              */
int pfn_mpn_export_gates(const pfn_mpn_config* cfg, const void* graph_ws, int64_t n_nodes, int64_t e_stored,
                         const float* const* params, const float* edge_attr, void* ws, size_t ws_bytes,
                         int64_t seg_nodes /* the value the forward call had: it decides the layout of saved tensors */,
                         int32_t kind, int32_t layer, uint8_t* out, void* stream);

/* ------------------------------------------------------------------------------------- single layers
 * EdgeAggregation(nfeature_dim, efeature_dim, hidden_dim, output_dim).forward (networks/MPN.py:30-56):
 *   out[i] = sum_{e -> i} ( W2 relu(W1 [x_i ; x_src(e) ; a_e] + b1) + b2 ).
 * x [N, ldx] (ldx = pfn_padded_ld(Fi), pad zero), out [N, ldo].  `ws` (pfn_edge_aggr_workspace_bytes)
 * keeps P|Q and S for backward.  The graph must have been built from the list the layer is given
 * (mode 0 when the caller already undirected it).                                                  */
size_t pfn_edge_aggr_workspace_bytes(int64_t n_nodes, int64_t e_stored, int fi, int fe, int h, int fo);
int pfn_edge_aggr_forward(const void* graph_ws, int64_t n_nodes, int64_t e_stored, int fi, int fe, int h,
                          int fo, const float* x, int64_t ldx, const float* edge_attr, const float* w1,
                          const float* b1, const float* w2, const float* b2, float* out, int64_t ldo,
                          void* ws, size_t ws_bytes, void* stream);
int pfn_edge_aggr_backward(const void* graph_ws, int64_t n_nodes, int64_t e_stored, int fi, int fe, int h,
                           int fo, const float* x, int64_t ldx, const float* edge_attr, const float* w1,
                           const float* b1, const float* w2, const float* b2, const float* grad_out,
                           int64_t ldgo, float* grad_x, int64_t ldgx, float* grad_edge_attr, float* grad_w1,
                           float* grad_b1, float* grad_w2, float* grad_b2, void* ws, size_t ws_bytes,
                           void* stream);

/* TAGConv(in, out, K).forward (PyG; call sites networks/MPN.py:477-484,:545):
 *   out = sum_k (A_hat^k x) W_k^T + b,  A_hat = D^-1/2 A D^-1/2, no self loops.
 * weights: host array of K+1 device pointers (lins.k.weight, (out, in) each).                     */
size_t pfn_tag_conv_workspace_bytes(int64_t n_nodes, int64_t e_stored, int cin, int cout, int K);
int pfn_tag_conv_forward(const void* graph_ws, int64_t n_nodes, int64_t e_stored, int cin, int cout, int K,
                         const float* x, int64_t ldx, const float* const* weights, const float* bias,
                         float* out, int64_t ldo, void* ws, size_t ws_bytes, int64_t seg_nodes, void* stream);
int pfn_tag_conv_backward(const void* graph_ws, int64_t n_nodes, int64_t e_stored, int cin, int cout, int K,
                          const float* x, int64_t ldx, const float* const* weights, const float* grad_out,
                          int64_t ldgo, float* grad_x, int64_t ldgx, float* const* grad_weights,
                          float* grad_bias, void* ws, size_t ws_bytes, int64_t seg_nodes, void* stream);

/* ----------------------------------------------------------------------------------------- utilities
 * The segmented scatter-add in isolation (PyG SumAggregation / scatter_add_ under propagate):
 * out[i] = sum_{e -> i} x[src(e)], F columns, ld = pfn_padded_ld(F).  Used for the roofline run.  */
int pfn_scatter_add(const void* graph_ws, int64_t n_nodes, int64_t e_stored, const float* x, float* out,
                    int64_t features, void* stream);
/* Row (un)padding between the caller's dense [N, F] tensors and the internal [N, ld] layout.      */
int pfn_pad_rows(const float* src, int64_t ld_src, float* dst, int64_t ld_dst, int64_t n_rows,
                 int64_t features, void* stream);
/* MSELoss(out, y) forward+backward in one pass (train.py:103; utils/training.py:72-74):
 * loss[0] = mean((out-y)^2), grad[i] = 2 (out[i]-y[i]) / count (grad may be NULL).  `ws`: >= 1028 bytes,
 * 256 float partials + one int32 arrival counter at byte 1024 that must be ZERO before the first call and is
 * left zero by every call (one launch: the last block to arrive sums the partials in block order).           */
int pfn_mse_loss(const float* out, const float* y, int64_t count, float* loss, float* grad, void* ws,
                 size_t ws_bytes, void* stream);
/* Masked_L2_loss(output, target, mask) forward+backward (utils/custom_loss_functions.py:10-46; the reference's default
 * --train_loss_fn, utils/argument_parser.py:36; dispatch utils/training.py:61-62), d = out - y:
 *   loss[0] = mean over {mask != 0} of d^2  +  (regularize ? regcoeff * mean over {(1 - mask) != 0} of d^2 : 0)
 *   grad[i] = 2 d_i ( [mask_i != 0] / n1 + regcoeff [mask_i != 1] / n0 )          (grad may be NULL)
 * An empty set gives NaN, as torch's mean of nothing does.  mask: `count` entries, mask_dtype 0 = int64, 1 = float32.
 * `ws`: >= 4128 bytes; its last int32 (byte 4124) is an arrival counter that must be ZERO before the first call and is
 * left zero by every call.                                                                                          */
int pfn_masked_l2_loss(const float* out, const float* y, const void* mask, int mask_dtype, int64_t count,
                       int regularize, float regcoeff, float* loss, float* grad, void* ws, size_t ws_bytes,
                       void* stream);
/* PowerImbalance(x, edge_index, edge_attr) forward+backward (utils/custom_loss_functions.py:99-286; --train_loss_fn
 * power_imbalance, train.py:95-97; dispatch utils/training.py:63-67).  `graph_ws`: the adjacency pfn_graph_build made from
 * the SAME stored-once edge_index with mode -1 (the class undirects by the model's rule, :136-157).  x [N, 4] =
 * normalised (Vm, Va deg, P, Q), edge_attr [e_stored, 2] = normalised (r, x), stats = 12 host floats
 * {xymean[4], xystd[4], edgemean[2], edgestd[2]} (de-normalisation x * std + mean, :127-132).
 *   loss[0] = mean_i (dP_i^2 + dQ_i^2),  grad_x [N, 4] = d loss / d x (NULL to skip).
 * dpq: scratch [N, 2] floats.  `ws`: >= 1280 bytes whose int32 at byte 1024 is an arrival counter that must be ZERO before
 * the first call and is left zero by every call.                                                                       */
int pfn_power_imbalance(const void* graph_ws, int64_t n_nodes, int64_t e_stored, const float* x, const float* edge_attr,
                        const float* stats, float* loss, float* grad_x, float* dpq, void* ws, size_t ws_bytes, void* stream);
/* Dropout bookkeeping (nn.Dropout, networks/MPN.py:496,546-547).  The train-mode epilogues draw their mask from
 * Philox4x32-10 with key {seed[31:0], seed[63:32] ^ offset[63:32]} and counter {row, column / 4, layer index i of
 * `layers`, offset[31:0]}; an element is KEPT (and scaled by 1/(1-p)) iff its uniform >= p.  This debug/verification entry
 * writes the keep mask (1.0 / 0.0, [rows, ncols] unpadded) layer `layer` uses for the CURRENT {seed, offset} of
 * `rng_state` (after a training forward: the mask that forward applied), so a train-mode pass can be replayed elsewhere. */
int pfn_dropout_mask(const uint64_t* rng_state, int32_t layer, int64_t rows, int64_t ncols, float p, float* keep,
                     void* stream);
/* AdamW on one flat buffer (train.py:123; torch defaults betas (0.9,0.999), eps 1e-8, wd 0.01).
 * `step` is a device int64[2] {completed steps, arrival scratch (zero)}; the call increments step[0] itself,
 * so one launch per update and the whole step stays hipGraph-replayable.                           */
int pfn_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t count,
                   float lr, float beta1, float beta2, float eps, float weight_decay, int64_t* step,
                   void* stream);
/* The same update with the five scalars {lr, beta1, beta2, eps, weight_decay} read from DEVICE memory (`hyper`, float[5]):
 * a launch captured into a hipGraph then follows a learning-rate schedule (train.py:129,145: OneCycleLR) by a 20-byte copy
 * into `hyper` between replays instead of a new capture.                                                              */
int pfn_adamw_step_dev(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t count,
                       const float* hyper, int64_t* step, void* stream);
/* ... and skipped ON THE DEVICE (nothing updated, the step not counted) when the device scalar `guard` -- normally the step's
 * loss -- is not finite: for a captured training step whose batch could not be validated on the host (a topology per batch: a
 * bad batch arrives as a NaN loss, pfn_graph_poison_if_bad; the reference's train loop, utils/training.py:55-77, would have
 * raised in the forward pass instead of stepping).  `step` is a device int64[3] here: {completed steps, arrival scratch,
 * SKIPPED updates} -- the caller's loop reads step[2] to tell a poisoned / diverged batch from a healthy one (ABI 6).
 * New layer, no reference counterpart.                                                                                   */
int pfn_adamw_step_guarded(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t count,
                           const float* hyper, int64_t* step, const float* guard, void* stream);

/* ----------------------------------------------------------------------- Diagnostic environment switches
 * The library reads these environment variables (each ONCE per process, through one function, pfn::diag_env).  They select
 * between kernels that compute the SAME result -- the parity tests use them to hold a fused kernel against the generic one it
 * replaces, the tuning scripts under tools/ to sweep a launch parameter -- and none of them is needed in production: unset, the
 * library behaves as DESIGN.md describes.  There is no switch that routes work off the GPU or through another backend.
 *   PFN_NO_SEG_EA=1         EdgeAggregation of small-graph batches: generic gemm_nt + edge walks instead of the graph-resident kernels
 *   PFN_NO_SEG_LIN_HOPS=1   small-graph batches: the Linear in front of a TAGConv's hops and the hops as two launches (gemm_nt + fused hops)
 *                           instead of one graph-resident launch (seg_lin_hops.hip; bit-identical results)
 *   PFN_NO_FUSED_FRONT=1    mask_embd + first P|Q as generic GEMMs instead of front.hip's one launch
 *   PFN_NO_SEG_FRONT=1      small-graph batches: front.hip's launch + the generic first edge walk instead of the one graph-resident
 *                           launch that does both (ea_seg.hip front_seg_fwd_kernel; bit-identical results)
 *   PFN_NO_FUSED_BACK=1     the last layer's Linear / dS outside the edge walks (generic GEMMs)
 *   PFN_NO_MSE_TAIL=1       pfn_mpn_mse_tail_ok answers 0: the output Linear, MSELoss and the backward pass as three calls
 *   PFN_FRONT_BLOCK_ROWS=1  front.hip: the block-per-row-group kernels instead of one row per wave
 *   PFN_NO_BIG_HOPS=1       TAGConv hops of large graphs (one LDS tile + registers per graph and column chunk, workgroups persistent over a graph's chunks): K generic hop launches instead
 *   PFN_NO_ROW_HOPS=1       TAGConv hops of big batches of small graphs: the two-tile column-slice kernel instead of whole rows per block
 *   PFN_NO_EDGE_ROWS=1      the edge stage of big inference batches of small graphs: the generic gather kernel instead of the LDS-resident one
 *   PFN_NO_L0_FLY=1         the front writes the first layer's P | Q and the edge walk gathers them (default: the walk forms them from x0)
 *   PFN_EDGE_FWD_BPC=N      workgroups per CU of the persistent generic forward edge walk (default 8; a large N = one workgroup per 256 items)
 *   PFN_FRONT_STORE_MEH=1   training beyond 32 k rows: mask_embd's hidden layer is stored and its weight gradients go through gemm_tn
 *                           (default: recomputed in the backward front, which forms those gradients itself)
 *   PFN_FRONT_NO_THREAD_ROWS=1   inference front: the row-per-wave / block kernels instead of one row per thread
 *   PFN_NO_SERPENTINE=1     the row-streaming kernels (gemm_nt, LDS-resident walks / hops, the generic forward walk) all visit their rows
 *                           first to last (default: consecutive launches alternate, so a consumer starts with what its producer wrote last)
 *   PFN_NT_CT=1|2           gemm_nt: quarters per wave
 *   PFN_NO_NT_ILF=1         gemm_nt, one-piece tiles at two quarters per wave: every tile flushed behind its own multiply (default: parked
 *                           and flushed inside the wave's next multiply, between its own MFMAs)
 *   PFN_NO_NT_PAIR=1        gemm_nt, products of >= 3 terms at one quarter per wave: ONE fp32 chain through all terms (default: every
 *                           term summed on its own, the terms added in order -- PyG's TAGConv dataflow)
 *   PFN_NT_TINY_MAX_TILES=<n> gemm_nt: the split-K small-batch kernel up to n row tiles of 32 rows (default 256; 0 = never)
 *   PFN_NT_WS_MIN_TILES=<n> gemm_nt: weight-streaming kernel from n row tiles per wave (default 2; 0 = never)                     */

/* --------------------------------------------------------------------------------------- profiling
 * Optional HIP-event bracket around every kernel launch (same stream), aggregated per kernel class with
 * the launch's algorithmic bytes / flops.  Inactive during hipGraph capture.
 * pfn_profile_report synchronises the device and writes a JSON object into `buf`.                  */
int pfn_profile_enable(int on);
int pfn_profile_report(char* buf, size_t buf_bytes, int reset);

#ifdef __cplusplus
}
#endif
#endif /* PFN_HIP_H */
