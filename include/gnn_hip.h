/* gnn_hip.h - C ABI of libgnn_hip.so: the SegmentClassifier message-passing forward
 * of jmduarte/gnn-fpga (gnn/model.py) as hand-written HIP kernels for gfx950 (MI355X).
 *
 * The reference has no native boundary: its "FFI" for this path is the set of ATen calls
 * in gnn/model.py.  Each entry point below names the reference interface it replaces.
 * A maintainer binds it with ctypes; no C++ or torch types cross.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer into caller-owned memory (e.g. tensor.data_ptr());
 *     the library never allocates, frees or retains caller buffers;
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream);
 *     all work is asynchronous on it, no implicit synchronisation;
 *   - return value 0 = success; negative = -(hipError_t) or one of GNN_ERR_*;
 *     gnn_last_error() returns a thread-local message for the last failing call;
 *   - matrices are row-major float32; weights use torch's nn.Linear layout [out, in];
 *   - F = input_dim, D = hidden_dim, C = D + F; hit-feature rows H[n] = [H'(D) | X(F) | 0-pad]
 *     with row stride ldh = gnn_h_stride(F, D) floats (C rounded up to a multiple of 4);
 *   - a padded segment has src = dst = -1 (the all-zero Ri/Ro column of
 *     gnn/trainSegmentClassifier.py:83-93): it scores sigmoid(W2 tanh(b1) + b2) and
 *     contributes nothing to any hit.
 */
#ifndef GNN_HIP_H
#define GNN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GNN_ABI_VERSION 5

#define GNN_ERR_UNSUPPORTED (-10001) /* (F, D) has no kernel instantiation            */
#define GNN_ERR_BADARG      (-10002) /* null pointer, negative size, bad stride ...   */
#define GNN_ERR_WORKSPACE   (-10003) /* workspace smaller than gnn_forward_workspace_bytes */

/* Effective weights (mask already applied: W*mask, gnn/model.py:28-33), device pointers.
 * Shapes follow the reference state_dict. */
typedef struct gnn_params {
    const float *Win, *bin; /* input_network.0           [D,F]  [D]   gnn/model.py:132-134 */
    const float *W1, *b1;   /* edge_network.network.0    [D,2C] [D]   gnn/model.py:45-46   */
    const float *W2, *b2;   /* edge_network.network.2    [1,D]  [1]   gnn/model.py:48      */
    const float *W3, *b3;   /* node_network.network.0    [D,3C] [D]   gnn/model.py:94-95   */
    const float *W4, *b4;   /* node_network.network.2    [D,D]  [D]   gnn/model.py:97      */
    int32_t F, D;
    int32_t flags;          /* GNN_FLAG_* (fused pipeline only)                                */
} gnn_params_t;

/* The caller asserts max(|P'|, |Q'|) <= 60 for every hit, with P' = 2 log2(e) (W1[:, :C] H + b1),
 * Q' = 2 log2(e) W1[:, C:] H.  Since |H'| <= 1 (tanh) a sufficient condition is
 *   2 log2(e) * max_i ( sum_{k<D} |W1[i,k]| + sum_{k<F} |W1[i,D+k]| max|X[:,k]| + |b1[i]| ) <= 60
 * (same for the columns C..2C without the bias); gnn_exp_product_bound computes the left side.
 * With the flag set the fused kernels publish 2^P', 2^Q' and form 2^(P'+Q') by multiplication
 * (one transcendental less per hidden unit and segment); without it they use the exact path. */
#define GNN_FLAG_EXP_PRODUCT 1

/* hidden_dim 32 / 64 only (ignored elsewhere): run the hit update (gnn/model.py:120-125, and the
 * per-hit halves of the next edge / node layers) on the matrix cores with bf16 operands and fp32
 * accumulation (v_mfma_f32_16x16x32_bf16).  Weights are rounded to bf16 once per forward,
 * activations per use; the per-hit gather records travel as bf16, everything per segment is
 * computed in fp32.  NOT within the 1e-5 parity bound of the fp32 path: scores move by ~1e-3
 * (tests state the bound). */
#define GNN_FLAG_BF16_MLP 2

/* One block-diagonal batch of hit graphs in index form (replaces the dense Ri/Ro of
 * gnn/graph.py:28-35).  CSR arrays list, per hit, the segments ending (in_*) / starting
 * (out_*) there in ascending segment id, and the hit at the other end of each. */
typedef struct gnn_graph {
    const float *X;                            /* [n_hits, F]                              */
    const int32_t *src, *dst;                  /* [n_segments]  start / end hit, -1 = pad  */
    const int32_t *in_ptr, *in_eid, *in_nbr;   /* [n_hits+1], [n_valid], [n_valid] (=src[in_eid])  */
    const int32_t *out_ptr, *out_eid, *out_nbr;/* [n_hits+1], [n_valid], [n_valid] (=dst[out_eid]) */
    int64_t n_hits, n_segments;
} gnn_graph_t;

/* Gradient outputs, same shapes as gnn_params_t; the caller zero-initialises them and the
 * backward ADDS into them (device pointers). */
typedef struct gnn_grads {
    float *Win, *bin, *W1, *b1, *W2, *b2, *W3, *b3, *W4, *b4;
} gnn_grads_t;

/* Execution plan of the fused pipeline (built once per batch by the host; see
 * gnn-fpga_amd/plan.py).  Hits are renumbered by (graph, topological level), cut into tiles of
 * <= tile_hits hits (one workgroup each), degree-sorted inside a tile and padded to 16-hit
 * slices; n_pad = padded hit count, NULL hit id = n_pad.  Neighbour lists are SELL-16: entry k
 * of hit i of slice s at off[s] + 16*k + i, padded with the NULL entry.  A tile / chunk whose
 * neighbour id windows fit the LDS budget (gnn_plan_limits) runs in LDS mode (mode = 1): its
 * list / endpoint entries are window-relative and NULL = window size; otherwise entries are
 * absolute padded ids.  Scores stay per segment in the caller's segment order.
 *   tile  descriptor, 8 ints: slice_begin, slice_end, in_lo, in_cnt, out_lo, out_cnt, mode, sched_base
 *   chunk descriptor, 8 ints: seg_begin, seg_end, src_lo, src_cnt, dst_lo, dst_cnt, mode, 0 */
typedef struct gnn_plan {
    const float *X;                      /* [n_pad+1+64, F] renumbered features; dummy, NULL and the 64
                                            tail rows (window DMA overrun) are zero */
    const int32_t *src, *dst;            /* [n_segments] endpoints (window-relative or absolute)   */
    const int32_t *in_off, *in_nbr;      /* [n_pad/16+1], [in_off[last]]  segments ending at a hit -> start hit */
    const int32_t *out_off, *out_nbr;    /* [n_pad/16+1], [out_off[last]] segments starting at a hit -> end hit */
    const int32_t *tiles, *chunks;       /* [n_tiles*8], [n_chunks*8] descriptors                  */
    /* 16-bit packed copy of the lists (window-relative entries, steps padded to 8 per slice,
     * word p of hit i of slice s = steps 2p | 2p+1 << 16 at off16[s] + 16*p + i) */
    const int32_t *in_off16, *in_nbr16, *out_off16, *out_nbr16;
    /* per-phase wave schedules of the phase-split kernel: sched[tile[7] + 16*round + wave] = slice
     * id (or -1) that wavefront `wave` of the tile's workgroup takes in `round` */
    const int32_t *sched_a, *sched_b;
    const int32_t *sd16;                 /* [n_segments] LDS-mode chunks: dst_rel << 16 | src_rel   */
    int64_t n_pad, n_segments, n_tiles, n_chunks;
    int64_t iter_lds_records;            /* max over LDS-mode tiles of in_cnt + out_cnt + 2        */
    int64_t edge_lds_rows;               /* max over LDS-mode chunks of src_cnt + dst_cnt + 2      */
    int64_t n_lds_tiles;                 /* number of tiles with mode = 1                          */
    int64_t iter_lds_in, iter_lds_out;   /* max in_cnt / out_cnt over LDS-mode tiles               */
    int64_t tile_hits_max;               /* largest tile, in (padded) hits                         */
    int64_t max_list_steps;              /* longest SELL list of any slice, in steps               */
} gnn_plan_t;

int gnn_abi_version(void);
const char *gnn_last_error(void);

/* 1 if kernels exist for this (input_dim, hidden_dim), else 0. */
int gnn_shape_supported(int32_t F, int32_t D);
/* Row stride (floats) of H for this shape; 0 if unsupported. */
int32_t gnn_h_stride(int32_t F, int32_t D);

/* input_network + skip concat: H[n] = [tanh(Win X[n] + bin) | X[n]].
 * Replaces gnn/model.py:144,146 (nn.Linear + Tanh + torch.cat). */
int gnn_input_fwd(const float *X, const float *Win, const float *bin, float *H,
                  int64_t n_hits, int32_t F, int32_t D, int32_t ldh, void *stream);

/* EdgeNetwork.forward: e[j] = sigmoid(W2 tanh(W1 [H[src j] | H[dst j]] + b1) + b2).
 * Replaces gnn/model.py:69-81 (two bmm gathers, cat, MaskedLinear, Tanh, MaskedLinear, Sigmoid).
 * pq_ws: scratch of n_hits * 2 * D floats (per-hit halves of the first layer). */
int gnn_edge_fwd(const float *H, int32_t ldh, const int32_t *src, const int32_t *dst,
                 const float *W1, const float *b1, const float *W2, const float *b2,
                 float *e, float *pq_ws, int64_t n_hits, int64_t n_segments,
                 int32_t F, int32_t D, void *stream);

/* NodeNetwork.forward + the following skip concat:
 *   mi[n] = sum_{j: dst j = n} e[j] H[src j],  mo[n] = sum_{j: src j = n} e[j] H[dst j],
 *   Hnext[n] = [tanh(W4 tanh(W3 [mi|mo|H[n]] + b3) + b4) | X[n]]   (X[n] = H[n][D:D+F]).
 * Replaces gnn/model.py:113-125 (four bmm, two broadcast multiplies, cat, 2x MaskedLinear+Tanh)
 * and :154.  Sums run in ascending segment id (fixed order: bit-reproducible).
 * Only X, in_*, out_* and n_hits of `g` are read.  Hnext must not alias H. */
int gnn_node_fwd(const float *H, int32_t ldh, const float *e, const gnn_graph_t *g,
                 const float *W3, const float *b3, const float *W4, const float *b4,
                 float *Hnext, int32_t F, int32_t D, void *stream);

/* Bytes of device scratch gnn_segclf_forward needs for this problem. */
size_t gnn_forward_workspace_bytes(int64_t n_hits, int64_t n_segments, int32_t F, int32_t D);

/* SegmentClassifier.forward (gnn/model.py:140-156): input network, n_iters x (edge pass,
 * node pass), final edge pass.  e_out [n_segments] receives the scores.
 * Optional traces for parity tests (NULL to skip): e_trace [(n_iters+1), n_segments],
 * H_trace [(n_iters+1), n_hits, C] (unpadded rows). */
int gnn_segclf_forward(const gnn_graph_t *g, const gnn_params_t *p, int32_t n_iters,
                       float *e_out, float *e_trace, float *H_trace,
                       void *workspace, size_t workspace_bytes, void *stream);

/* Small events (the reference's muon graphs, gnn/prepareMuonGraphs.py: tens of hits): the same
 * forward in ONE launch, one workgroup per graph of the block-diagonal batch, hit features and
 * scores resident in LDS across all iterations (no workspace).  hit_ptr / seg_ptr [n_graphs+1]
 * are device arrays of the graphs' first hit / segment; every segment in
 * [seg_ptr[i], seg_ptr[i+1]) must join hits in [hit_ptr[i], hit_ptr[i+1]) (or be padded,
 * src = dst = -1); max_hits / max_segments bound the graph sizes.  Bit-identical to
 * gnn_segclf_forward.  gnn_events_supported: 1 if graphs of that size fit one workgroup.
 * A never-seen event needs nothing prepared (gnn/Inference.ipynb cell 3: one graph in, scores out): with
 * all six list pointers of `g` NULL the kernel builds the lists itself, in LDS, from (src, dst) - the lists
 * gnn_csr_build makes, so the scores are the same bits -, and with n_graphs == 1 hit_ptr / seg_ptr may be
 * NULL (the graph is [0, n_hits) x [0, n_segments)).  (gnn_segclf_forward only; the training entry points
 * below take caller-built lists and offsets.) */
int gnn_events_supported(int32_t F, int32_t D, int64_t max_hits, int64_t max_segments);
int gnn_segclf_forward_events(const gnn_graph_t *g, const gnn_params_t *p, const int32_t *hit_ptr,
                              const int32_t *seg_ptr, int64_t n_graphs, int32_t max_hits,
                              int32_t max_segments, int32_t n_iters, float *e_out, void *stream);

/* Training forward: like gnn_segclf_forward but keeps what the backward needs - the scores of
 * every edge pass e_all [(n_iters+1), n_segments] (the last row is the model output), the hit
 * features of every iteration H_all [(n_iters+1), n_hits, ldh] (padded rows) and, optionally, the
 * hidden layer of every node pass Q_all [n_iters, n_hits, hidden_dim] (NULL: not kept; the backward
 * then rebuilds it with a second walk over both segment lists). */
int gnn_segclf_forward_train(const gnn_graph_t *g, const gnn_params_t *p, int32_t n_iters,
                             float *e_all, float *H_all, float *Q_all, void *workspace,
                             size_t workspace_bytes, void *stream);

/* The same for a batch of small graphs in one launch (gnn_segclf_forward_events that also keeps
 * e_all / H_all; bit-identical to gnn_segclf_forward_train). */
int gnn_segclf_forward_train_events(const gnn_graph_t *g, const gnn_params_t *p, const int32_t *hit_ptr,
                                    const int32_t *seg_ptr, int64_t n_graphs, int32_t max_hits,
                                    int32_t max_segments, int32_t n_iters, float *e_all, float *H_all,
                                    void *stream);

/* Gradient of a scalar loss w.r.t. the ten parameter tensors, given grad_out [n_segments] =
 * dLoss/d(scores) and the tensors saved by gnn_segclf_forward_train.  Replaces autograd through
 * gnn/model.py:140-156 as triggered by loss.backward() in gnn/estimator.py:58.  Adds into
 * `grads` (zero them first).  Workspace size from gnn_backward_workspace_bytes. */
size_t gnn_backward_workspace_bytes(int64_t n_hits, int64_t n_segments, int32_t F, int32_t D);

/* The reference's loss, nn.BCELoss()(scores, targets) (gnn/trainSegmentClassifier.py:164,
 * gnn/estimator.py:57), value and gradient in one pass over the scores:
 *   loss_out[0] = scale * sum_j -( y_j max(log e_j, -100) + (1 - y_j) max(log(1 - e_j), -100) )
 *   grad_e[j]   = scale * (e_j - y_j) / max(e_j (1 - e_j), 1e-12)        (NULL to skip)
 * scale = 1/n for the reference's "mean" reduction (n counts padded segments too, like the
 * reference's mean over B x E_max), 1 for "sum".  Same clamps as torch; deterministic (fixed
 * summation order).  workspace: GNN_BCE_WORKSPACE_BYTES of device scratch. */
#define GNN_BCE_WORKSPACE_BYTES 4096
int gnn_bce_loss(const float *e, const float *y, int64_t n, float scale, float *loss_out,
                 float *grad_e, void *workspace, void *stream);
int gnn_segclf_backward(const gnn_graph_t *g, const gnn_params_t *p, int32_t n_iters,
                        const float *e_all, const float *H_all, const float *Q_all /* or NULL */,
                        const float *grad_out, const gnn_grads_t *grads, void *workspace,
                        size_t workspace_bytes, void *stream);

/* The same gradients for a batch of SMALL graphs (gnn/prepareMuonGraphs.py sizes) in ONE launch:
 * one workgroup per graph keeps the graph's saved rows and every intermediate in LDS (counterpart
 * of gnn_segclf_forward_events; same layout contract for hit_ptr / seg_ptr).  Workspace:
 * gnn_backward_events_workspace_bytes.  gnn_events_backward_supported: 1 if graphs of that size
 * fit one workgroup for this (input_dim, hidden_dim). */
int gnn_events_backward_supported(int32_t F, int32_t D, int64_t max_hits, int64_t max_segments);
size_t gnn_backward_events_workspace_bytes(int64_t n_graphs, int32_t F, int32_t D);
int gnn_segclf_backward_events(const gnn_graph_t *g, const gnn_params_t *p, const int32_t *hit_ptr,
                               const int32_t *seg_ptr, int64_t n_graphs, int32_t max_hits,
                               int32_t max_segments, int32_t n_iters, const float *e_all,
                               const float *H_all, const float *grad_out, const gnn_grads_t *grads,
                               void *workspace, size_t workspace_bytes, void *stream);

/* SegmentClassifier.forward (gnn/model.py:140-156) on a planned batch: the fast path.
 * One fused kernel per message-passing iteration (edge scores are recomputed at both
 * endpoints from per-hit partial products instead of being stored), one final edge kernel.
 * e_out [n_segments].  Workspace size from gnn_plan_workspace_bytes. */
size_t gnn_plan_workspace_bytes(int64_t n_pad, int64_t n_segments, int32_t F, int32_t D);
int gnn_segclf_forward_plan(const gnn_plan_t *plan, const gnn_params_t *p, int32_t n_iters,
                            float *e_out, void *workspace, size_t workspace_bytes, void *stream);
/* The TRAINING forward on a planned batch (ABI 4): the same tile kernels, keeping what the backward
 * (gnn_segclf_backward) needs - what gnn_segclf_forward_train keeps with the per-module kernels
 * (autograd for gnn/estimator.py:53,57-58).  The backward's batch must be the plan-space form of the
 * planned batch: hits numbered by the plan's padded ids (n_pad hits, dummies without segments),
 * segments sorted by end hit (stable), `seg_ptr` [n_pad + 1] = its CSR pointer over end hits
 * (in_ptr).  Written: e_all rows 0 .. n_iters-1 ([n_iters + 1, n_segments]; the scores of every pass
 * in THAT segment order, valid segments only - padded ones are never read by the backward's list
 * walks), H_all [(n_iters + 1), n_pad, ldh] (ldh = gnn_h_stride), Q_all [n_iters, n_pad, D] and the
 * final scores e_out [n_segments] in the PLAN's segment order (the caller's; NULL: not wanted).  Row
 * n_iters of e_all - the final scores in the backward's order - is written when tw_src / tw_dst
 * [n_segments] are given: that batch's segment endpoints as plan hit ids, -1 = padded (same bits as
 * e_out for the same segment); with NULL the row is the caller's to fill (a gather of e_out).  Shapes
 * on the general tile kernel only (hidden_dim <= 16 without the 16-lanes-per-hit route):
 * GNN_ERR_UNSUPPORTED otherwise.  Workspace as gnn_segclf_forward_plan. */
int gnn_segclf_forward_train_plan(const gnn_plan_t *plan, const gnn_params_t *p, int32_t n_iters,
                                  const int32_t *seg_ptr, const int32_t *tw_src, const int32_t *tw_dst,
                                  float *e_all, float *H_all, float *Q_all, float *e_out /* or NULL */,
                                  void *workspace, size_t workspace_bytes, void *stream);
/* 1 if the fused pipeline has kernels for this (input_dim, hidden_dim). */
int gnn_plan_shape_supported(int32_t F, int32_t D);
/* Plan-building limits for a shape: out4 = { tile_hits, iter_records, chunk_segments,
 * edge_records } - the tile / chunk sizes and the LDS window budgets (in records of 2D floats
 * for the iteration kernel, rows of D floats for the edge kernel). */
int gnn_plan_limits(int32_t F, int32_t D, int32_t *out4);

/* The reference's dense input contract -> index form, on the device: Ri, Ro [B, N, E] float32 with one
 * non-zero per real column and all-zero padded columns (gnn/graph.py:28-35,
 * gnn/trainSegmentClassifier.py:66-95) -> src, dst [B*E] int32 global hit ids b*N + n, -1 for a padded
 * column.  flags [1] (device): bit 0 = a column with more than one non-zero, bit 1 = a column set in
 * only one of Ri / Ro (such columns come out as -1).  Asynchronous on `stream`. */
int gnn_dense_to_index(const float *Ri, const float *Ro, int64_t B, int64_t N, int64_t E, int32_t *src,
                       int32_t *dst, int32_t *flags, void *stream);

/* ---- per-module backward ------------------------------------------------------------------------
 * The reference's sub-modules are ordinary autograd modules (model.edge_network(H, Ri, Ro),
 * model.node_network(H, e, Ri, Ro): gnn/model.py:69-81, 113-125; called on their own in
 * gnn/MPNN_Seg_ACTS_maskedlinear.ipynb cells 42, 46).  These two entry points are their backward:
 * H rows of ldh = gnn_h_stride(F, D) floats, gradient tensors as in gnn_segclf_backward (all ten
 * pointers valid, the backward ADDS into them; an edge backward touches W1, b1, W2, b2 only, a node
 * backward W3, b3, W4, b4 only).  Workspace: gnn_backward_workspace_bytes.
 *   gnn_edge_bwd: grad_e [n_segments] -> grad_H [n_hits, ldh] (caller-zeroed, ADDED into)
 *   gnn_node_bwd: Hnext = the forward's output [n_hits, ldh], grad_Hnext [n_hits, ldh] (first D
 *                 columns used) -> grad_H [n_hits, ldh] (written), grad_e [n_segments] (written) */
int gnn_edge_bwd(const float *H, int32_t ldh, const gnn_graph_t *g, const gnn_params_t *p, const float *e,
                 const float *grad_e, float *grad_H, const gnn_grads_t *grads, void *workspace,
                 size_t workspace_bytes, void *stream);
int gnn_node_bwd(const float *H, int32_t ldh, const float *e, const float *Hnext, const gnn_graph_t *g,
                 const gnn_params_t *p, const float *grad_Hnext, float *grad_H, float *grad_e,
                 const gnn_grads_t *grads, void *workspace, size_t workspace_bytes, void *stream);

/* ---- the plan, built on the GPU (csrc/plan_build.hip) -------------------------------------------
 * Index-form counterpart of the reference's per-batch host work (graph_from_sparse densifies,
 * merge_graphs zero-pads: gnn/graph.py:28-35, gnn/trainSegmentClassifier.py:66-111): the same
 * plan gnn-fpga_amd/plan.py specifies in numpy, array for array, in two calls around ONE host
 * read-back of the sizes:
 *   gnn_plan_build_sizes  levels, tiles, renumbering, sorted neighbour lists, windows, chunks ->
 *                         *sizes_out (DEVICE memory, written asynchronously on `stream`)
 *   gnn_plan_build_fill   fills the arrays the caller allocated from those sizes (a HOST copy of
 *                         the sizes is passed back in)
 * src / dst [n_segments] int32 (-1 = padded), hit_ptr [n_graphs+1] int64 on the DEVICE (graph
 * boundaries, non-decreasing); tile_hits / iter_records / chunk_segments / edge_records as
 * gnn_plan_limits gives them (after the caller's small-batch adjustments, plan.py).  sizes.status
 * != 0: this batch is outside the builder's static bounds (degree >= 65536, > n/16 + 1024 tiles,
 * ...) - build the plan with plan.py / plan_device.py instead; bit 64 of status: a segment end
 * outside [0, n_hits) or a segment with exactly one negative end (malformed batch: the kernels
 * skip such segments, the caller must raise).  n_hits, n_segments > 0. */
typedef struct gnn_plan_sizes {
    int64_t n_pad, n_tiles, n_slices, n_chunks;
    int64_t in_total, out_total;        /* entries of in_nbr / out_nbr before their 64 zero entries */
    int64_t in16_words, out16_words;    /* words of in_nbr16 / out_nbr16 before their 64 zero words */
    int64_t n_sched;                    /* entries of sched_a, sched_b                              */
    int64_t iter_lds_records, edge_lds_rows, n_lds_tiles, n_lds_chunks, iter_lds_in, iter_lds_out;
    int64_t tile_hits_max, max_list_steps, n_valid, max_level, status;
    int64_t list_mode;                  /* ABI 5, graph-local form: 1 = neighbour lists built per tile in LDS, 0 = by scattered
                                           pairs and a sort per list (segments of a tile's lists not contiguous, hub hits) */
} gnn_plan_sizes_t;

typedef struct gnn_plan_out {           /* device arrays gnn_plan_build_fill writes (sizes: gnn_plan_t) */
    float *X;                           /* [(n_pad + 1 + 64) * F]                                     */
    float *x_absmax;                    /* [F] per-feature max |X| (gnn_exp_product_bound)            */
    int32_t *src, *dst, *sd16;          /* [n_segments]                                               */
    int32_t *in_off, *in_nbr;           /* [n_slices + 1], [in_total + 64]                            */
    int32_t *out_off, *out_nbr;         /* [n_slices + 1], [out_total + 64]                           */
    int32_t *in_off16, *in_nbr16;       /* [n_slices + 1], [in16_words + 64]                          */
    int32_t *out_off16, *out_nbr16;     /* [n_slices + 1], [out16_words + 64]                         */
    int32_t *tiles, *chunks;            /* [8 n_tiles], [8 n_chunks]                                  */
    int32_t *sched_a, *sched_b;         /* [n_sched]                                                  */
    int32_t *perm;                      /* [n_pad] new id -> caller's hit id, -1 = padding            */
    int32_t *src_abs, *dst_abs, *level; /* optional (NULL): renumbered endpoints [n_segments], levels [n_hits] */
} gnn_plan_out_t;

size_t gnn_plan_build_workspace_bytes(int64_t n_hits, int64_t n_segments, int32_t chunk_segments);
int gnn_plan_build_sizes(const int32_t *src, const int32_t *dst, const int64_t *hit_ptr, int64_t n_hits,
                         int64_t n_segments, int64_t n_graphs, int32_t tile_hits, int32_t iter_records,
                         int32_t chunk_segments, int32_t edge_records, void *workspace,
                         size_t workspace_bytes, gnn_plan_sizes_t *sizes_out, void *stream);
/* ABI 5: the same stage 1 for a batch whose graphs are laid out one after the other, as merge_graphs' block-diagonal
 * batches are (gnn/trainSegmentClassifier.py:66-95): seg_ptr [n_graphs+1] int64 on the DEVICE, graph g owns segments
 * [seg_ptr[g], seg_ptr[g+1]) and they join hits of [hit_ptr[g], hit_ptr[g+1]) only; max_graph_hits /
 * max_graph_segments = the largest graph (host values).  Degrees, levels, renumbered endpoints and both neighbour
 * lists are then built by one workgroup per graph in LDS instead of device-wide sweeps and sorts (same arrays, entry
 * for entry).  The kernels CHECK the layout they were told: status bit 128 = it does not hold for this batch (an end
 * outside its graph's hit range, ranges that do not tile [0, n_hits) / [0, n_segments), a level above 64, a hit with
 * more than 1024 segments in one direction, a graph of more than 16384 hits whose level / degree / id bits exceed a 32-bit sort key) - call gnn_plan_build_sizes instead.  max_graph_hits > 19456:
 * GNN_ERR_UNSUPPORTED. */
int gnn_plan_build_sizes_graphs(const int32_t *src, const int32_t *dst, const int64_t *hit_ptr, const int64_t *seg_ptr,
                                int64_t max_graph_hits, int64_t max_graph_segments, int64_t n_hits,
                                int64_t n_segments, int64_t n_graphs, int32_t tile_hits, int32_t iter_records,
                                int32_t chunk_segments, int32_t edge_records, void *workspace,
                                size_t workspace_bytes, gnn_plan_sizes_t *sizes_out, void *stream);
int gnn_plan_build_fill(const float *X, int32_t F, const int32_t *src, const int32_t *dst, int64_t n_hits,
                        int64_t n_segments, int32_t chunk_segments, const gnn_plan_sizes_t *sizes,
                        void *workspace, size_t workspace_bytes, const gnn_plan_out_t *out, void *stream);

/* ---- the two segment lists, built on the GPU (csrc/csr_build.hip; ABI 4) --------------------------
 * What gnn_graph_t's in_ptr / in_eid / in_nbr and out_ptr / out_eid / out_nbr hold, from the index form: for every
 * hit the ids of the segments that end (start) there, ASCENDING - the reference's on-disk order, `Ri.nonzero()` /
 * `Ro.nonzero()` row-major (gnn/graph.py:20-26), and the order its bmm against Ri / Ro sums in
 * (gnn/model.py:114-119).  Replaces two stable sorts and a read-back per never-seen batch.
 * src / dst [n_segments] int32 (-1 / -1 = padded segment, left out of both lists).  Written: in_ptr, out_ptr
 * [n_hits + 1]; in_eid, in_nbr, out_eid, out_nbr [n_segments] - the first in_ptr[n_hits] (= out_ptr[n_hits]) entries
 * are the lists, the rest is -1; status [1] (device): bit 0 = a segment with an end outside [0, n_hits) or with
 * exactly one negative end (skipped like a padded one; the caller must raise).  The arrays are the same in every
 * run (no dependence on the order atomics arrive in).  Asynchronous on `stream`; no host synchronisation. */
size_t gnn_csr_build_workspace_bytes(int64_t n_hits, int64_t n_segments);
int gnn_csr_build(const int32_t *src, const int32_t *dst, int64_t n_hits, int64_t n_segments, int32_t *in_ptr,
                  int32_t *in_eid, int32_t *in_nbr, int32_t *out_ptr, int32_t *out_eid, int32_t *out_nbr,
                  int32_t *status, void *workspace, size_t workspace_bytes, void *stream);

/* bound_out (device, 1 float) = the left side of the GNN_FLAG_EXP_PRODUCT condition;
 * x_absmax (device, [F]) = per-feature max |X|.  Asynchronous on `stream`. */
int gnn_exp_product_bound(const gnn_params_t *p, const float *x_absmax, float *bound_out,
                          void *stream);

/* Per-kernel timing with HIP events on the launch stream (bench.py's roofline leg).
 * gnn_profile_begin(capacity) arms recording of up to `capacity` kernel launches;
 * gnn_profile_end synchronises the recorded events and returns the number of records,
 * filling names[i] (static strings) and ms[i] for i < min(count, capacity_out). */
int gnn_profile_begin(int32_t capacity);
int gnn_profile_end(const char **names, float *ms, int32_t capacity_out);

#ifdef __cplusplus
}
#endif
#endif /* GNN_HIP_H */
