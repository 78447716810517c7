// backward.hip - gradient of the SegmentClassifier forward w.r.t. its ten parameter tensors.
//
// The reference gets this from autograd through gnn/model.py:140-156 when
// gnn/estimator.py:58 calls loss.backward().  Here the training forward (gnn_kernels.hip) saves
// the hit features H_t and edge scores e_t of every iteration and this file walks the iterations
// backwards with explicit kernels (formulas: SURVEY.md appendix A, checked there against
// autograd in fp64).  Index form, CSR pulls, fp32:
//
//   edge pass t   e = sigmoid(u), u = W2 a + b2, a = tanh(z), z = P[s] + Q[d]
//       k_edge_bwd   per segment:  gu = ge e (1-e) -> gu[E];  gW2, gb2 (+ gb1 of the padded segments)
//       k_pq_bwd     per hit:      gz = gu W2 (1-a^2) rebuilt per CSR entry from the other end's
//                                  P / Q row;  gP = sum_out gz, gQ = sum_in gz (fixed order)
//                                  gH += W1[:, :C]^T gP + W1[:, C:]^T gQ;  gW1, gb1
//   node pass t   H' = tanh(W4 q + b4), q = tanh(W3 M + b3), M = [mi | mo | H]
//       k_node_bwd   per hit: recompute M, q; gr = gH' (1-H'^2); gp = W4^T gr (1-q^2);
//                             gM = W3^T gp -> gmi, gmo stored, gH_prev = gHself;  gW3, gb3, gW4, gb4
//       (the gradient of that pass's scores, ge = <gmi[d], H[s]> + <gmo[s], H[d]>, is rebuilt per
//        segment inside the k_edge_bwd of the earlier edge pass)
//       k_agg_bwd_n  per hit:      gH_prev += sum_out e gmi[d] + sum_in e gmo[s]
//   input         k_input_bwd per hit: g = gH0[:D] (1-H0^2);  gWin, gbin
//
// hidden_dim <= 16 takes the PULL FORM further down instead of the node / aggregation kernels above
// (D-wide records [P | R | gp], [Q | S | gp]; a hit walks its two lists once per iteration): with the
// forward's kept hidden layers (Q_all) on four lanes per hit - k_hit_bwd4, k_seg_bwd4, k_seg_fin -
// else on one lane per hit - kb_prs, k_hit_bwd, k_seg_bwd.
//
// Weight gradients are sums of per-item outer products: a workgroup parks its 256 items' factors
// in LDS, each thread then owns output elements and sums over the 256 items (fixed order) and adds
// the result to ITS WORKGROUP'S OWN ROW of a partial-sum table (row = blockIdx.x; one writer per
// element, so the adds of successive launches land in launch order).  k_grad_fold1 / 2 then sum the
// rows in row order, 32 rows per chunk, chunks in order: every float addition of a backward happens
// in a fixed order, and two runs give bit-identical gradients (deterministic by default).
#include "common.h"

namespace {
using namespace gnn;

template <int F, int D>
struct Shape {
    static constexpr int C = F + D;
    static constexpr int LDH = (C + 3) & ~3;
};

// row of LDH floats (16-byte aligned: LDH % 4 == 0) as float4 loads
template <int N4>
__device__ __forceinline__ void load_row4(const float *__restrict__ row, float *v)
{
    const float4 *r = reinterpret_cast<const float4 *>(row);
#pragma unroll
    for (int i = 0; i < N4; ++i) {
        const float4 a = r[i];
        v[4 * i] = a.x; v[4 * i + 1] = a.y; v[4 * i + 2] = a.z; v[4 * i + 3] = a.w;
    }
}
template <int N4>
__device__ __forceinline__ void store_row4(float *__restrict__ row, const float *v)
{
    float4 *r = reinterpret_cast<float4 *>(row);
#pragma unroll
    for (int i = 0; i < N4; ++i) r[i] = make_float4(v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]);
}

// Sum over the workgroup's 256 items of L (x) [R | 1], added into g[i * ldg + col0 + k] and (the
// ones column, gb != null) into gb[i].  The items' factors are parked in LDS TRANSPOSED - row c holds
// factor c of all 256 items, rows padded by 4 floats - so that the thread that owns an output
// element walks two rows with 16-byte reads (4 items per read, bank-conflict free across the
// wavefront: neighbouring threads own neighbouring k, i.e. rows 4 banks apart), in item order.
// One add per element per workgroup goes to the workgroup's own row of the partial table.
// Wide right factors: the staging area is capped at kOuterCap rows (133 KB), the columns of R are
// taken in chunks that fit beside L.
constexpr int kOuterCap = 128, kOuterStride = kBlock + 4;

// A workgroup's partial gradient sums live in row blockIdx.x of the partial table (GradLayout::stride
// floats per row): no other workgroup touches the row, so there is nothing to bounce between the
// 8 L2s and nothing whose order could vary; the fold kernels add the rows up in a fixed order.
constexpr int kFoldChunk = 32;            // rows summed sequentially by one thread of k_grad_fold1
__device__ __forceinline__ float *my_replica(float *g, int rep_stride)
{
    return g + (size_t)blockIdx.x * rep_stride;
}
inline int64_t fold_chunks(int64_t rows) { return (rows + kFoldChunk - 1) / kFoldChunk; }
template <int NL, int NR>
constexpr int outer_lds_floats() { return kOuterStride * ((NL + NR + 1) < kOuterCap ? (NL + NR + 1) : kOuterCap); }

template <int NL, int NR, int C0 = 0>
__device__ __forceinline__ void accum_outer(const float *L, const float *R, bool active, float *g,
                                            int ldg, int col0, float *gb, float *lds)
{
    static_assert(NL + 1 < kOuterCap, "left factor too wide");
#ifdef GNN_ABLATE_OUTER           // timing experiment (results invalid): what the outer-product sums cost
    return;
#endif
    // rows: L (NL) | this chunk of R (CH) | ones (the bias column, first chunk only)
    constexpr bool ONES = (C0 == 0);
    constexpr int ROOM = kOuterCap - NL - (ONES ? 1 : 0);
    constexpr int CH = (NR - C0 <= ROOM) ? NR - C0 : ROOM, RS = kOuterStride;
#pragma unroll
    for (int i = 0; i < NL; ++i) lds[i * RS + threadIdx.x] = active ? L[i] : 0.0f;
#pragma unroll
    for (int k = 0; k < CH; ++k) lds[(NL + k) * RS + threadIdx.x] = active ? R[C0 + k] : 0.0f;
    if constexpr (ONES) lds[(NL + CH) * RS + threadIdx.x] = active ? 1.0f : 0.0f;
    __syncthreads();
    const int cols = CH + ((ONES && gb) ? 1 : 0);
    for (int o = threadIdx.x; o < NL * cols; o += kBlock) {
        const int i = o / cols, k = o % cols;
        const float4 *a = reinterpret_cast<const float4 *>(lds + i * RS);
        const float4 *b = reinterpret_cast<const float4 *>(lds + (NL + k) * RS);
        float acc = 0.0f;
#pragma unroll 4
        for (int t = 0; t < kBlock / 4; ++t) {
            const float4 x = a[t], y = b[t];
            acc = fmaf(x.x, y.x, acc); acc = fmaf(x.y, y.y, acc);
            acc = fmaf(x.z, y.z, acc); acc = fmaf(x.w, y.w, acc);
        }
        float *dst = k < CH ? &g[i * ldg + col0 + C0 + k] : &gb[i];      // own row: the only writer
        *dst += acc;
    }
    __syncthreads();
    if constexpr (C0 + CH < NR) accum_outer<NL, NR, C0 + CH>(L, R, active, g, ldg, col0, gb, lds);
}

// Several outer-product sums in ONE staging (two barriers instead of two per sum, the right factor staged once when
// the sums share it, and all 256 threads busy: one sum of D x (C + 1) = 96 outputs keeps 96 of them): job j adds
// sum_items L_j (x) [R_j | 1] into g_j[i * ldg_j + col0_j + k] (and, gb_j != null, the ones column into gb_j[i]).
// The caller has parked its factors in LDS rows (stage_row) - L_j in rows [lrow_j, lrow_j + nl), R_j in
// [rrow_j, rrow_j + nr_j), the ones row last; every output is the same 256-term dot product in item order as in
// accum_outer: the sums come out bit-identical to separate calls.  k_seg_fin's four sums at D = 8 (44 rows, 46 KB)
// would leave three workgroups per CU where the kernel wants five: the callers stage two sums at a time.
constexpr int kOuterJobRows = 62;                  // 62 x 260 x 4 B = 64,480 B: what a static __shared__ array may hold
struct OuterJob { int lrow, rrow, nr, ldg, col0; float *g, *gb; };
__device__ __forceinline__ void stage_row(float *lds, int row, float v) { lds[row * kOuterStride + threadIdx.x] = v; }
template <int NJ>
__device__ __forceinline__ void accum_outer_jobs(const OuterJob (&jobs)[NJ], int nl, int ones_row, float *lds)
{
    constexpr int RS = kOuterStride;
    __syncthreads();
    int total = 0;
#pragma unroll
    for (int j = 0; j < NJ; ++j) total += nl * (jobs[j].nr + (jobs[j].gb ? 1 : 0));
    for (int o = threadIdx.x; o < total; o += kBlock) {
        int oo = o, j = 0;
#pragma unroll
        for (int t = 0; t < NJ - 1; ++t) {
            const int n = nl * (jobs[t].nr + (jobs[t].gb ? 1 : 0));
            if (j == t && oo >= n) { oo -= n; j = t + 1; }
        }
        const OuterJob &J = jobs[j];
        const int cols = J.nr + (J.gb ? 1 : 0), i = oo / cols, k = oo % cols;
        const float4 *a = reinterpret_cast<const float4 *>(lds + (J.lrow + i) * RS);
        const float4 *b = reinterpret_cast<const float4 *>(lds + (k < J.nr ? J.rrow + k : ones_row) * RS);
        float acc = 0.0f;
#pragma unroll 4
        for (int t = 0; t < kBlock / 4; ++t) {
            const float4 x = a[t], y = b[t];
            acc = fmaf(x.x, y.x, acc); acc = fmaf(x.y, y.y, acc);
            acc = fmaf(x.z, y.z, acc); acc = fmaf(x.w, y.w, acc);
        }
        float *dst = k < J.nr ? &J.g[i * J.ldg + J.col0 + k] : &J.gb[i];          // own row: the only writer
        *dst += acc;
    }
    __syncthreads();
}
// two left factors against one right factor: (L0 (x) [R | 1]) -> g[.., col0a + k], gb;  (L1 (x) R) -> g[.., col0b + k]
template <int NL, int NR>
constexpr int outer2_lds_floats() { return kOuterStride * (2 * NL + NR + 1); }
template <int NL, int NR>
__device__ __forceinline__ void accum_outer2(const float *L0, const float *L1, const float *R, bool active, float *g,
                                             int ldg, int col0a, int col0b, float *gb, float *lds)
{
    static_assert(2 * NL + NR + 1 <= kOuterJobRows, "two left factors and the right one must fit the static LDS array");
#ifdef GNN_ABLATE_OUTER
    return;
#endif
#pragma unroll
    for (int i = 0; i < NL; ++i) { stage_row(lds, i, active ? L0[i] : 0.0f); stage_row(lds, NL + i, active ? L1[i] : 0.0f); }
#pragma unroll
    for (int k = 0; k < NR; ++k) stage_row(lds, 2 * NL + k, active ? R[k] : 0.0f);
    stage_row(lds, 2 * NL + NR, active ? 1.0f : 0.0f);
    const OuterJob jobs[2] = {{0, 2 * NL, NR, ldg, col0a, g, gb}, {NL, 2 * NL, NR, ldg, col0b, g, nullptr}};
    accum_outer_jobs<2>(jobs, NL, 2 * NL + NR, lds);
}

// The same sums on the matrix cores for wide left factors (NL = 32 / 64): the outer-product sum over
// the workgroup's 256 items is a [NL x 256] . [256 x NR] product.  v_mfma_f32_16x16x4_f32 multiplies
// and accumulates in fp32 in k (= item) order: nothing is rounded, the order is fixed.  Lane l supplies
// A[row l & 15][item 4 s + (l >> 4)] and B[item 4 s + (l >> 4)][col l & 15] straight from the transposed
// staging (stride 260: the 64 lanes hit 64 different banks) and receives rows 4 (l >> 4) .. + 3 of
// column l & 15; the workgroup's 4 waves deal the 16 x 16 output tiles.  One thread per output was
// 26 k LDS-bound instructions per thread at D = 64 (0.8 of k_seg_finW's 0.96 ms).
template <int NL, int NR, int C0 = 0>
__device__ __forceinline__ void accum_outer_mfma(const float *L, const float *R, bool active, float *g,
                                                 int ldg, int col0, float *gb, float *lds)
{
    static_assert(NL % 16 == 0 && NL + 16 <= kOuterCap, "left factor: whole 16-row tiles");
    typedef float f4v __attribute__((ext_vector_type(4)));
    constexpr bool ONES = (C0 == 0);
    constexpr int ROOM = ((kOuterCap - NL) / 16) * 16 - (ONES ? 1 : 0);     // columns per chunk (tiles are whole)
    constexpr int CH = (NR - C0 <= ROOM) ? NR - C0 : ROOM, RS = kOuterStride;
    constexpr int COLS = CH + (ONES ? 1 : 0), CT = (COLS + 15) / 16, RT = NL / 16;
#pragma unroll
    for (int i = 0; i < NL; ++i) lds[i * RS + threadIdx.x] = active ? L[i] : 0.0f;
#pragma unroll
    for (int k = 0; k < CH; ++k) lds[(NL + k) * RS + threadIdx.x] = active ? R[C0 + k] : 0.0f;
    if constexpr (ONES) lds[(NL + CH) * RS + threadIdx.x] = active ? 1.0f : 0.0f;
#pragma unroll
    for (int k = COLS; k < 16 * CT; ++k) lds[(NL + k) * RS + threadIdx.x] = 0.0f;      // pad the last column tile
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, r16 = lane & 15, g4 = lane >> 4;
    for (int t = wv; t < RT * CT; t += kBlock / 64) {
        const int it = t / CT, jt = t % CT;
        const float *a = lds + (16 * it + r16) * RS + g4, *b = lds + (NL + 16 * jt + r16) * RS + g4;
        f4v c = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll 8
        for (int st = 0; st < kBlock / 4; ++st) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[4 * st], b[4 * st], c, 0, 0, 0);
        const int col = 16 * jt + r16;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = 16 * it + 4 * g4 + r;
            const float v = r == 0 ? c.x : r == 1 ? c.y : r == 2 ? c.z : c.w;
            if (col < CH) g[i * ldg + col0 + C0 + col] += v;                // own row: the only writer
            else if (ONES && col == CH && gb) gb[i] += v;
        }
    }
    __syncthreads();
    if constexpr (C0 + CH < NR) accum_outer_mfma<NL, NR, C0 + CH>(L, R, active, g, ldg, col0, gb, lds);
}

// P/Q rows from H (same as the forward's k_pq; kept local to this file)
template <int F, int D>
__global__ __launch_bounds__(kBlock) void kb_pq(const float *__restrict__ H, int ldh,
                                                const float *__restrict__ W1,
                                                const float *__restrict__ b1,
                                                float *__restrict__ PQ, int64_t n_hits)
{
    constexpr int C = F + D;
    const int64_t n = xcd_block() * kBlock + threadIdx.x;
    if (n >= n_hits) return;
    float h[Shape<F, D>::LDH], pq[2 * D];
    load_row4<Shape<F, D>::LDH / 4>(H + n * ldh, h);
#pragma unroll
    for (int d = 0; d < D; ++d) {
        float p = b1[d], q = 0.0f;
#pragma unroll
        for (int k = 0; k < C; ++k) {
            p = fmaf(W1[d * 2 * C + k], h[k], p);
            q = fmaf(W1[d * 2 * C + C + k], h[k], q);
        }
        pq[d] = p;
        pq[D + d] = q;
    }
    store_row4<2 * D / 4>(PQ + n * 2 * D, pq);
}

// Per-segment kernels are grid-stride with a bounded grid: every thread keeps private sums of its
// segments' contributions, the workgroup adds them up once at the end (wavefront butterfly, one
// LDS slot per wavefront) and issues ONE atomic per gradient element - a few hundred per launch
// instead of one per 256 segments (12 500 at 3.2 M segments: they serialised and were the launch).
// per segment; padded segments (src = -1) score sigmoid(W2 tanh(b1) + b2): their gz flows into
// b1 only (summed here, the hits' share of gb1 comes from k_pq_bwd)
// ge: gradient w.r.t. this pass's scores - the loss gradient for the last pass (ge != null), else
// rebuilt here from the node pass that consumed them (k_node_bwd's gmio and the H it read):
// ge = <gmi[d], H[s]> + <gmo[s], H[d]>, zero for padded segments.
template <int F, int D>
__global__ __launch_bounds__(kBlock) void k_edge_bwd(
    const int32_t *__restrict__ src, const int32_t *__restrict__ dst,
    const float *__restrict__ PQ, const float *__restrict__ b1, const float *__restrict__ W2,
    const float *__restrict__ e, const float *__restrict__ ge, const float *__restrict__ H,
    const float *__restrict__ gmio, float *__restrict__ gu_out,
    float *__restrict__ gW2, float *__restrict__ gb2, float *__restrict__ gb1, int rep_stride,
    int64_t n_segments, int pad_only = 0)
{
    constexpr int C = F + D, LDH = Shape<F, D>::LDH;
    gW2 = my_replica(gW2, rep_stride);
    gb2 = my_replica(gb2, rep_stride);
    gb1 = my_replica(gb1, rep_stride);
    __shared__ float lds[(kBlock / 64) * (D + 2)];
    // private sums: gW2[D] | gb2 | sum of gu over the padded segments.  A padded segment has
    // z = b1, the same for all of them, so its gz = gu W2 (1 - tanh(b1)^2) needs only that sum.
    float sum[D + 2];
#pragma unroll
    for (int i = 0; i < D + 2; ++i) sum[i] = 0.0f;
    // XCD x (= blockIdx & 7, see xcd_block) walks its own contiguous eighth of the segments, so
    // the rows its workgroups gather at any one time belong to a fraction of one graph
    const bool split = (gridDim.x & 7) == 0;
    const int64_t per = split ? (n_segments + 7) / 8 : n_segments;
    const int64_t lo = split ? (int64_t)(blockIdx.x & 7) * per : 0;
    const int64_t hi = lo + per < n_segments ? lo + per : n_segments;
    const int64_t lb = split ? blockIdx.x >> 3 : blockIdx.x, nlb = split ? gridDim.x >> 3 : gridDim.x;
    for (int64_t j = lo + lb * kBlock + threadIdx.x; j < hi; j += nlb * kBlock) {
        const int s = src[j], d = dst[j];
        if (pad_only && s >= 0) continue;          // (wide pull form: the hits' walks cover the real segments)
        const float ev = e[j];
        float gej = 0.0f;
        if (ge) {
            gej = ge[j];
        } else if (s >= 0) {
            float hs[LDH], hd[LDH], gmi_d[LDH], gmo_s[LDH];
            load_row4<LDH / 4>(H + (int64_t)s * LDH, hs);
            load_row4<LDH / 4>(H + (int64_t)d * LDH, hd);
            load_row4<LDH / 4>(gmio + (int64_t)d * 2 * LDH, gmi_d);
            load_row4<LDH / 4>(gmio + (int64_t)s * 2 * LDH + LDH, gmo_s);
#pragma unroll
            for (int c = 0; c < C; ++c) gej = fmaf(gmi_d[c], hs[c], gej);
#pragma unroll
            for (int c = 0; c < C; ++c) gej = fmaf(gmo_s[c], hd[c], gej);
        }
        const float gu = gej * ev * (1.0f - ev);
        if (s >= 0) {
            const float4 *p = reinterpret_cast<const float4 *>(PQ + (int64_t)s * 2 * D);
            const float4 *q = reinterpret_cast<const float4 *>(PQ + (int64_t)d * 2 * D + D);
#pragma unroll
            for (int v = 0; v < D / 4; ++v) {
                const float4 a = p[v], b = q[v];
                sum[4 * v] = fmaf(gu, tanh_f(a.x + b.x), sum[4 * v]);
                sum[4 * v + 1] = fmaf(gu, tanh_f(a.y + b.y), sum[4 * v + 1]);
                sum[4 * v + 2] = fmaf(gu, tanh_f(a.z + b.z), sum[4 * v + 2]);
                sum[4 * v + 3] = fmaf(gu, tanh_f(a.w + b.w), sum[4 * v + 3]);
            }
        } else {
#pragma unroll
            for (int i = 0; i < D; ++i) sum[i] = fmaf(gu, tanh_f(b1[i]), sum[i]);
            sum[D + 1] += gu;
        }
        sum[D] += gu;
        if (!pad_only) gu_out[j] = gu;
    }
    {
        constexpr int NW = kBlock / 64, N = D + 2;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            float x = sum[i];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
            if ((threadIdx.x & 63) == 0) lds[(threadIdx.x >> 6) * N + i] = x;
        }
        __syncthreads();
        if ((int)threadIdx.x <= D) {
            float x = 0.0f;
#pragma unroll
            for (int w = 0; w < NW; ++w) x += lds[w * N + threadIdx.x];
            *((int)threadIdx.x < D ? gW2 + threadIdx.x : gb2) += x;          // own row, one writer per element
        } else if ((int)threadIdx.x < 2 * D + 1) {
            const int i = threadIdx.x - D - 1;
            float x = 0.0f;
#pragma unroll
            for (int w = 0; w < NW; ++w) x += lds[w * N + D + 1];
            const float a = tanh_f(b1[i]);
            if (x != 0.0f) gb1[i] += x * W2[i] * (1.0f - a * a);
        }
    }
}

template <int F, int D>
__global__ __launch_bounds__(kBlock) void k_pq_bwd(
    const float *__restrict__ H, int ldh, const float *__restrict__ PQ, const float *__restrict__ gu,
    const int32_t *__restrict__ in_ptr, const int32_t *__restrict__ in_eid,
    const int32_t *__restrict__ in_nbr, const int32_t *__restrict__ out_ptr,
    const int32_t *__restrict__ out_eid, const int32_t *__restrict__ out_nbr,
    const float *__restrict__ W1, const float *__restrict__ W2, float *__restrict__ gH,
    float *__restrict__ gW1, float *__restrict__ gb1, int rep_stride, int64_t n_hits)
{
    gW1 = my_replica(gW1, rep_stride);
    gb1 = my_replica(gb1, rep_stride);
    constexpr int C = F + D, LDH = Shape<F, D>::LDH;
    constexpr int kLds = (2 * D + C + 1 <= kOuterJobRows && outer2_lds_floats<D, C>() > outer_lds_floats<D, C>())
                             ? outer2_lds_floats<D, C>() : outer_lds_floats<D, C>();
    __shared__ __attribute__((aligned(16))) float lds[kLds];
    const int64_t n = xcd_block() * kBlock + threadIdx.x;
    const bool active = n < n_hits;
    float gP[D], gQ[D], h[C];
#pragma unroll
    for (int i = 0; i < D; ++i) gP[i] = gQ[i] = 0.0f;
#pragma unroll
    for (int k = 0; k < C; ++k) h[k] = 0.0f;
    if (active) {
        // gz of a segment is rebuilt from the other end's P / Q row (small, cache-resident) and
        // the segment's gu instead of being stored per segment by k_edge_bwd and gathered here
        // (E x D floats through HBM, twice)
        constexpr int U = D <= 16 ? 4 : 2;
        float own[2 * D], w2[D];
        load_row4<2 * D / 4>(PQ + n * 2 * D, own);
#pragma unroll
        for (int i = 0; i < D; ++i) w2[i] = W2[i];
        csr_walk<D / 4, U>(out_ptr[n], out_ptr[n + 1],
                           [&](int k) { return PQ + (int64_t)out_nbr[k] * 2 * D + D; },      // Q[d]
                           [&](int k) { return gu[out_eid[k]]; },
                           [&](float g, const float *r) {
#pragma unroll
                               for (int i = 0; i < D; ++i) {
                                   const float a = tanh_f(own[i] + r[i]);
                                   gP[i] += g * w2[i] * (1.0f - a * a);
                               }
                           });
        csr_walk<D / 4, U>(in_ptr[n], in_ptr[n + 1],
                           [&](int k) { return PQ + (int64_t)in_nbr[k] * 2 * D; },           // P[s]
                           [&](int k) { return gu[in_eid[k]]; },
                           [&](float g, const float *r) {
#pragma unroll
                               for (int i = 0; i < D; ++i) {
                                   const float a = tanh_f(r[i] + own[D + i]);
                                   gQ[i] += g * w2[i] * (1.0f - a * a);
                               }
                           });
        float hr[LDH], gh[LDH];
        load_row4<LDH / 4>(H + n * ldh, hr);
        load_row4<LDH / 4>(gH + n * ldh, gh);
#pragma unroll
        for (int k = 0; k < C; ++k) h[k] = hr[k];
#pragma unroll
        for (int k = 0; k < C; ++k) {
            float acc = 0.0f;
#pragma unroll
            for (int i = 0; i < D; ++i) {
                acc = fmaf(W1[i * 2 * C + k], gP[i], acc);
                acc = fmaf(W1[i * 2 * C + C + k], gQ[i], acc);
            }
            gh[k] += acc;
        }
        store_row4<LDH / 4>(gH + n * ldh, gh);
    }
    if constexpr (2 * D + C + 1 <= kOuterJobRows) {
        accum_outer2<D, C>(gP, gQ, h, active, gW1, 2 * C, 0, C, gb1, lds);
    } else {
        accum_outer<D, C>(gP, h, active, gW1, 2 * C, 0, gb1, lds);
        accum_outer<D, C>(gQ, h, active, gW1, 2 * C, C, nullptr, lds);
    }
}

template <int F, int D>
__global__ __launch_bounds__(kBlock) void k_node_bwd(
    const float *__restrict__ H, const float *__restrict__ Hn, int ldh,
    const float *__restrict__ e, const int32_t *__restrict__ in_ptr,
    const int32_t *__restrict__ in_eid, const int32_t *__restrict__ in_nbr,
    const int32_t *__restrict__ out_ptr, const int32_t *__restrict__ out_eid,
    const int32_t *__restrict__ out_nbr, const float *__restrict__ W3,
    const float *__restrict__ b3, const float *__restrict__ W4, const float *__restrict__ gHn,
    float *__restrict__ gH, float *__restrict__ gmio, float *__restrict__ gW3,
    float *__restrict__ gb3, float *__restrict__ gW4, float *__restrict__ gb4, int rep_stride,
    int64_t n_hits)
{
    gW3 = my_replica(gW3, rep_stride);
    gb3 = my_replica(gb3, rep_stride);
    gW4 = my_replica(gW4, rep_stride);
    gb4 = my_replica(gb4, rep_stride);
    constexpr int C = Shape<F, D>::C, LDH = Shape<F, D>::LDH;
    __shared__ __attribute__((aligned(16))) float lds[outer_lds_floats<D, 3 * C>()];
    const int64_t n = xcd_block() * kBlock + threadIdx.x;
    const bool active = n < n_hits;
    float M[3 * C], q[D], gr[D], gp[D];
#pragma unroll
    for (int k = 0; k < 3 * C; ++k) M[k] = 0.0f;
#pragma unroll
    for (int i = 0; i < D; ++i) q[i] = gr[i] = gp[i] = 0.0f;
    if (active) {
        constexpr int U = D <= 16 ? 4 : 1;
        csr_walk<LDH / 4, U>(in_ptr[n], in_ptr[n + 1],
                             [&](int k) { return H + (int64_t)in_nbr[k] * ldh; },
                             [&](int k) { return e[in_eid[k]]; },
                             [&](float w, const float *hp) {
#pragma unroll
                                 for (int c = 0; c < C; ++c) M[c] = fmaf(w, hp[c], M[c]);
                             });
        csr_walk<LDH / 4, U>(out_ptr[n], out_ptr[n + 1],
                             [&](int k) { return H + (int64_t)out_nbr[k] * ldh; },
                             [&](int k) { return e[out_eid[k]]; },
                             [&](float w, const float *hp) {
#pragma unroll
                                 for (int c = 0; c < C; ++c) M[C + c] = fmaf(w, hp[c], M[C + c]);
                             });
        {
            float hp[LDH];
            load_row4<LDH / 4>(H + n * ldh, hp);
#pragma unroll
            for (int c = 0; c < C; ++c) M[2 * C + c] = hp[c];
        }
#pragma unroll
        for (int i = 0; i < D; ++i) {
            float acc = b3[i];
#pragma unroll
            for (int k = 0; k < 3 * C; ++k) acc = fmaf(W3[i * 3 * C + k], M[k], acc);
            q[i] = tanh_f(acc);
        }
        {
            float hn[D], gn[D];
            load_row4<D / 4>(Hn + n * ldh, hn);
            load_row4<D / 4>(gHn + n * ldh, gn);
#pragma unroll
            for (int i = 0; i < D; ++i) gr[i] = gn[i] * (1.0f - hn[i] * hn[i]);
        }
#pragma unroll
        for (int k = 0; k < D; ++k) {
            float acc = 0.0f;
#pragma unroll
            for (int i = 0; i < D; ++i) acc = fmaf(W4[i * D + k], gr[i], acc);
            gp[k] = acc * (1.0f - q[k] * q[k]);
        }
        float gm[3][LDH];                          // gmi | gmo | gHself, rows padded to LDH
#pragma unroll
        for (int k = 0; k < 3 * LDH; ++k) gm[k / LDH][k % LDH] = 0.0f;
#pragma unroll
        for (int k = 0; k < 3 * C; ++k) {
            float acc = 0.0f;
#pragma unroll
            for (int i = 0; i < D; ++i) acc = fmaf(W3[i * 3 * C + k], gp[i], acc);
            gm[k / C][k % C] = acc;
        }
        store_row4<LDH / 4>(gmio + n * 2 * LDH, gm[0]);          // [gmi | gmo], rows of LDH
        store_row4<LDH / 4>(gmio + n * 2 * LDH + LDH, gm[1]);
        store_row4<LDH / 4>(gH + n * ldh, gm[2]);                // gHself initialises gH_prev
    }
    accum_outer<D, 3 * C>(gp, M, active, gW3, 3 * C, 0, gb3, lds);
    accum_outer<D, D>(gr, q, active, gW4, D, 0, gb4, lds);
}

template <int F, int D>
__global__ __launch_bounds__(kBlock) void k_agg_bwd_n(
    const float *__restrict__ e, const float *__restrict__ gmio,
    const int32_t *__restrict__ in_ptr, const int32_t *__restrict__ in_eid,
    const int32_t *__restrict__ in_nbr, const int32_t *__restrict__ out_ptr,
    const int32_t *__restrict__ out_eid, const int32_t *__restrict__ out_nbr,
    float *__restrict__ gH, int ldh, int64_t n_hits)
{
    constexpr int C = F + D, LDH = Shape<F, D>::LDH;
    const int64_t n = xcd_block() * kBlock + threadIdx.x;
    if (n >= n_hits) return;
    float acc[LDH];
    load_row4<LDH / 4>(gH + n * ldh, acc);
    float sum[C];
#pragma unroll
    for (int c = 0; c < C; ++c) sum[c] = 0.0f;
    // n is the START of these segments: H[n] entered mi of the end hit d
    constexpr int U = D <= 16 ? 4 : 2;
    auto add = [&](float w, const float *g) {
#pragma unroll
        for (int c = 0; c < C; ++c) sum[c] = fmaf(w, g[c], sum[c]);
    };
    csr_walk<LDH / 4, U>(out_ptr[n], out_ptr[n + 1],
                         [&](int k) { return gmio + (int64_t)out_nbr[k] * 2 * LDH; },            // gmi[d]
                         [&](int k) { return e[out_eid[k]]; }, add);
    // n is the END of these segments: H[n] entered mo of the start hit s
    csr_walk<LDH / 4, U>(in_ptr[n], in_ptr[n + 1],
                         [&](int k) { return gmio + (int64_t)in_nbr[k] * 2 * LDH + LDH; },       // gmo[s]
                         [&](int k) { return e[in_eid[k]]; }, add);
#pragma unroll
    for (int c = 0; c < C; ++c) acc[c] += sum[c];
    store_row4<LDH / 4>(gH + n * ldh, acc);
}

template <int F, int D>
__global__ __launch_bounds__(kBlock) void k_input_bwd(const float *__restrict__ X,
                                                      const float *__restrict__ H0, int ldh,
                                                      const float *__restrict__ gH,
                                                      float *__restrict__ gWin,
                                                      float *__restrict__ gbin, int rep_stride,
                                                      int64_t n_hits)
{
    gWin = my_replica(gWin, rep_stride);
    gbin = my_replica(gbin, rep_stride);
    __shared__ __attribute__((aligned(16))) float lds[outer_lds_floats<D, F>()];
    const int64_t n = xcd_block() * kBlock + threadIdx.x;
    const bool active = n < n_hits;
    float g[D], x[F];
#pragma unroll
    for (int i = 0; i < D; ++i) g[i] = 0.0f;
#pragma unroll
    for (int k = 0; k < F; ++k) x[k] = 0.0f;
    if (active) {
#pragma unroll
        for (int i = 0; i < D; ++i) {
            const float h = H0[n * ldh + i];
            g[i] = gH[n * ldh + i] * (1.0f - h * h);
        }
#pragma unroll
        for (int k = 0; k < F; ++k) x[k] = X[n * F + k];
    }
    accum_outer<D, F>(g, x, active, gWin, F, 0, gbin, lds);
}

// ---------------------------------------------------------------------------------------------
// Pull-form backward of one message-passing iteration (node pass + the edge pass that fed it),
// hidden_dim <= 16.  The four kernels above move C-wide rows (H, gmi, gmo) and visit every segment
// six times per iteration with twelve row gathers.  The same algebra that makes the forward cheap
// (sell_pipeline.hip) applies to the gradients:
//     mi = sum_in e H[s]  enters only as W3a mi = sum_in e R[s],  R = W3a H  (D wide)
//     <gmi[d], H[s]> = gp[d] . R[s],   <gmo[s], H[d]> = gp[s] . S[d],   S = W3b H
//     sum_out e gmi[d] = W3a^T Gout,  Gout[s] = sum_out e gp[d];   sum_in e gmo[s] = W3b^T Gin
//     gW3a = sum_s Gout[s] H[s]^T,  gW3b = sum_d Gin[d] H[d]^T     (no aggregated M needed)
// so per hit two records A = [P | R | gp], B = [Q | S | gp] (3D floats) carry everything a neighbour
// must supply, a hit PULLS along its two CSR lists (fixed order, no atomics), and a segment's score
// gradient is rebuilt at both of its ends:
//   kb_prs    per hit:  P, R, Q, S from H                                   (fills A, B)
//   k_hit_bwd per hit:  acc = U + sum_in e R[s] + sum_out e S[d] (the forward's sum), q, gp;
//                       gp into A and B;  gH_prev = W3c^T gp;  gW3c, gb3, gW4, gb4
//   k_seg_bwd per hit:  out list: ge, gu, gz from (Q, S, gp)[d] -> gP, Gout;  in list: from
//                       (P, R, gp)[s] -> gQ, Gin (+ gW2, gb2 once per segment);
//                       gH_prev += W1a^T gP + W1b^T gQ + W3a^T Gout + W3b^T Gin;  gW1, gb1, gW3a, gW3b
// Two kernels and four row gathers per segment and iteration instead of four and twelve.
template <int F, int D>
__global__ __launch_bounds__(kBlock) void kb_prs(const float *__restrict__ H, int ldh, const float *__restrict__ W1,
                                                 const float *__restrict__ b1, const float *__restrict__ W3,
                                                 float *__restrict__ A, float *__restrict__ B, int64_t n_hits)
{
    constexpr int C = F + D;
    const int64_t n = xcd_block() * kBlock + threadIdx.x;
    if (n >= n_hits) return;
    float h[C];
#pragma unroll
    for (int k = 0; k < C; ++k) h[k] = H[n * ldh + k];
    float a[2 * D], b[2 * D];
#pragma unroll
    for (int i = 0; i < D; ++i) {
        float pp = b1[i], qq = 0.0f, rr = 0.0f, ss = 0.0f;
#pragma unroll
        for (int k = 0; k < C; ++k) {
            pp = fmaf(W1[i * 2 * C + k], h[k], pp);
            qq = fmaf(W1[i * 2 * C + C + k], h[k], qq);
            rr = fmaf(W3[i * 3 * C + k], h[k], rr);
            ss = fmaf(W3[i * 3 * C + C + k], h[k], ss);
        }
        a[i] = pp; a[D + i] = rr; b[i] = qq; b[D + i] = ss;
    }
    store_row4<2 * D / 4>(A + n * 3 * D, a);
    store_row4<2 * D / 4>(B + n * 3 * D, b);
}

template <int F, int D, bool KEPT>
__global__ __launch_bounds__(kBlock) void k_hit_bwd(
    const float *__restrict__ H, const float *__restrict__ Hn, const float *__restrict__ Qk, int ldh,
    const float *__restrict__ e,
    const int32_t *__restrict__ in_ptr, const int32_t *__restrict__ in_eid, const int32_t *__restrict__ in_nbr,
    const int32_t *__restrict__ out_ptr, const int32_t *__restrict__ out_eid, const int32_t *__restrict__ out_nbr,
    const float *__restrict__ W3, const float *__restrict__ b3, const float *__restrict__ W4,
    const float *__restrict__ gHn, float *__restrict__ gH, float *A, float *B, float *__restrict__ gW3,
    float *__restrict__ gb3, float *__restrict__ gW4, float *__restrict__ gb4, int rep_stride, int64_t n_hits)
{
    gW3 = my_replica(gW3, rep_stride);
    gb3 = my_replica(gb3, rep_stride);
    gW4 = my_replica(gW4, rep_stride);
    gb4 = my_replica(gb4, rep_stride);
    constexpr int C = Shape<F, D>::C, LDH = Shape<F, D>::LDH;
    __shared__ __attribute__((aligned(16))) float lds[outer_lds_floats<D, C>()];
    const int64_t n = xcd_block() * kBlock + threadIdx.x;
    const bool active = n < n_hits;
    float h[C], q[D], gr[D], gp[D];
#pragma unroll
    for (int k = 0; k < C; ++k) h[k] = 0.0f;
#pragma unroll
    for (int i = 0; i < D; ++i) q[i] = gr[i] = gp[i] = 0.0f;
    if (active) {
        float hp[LDH];
        load_row4<LDH / 4>(H + n * ldh, hp);
#pragma unroll
        for (int k = 0; k < C; ++k) h[k] = hp[k];
        if constexpr (KEPT) {
            load_row4<D / 4>(Qk + n * D, q);                     // kept by the training forward (k_node)
        } else {
            float acc[D];
    #pragma unroll
            for (int i = 0; i < D; ++i) {
                float u = b3[i];
    #pragma unroll
                for (int k = 0; k < C; ++k) u = fmaf(W3[i * 3 * C + 2 * C + k], h[k], u);
                acc[i] = u;
            }
            auto add = [&](float w, const float *r) {
    #pragma unroll
                for (int i = 0; i < D; ++i) acc[i] = fmaf(w, r[i], acc[i]);
            };
            csr_walk<D / 4, 4>(in_ptr[n], in_ptr[n + 1],
                               [&](int k) { return A + (int64_t)in_nbr[k] * 3 * D + D; },            // R[s]
                               [&](int k) { return e[in_eid[k]]; }, add);
            csr_walk<D / 4, 4>(out_ptr[n], out_ptr[n + 1],
                               [&](int k) { return B + (int64_t)out_nbr[k] * 3 * D + D; },           // S[d]
                               [&](int k) { return e[out_eid[k]]; }, add);
    #pragma unroll
            for (int i = 0; i < D; ++i) q[i] = tanh_f(acc[i]);
        }
        float hn[D], gn[D];
        load_row4<D / 4>(Hn + n * ldh, hn);
        load_row4<D / 4>(gHn + n * ldh, gn);
#pragma unroll
        for (int i = 0; i < D; ++i) gr[i] = gn[i] * (1.0f - hn[i] * hn[i]);
#pragma unroll
        for (int k = 0; k < D; ++k) {
            float s = 0.0f;
#pragma unroll
            for (int i = 0; i < D; ++i) s = fmaf(W4[i * D + k], gr[i], s);
            gp[k] = s * (1.0f - q[k] * q[k]);
        }
        float gh[LDH];
#pragma unroll
        for (int k = 0; k < LDH; ++k) gh[k] = 0.0f;
#pragma unroll
        for (int k = 0; k < C; ++k) {
            float s = 0.0f;
#pragma unroll
            for (int i = 0; i < D; ++i) s = fmaf(W3[i * 3 * C + 2 * C + k], gp[i], s);
            gh[k] = s;
        }
        store_row4<LDH / 4>(gH + n * ldh, gh);                     // gHself initialises gH_prev
    }
    // every lane has finished READING neighbours' R / S before anyone writes its gp next to them?
    // R / S and gp are different words of a record: no ordering is needed.
    if (active) {
        store_row4<D / 4>(A + n * 3 * D + 2 * D, gp);
        store_row4<D / 4>(B + n * 3 * D + 2 * D, gp);
    }
    accum_outer<D, C>(gp, h, active, gW3, 3 * C, 2 * C, gb3, lds);
    accum_outer<D, D>(gr, q, active, gW4, D, 0, gb4, lds);
}

template <int F, int D>
__global__ __launch_bounds__(kBlock) void k_seg_bwd(
    const float *__restrict__ H, int ldh, const float *__restrict__ A, const float *__restrict__ B,
    const float *__restrict__ e, const int32_t *__restrict__ in_ptr, const int32_t *__restrict__ in_eid,
    const int32_t *__restrict__ in_nbr, const int32_t *__restrict__ out_ptr, const int32_t *__restrict__ out_eid,
    const int32_t *__restrict__ out_nbr, const float *__restrict__ W1, const float *__restrict__ W2,
    const float *__restrict__ W3, float *__restrict__ gH, float *__restrict__ gW1, float *__restrict__ gb1,
    float *__restrict__ gW2, float *__restrict__ gb2, float *__restrict__ gW3, int rep_stride, int64_t n_hits)
{
    gW1 = my_replica(gW1, rep_stride);
    gb1 = my_replica(gb1, rep_stride);
    gW2 = my_replica(gW2, rep_stride);
    gb2 = my_replica(gb2, rep_stride);
    gW3 = my_replica(gW3, rep_stride);
    constexpr int C = Shape<F, D>::C, LDH = Shape<F, D>::LDH;
    __shared__ __attribute__((aligned(16))) float lds[outer_lds_floats<D, C>()];
    const int64_t n = xcd_block() * kBlock + threadIdx.x;
    const bool active = n < n_hits;
    float gP[D], gQ[D], Gout[D], Gin[D], h[C], sw2[D + 1];
#pragma unroll
    for (int i = 0; i < D; ++i) gP[i] = gQ[i] = Gout[i] = Gin[i] = 0.0f;
#pragma unroll
    for (int i = 0; i <= D; ++i) sw2[i] = 0.0f;
#pragma unroll
    for (int k = 0; k < C; ++k) h[k] = 0.0f;
    if (active) {
        float a[3 * D], b[3 * D], w2[D];                 // own [P | R | gp], [Q | S | gp]
        load_row4<3 * D / 4>(A + n * 3 * D, a);
        load_row4<3 * D / 4>(B + n * 3 * D, b);
#pragma unroll
        for (int i = 0; i < D; ++i) w2[i] = W2[i];
        constexpr int U = 2;                             // (4 rows of 3D floats in flight cost half the occupancy)
        // segments starting here (n -> d): the end hit supplies [Q | S | gp]
        csr_walk<3 * D / 4, U>(out_ptr[n], out_ptr[n + 1],
                               [&](int k) { return B + (int64_t)out_nbr[k] * 3 * D; },
                               [&](int k) { return e[out_eid[k]]; },
                               [&](float ev, const float *r) {
                                   float ge = 0.0f;
#pragma unroll
                                   for (int i = 0; i < D; ++i) ge = fmaf(r[2 * D + i], a[D + i], ge);     // gp[d] . R[n]
#pragma unroll
                                   for (int i = 0; i < D; ++i) ge = fmaf(a[2 * D + i], r[D + i], ge);     // gp[n] . S[d]
                                   const float gu = ge * ev * (1.0f - ev);
#pragma unroll
                                   for (int i = 0; i < D; ++i) {
                                       const float t = tanh_f(a[i] + r[i]);
                                       gP[i] = fmaf(gu * w2[i], 1.0f - t * t, gP[i]);
                                       Gout[i] = fmaf(ev, r[2 * D + i], Gout[i]);
                                   }
                               });
        // segments ending here (s -> n): the start hit supplies [P | R | gp]; W2 / b2 sums are taken here
        csr_walk<3 * D / 4, U>(in_ptr[n], in_ptr[n + 1],
                               [&](int k) { return A + (int64_t)in_nbr[k] * 3 * D; },
                               [&](int k) { return e[in_eid[k]]; },
                               [&](float ev, const float *r) {
                                   float ge = 0.0f;
#pragma unroll
                                   for (int i = 0; i < D; ++i) ge = fmaf(b[2 * D + i], r[D + i], ge);     // gp[n] . R[s]
#pragma unroll
                                   for (int i = 0; i < D; ++i) ge = fmaf(r[2 * D + i], b[D + i], ge);     // gp[s] . S[n]
                                   const float gu = ge * ev * (1.0f - ev);
#pragma unroll
                                   for (int i = 0; i < D; ++i) {
                                       const float t = tanh_f(r[i] + b[i]);
                                       gQ[i] = fmaf(gu * w2[i], 1.0f - t * t, gQ[i]);
                                       Gin[i] = fmaf(ev, r[2 * D + i], Gin[i]);
                                       sw2[i] = fmaf(gu, t, sw2[i]);
                                   }
                                   sw2[D] += gu;
                               });
        float hr[LDH], gh[LDH];
        load_row4<LDH / 4>(H + n * ldh, hr);
        load_row4<LDH / 4>(gH + n * ldh, gh);
#pragma unroll
        for (int k = 0; k < C; ++k) h[k] = hr[k];
#pragma unroll
        for (int k = 0; k < C; ++k) {
            float s = 0.0f;
#pragma unroll
            for (int i = 0; i < D; ++i) {
                s = fmaf(W1[i * 2 * C + k], gP[i], s);
                s = fmaf(W1[i * 2 * C + C + k], gQ[i], s);
                s = fmaf(W3[i * 3 * C + k], Gout[i], s);
                s = fmaf(W3[i * 3 * C + C + k], Gin[i], s);
            }
            gh[k] += s;
        }
        store_row4<LDH / 4>(gH + n * ldh, gh);
    }
    accum_outer<D, C>(gP, h, active, gW1, 2 * C, 0, gb1, lds);
    accum_outer<D, C>(gQ, h, active, gW1, 2 * C, C, nullptr, lds);
    accum_outer<D, C>(Gout, h, active, gW3, 3 * C, 0, nullptr, lds);
    accum_outer<D, C>(Gin, h, active, gW3, 3 * C, C, nullptr, lds);
    {   // gW2[D] | gb2: wave sums, then one writer per element of this workgroup's row
        constexpr int NW = kBlock / 64, NS = D + 1;
        float *red = lds;
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            float x = sw2[i];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
            if ((threadIdx.x & 63) == 0) red[(threadIdx.x >> 6) * NS + i] = x;
        }
        __syncthreads();
        if ((int)threadIdx.x <= D) {
            float x = 0.0f;
#pragma unroll
            for (int w = 0; w < NW; ++w) x += red[w * NS + threadIdx.x];
            *((int)threadIdx.x < D ? gW2 + threadIdx.x : gb2) += x;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// The same pull with FOUR LANES PER HIT (Q_all kept by the forward).  One lane per hit reading 96-byte
// rows touches 64 different lines per load instruction and keeps 243 registers; here a quad owns a
// hit, lane q holds dims [q D/4, (q+1) D/4) of every vector, a record is stored so that a lane's
// share is contiguous ([P(dl) R(dl)] x 4 | gp), a wave instruction touches 16 rows instead of 64, four
// list entries are taken per step (indices and scores loaded one per lane, rows broadcast in the
// quad), and a segment's ge is finished by two quad adds.  The per-hit dense tail (W^T products and
// the outer-product sums) stays with one lane per hit in k_seg_fin, through 4D floats per hit.
//   k_hit_bwd4  per hit (1 lane): P R Q S from H, gp from the kept q -> A, B;  gH_prev = W3c^T gp; gW3c, gb3, gW4, gb4
//   k_seg_bwd4  per hit (4 lanes): both list walks -> G4 = [gP gQ Gout Gin];  gW2, gb2
//   k_seg_fin   per hit (1 lane): gH_prev += W1a^T gP + W1b^T gQ + W3a^T Gout + W3b^T Gin;  gW1, gb1, gW3a, gW3b
template <int SEL>
__device__ __forceinline__ float quad_bcast(float x)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), SEL * 0x55, 0xf, 0xf, true));
}
template <int SEL>
__device__ __forceinline__ int quad_bcast(int x) { return __builtin_amdgcn_mov_dpp(x, SEL * 0x55, 0xf, 0xf, true); }
__device__ __forceinline__ float quad_sum(float x)
{
    x += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0xB1, 0xf, 0xf, true));   // lane ^ 1
    x += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0x4E, 0xf, 0xf, true));   // lane ^ 2
    return x;
}
template <int N>
__device__ __forceinline__ void load_vec(const float *__restrict__ p, float *v)
{
    if constexpr (N == 1) {
        v[0] = p[0];
    } else if constexpr (N == 2) {
        const float2 a = *reinterpret_cast<const float2 *>(p);
        v[0] = a.x; v[1] = a.y;
    } else {
#pragma unroll
        for (int i = 0; i < N / 4; ++i) {
            const float4 a = reinterpret_cast<const float4 *>(p)[i];
            v[4 * i] = a.x; v[4 * i + 1] = a.y; v[4 * i + 2] = a.z; v[4 * i + 3] = a.w;
        }
    }
}
template <int N>
__device__ __forceinline__ void store_vec(float *__restrict__ p, const float *v)
{
    if constexpr (N == 1) {
        p[0] = v[0];
    } else if constexpr (N == 2) {
        *reinterpret_cast<float2 *>(p) = make_float2(v[0], v[1]);
    } else {
#pragma unroll
        for (int i = 0; i < N / 4; ++i)
            reinterpret_cast<float4 *>(p)[i] = make_float4(v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]);
    }
}

// LDS floats of the per-hit kernels' outer-product stagings (hit: gp | gr | h | q | ones when that fits, fin: two left
// factors | h | ones)
template <int F, int D>
constexpr int hit_lds_floats()
{
    constexpr int C = Shape<F, D>::C;
    return (3 * D + C + 1 <= kOuterJobRows && kOuterStride * (3 * D + C + 1) > outer_lds_floats<D, C>())
               ? kOuterStride * (3 * D + C + 1) : outer_lds_floats<D, C>();
}

// k_hit_bwd4's work for hit n, given the next pass's features hn[D] and their gradient gn[D] in registers
// (gW3 .. gb4: the workgroup's own rows of the partial table)
template <int F, int D>
__device__ __forceinline__ void hit_bwd4_core(
    int64_t n, bool active, const float *__restrict__ H, const float *__restrict__ Qk, int ldh, const float *hn,
    const float *gn, const float *__restrict__ W1, const float *__restrict__ b1, const float *__restrict__ W3,
    const float *__restrict__ W4, float *__restrict__ gH, float *__restrict__ A, float *__restrict__ B, float *gW3,
    float *gb3, float *gW4, float *gb4, float *lds)
{
    constexpr int C = Shape<F, D>::C, LDH = Shape<F, D>::LDH, DL = D / 4;
    float h[C], q[D], gr[D], gp[D];
#pragma unroll
    for (int k = 0; k < C; ++k) h[k] = 0.0f;
#pragma unroll
    for (int i = 0; i < D; ++i) q[i] = gr[i] = gp[i] = 0.0f;
    if (active) {
        float hp[LDH];
        load_row4<LDH / 4>(H + n * ldh, hp);
#pragma unroll
        for (int k = 0; k < C; ++k) h[k] = hp[k];
        load_row4<D / 4>(Qk + n * D, q);
#pragma unroll
        for (int i = 0; i < D; ++i) gr[i] = gn[i] * (1.0f - hn[i] * hn[i]);
#pragma unroll
        for (int k = 0; k < D; ++k) {
            float s = 0.0f;
#pragma unroll
            for (int i = 0; i < D; ++i) s = fmaf(W4[i * D + k], gr[i], s);
            gp[k] = s * (1.0f - q[k] * q[k]);
        }
        float gh[LDH];
#pragma unroll
        for (int k = 0; k < LDH; ++k) gh[k] = 0.0f;
#pragma unroll
        for (int k = 0; k < C; ++k) {
            float s = 0.0f;
#pragma unroll
            for (int i = 0; i < D; ++i) s = fmaf(W3[i * 3 * C + 2 * C + k], gp[i], s);
            gh[k] = s;
        }
        store_row4<LDH / 4>(gH + n * ldh, gh);                     // gHself initialises gH_prev
        float a[3 * D], b[3 * D];
#pragma unroll
        for (int i = 0; i < D; ++i) {
            float pp = b1[i], qq = 0.0f, rr = 0.0f, ss = 0.0f;
#pragma unroll
            for (int k = 0; k < C; ++k) {
                pp = fmaf(W1[i * 2 * C + k], h[k], pp);
                qq = fmaf(W1[i * 2 * C + C + k], h[k], qq);
                rr = fmaf(W3[i * 3 * C + k], h[k], rr);
                ss = fmaf(W3[i * 3 * C + C + k], h[k], ss);
            }
            const int at = (i / DL) * 2 * DL + (i % DL);            // lane i / DL's share: [P(DL) R(DL)]
            a[at] = pp; a[at + DL] = rr; b[at] = qq; b[at + DL] = ss;
            a[2 * D + i] = gp[i]; b[2 * D + i] = gp[i];
        }
        store_row4<3 * D / 4>(A + n * 3 * D, a);
        store_row4<3 * D / 4>(B + n * 3 * D, b);
    }
    if constexpr (3 * D + C + 1 <= kOuterJobRows) {
        // both sums in one staging: rows gp | gr | h | q | ones
#ifndef GNN_ABLATE_OUTER
#pragma unroll
        for (int i = 0; i < D; ++i) {
            stage_row(lds, i, active ? gp[i] : 0.0f);
            stage_row(lds, D + i, active ? gr[i] : 0.0f);
            stage_row(lds, 2 * D + C + i, active ? q[i] : 0.0f);
        }
#pragma unroll
        for (int k = 0; k < C; ++k) stage_row(lds, 2 * D + k, active ? h[k] : 0.0f);
        stage_row(lds, 3 * D + C, active ? 1.0f : 0.0f);
        const OuterJob jobs[2] = {{0, 2 * D, C, 3 * C, 2 * C, gW3, gb3}, {D, 2 * D + C, D, D, 0, gW4, gb4}};
        accum_outer_jobs<2>(jobs, D, 3 * D + C, lds);
#endif
    } else {
        accum_outer<D, C>(gp, h, active, gW3, 3 * C, 2 * C, gb3, lds);
        accum_outer<D, D>(gr, q, active, gW4, D, 0, gb4, lds);
    }
}

template <int F, int D>
__global__ __launch_bounds__(kBlock) void k_hit_bwd4(
    const float *__restrict__ H, const float *__restrict__ Hn, const float *__restrict__ Qk, int ldh,
    const float *__restrict__ W1, const float *__restrict__ b1, const float *__restrict__ W3,
    const float *__restrict__ W4, const float *__restrict__ gHn, float *__restrict__ gH, float *__restrict__ A,
    float *__restrict__ B, float *__restrict__ gW3, float *__restrict__ gb3, float *__restrict__ gW4,
    float *__restrict__ gb4, int rep_stride, int64_t n_hits)
{
    __shared__ __attribute__((aligned(16))) float lds[hit_lds_floats<F, D>()];
    const int64_t n = xcd_block() * kBlock + threadIdx.x;
    const bool active = n < n_hits;
    float hn[D], gn[D];
#pragma unroll
    for (int i = 0; i < D; ++i) hn[i] = gn[i] = 0.0f;
    if (active) {
        load_row4<D / 4>(Hn + n * ldh, hn);
        load_row4<D / 4>(gHn + n * ldh, gn);
    }
    hit_bwd4_core<F, D>(n, active, H, Qk, ldh, hn, gn, W1, b1, W3, W4, gH, A, B, my_replica(gW3, rep_stride),
                        my_replica(gb3, rep_stride), my_replica(gW4, rep_stride), my_replica(gb4, rep_stride), lds);
}

// one direction of a hit's pull: REC = the far ends' records, own_pq / own_r / own_gp the hit's own
// shares (out list: P, R of the start hit against [Q S gp] of the end hits; in list: Q, S against [P R gp])
template <int D, bool IN>
__device__ __forceinline__ void quad_walk(int beg, int end, int n, int q, const int32_t *__restrict__ nbr,
                                          const int32_t *__restrict__ eid, const float *__restrict__ e,
                                          const float *__restrict__ REC, const float *own_pq, const float *own_r,
                                          const float *own_gp, const float *w2, float *gZ, float *G, float *sw2)
{
    constexpr int DL = D / 4;
    for (int k = beg; k < end; k += 4) {
        const int kk = k + q;
        const bool ok = kk < end;
        const int nb = ok ? nbr[kk] : n;                        // (a masked entry reads the own record with score 0)
        const float ev = ok ? e[eid[kk]] : 0.0f;
        float pr[4][2 * DL], gv[4][DL], part[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int nbj = j == 0 ? quad_bcast<0>(nb) : j == 1 ? quad_bcast<1>(nb) : j == 2 ? quad_bcast<2>(nb) : quad_bcast<3>(nb);
            const float *r = REC + (int64_t)nbj * 3 * D;
            load_vec<2 * DL>(r + q * 2 * DL, pr[j]);
            load_vec<DL>(r + 2 * D + q * DL, gv[j]);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float p = 0.0f;
#pragma unroll
            for (int i = 0; i < DL; ++i) {
                p = fmaf(gv[j][i], own_r[i], p);
                p = fmaf(own_gp[i], pr[j][DL + i], p);
            }
            part[j] = quad_sum(p);
        }
        const float ge = q == 0 ? part[0] : q == 1 ? part[1] : q == 2 ? part[2] : part[3];
        const float gu = ge * ev * (1.0f - ev);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float guj = j == 0 ? quad_bcast<0>(gu) : j == 1 ? quad_bcast<1>(gu) : j == 2 ? quad_bcast<2>(gu) : quad_bcast<3>(gu);
            const float evj = j == 0 ? quad_bcast<0>(ev) : j == 1 ? quad_bcast<1>(ev) : j == 2 ? quad_bcast<2>(ev) : quad_bcast<3>(ev);
#pragma unroll
            for (int i = 0; i < DL; ++i) {
                const float t = tanh_f(own_pq[i] + pr[j][i]);
                gZ[i] = fmaf(guj * w2[i], 1.0f - t * t, gZ[i]);
                G[i] = fmaf(evj, gv[j][i], G[i]);
                if constexpr (IN) sw2[i] = fmaf(guj, t, sw2[i]);
            }
            if constexpr (IN) sw2[DL] += guj;
        }
    }
}

constexpr int kQuadBlock = 1024;          // 256 hits per workgroup: the same partial-table rows as the 1-lane kernels
template <int F, int D>
__global__ __launch_bounds__(kQuadBlock) void k_seg_bwd4(
    const float *__restrict__ A, const float *__restrict__ B, const float *__restrict__ e,
    const int32_t *__restrict__ in_ptr, const int32_t *__restrict__ in_eid, const int32_t *__restrict__ in_nbr,
    const int32_t *__restrict__ out_ptr, const int32_t *__restrict__ out_eid, const int32_t *__restrict__ out_nbr,
    const float *__restrict__ W2, float *__restrict__ G4, float *__restrict__ gW2, float *__restrict__ gb2,
    int rep_stride, int64_t n_hits)
{
    gW2 = my_replica(gW2, rep_stride);
    gb2 = my_replica(gb2, rep_stride);
    constexpr int DL = D / 4, NW = kQuadBlock / 64;
    __shared__ float red[NW * (D + 1)];
    const int q = threadIdx.x & 3;
    const int64_t n = xcd_block() * (kQuadBlock / 4) + (threadIdx.x >> 2);
    const bool active = n < n_hits;
    float gP[DL], gQ[DL], Gout[DL], Gin[DL], sw2[DL + 1], w2[DL];
#pragma unroll
    for (int i = 0; i < DL; ++i) {
        gP[i] = gQ[i] = Gout[i] = Gin[i] = sw2[i] = 0.0f;
        w2[i] = W2[q * DL + i];
    }
    sw2[DL] = 0.0f;
    if (active) {
        float a[2 * DL], b[2 * DL], gp[DL];
        load_vec<2 * DL>(A + n * 3 * D + q * 2 * DL, a);
        load_vec<2 * DL>(B + n * 3 * D + q * 2 * DL, b);
        load_vec<DL>(A + n * 3 * D + 2 * D + q * DL, gp);
        // segments starting here (n -> d): [Q | S | gp] of the end hits
        quad_walk<D, false>(out_ptr[n], out_ptr[n + 1], (int)n, q, out_nbr, out_eid, e, B, a, a + DL, gp, w2, gP, Gout, sw2);
        // segments ending here (s -> n): [P | R | gp] of the start hits; the W2 / b2 sums are taken here
        quad_walk<D, true>(in_ptr[n], in_ptr[n + 1], (int)n, q, in_nbr, in_eid, e, A, b, b + DL, gp, w2, gQ, Gin, sw2);
        float out[4 * DL];
#pragma unroll
        for (int i = 0; i < DL; ++i) {
            out[i] = gP[i]; out[DL + i] = gQ[i]; out[2 * DL + i] = Gout[i]; out[3 * DL + i] = Gin[i];
        }
        store_vec<4 * DL>(G4 + n * 4 * D + q * 4 * DL, out);
    }
    // gW2[D] | gb2: lanes with the same q across the wave, then the waves in order
#pragma unroll
    for (int i = 0; i <= DL; ++i) {
        float x = sw2[i];
#pragma unroll
        for (int o = 32; o >= 4; o >>= 1) x += __shfl_xor(x, o, 64);
        if ((threadIdx.x & 63) < 4) {
            if (i < DL) red[(threadIdx.x >> 6) * (D + 1) + q * DL + i] = x;
            else if (q == 0) red[(threadIdx.x >> 6) * (D + 1) + D] = x;
        }
    }
    __syncthreads();
    if ((int)threadIdx.x <= D) {
        float x = 0.0f;
#pragma unroll
        for (int w = 0; w < NW; ++w) x += red[w * (D + 1) + threadIdx.x];
        *((int)threadIdx.x < D ? gW2 + threadIdx.x : gb2) += x;
    }
}

// k_seg_fin's work for hit n: gh[LDH] comes in as the row k_hit_bwd4 initialised (gHself) and leaves as the complete
// gradient of the hit's features; hr[LDH] = the hit's feature row (handed on to a fused next step)
template <int F, int D>
__device__ __forceinline__ void seg_fin_core(int64_t n, bool active, const float *__restrict__ H, int ldh,
                                             const float *__restrict__ G4, const float *__restrict__ W1,
                                             const float *__restrict__ W3, float *gh, float *hr, float *gW1, float *gb1,
                                             float *gW3, float *lds)
{
    constexpr int C = Shape<F, D>::C, LDH = Shape<F, D>::LDH, DL = D / 4;
    float gP[D], gQ[D], Gout[D], Gin[D], h[C];
#pragma unroll
    for (int i = 0; i < D; ++i) gP[i] = gQ[i] = Gout[i] = Gin[i] = 0.0f;
#pragma unroll
    for (int k = 0; k < C; ++k) h[k] = 0.0f;
    if (active) {
        float g4[4 * D];
        load_row4<D>(G4 + n * 4 * D, g4);
#pragma unroll
        for (int i = 0; i < D; ++i) {
            const int at = (i / DL) * 4 * DL + (i % DL);
            gP[i] = g4[at]; gQ[i] = g4[at + DL]; Gout[i] = g4[at + 2 * DL]; Gin[i] = g4[at + 3 * DL];
        }
        load_row4<LDH / 4>(H + n * ldh, hr);
#pragma unroll
        for (int k = 0; k < C; ++k) h[k] = hr[k];
#pragma unroll
        for (int k = 0; k < C; ++k) {
            float s = 0.0f;
#pragma unroll
            for (int i = 0; i < D; ++i) {
                s = fmaf(W1[i * 2 * C + k], gP[i], s);
                s = fmaf(W1[i * 2 * C + C + k], gQ[i], s);
                s = fmaf(W3[i * 3 * C + k], Gout[i], s);
                s = fmaf(W3[i * 3 * C + C + k], Gin[i], s);
            }
            gh[k] += s;
        }
    }
    accum_outer2<D, C>(gP, gQ, h, active, gW1, 2 * C, 0, C, gb1, lds);
    accum_outer2<D, C>(Gout, Gin, h, active, gW3, 3 * C, 0, C, nullptr, lds);
}

template <int F, int D>
__global__ __launch_bounds__(kBlock) void k_seg_fin(
    const float *__restrict__ H, int ldh, const float *__restrict__ G4, const float *__restrict__ W1,
    const float *__restrict__ W3, float *__restrict__ gH, float *__restrict__ gW1, float *__restrict__ gb1,
    float *__restrict__ gW3, int rep_stride, int64_t n_hits)
{
    constexpr int C = Shape<F, D>::C, LDH = Shape<F, D>::LDH;
    __shared__ __attribute__((aligned(16))) float lds[outer2_lds_floats<D, C>()];
    const int64_t n = xcd_block() * kBlock + threadIdx.x;
    const bool active = n < n_hits;
    float gh[LDH], hr[LDH];
#pragma unroll
    for (int k = 0; k < LDH; ++k) gh[k] = hr[k] = 0.0f;
    if (active) load_row4<LDH / 4>(gH + n * ldh, gh);
    seg_fin_core<F, D>(n, active, H, ldh, G4, W1, W3, gh, hr, my_replica(gW1, rep_stride), my_replica(gb1, rep_stride),
                       my_replica(gW3, rep_stride), lds);
    if (active) store_row4<LDH / 4>(gH + n * ldh, gh);
}

// k_seg_fin of iteration u and k_hit_bwd4 of iteration u - 1 in one launch: both are one lane per hit, and what the
// first hands the second - the finished gradient row of H_{u-1} and that row itself - stays in registers (130 bytes
// per hit less through memory, one launch less per iteration).  H = H_{u-1}, Hpp = H_{u-2}, Qk = the node network's
// hidden layer of pass u - 2; gH = the row k_hit_bwd4 (u) initialised (read only), gHpp = the buffer the next
// iteration accumulates into; A / B are rewritten for iteration u - 1 (k_seg_bwd4 (u) has finished with them).
template <int F, int D>
__global__ __launch_bounds__(kBlock) void k_fin_hit(
    const float *__restrict__ H, const float *__restrict__ Hpp, const float *__restrict__ Qk, int ldh,
    const float *__restrict__ G4, const float *__restrict__ W1, const float *__restrict__ b1,
    const float *__restrict__ W3, const float *__restrict__ W4, const float *__restrict__ gH, float *__restrict__ gHpp,
    float *__restrict__ A, float *__restrict__ B, float *__restrict__ gW1, float *__restrict__ gb1,
    float *__restrict__ gW3, float *__restrict__ gb3, float *__restrict__ gW4, float *__restrict__ gb4, int rep_stride,
    int64_t n_hits)
{
    constexpr int C = Shape<F, D>::C, LDH = Shape<F, D>::LDH;
    constexpr int kLds = hit_lds_floats<F, D>() > outer2_lds_floats<D, C>() ? hit_lds_floats<F, D>() : outer2_lds_floats<D, C>();
    __shared__ __attribute__((aligned(16))) float lds[kLds];
    const int64_t n = xcd_block() * kBlock + threadIdx.x;
    const bool active = n < n_hits;
    float gh[LDH], hr[LDH];
#pragma unroll
    for (int k = 0; k < LDH; ++k) gh[k] = hr[k] = 0.0f;
    if (active) load_row4<LDH / 4>(gH + n * ldh, gh);
    float *const gW3r = my_replica(gW3, rep_stride);
    seg_fin_core<F, D>(n, active, H, ldh, G4, W1, W3, gh, hr, my_replica(gW1, rep_stride), my_replica(gb1, rep_stride),
                       gW3r, lds);
    hit_bwd4_core<F, D>(n, active, Hpp, Qk, ldh, hr, gh, W1, b1, W3, W4, gHpp, A, B, gW3r, my_replica(gb3, rep_stride),
                        my_replica(gW4, rep_stride), my_replica(gb4, rep_stride), lds);
}

// ---------------------------------------------------------------------------------------------
// WIDE hidden layers (hidden_dim 32 / 64, the reference's toy / ACTS / mu200 models): the same pull
// form with SIXTEEN lanes per hit in the list walk.  The per-pass kernels above keep C-wide rows in
// one lane's registers (3C = 201 floats of M at D = 64) and ran 25x the inference forward.  Here lane p
// of a hit's 16 owns dims [p DL, (p+1) DL), DL = D / 16; a record is stored [P(DL) R(DL)] x 16 | gp so
// that a lane's share is contiguous and the 16 lanes read a row as whole lines; a segment's ge is
// finished by the 4x4 transpose-add in the quad plus two row rotations (as k_iter_w's scores).
//   k_hit_bwdW  per hit (1 lane): P R Q S from H, gp from the kept q -> A, B; gH_prev = W3c^T gp; gW3c, gb3, gW4, gb4
//   k_seg_bwdW  per hit (16 lanes): both list walks -> G4 = [gP gQ Gout Gin] and the hit's W2 / b2 terms
//   k_seg_finW  per hit (1 lane): gH_prev += W^T G4;  gW1, gb1, gW3a, gW3b, gW2, gb2
template <int CTRL>
__device__ __forceinline__ float dppf(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}

// the dense wide kernels run 512 threads for their 256 hits: eight waves share the tiles, two per SIMD instead of one
constexpr int kFinThreads = 512;
// Columns [col0, col0 + 4 K4) of rows n0 .. n0 + 255 of a row-major array (row stride `ld` floats, 16-byte
// aligned pieces) -> LDS, transposed: dst[k * RS + hit].  Thread = hit reading its own row touches 64
// different lines per load instruction (the L1's lookup rate bounded the dense kernels: 1.1e7 lookups per
// launch in k_seg_finW); here consecutive threads read consecutive 16-byte pieces, a wave instruction
// covers 64 / K4 rows as whole lines.  Rows beyond n_hits read as zeros.
template <int K4, int NTH = kBlock>
__device__ __forceinline__ void stage_cols_T(const float *__restrict__ src, int ld, int col0, int64_t n0,
                                             int64_t n_hits, float *dst)
{
    constexpr int RS = kOuterStride;
#pragma unroll 4
    for (int j = threadIdx.x; j < kBlock * K4; j += NTH) {
        const int hit = j / K4, c = j % K4;
        float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (n0 + hit < n_hits) v = *reinterpret_cast<const float4 *>(src + (n0 + hit) * ld + col0 + 4 * c);
        float *d = dst + (4 * c) * RS + hit;
        d[0] = v.x; d[RS] = v.y; d[2 * RS] = v.z; d[3 * RS] = v.w;
    }
}

// All of it as exact fp32 matrix-core products over the workgroup's 256 hits (thread = hit only for
// loading and staging; lane l of a product holds rows 4 (l >> 4) .. + 3 of hit column l & 15):
//   phase 1  LDS [gr | q 1]:  gW4, gb4 += gr (x) [q 1];   gp = (W4^T gr) (1 - q^2)   (this wave's 64 hits)
//   phase 2  LDS [gp | h 1]:  gW3c, gb3 += gp (x) [h 1];  gH_prev = W3c^T gp;  records P R Q S = W h (+ b1)
// One lane per hit with scalar weight operands streamed 100 KB of weights per wave through the scalar
// cache (0.35 ms per launch at D = 64).
template <int F, int D>
__global__ __launch_bounds__(kFinThreads) void k_hit_bwdW(
    const float *__restrict__ H, const float *__restrict__ Hn, const float *__restrict__ Qk, int ldh,
    const float *__restrict__ W1, const float *__restrict__ b1, const float *__restrict__ W3,
    const float *__restrict__ W4, const float *__restrict__ gHn, float *__restrict__ gH, float *__restrict__ A,
    float *__restrict__ B, float *__restrict__ gW3, float *__restrict__ gb3, float *__restrict__ gW4,
    float *__restrict__ gb4, int rep_stride, int64_t n_hits)
{
    gW3 = my_replica(gW3, rep_stride);
    gb3 = my_replica(gb3, rep_stride);
    gW4 = my_replica(gW4, rep_stride);
    gb4 = my_replica(gb4, rep_stride);
    typedef float f4v __attribute__((ext_vector_type(4)));
    constexpr int C = Shape<F, D>::C, LDH = Shape<F, D>::LDH, DL = D / 16, RS = kOuterStride;
    constexpr int RT = D / 16, CT = (C + 1 + 15) / 16, KT = (LDH + 15) / 16, YR = 16 * CT;
    static_assert(D + 1 <= YR && LDH <= YR, "the [q | 1] block and the padded h rows fit the [h | 1] rows");
    __shared__ __attribute__((aligned(16))) float lds[(D + YR) * RS];
    float *X = lds, *Y = lds + D * RS;
    constexpr int NT = kFinThreads, NW = NT / 64, HW = kBlock / NW, HT = HW / 16;    // 512 threads for 256 hits
    const int64_t n0 = xcd_block() * kBlock, n = n0 + threadIdx.x;
    const bool is_hit = threadIdx.x < kBlock, active = is_hit && n < n_hits;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, r16 = lane & 15, g4 = lane >> 4;
    const int hcol = HW * wv + r16;                      // + 16 ht: this lane's hit column of tile ht
    // ---- phase 1 staging: X = gr, Y = [q | 1 | 0]
    if (is_hit) {
        float qv[D], hn[D], gn[D];
#pragma unroll
        for (int i = 0; i < D; ++i) qv[i] = hn[i] = gn[i] = 0.0f;
        if (active) {
            load_row4<D / 4>(Qk + n * D, qv);
            load_row4<D / 4>(Hn + n * ldh, hn);
            load_row4<D / 4>(gHn + n * ldh, gn);
        }
#pragma unroll
        for (int i = 0; i < D; ++i) {
            X[i * RS + threadIdx.x] = gn[i] * (1.0f - hn[i] * hn[i]);
            Y[i * RS + threadIdx.x] = qv[i];
        }
        Y[D * RS + threadIdx.x] = active ? 1.0f : 0.0f;
#pragma unroll
        for (int k = D + 1; k < YR; ++k) Y[k * RS + threadIdx.x] = 0.0f;
    }
    __syncthreads();
    auto outer = [&](int ncol, float *g, int ldg, int col0, float *gb) {      // X (x) Y[0 .. ncol) | ones at ncol
        const int ct = (ncol + 1 + 15) / 16;
        for (int t = wv; t < RT * ct; t += NW) {
            const int it = t / ct, jt = t % ct;
            const float *a = X + (16 * it + r16) * RS + g4, *b = Y + (16 * jt + r16) * RS + g4;
            f4v c = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll 8
            for (int st = 0; st < kBlock / 4; ++st) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[4 * st], b[4 * st], c, 0, 0, 0);
            const int col = 16 * jt + r16;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 16 * it + 4 * g4 + r;
                const float x = r == 0 ? c.x : r == 1 ? c.y : r == 2 ? c.z : c.w;
                if (col < ncol) g[i * ldg + col0 + col] += x;               // own row: the only writer
                else if (col == ncol) gb[i] += x;
            }
        }
    };
    outer(D, gW4, D, 0, gb4);
    f4v gp[RT][HT];                                       // gp rows 16 kt + 4 g4 + r of hits hcol + 16 ht
#pragma unroll
    for (int kt = 0; kt < RT; ++kt) {
#pragma unroll
        for (int ht = 0; ht < HT; ++ht) gp[kt][ht] = f4v{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll 4
        for (int st = 0; st < D / 4; ++st) {
            const float aw = W4[(4 * st + g4) * D + 16 * kt + r16];
            const float *bv = X + (4 * st + g4) * RS + hcol;
#pragma unroll
            for (int ht = 0; ht < HT; ++ht) gp[kt][ht] = __builtin_amdgcn_mfma_f32_16x16x4f32(aw, bv[16 * ht], gp[kt][ht], 0, 0, 0);
        }
#pragma unroll
        for (int ht = 0; ht < HT; ++ht) {
            const float *qy = Y + (16 * kt + 4 * g4) * RS + hcol + 16 * ht;
            const float q0 = qy[0], q1 = qy[RS], q2 = qy[2 * RS], q3 = qy[3 * RS];
            gp[kt][ht].x *= 1.0f - q0 * q0; gp[kt][ht].y *= 1.0f - q1 * q1;
            gp[kt][ht].z *= 1.0f - q2 * q2; gp[kt][ht].w *= 1.0f - q3 * q3;
        }
    }
    __syncthreads();                                     // everyone is done with gr and q
    // ---- phase 2 staging: X = gp (from the product's layout), Y = [h | 1 | 0]
#pragma unroll
    for (int kt = 0; kt < RT; ++kt)
#pragma unroll
        for (int ht = 0; ht < HT; ++ht) {
            float *x = X + (16 * kt + 4 * g4) * RS + hcol + 16 * ht;
            x[0] = gp[kt][ht].x; x[RS] = gp[kt][ht].y; x[2 * RS] = gp[kt][ht].z; x[3 * RS] = gp[kt][ht].w;
        }
    stage_cols_T<LDH / 4, NT>(H, ldh, 0, n0, n_hits, Y);             // h (and its padding columns)
    __syncthreads();
    if (is_hit) {
#pragma unroll
        for (int k = C; k < YR; ++k) Y[k * RS + threadIdx.x] = (k == C && active) ? 1.0f : 0.0f;    // ones row, zero pad
    }
    __syncthreads();
    outer(C, gW3 + 2 * C, 3 * C, 0, gb3);
    // gH_prev = W3c^T gp (initialises the row; k_seg_finW adds the segment terms)
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
        f4v c[HT];
#pragma unroll
        for (int ht = 0; ht < HT; ++ht) c[ht] = f4v{0.0f, 0.0f, 0.0f, 0.0f};
        const int k = 16 * kt + r16;
#pragma unroll 4
        for (int st = 0; st < D / 4; ++st) {
            const float aw = k < C ? W3[(4 * st + g4) * 3 * C + 2 * C + k] : 0.0f;
            const float *bv = X + (4 * st + g4) * RS + hcol;
#pragma unroll
            for (int ht = 0; ht < HT; ++ht) c[ht] = __builtin_amdgcn_mfma_f32_16x16x4f32(aw, bv[16 * ht], c[ht], 0, 0, 0);
        }
        const int k0 = 16 * kt + 4 * g4;
#pragma unroll
        for (int ht = 0; ht < HT; ++ht) {
            const int64_t nn = n0 + hcol + 16 * ht;
            if (nn < n_hits && k0 < LDH) *reinterpret_cast<f4v *>(gH + nn * ldh + k0) = c[ht];
        }
    }
    // records: [P R] -> A, [Q S] -> B (rows i of the products = dims; the ones row of Y carries b1 into P)
    constexpr int KS = (C + 1 + 3) / 4;                  // k-steps over [h | 1]
#pragma unroll 1
    for (int it = 0; it < RT; ++it) {
        const int i = 16 * it + r16;
#pragma unroll 1
        for (int half = 0; half < 2; ++half) {           // 0: P, R (W1a, W3a) -> A;  1: Q, S (W1b, W3b) -> B
            f4v c1[HT], c3[HT];
#pragma unroll
            for (int ht = 0; ht < HT; ++ht) c1[ht] = c3[ht] = f4v{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll 4
            for (int st = 0; st < KS; ++st) {
                const int kk = 4 * st + g4;
                float a1 = 0.0f, a3 = 0.0f;
                if (kk < C) {
                    a1 = W1[i * 2 * C + half * C + kk];
                    a3 = W3[i * 3 * C + half * C + kk];
                } else if (kk == C && half == 0) {
                    a1 = b1[i];
                }
                const float *bv = Y + kk * RS + hcol;
#pragma unroll
                for (int ht = 0; ht < HT; ++ht) {
                    const float bb = bv[16 * ht];
                    c1[ht] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, bb, c1[ht], 0, 0, 0);
                    c3[ht] = __builtin_amdgcn_mfma_f32_16x16x4f32(a3, bb, c3[ht], 0, 0, 0);
                }
            }
            float *REC = half == 0 ? A : B;
#pragma unroll
            for (int ht = 0; ht < HT; ++ht) {
                const int64_t nn = n0 + hcol + 16 * ht;
                if (nn >= n_hits) continue;
                float *rec = REC + nn * 3 * D;
                // this lane holds dims i0 .. i0 + 3 (i0 = 16 it + 4 g4): lane share i0 / DL (+1 at DL = 2)
                const int i0 = 16 * it + 4 * g4;
                if constexpr (DL == 4) {
                    *reinterpret_cast<f4v *>(rec + (i0 / 4) * 8) = c1[ht];
                    *reinterpret_cast<f4v *>(rec + (i0 / 4) * 8 + 4) = c3[ht];
                } else {
                    static_assert(DL == 2 || DL == 4, "record shares of 2 or 4 dims");
                    *reinterpret_cast<f4v *>(rec + (i0 / 2) * 4) = f4v{c1[ht].x, c1[ht].y, c3[ht].x, c3[ht].y};
                    *reinterpret_cast<f4v *>(rec + (i0 / 2 + 1) * 4) = f4v{c1[ht].z, c1[ht].w, c3[ht].z, c3[ht].w};
                }
            }
        }
    }
    // the gp field of both records (natural dim order)
#pragma unroll
    for (int kt = 0; kt < RT; ++kt)
#pragma unroll
        for (int ht = 0; ht < HT; ++ht) {
            const int64_t nn = n0 + hcol + 16 * ht;
            if (nn < n_hits) {
                *reinterpret_cast<f4v *>(A + nn * 3 * D + 2 * D + 16 * kt + 4 * g4) = gp[kt][ht];
                *reinterpret_cast<f4v *>(B + nn * 3 * D + 2 * D + 16 * kt + 4 * g4) = gp[kt][ht];
            }
        }
}

// one direction of a hit's pull, 16 lanes per hit (see quad_walk for the roles of the arguments)
template <int D, bool IN>
__device__ __forceinline__ void row_walk(int beg, int end, int n, int p, const int32_t *__restrict__ nbr,
                                         const int32_t *__restrict__ eid, const float *__restrict__ e,
                                         const float *__restrict__ ge_ext, const float *__restrict__ REC,
                                         const float *own_pq, const float *own_r, const float *own_gp, const float *w2,
                                         float *gZ, float *G, float *sw2)
{
    constexpr int DL = D / 16;
    const int q = p & 3;
    for (int k = beg; k < end; k += 4) {
        const int kk = k + q;
        const bool ok = kk < end;
        const int nb = ok ? nbr[kk] : n;                        // (a masked entry reads the own record with score 0)
        const int se = ok ? eid[kk] : 0;
        const float ev = ok ? e[se] : 0.0f;
        const float gx = (ok && ge_ext) ? ge_ext[se] : 0.0f;    // final edge pass: the loss gradient of the scores
        float pr[4][2 * DL], gv[4][DL], part[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int nbj = j == 0 ? quad_bcast<0>(nb) : j == 1 ? quad_bcast<1>(nb) : j == 2 ? quad_bcast<2>(nb) : quad_bcast<3>(nb);
            const float *r = REC + (int64_t)nbj * 3 * D;
            load_vec<2 * DL>(r + p * 2 * DL, pr[j]);
            load_vec<DL>(r + 2 * D + p * DL, gv[j]);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float x = 0.0f;
#pragma unroll
            for (int i = 0; i < DL; ++i) {
                x = fmaf(gv[j][i], own_r[i], x);
                x = fmaf(own_gp[i], pr[j][DL + i], x);
            }
            part[j] = x;
        }
        // lane q of every quad ends with the sum over the hit's 16 lanes of part[q]
        const bool odd = q & 1, hi = q & 2;
        const float s0 = odd ? part[0] : part[1], s1 = odd ? part[2] : part[3];
        const float k0 = odd ? part[1] : part[0], k1 = odd ? part[3] : part[2];
        const float t0 = k0 + dppf<0xB1>(s0), t1 = k1 + dppf<0xB1>(s1);
        const float give = hi ? t0 : t1, keep = hi ? t1 : t0;
        float ge = keep + dppf<0x4E>(give);
        ge += dppf<0x124>(ge);                                 // row_ror:4
        ge += dppf<0x128>(ge);                                 // row_ror:8
        ge += gx;
        const float gu = ge * ev * (1.0f - ev);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float guj = j == 0 ? quad_bcast<0>(gu) : j == 1 ? quad_bcast<1>(gu) : j == 2 ? quad_bcast<2>(gu) : quad_bcast<3>(gu);
            const float evj = j == 0 ? quad_bcast<0>(ev) : j == 1 ? quad_bcast<1>(ev) : j == 2 ? quad_bcast<2>(ev) : quad_bcast<3>(ev);
#pragma unroll
            for (int i = 0; i < DL; ++i) {
                const float t = tanh_f(own_pq[i] + pr[j][i]);
                gZ[i] = fmaf(guj * w2[i], 1.0f - t * t, gZ[i]);
                G[i] = fmaf(evj, gv[j][i], G[i]);
                if constexpr (IN) sw2[i] = fmaf(guj, t, sw2[i]);
            }
            if constexpr (IN) sw2[DL] += guj;
        }
    }
}

constexpr int kRowHits = kQuadBlock / 16;             // 64 hits per workgroup
inline unsigned grid_rows(int64_t n)
{
    const unsigned g = (unsigned)((n + kRowHits - 1) / kRowHits);
    return g > 8 ? (g + 7) & ~7u : g;
}
template <int D> constexpr int kSwStride = D + 4;    // per hit: gW2 terms [D] | gb2 term | pad

template <int F, int D>
__global__ __launch_bounds__(kQuadBlock) void k_seg_bwdW(
    const float *__restrict__ A, const float *__restrict__ B, const float *__restrict__ e,
    const int32_t *__restrict__ in_ptr, const int32_t *__restrict__ in_eid, const int32_t *__restrict__ in_nbr,
    const int32_t *__restrict__ out_ptr, const int32_t *__restrict__ out_eid, const int32_t *__restrict__ out_nbr,
    const float *__restrict__ W2, float *__restrict__ G4, float *__restrict__ SW, int64_t n_hits,
    const float *__restrict__ ge_ext)
{
    constexpr int DL = D / 16;
    const int p = threadIdx.x & 15;
    const int64_t n = xcd_block() * kRowHits + (threadIdx.x >> 4);
    if (n >= n_hits) return;                            // (whole 16-lane rows leave together)
    float gP[DL], gQ[DL], Gout[DL], Gin[DL], sw2[DL + 1], w2[DL], a[2 * DL], b[2 * DL], gp[DL];
#pragma unroll
    for (int i = 0; i < DL; ++i) {
        gP[i] = gQ[i] = Gout[i] = Gin[i] = sw2[i] = 0.0f;
        w2[i] = W2[p * DL + i];
    }
    sw2[DL] = 0.0f;
    load_vec<2 * DL>(A + n * 3 * D + p * 2 * DL, a);
    load_vec<2 * DL>(B + n * 3 * D + p * 2 * DL, b);
    load_vec<DL>(A + n * 3 * D + 2 * D + p * DL, gp);
    // segments starting here (n -> d): [Q | S | gp] of the end hits; then ending here: [P | R | gp]
    row_walk<D, false>(out_ptr[n], out_ptr[n + 1], (int)n, p, out_nbr, out_eid, e, ge_ext, B, a, a + DL, gp, w2, gP, Gout, sw2);
    row_walk<D, true>(in_ptr[n], in_ptr[n + 1], (int)n, p, in_nbr, in_eid, e, ge_ext, A, b, b + DL, gp, w2, gQ, Gin, sw2);
    // G4 row = [gP(D) | gQ(D) | Gout(D) | Gin(D)], natural dim order: the 16 lanes write whole lines,
    // and k_seg_finW reads one vector of 256 hits as 256-byte pieces (stage_cols_T)
    store_vec<DL>(G4 + n * 4 * D + p * DL, gP);
    store_vec<DL>(G4 + n * 4 * D + D + p * DL, gQ);
    store_vec<DL>(G4 + n * 4 * D + 2 * D + p * DL, Gout);
    store_vec<DL>(G4 + n * 4 * D + 3 * D + p * DL, Gin);
    store_vec<DL>(SW + n * kSwStride<D> + p * DL, sw2);
    if (p == 0) SW[n * kSwStride<D> + D] = sw2[DL];
}

// Everything dense in this kernel is a product over the workgroup's 256 hits and runs on the matrix
// cores in exact fp32 from ONE transposed staging of the operands: the outer-product sums
// [D x 256] . [256 x (C + 1)] (as accum_outer_mfma) and gH += W^T v as [C x D] . [D x 256] - one lane
// per hit with scalar weight operands streamed 68 KB of weights per wave through the scalar cache
// (0.96 ms per launch at D = 64).  Wave w keeps the gh tiles of hits 64 w .. 64 w + 63 in registers
// over the four vectors.
template <int F, int D>
__global__ __launch_bounds__(kFinThreads) void k_seg_finW(
    const float *__restrict__ H, int ldh, const float *__restrict__ G4, const float *__restrict__ SW,
    const float *__restrict__ W1, const float *__restrict__ W3, float *__restrict__ gH, float *__restrict__ gW1,
    float *__restrict__ gb1, float *__restrict__ gW2, float *__restrict__ gb2, float *__restrict__ gW3,
    int rep_stride, int64_t n_hits)
{
    gW1 = my_replica(gW1, rep_stride);
    gb1 = my_replica(gb1, rep_stride);
    gW2 = my_replica(gW2, rep_stride);
    gb2 = my_replica(gb2, rep_stride);
    gW3 = my_replica(gW3, rep_stride);
    typedef float f4v __attribute__((ext_vector_type(4)));
    constexpr int C = Shape<F, D>::C, LDH = Shape<F, D>::LDH, DL = D / 16, RS = kOuterStride;
    constexpr int RT = D / 16, CT = (C + 1 + 15) / 16, KT = (LDH + 15) / 16, ROWS = D + 16 * CT;
    __shared__ __attribute__((aligned(16))) float lds[ROWS * RS];   // [v (D rows) | h (C) | ones | zero pad] x 256 hits
    constexpr int NT = kFinThreads, NW = NT / 64, HW = kBlock / NW, HT = HW / 16;    // hits (hit tiles) per wave
    const int64_t n0 = xcd_block() * kBlock, n = n0 + threadIdx.x;
    const bool is_hit = threadIdx.x < kBlock, active = is_hit && n < n_hits;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, r16 = lane & 15, g4 = lane >> 4;
    static_assert(LDH <= 16 * CT, "the padded h rows fit the column tiles");
    stage_cols_T<LDH / 4, NT>(H, ldh, 0, n0, n_hits, lds + D * RS);      // rows D .. D + LDH: h (and its padding)
    __syncthreads();                                     // (the padding columns are overwritten next, by other threads)
    if (is_hit) {
#pragma unroll
        for (int k = C; k < 16 * CT; ++k) lds[(D + k) * RS + threadIdx.x] = (k == C && active) ? 1.0f : 0.0f;   // ones row, zero pad
    }
    f4v cg[KT][HT];                                      // gh rows 16 kt + 4 g4 + r of hits HW wv + 16 ht + r16
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int ht = 0; ht < HT; ++ht) cg[kt][ht] = f4v{0.0f, 0.0f, 0.0f, 0.0f};
    // the four per-hit vectors one at a time: m = 0 gP (W1a, gb1), 1 gQ (W1b), 2 Gout (W3a), 3 Gin (W3b)
#pragma unroll 1
    for (int m = 0; m < 4; ++m) {
        __syncthreads();                                 // the previous vector's readers are done
        stage_cols_T<D / 4, NT>(G4, 4 * D, m * D, n0, n_hits, lds);
        __syncthreads();
        const float *W = m < 2 ? W1 : W3;
        float *gW = m < 2 ? gW1 : gW3;
        const int ld = m < 2 ? 2 * C : 3 * C, c0 = (m & 1) * C;
        // outer-product sums over the 256 hits: [v] . [h | 1]
        for (int t = wv; t < RT * CT; t += NW) {
            const int it = t / CT, jt = t % CT;
            const float *a = lds + (16 * it + r16) * RS + g4, *b = lds + (D + 16 * jt + r16) * RS + g4;
            f4v c = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll 8
            for (int st = 0; st < kBlock / 4; ++st) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[4 * st], b[4 * st], c, 0, 0, 0);
            const int col = 16 * jt + r16;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 16 * it + 4 * g4 + r;
                const float x = r == 0 ? c.x : r == 1 ? c.y : r == 2 ? c.z : c.w;
                if (col < C) gW[i * ld + c0 + col] += x;               // own row: the only writer
                else if (col == C && m == 0) gb1[i] += x;
            }
        }
        // gH += W^T v for this wave's 64 hits
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            const int k = 16 * kt + r16;
#pragma unroll 4
            for (int st = 0; st < D / 4; ++st) {
                const float aw = k < C ? W[(4 * st + g4) * ld + c0 + k] : 0.0f;
                const float *bv = lds + (4 * st + g4) * RS + HW * wv + r16;
#pragma unroll
                for (int ht = 0; ht < HT; ++ht)
                    cg[kt][ht] = __builtin_amdgcn_mfma_f32_16x16x4f32(aw, bv[16 * ht], cg[kt][ht], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int ht = 0; ht < HT; ++ht) {
            const int64_t nn = n0 + HW * wv + 16 * ht + r16;
            const int k0 = 16 * kt + 4 * g4;
            if (nn < n_hits && k0 < LDH) {
                f4v *dst = reinterpret_cast<f4v *>(gH + nn * ldh + k0);
                *dst = *dst + cg[kt][ht];
            }
        }
    {   // gW2[D] | gb2 from the hits' terms: wave sums, then one writer per element of this workgroup's row
        // the hits' terms, transposed into LDS (rows = dims | gb2 term), then one thread per element adds
        // its row of 256 in hit order
        __syncthreads();
        stage_cols_T<kSwStride<D> / 4, NT>(SW, kSwStride<D>, 0, n0, n_hits, lds);
        __syncthreads();
        if ((int)threadIdx.x <= D) {
            const float4 *row = reinterpret_cast<const float4 *>(lds + threadIdx.x * RS);
            float x = 0.0f;
#pragma unroll 4
            for (int t = 0; t < kBlock / 4; ++t) {
                const float4 a = row[t];
                x += a.x; x += a.y; x += a.z; x += a.w;
            }
            *((int)threadIdx.x < D ? gW2 + threadIdx.x : gb2) += x;
        }
    }
}

// gradient w.r.t. the scores a node pass consumed: ge[j] = <gmi[d], H[s]> + <gmo[s], H[d]>, zero for
// padded segments (the whole-model backward folds this into k_edge_bwd; the per-module entry point
// gnn_node_bwd hands it to the caller)
template <int F, int D>
__global__ __launch_bounds__(kBlock) void k_seg_grad(const int32_t *__restrict__ src, const int32_t *__restrict__ dst,
                                                     const float *__restrict__ H, const float *__restrict__ gmio,
                                                     float *__restrict__ ge, int64_t n_segments)
{
    constexpr int C = F + D, LDH = Shape<F, D>::LDH;
    for (int64_t j = xcd_block() * kBlock + threadIdx.x; j < n_segments; j += (int64_t)gridDim.x * kBlock) {
        const int s = src[j], d = dst[j];
        float g = 0.0f;
        if (s >= 0) {
            float hs[LDH], hd[LDH], gmi_d[LDH], gmo_s[LDH];
            load_row4<LDH / 4>(H + (int64_t)s * LDH, hs);
            load_row4<LDH / 4>(H + (int64_t)d * LDH, hd);
            load_row4<LDH / 4>(gmio + (int64_t)d * 2 * LDH, gmi_d);
            load_row4<LDH / 4>(gmio + (int64_t)s * 2 * LDH + LDH, gmo_s);
#pragma unroll
            for (int c = 0; c < C; ++c) g = fmaf(gmi_d[c], hs[c], g);
#pragma unroll
            for (int c = 0; c < C; ++c) g = fmaf(gmo_s[c], hd[c], g);
        }
        ge[j] = g;
    }
}

// layout of one gradient replica (floats): the ten tensors in state_dict order
template <int F, int D>
struct GradLayout {
    static constexpr int C = F + D;
    static constexpr int oWin = 0, obin = oWin + D * F, oW1 = obin + D, ob1 = oW1 + D * 2 * C,
                         oW2 = ob1 + D, ob2 = oW2 + D, oW3 = ob2 + 1, ob3 = oW3 + D * 3 * C,
                         oW4 = ob3 + D, ob4 = oW4 + D * D, total = ob4 + D;
    static constexpr int stride = (total + 63) & ~63;      // replicas on separate 256-byte lines
};

// rows of the partial table -> the caller's gradient tensors, in a fixed order: thread (element i,
// chunk c) of k_grad_fold1 adds rows 32 c .. 32 c + 31 sequentially into tmp[c][i]; k_grad_fold2
// adds the chunks of an element sequentially into the tensor.
template <int F, int D>
__global__ __launch_bounds__(kBlock) void k_grad_fold1(const float *__restrict__ rep, int64_t n_rows,
                                                       float *__restrict__ tmp)
{
    using G = GradLayout<F, D>;
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= G::total) return;
    const int64_t r0 = (int64_t)blockIdx.y * kFoldChunk;
    const int64_t r1 = r0 + kFoldChunk < n_rows ? r0 + kFoldChunk : n_rows;
    // all 32 loads in flight, then the adds in row order (one dependent load per add was 61 us
    // for 3 MB of partial rows)
    float v[kFoldChunk];
#pragma unroll
    for (int k = 0; k < kFoldChunk; ++k) v[k] = r0 + k < r1 ? rep[(r0 + k) * G::stride + i] : 0.0f;
    float sum = 0.0f;
#pragma unroll
    for (int k = 0; k < kFoldChunk; ++k) sum += v[k];
    tmp[(int64_t)blockIdx.y * G::stride + i] = sum;
}

template <int F, int D>
__global__ __launch_bounds__(kBlock) void k_grad_fold2(const float *__restrict__ tmp, int64_t n_chunks, gnn_grads_t gr)
{
    using G = GradLayout<F, D>;
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= G::total) return;
    float sum = 0.0f;
    for (int64_t c = 0; c < n_chunks; ++c) sum += tmp[c * G::stride + i];
    float *dst = i < G::obin ? gr.Win + (i - G::oWin)
               : i < G::oW1 ? gr.bin + (i - G::obin)
               : i < G::ob1 ? gr.W1 + (i - G::oW1)
               : i < G::oW2 ? gr.b1 + (i - G::ob1)
               : i < G::ob2 ? gr.W2 + (i - G::oW2)
               : i < G::oW3 ? gr.b2 + (i - G::ob2)
               : i < G::ob3 ? gr.W3 + (i - G::oW3)
               : i < G::oW4 ? gr.b3 + (i - G::ob3)
               : i < G::ob4 ? gr.W4 + (i - G::oW4)
                            : gr.b4 + (i - G::ob4);
    *dst += sum;
}

template <int F, int D>
int grad_fold(const float *rep, int64_t n_rows, float *tmp, const gnn_grads_t *gr, hipStream_t s)
{
    using GL = GradLayout<F, D>;
    const int64_t nc = fold_chunks(n_rows);
    if (nc > 65535) return fail(GNN_ERR_UNSUPPORTED, "too many partial-gradient rows (%lld)", (long long)n_rows);
    const unsigned gx = (GL::total + kBlock - 1) / kBlock;
    GNN_LAUNCH("k_grad_fold", (k_grad_fold1<F, D>), dim3(gx, (unsigned)(nc < 1 ? 1 : nc)), kBlock, s, rep, n_rows, tmp);
    GNN_LAUNCH("k_grad_fold", (k_grad_fold2<F, D>), gx, kBlock, s, tmp, nc, *gr);
    return 0;
}

// rows of the partial table a backward over N hits / E segments needs: one per workgroup of its
// widest launch (hit kernels: grid_for(N); k_edge_bwd: at most kSegGrid)
constexpr int kSegGrid = 1024;
inline int64_t bwd_rows(int64_t N, int64_t E)
{
    int64_t r = grid_for(N);
    const int64_t ge = grid_for(E) < (unsigned)kSegGrid ? grid_for(E) : kSegGrid;
    r = r > ge ? r : ge;
    return r < 8 ? 8 : r;
}

struct BwdWs {
    float *PQ, *gu, *gHa, *gHb, *gmio, *rep, *tmp, *A, *B, *G4, *SW;
    char *rep_end;
    int64_t rows;
    size_t bytes;
};

BwdWs carve_bwd(char *b, int64_t N, int64_t E, int ldh, int C, int D)
{
    BwdWs w;
    size_t off = 0;
    auto take = [&](size_t nfloat) {
        float *p = reinterpret_cast<float *>(b + off);
        off += align256(nfloat * sizeof(float));
        return p;
    };
    w.PQ = take((size_t)N * 2 * D);
    w.gu = take((size_t)E);
    w.gHb = take((size_t)N * ldh);
    w.gmio = take((size_t)N * 2 * ldh);      // [gmi | gmo], rows padded to LDH
    // gHa and the gradient replicas (GradLayout<F, D>::stride each; F = C - D) are adjacent: one
    // memset clears both
    w.gHa = take((size_t)N * ldh);
    const int tot = D * (C - D) + D + D * 2 * C + D + D + 1 + D * 3 * C + D + D * D + D;
    const size_t stride = (size_t)((tot + 63) & ~63);
    w.rows = bwd_rows(N, E);
    w.rep = take((size_t)w.rows * stride);
    w.rep_end = b + off;
    w.tmp = take((size_t)fold_chunks(w.rows) * stride);
    w.A = take((size_t)N * 3 * D);           // [P | R | gp], [Q | S | gp] of the pull-form kernels
    w.B = take((size_t)N * 3 * D);
    w.G4 = take((size_t)N * 4 * D);           // [gP gQ Gout Gin] between k_seg_bwd4 and k_seg_fin
    w.SW = take(D >= 32 ? (size_t)N * (D + 4) : 0);   // wide shapes: a hit's gW2 / gb2 terms (k_seg_bwdW -> k_seg_finW)
    w.bytes = off;
    return w;
}

template <int F, int D>
int backward_t(const gnn_graph_t *g, const gnn_params_t *p, int T, const float *e_all,
               const float *H_all, const float *Q_all, const float *grad_out, const gnn_grads_t *gr,
               char *ws, hipStream_t s)
{
    constexpr int C = Shape<F, D>::C, LDH = Shape<F, D>::LDH;
    const int64_t N = g->n_hits, E = g->n_segments;
    BwdWs w = carve_bwd(ws, N, E, LDH, C, D);
    using GL = GradLayout<F, D>;
    float *gH = w.gHa, *gHprev = w.gHb;
    {   // hit gradient of the last iteration and the gradient replicas start at zero
        hipError_t err = hipMemsetAsync(gH, 0, (size_t)(w.rep_end - reinterpret_cast<char *>(gH)), s);
        if (err != hipSuccess) return fail(-(int)err, "memset of the gradient workspace failed");
    }
    float *const rp = w.rep;
    constexpr int RS = GL::stride;
    if (w.rep_end - reinterpret_cast<char *>(rp) < (ptrdiff_t)((size_t)w.rows * RS * sizeof(float)))
        return fail(GNN_ERR_WORKSPACE, "partial-gradient table does not match GradLayout");
    const float *ge = grad_out;
    for (int t = T; t >= 0; --t) {
        const float *Ht = H_all + (size_t)t * N * LDH;
        const float *et = e_all + (size_t)t * E;
        // edge pass t backward: adds into gH (gradient w.r.t. H_t)
        bool edge_done = false;
        if constexpr (D >= 32) {
            // wide shapes, final pass (the only edge pass this loop still runs for them): the pull-form
            // kernels with the loss gradient as ge - k_hit_bwdW on a zero gradient row builds the records
            // ([P R 0], [Q S 0]), the walks rebuild gu at both ends, k_seg_finW adds W1^T (gP, gQ); the
            // padded segments (in no hit's list) keep their own pass
            if (Q_all && t == T && T > 0 && ge && !getenv("GNN_BWD_WIDE_PER_PASS")) {
                if (N > 0) {
                    GNN_LAUNCH("k_hit_bwdW", (k_hit_bwdW<F, D>), grid_for(N), kFinThreads, s, Ht, Ht, Q_all, LDH, p->W1, p->b1,
                               p->W3, p->W4, gH, gHprev, w.A, w.B, rp + GL::oW3, rp + GL::ob3, rp + GL::oW4, rp + GL::ob4, RS, N);
                    GNN_LAUNCH("k_seg_bwdW", (k_seg_bwdW<F, D>), grid_rows(N), kQuadBlock, s, w.A, w.B, et, g->in_ptr,
                               g->in_eid, g->in_nbr, g->out_ptr, g->out_eid, g->out_nbr, p->W2, w.G4, w.SW, N, ge);
                    GNN_LAUNCH("k_seg_finW", (k_seg_finW<F, D>), grid_for(N), kFinThreads, s, Ht, LDH, w.G4, w.SW, p->W1, p->W3,
                               gH, rp + GL::oW1, rp + GL::ob1, rp + GL::oW2, rp + GL::ob2, rp + GL::oW3, RS, N);
                }
                if (E > 0) {
                    const unsigned ge_grid = grid_for(E) < (unsigned)kSegGrid ? grid_for(E) : (unsigned)kSegGrid;
                    GNN_LAUNCH("k_edge_bwd", (k_edge_bwd<F, D>), ge_grid, kBlock, s, g->src, g->dst, w.PQ, p->b1, p->W2, et,
                               ge, Ht, w.gmio, w.gu, rp + GL::oW2, rp + GL::ob2, rp + GL::ob1, RS, E, 1);
                }
                edge_done = true;
            }
        }
        if (!edge_done && N > 0) GNN_LAUNCH("kb_pq", (kb_pq<F, D>), grid_for(N), kBlock, s, Ht, LDH, p->W1, p->b1, w.PQ, N);
        if (!edge_done && E > 0) {
            const unsigned ge_grid = grid_for(E) < (unsigned)kSegGrid ? grid_for(E) : (unsigned)kSegGrid;
            GNN_LAUNCH("k_edge_bwd", (k_edge_bwd<F, D>), ge_grid, kBlock, s, g->src, g->dst, w.PQ,
                       p->b1, p->W2, et, ge, Ht, w.gmio, w.gu, rp + GL::oW2, rp + GL::ob2, rp + GL::ob1,
                       RS, E);
        }
        if (!edge_done && N > 0)
            GNN_LAUNCH("k_pq_bwd", (k_pq_bwd<F, D>), grid_for(N), kBlock, s, Ht, LDH, w.PQ, w.gu,
                       g->in_ptr, g->in_eid, g->in_nbr, g->out_ptr, g->out_eid, g->out_nbr, p->W1, p->W2,
                       gH, rp + GL::oW1, rp + GL::ob1, RS, N);
        if (t == 0) break;
        if constexpr (D <= 16) {
            // iterations t-1 .. 0 in pull form: node pass (H_{t-1}, e_{t-1} -> H_t) together with the
            // edge pass that produced e_{t-1}
            for (int u = t; u >= 1; --u) {
                const float *Hu = H_all + (size_t)u * N * LDH;
                const float *Hp = H_all + (size_t)(u - 1) * N * LDH;
                const float *ep = e_all + (size_t)(u - 1) * E;
                if (N > 0) {
                    if (Q_all) {
                        // k_hit_bwd4 (u) ran as the tail of the previous iteration's k_fin_hit, except for the first
                        static const bool unfused = getenv("GNN_BWD_NO_FIN_HIT") != nullptr;      // A / B runs
                        if (u == t || unfused)
                            GNN_LAUNCH("k_hit_bwd4", (k_hit_bwd4<F, D>), grid_for(N), kBlock, s, Hp, Hu,
                                       Q_all + (size_t)(u - 1) * N * D, LDH, p->W1, p->b1, p->W3, p->W4, gH, gHprev, w.A,
                                       w.B, rp + GL::oW3, rp + GL::ob3, rp + GL::oW4, rp + GL::ob4, RS, N);
                        GNN_LAUNCH("k_seg_bwd4", (k_seg_bwd4<F, D>), grid_for(N), kQuadBlock, s, w.A, w.B, ep, g->in_ptr,
                                   g->in_eid, g->in_nbr, g->out_ptr, g->out_eid, g->out_nbr, p->W2, w.G4, rp + GL::oW2,
                                   rp + GL::ob2, RS, N);
                        if (u > 1 && !unfused) {
                            // the gradient row of H_{u-1} is finished in registers and feeds iteration u - 1's per-hit
                            // pass at once: gHprev (its initial row) is read, gH (free by now) takes iteration u - 1's
                            // initial row - after the swap below the roles are as a separate k_hit_bwd4 leaves them
                            GNN_LAUNCH("k_fin_hit", (k_fin_hit<F, D>), grid_for(N), kBlock, s, Hp,
                                       H_all + (size_t)(u - 2) * N * LDH, Q_all + (size_t)(u - 2) * N * D, LDH, w.G4, p->W1,
                                       p->b1, p->W3, p->W4, gHprev, gH, w.A, w.B, rp + GL::oW1, rp + GL::ob1, rp + GL::oW3,
                                       rp + GL::ob3, rp + GL::oW4, rp + GL::ob4, RS, N);
                        } else {
                            GNN_LAUNCH("k_seg_fin", (k_seg_fin<F, D>), grid_for(N), kBlock, s, Hp, LDH, w.G4, p->W1, p->W3,
                                       gHprev, rp + GL::oW1, rp + GL::ob1, rp + GL::oW3, RS, N);
                        }
                        float *tmp = gH; gH = gHprev; gHprev = tmp;
                        continue;
                    }
                    GNN_LAUNCH("kb_prs", (kb_prs<F, D>), grid_for(N), kBlock, s, Hp, LDH, p->W1, p->b1, p->W3, w.A, w.B, N);
                    GNN_LAUNCH("k_hit_bwd", (k_hit_bwd<F, D, false>), grid_for(N), kBlock, s, Hp, Hu, nullptr, LDH, ep,
                               g->in_ptr, g->in_eid, g->in_nbr, g->out_ptr, g->out_eid, g->out_nbr, p->W3, p->b3,
                               p->W4, gH, gHprev, w.A, w.B, rp + GL::oW3, rp + GL::ob3, rp + GL::oW4, rp + GL::ob4, RS,
                               N);
                    GNN_LAUNCH("k_seg_bwd", (k_seg_bwd<F, D>), grid_for(N), kBlock, s, Hp, LDH, w.A, w.B, ep, g->in_ptr,
                               g->in_eid, g->in_nbr, g->out_ptr, g->out_eid, g->out_nbr, p->W1, p->W2, p->W3, gHprev,
                               rp + GL::oW1, rp + GL::ob1, rp + GL::oW2, rp + GL::ob2, rp + GL::oW3, RS, N);
                }
                float *tmp = gH; gH = gHprev; gHprev = tmp;
            }
            break;
        }
        if constexpr (D >= 32) {
            // wide hidden layers with the forward's kept hidden layers: pull form, 16 lanes per hit
            if (Q_all && !getenv("GNN_BWD_WIDE_PER_PASS")) {
                for (int u = t; u >= 1; --u) {
                    const float *Hu = H_all + (size_t)u * N * LDH;
                    const float *Hp = H_all + (size_t)(u - 1) * N * LDH;
                    const float *ep = e_all + (size_t)(u - 1) * E;
                    if (N > 0) {
                        GNN_LAUNCH("k_hit_bwdW", (k_hit_bwdW<F, D>), grid_for(N), kFinThreads, s, Hp, Hu,
                                   Q_all + (size_t)(u - 1) * N * D, LDH, p->W1, p->b1, p->W3, p->W4, gH, gHprev, w.A, w.B,
                                   rp + GL::oW3, rp + GL::ob3, rp + GL::oW4, rp + GL::ob4, RS, N);
                        GNN_LAUNCH("k_seg_bwdW", (k_seg_bwdW<F, D>), grid_rows(N), kQuadBlock, s, w.A, w.B, ep, g->in_ptr,
                                   g->in_eid, g->in_nbr, g->out_ptr, g->out_eid, g->out_nbr, p->W2, w.G4, w.SW, N,
                                   (const float *)nullptr);
                        GNN_LAUNCH("k_seg_finW", (k_seg_finW<F, D>), grid_for(N), kFinThreads, s, Hp, LDH, w.G4, w.SW, p->W1, p->W3,
                                   gHprev, rp + GL::oW1, rp + GL::ob1, rp + GL::oW2, rp + GL::ob2, rp + GL::oW3, RS, N);
                    }
                    float *tmp = gH; gH = gHprev; gHprev = tmp;
                }
                break;
            }
        }
        // (wide hidden layers without Q_all: the per-pass kernels) node pass t-1 backward: H_{t-1}, e_{t-1} -> H_t
        const float *Hp = H_all + (size_t)(t - 1) * N * LDH;
        const float *ep = e_all + (size_t)(t - 1) * E;
        if (N > 0) {
            GNN_LAUNCH("k_node_bwd", (k_node_bwd<F, D>), grid_for(N), kBlock, s, Hp, Ht, LDH, ep,
                       g->in_ptr, g->in_eid, g->in_nbr, g->out_ptr, g->out_eid, g->out_nbr, p->W3,
                       p->b3, p->W4, gH, gHprev, w.gmio, rp + GL::oW3, rp + GL::ob3, rp + GL::oW4,
                       rp + GL::ob4, RS, N);
            GNN_LAUNCH("k_agg_bwd_n", (k_agg_bwd_n<F, D>), grid_for(N), kBlock, s, ep, w.gmio,
                       g->in_ptr, g->in_eid, g->in_nbr, g->out_ptr, g->out_eid, g->out_nbr, gHprev,
                       LDH, N);
        }
        float *tmp = gH; gH = gHprev; gHprev = tmp;
        ge = nullptr;      // the next (earlier) edge pass rebuilds its ge from gmio and H_{t-1}
    }
    if (N > 0)
        GNN_LAUNCH("k_input_bwd", (k_input_bwd<F, D>), grid_for(N), kBlock, s, g->X, H_all, LDH, gH,
                   rp + GL::oWin, rp + GL::obin, RS, N);
    return grad_fold<F, D>(rp, w.rows, w.tmp, gr, s);
}

// ---- per-module backward (the reference's sub-modules are ordinary autograd modules:
// model.edge_network(H, Ri, Ro) / model.node_network(H, e, Ri, Ro), gnn/model.py:69-81,113-125, called
// directly in gnn/MPNN_Seg_ACTS_maskedlinear.ipynb cells 42, 46) --------------------------------
// EdgeNetwork: e = sigmoid(W2 tanh(W1 [H_s | H_d] + b1) + b2).  Given ge = dL/de: adds dL/dH into gH
// (rows of LDH floats, caller-zeroed) and the gradients of W1, b1, W2, b2 into gr.
template <int F, int D>
int edge_bwd_t(const float *H, const gnn_graph_t *g, const gnn_params_t *p, const float *e, const float *ge,
               float *gH, const gnn_grads_t *gr, char *ws, hipStream_t s)
{
    constexpr int C = Shape<F, D>::C, LDH = Shape<F, D>::LDH;
    const int64_t N = g->n_hits, E = g->n_segments;
    BwdWs w = carve_bwd(ws, N, E, LDH, C, D);
    using GL = GradLayout<F, D>;
    float *const rp = w.rep;
    constexpr int RS = GL::stride;
    hipError_t err = hipMemsetAsync(rp, 0, (size_t)w.rows * RS * sizeof(float), s);
    if (err != hipSuccess) return fail(-(int)err, "memset of the partial-gradient table failed");
    if (N > 0) GNN_LAUNCH("kb_pq", (kb_pq<F, D>), grid_for(N), kBlock, s, H, LDH, p->W1, p->b1, w.PQ, N);
    if (E > 0) {
        const unsigned ge_grid = grid_for(E) < (unsigned)kSegGrid ? grid_for(E) : (unsigned)kSegGrid;
        GNN_LAUNCH("k_edge_bwd", (k_edge_bwd<F, D>), ge_grid, kBlock, s, g->src, g->dst, w.PQ, p->b1, p->W2, e, ge, H,
                   w.gmio, w.gu, rp + GL::oW2, rp + GL::ob2, rp + GL::ob1, RS, E);
    }
    if (N > 0)
        GNN_LAUNCH("k_pq_bwd", (k_pq_bwd<F, D>), grid_for(N), kBlock, s, H, LDH, w.PQ, w.gu, g->in_ptr, g->in_eid,
                   g->in_nbr, g->out_ptr, g->out_eid, g->out_nbr, p->W1, p->W2, gH, rp + GL::oW1, rp + GL::ob1, RS, N);
    return grad_fold<F, D>(rp, w.rows, w.tmp, gr, s);
}

// NodeNetwork: H' = tanh(W4 tanh(W3 [mi | mo | H] + b3) + b4).  Given gHn = dL/dH' (rows of LDH floats,
// first D used) and the forward's H, e, H': writes dL/dH into gH (rows of LDH), dL/de into ge, adds the
// gradients of W3, b3, W4, b4 into gr.
template <int F, int D>
int node_bwd_t(const float *H, const float *e, const float *Hn, const gnn_graph_t *g, const gnn_params_t *p,
               const float *gHn, float *gH, float *ge, const gnn_grads_t *gr, char *ws, hipStream_t s)
{
    constexpr int C = Shape<F, D>::C, LDH = Shape<F, D>::LDH;
    const int64_t N = g->n_hits, E = g->n_segments;
    BwdWs w = carve_bwd(ws, N, E, LDH, C, D);
    using GL = GradLayout<F, D>;
    float *const rp = w.rep;
    constexpr int RS = GL::stride;
    hipError_t err = hipMemsetAsync(rp, 0, (size_t)w.rows * RS * sizeof(float), s);
    if (err != hipSuccess) return fail(-(int)err, "memset of the partial-gradient table failed");
    if (N > 0) {
        GNN_LAUNCH("k_node_bwd", (k_node_bwd<F, D>), grid_for(N), kBlock, s, H, Hn, LDH, e, g->in_ptr, g->in_eid,
                   g->in_nbr, g->out_ptr, g->out_eid, g->out_nbr, p->W3, p->b3, p->W4, gHn, gH, w.gmio,
                   rp + GL::oW3, rp + GL::ob3, rp + GL::oW4, rp + GL::ob4, RS, N);
        GNN_LAUNCH("k_agg_bwd_n", (k_agg_bwd_n<F, D>), grid_for(N), kBlock, s, e, w.gmio, g->in_ptr, g->in_eid,
                   g->in_nbr, g->out_ptr, g->out_eid, g->out_nbr, gH, LDH, N);
    }
    if (E > 0) GNN_LAUNCH("k_seg_grad", (k_seg_grad<F, D>), grid_for(E), kBlock, s, g->src, g->dst, H, w.gmio, ge, E);
    return grad_fold<F, D>(rp, w.rows, w.tmp, gr, s);
}

// ---------------------------------------------------------------------------------------------
// small events: the WHOLE backward of one graph in one workgroup, one launch for the batch
// ---------------------------------------------------------------------------------------------
// Counterpart of k_event (gnn_kernels.hip).  The per-pass kernels above cost 5T + 4 launches of a
// few microseconds each, which for the reference's muon graphs (tens of hits) is all of the
// backward.  Here a workgroup owns one graph: its saved H_t / e_t rows, the P/Q rows, the hit
// gradients and every intermediate of backward_t live in LDS, the stages are the same statements
// in the same order per item, a barrier separates them.  Weight gradients: every thread owns a few
// elements of each gradient tensor, sums its graph's contributions in registers over all
// iterations and issues one atomic per element at the end (to the workgroup's gradient replica).
template <int F, int D>
struct EvBwd {
    static constexpr int C = F + D, LDH = (C + 3) & ~3;
    // weights in LDS: W1, b1, W2, W3, b3, W4; the rows of W1 / W3 are laid out like the feature rows
    // they multiply (blocks of C columns padded to LDH, zeros) so that products run on 4-float pieces
    static constexpr int oW1 = 0, ob1 = oW1 + D * 2 * LDH, oW2 = ob1 + D, oW3 = oW2 + D,
                         ob3 = oW3 + D * 3 * LDH, oW4 = ob3 + D, w_total = oW4 + D * D;
    static constexpr int per_hit = 9 * LDH + 5 * D;     // Hc Hp gH gHp (4) + gmio (2) + M (3) | PQ fa (2D each) qb
    static size_t lds_bytes(int64_t cap_h, int64_t cap_s)
    {
        const int64_t s4 = (cap_s + 3) & ~3;
        return (size_t)(cap_h * per_hit + 2 * s4 + w_total + 2 * (cap_h + 1) + 6 * cap_s + 8) * sizeof(float);
    }
    static constexpr int n1 = D * 2 * C + D, n3 = D * 3 * C + D, n4 = D * D + D, nin = D * F + D;
};

// Weight gradients inside k_event_bwd: sum over the graph's hits of L (x) R, R taken in 4-float
// pieces.  Role r of a thread (r = tid + u * 256): r < NL * NR4 -> the 4 products of L row r / NR4
// with piece r % NR4 of R; the next NB roles -> the plain sums of L rows 0 .. NB-1 (the bias).
template <int NL, int NR4, int NB>
struct Outer4 {
    static constexpr int roles = NL * NR4 + NB, NA = (roles + kBlock - 1) / kBlock;
    float4 acc[NA];
    __device__ __forceinline__ void clear()
    {
#pragma unroll
        for (int u = 0; u < NA; ++u) acc[u] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
    __device__ __forceinline__ void add(int nh, const float *L, int ls, const float *R, int rs)
    {
#pragma unroll
        for (int u = 0; u < NA; ++u) {
            const int r = threadIdx.x + u * kBlock;
            if (r < NL * NR4) {
                const float *l = L + r / NR4, *rr = R + 4 * (r % NR4);
                float4 a = acc[u];
                for (int n = 0; n < nh; ++n) {
                    const float lv = l[n * ls];
                    const float4 rv = *reinterpret_cast<const float4 *>(rr + n * rs);
                    a.x = fmaf(lv, rv.x, a.x); a.y = fmaf(lv, rv.y, a.y);
                    a.z = fmaf(lv, rv.z, a.z); a.w = fmaf(lv, rv.w, a.w);
                }
                acc[u] = a;
            } else if (r < roles) {
                float a = acc[u].x;
                for (int n = 0; n < nh; ++n) a += L[n * ls + (r - NL * NR4)];
                acc[u].x = a;
            }
        }
    }
    // index(l, kk) -> element of the gradient replica for L row l and R column kk, or -1 (padding)
    template <typename Index>
    __device__ __forceinline__ void flush(float *rep, Index index, int bias0) const
    {
#pragma unroll
        for (int u = 0; u < NA; ++u) {
            const int r = threadIdx.x + u * kBlock;
            if (r < NL * NR4) {
                const float v[4] = {acc[u].x, acc[u].y, acc[u].z, acc[u].w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int o = index(r / NR4, 4 * (r % NR4) + j);
                    if (o >= 0) atomicAdd(rep + o, v[j]);
                }
            } else if (r < roles) {
                atomicAdd(rep + bias0 + (r - NL * NR4), acc[u].x);
            }
        }
    }
};

template <int F, int D>
__global__ __launch_bounds__(kBlock) void k_event_bwd(
    gnn_graph_t g, gnn_params_t p, const int32_t *__restrict__ hit_ptr,
    const int32_t *__restrict__ seg_ptr, int T, const float *__restrict__ e_all,
    const float *__restrict__ H_all, const float *__restrict__ grad_out, float *__restrict__ rep,
    int cap_h, int cap_s)
{
    using B = EvBwd<F, D>;
    using GL = GradLayout<F, D>;
    constexpr int C = B::C, LDH = B::LDH, NT = kBlock;
    extern __shared__ __attribute__((aligned(16))) float eb_lds[];
    const int s4 = (cap_s + 3) & ~3;
    float *Hc = eb_lds, *Hp = Hc + cap_h * LDH, *gH = Hp + cap_h * LDH, *gHp = gH + cap_h * LDH,
          *gmio = gHp + cap_h * LDH, *Mr = gmio + cap_h * 2 * LDH, *PQ = Mr + cap_h * 3 * LDH,
          *fa = PQ + cap_h * 2 * D, *qb = fa + cap_h * 2 * D, *ec = qb + cap_h * D, *gu = ec + s4,
          *wl = gu + s4;
    int32_t *ip = reinterpret_cast<int32_t *>(wl + B::w_total), *op = ip + cap_h + 1,
            *ie = op + cap_h + 1, *inb = ie + cap_s, *oe = inb + cap_s, *onb = oe + cap_s,
            *sl = onb + cap_s, *dl = sl + cap_s;
    const int tid = threadIdx.x;
    const int h0 = hit_ptr[blockIdx.x], nh = hit_ptr[blockIdx.x + 1] - h0;
    const int s0 = seg_ptr[blockIdx.x], ns = seg_ptr[blockIdx.x + 1] - s0;
    const int64_t N = g.n_hits, E = g.n_segments;
    rep = my_replica(rep, GL::stride);

    for (int i = tid; i < D * 2 * LDH; i += NT) {
        const int d = i / (2 * LDH), blk = (i / LDH) % 2, k = i % LDH;
        wl[B::oW1 + i] = k < C ? p.W1[d * 2 * C + blk * C + k] : 0.0f;
    }
    for (int i = tid; i < D * 3 * LDH; i += NT) {
        const int d = i / (3 * LDH), blk = (i / LDH) % 3, k = i % LDH;
        wl[B::oW3 + i] = k < C ? p.W3[d * 3 * C + blk * C + k] : 0.0f;
    }
    for (int i = tid; i < D * D; i += NT) wl[B::oW4 + i] = p.W4[i];
    for (int i = tid; i < D; i += NT) {
        wl[B::ob1 + i] = p.b1[i];
        wl[B::oW2 + i] = p.W2[i];
        wl[B::ob3 + i] = p.b3[i];
    }
    const float *W1 = wl + B::oW1, *b1 = wl + B::ob1, *W2 = wl + B::oW2, *W3 = wl + B::oW3,
                *b3 = wl + B::ob3, *W4 = wl + B::oW4;
    if (nh > 0) {       // the graph's index arrays as LOCAL ids (see k_event)
        const int ib = g.in_ptr[h0], ob = g.out_ptr[h0];
        for (int n = tid; n <= nh; n += NT) {
            ip[n] = g.in_ptr[h0 + n] - ib;
            op[n] = g.out_ptr[h0 + n] - ob;
        }
        const int ni = g.in_ptr[h0 + nh] - ib, no = g.out_ptr[h0 + nh] - ob;
        for (int k = tid; k < ni; k += NT) {
            ie[k] = g.in_eid[ib + k] - s0;
            inb[k] = g.in_nbr[ib + k] - h0;
        }
        for (int k = tid; k < no; k += NT) {
            oe[k] = g.out_eid[ob + k] - s0;
            onb[k] = g.out_nbr[ob + k] - h0;
        }
    }
    for (int j = tid; j < ns; j += NT) {
        const int sg = g.src[s0 + j];
        sl[j] = sg < 0 ? -1 : sg - h0;
        dl[j] = sg < 0 ? -1 : g.dst[s0 + j] - h0;
    }
    for (int i = tid; i < nh * LDH; i += NT) {
        gH[i] = gHp[i] = 0.0f;                  // (padding columns stay zero: nothing writes them)
        gmio[2 * i] = gmio[2 * i + 1] = 0.0f;
        Hc[i] = H_all[((int64_t)T * N + h0) * LDH + i];
    }
    for (int j = tid; j < ns; j += NT) ec[j] = e_all[(int64_t)T * E + s0 + j];

    // this thread's elements of the gradient tensors (summed over the graph's hits and over t)
    Outer4<2 * D, LDH / 4, D> g1;               // [gP | gQ] (x) H_t      -> gW1, gb1
    Outer4<D, 3 * LDH / 4, D> g3;               // gp (x) [mi | mo | h]   -> gW3, gb3
    Outer4<D, D / 4, D> g4;                     // gr (x) q               -> gW4, gb4
    Outer4<D, (LDH - D) / 4, D> gin;            // g (x) x                -> gWin, gbin
    g1.clear();
    g3.clear();
    g4.clear();
    gin.clear();
    float sW2[D], sb2 = 0.0f, spad = 0.0f;      // per-segment sums (k_edge_bwd)
#pragma unroll
    for (int i = 0; i < D; ++i) sW2[i] = 0.0f;
    __syncthreads();

    for (int t = T;; --t) {
        // P/Q rows of H_t (kb_pq): one (hit, row) per thread
        for (int i = tid; i < nh * 2 * D; i += NT) {
            const int n = i / (2 * D), r = i % (2 * D), d = r % D;
            const bool isq = r >= D;
            float acc = isq ? 0.0f : b1[d];
            const float4 *w = reinterpret_cast<const float4 *>(W1 + d * 2 * LDH + (isq ? LDH : 0));
            const float4 *h = reinterpret_cast<const float4 *>(Hc + n * LDH);
#pragma unroll
            for (int c = 0; c < LDH / 4; ++c) {      // (padding columns add 0 * 0: the sums are unchanged)
                const float4 a = w[c], b = h[c];
                acc = fmaf(a.x, b.x, acc); acc = fmaf(a.y, b.y, acc);
                acc = fmaf(a.z, b.z, acc); acc = fmaf(a.w, b.w, acc);
            }
            PQ[n * 2 * D + r] = acc;
        }
        __syncthreads();
        // edge pass t (k_edge_bwd): gu per segment, sums for gW2 / gb2 / padded share of gb1
        for (int j = tid; j < ns; j += NT) {
            const int s = sl[j], d = dl[j];
            const float ev = ec[j];
            float gej = 0.0f;
            if (t == T) {
                gej = grad_out[s0 + j];
            } else if (s >= 0) {
#pragma unroll
                for (int c = 0; c < C; ++c) gej = fmaf(gmio[d * 2 * LDH + c], Hc[s * LDH + c], gej);
#pragma unroll
                for (int c = 0; c < C; ++c) gej = fmaf(gmio[s * 2 * LDH + LDH + c], Hc[d * LDH + c], gej);
            }
            const float guv = gej * ev * (1.0f - ev);
            if (s >= 0) {
#pragma unroll
                for (int i = 0; i < D; ++i)
                    sW2[i] = fmaf(guv, tanh_f(PQ[s * 2 * D + i] + PQ[d * 2 * D + D + i]), sW2[i]);
            } else {
#pragma unroll
                for (int i = 0; i < D; ++i) sW2[i] = fmaf(guv, tanh_f(b1[i]), sW2[i]);
                spad += guv;
            }
            sb2 += guv;
            gu[j] = guv;
        }
        __syncthreads();
        // k_pq_bwd: gP = sum_out gz, gQ = sum_in gz, one (hit, P|Q, 4 hidden units) per thread
        // -> fa = [gP | gQ]
        for (int i = tid; i < nh * 2 * (D / 4); i += NT) {
            const int n = i / (2 * (D / 4)), part = (i / (D / 4)) & 1, v = i % (D / 4);
            const float4 own = *reinterpret_cast<const float4 *>(PQ + n * 2 * D + part * D + 4 * v);
            const float4 w2 = *reinterpret_cast<const float4 *>(W2 + 4 * v);
            const int32_t *el = part ? ie : oe, *nl = part ? inb : onb;
            float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            for (int k = part ? ip[n] : op[n], k1 = part ? ip[n + 1] : op[n + 1]; k < k1; ++k) {
                const float gk = gu[el[k]];
                // out-walk: P[n] + Q[end hit];  in-walk: P[start hit] + Q[n]
                const float4 o = *reinterpret_cast<const float4 *>(PQ + nl[k] * 2 * D + (part ? 0 : D) + 4 * v);
                const float ax = tanh_f(own.x + o.x), ay = tanh_f(own.y + o.y), az = tanh_f(own.z + o.z),
                            aw = tanh_f(own.w + o.w);
                acc.x += gk * w2.x * (1.0f - ax * ax);
                acc.y += gk * w2.y * (1.0f - ay * ay);
                acc.z += gk * w2.z * (1.0f - az * az);
                acc.w += gk * w2.w * (1.0f - aw * aw);
            }
            *reinterpret_cast<float4 *>(fa + n * 2 * D + part * D + 4 * v) = acc;
        }
        __syncthreads();
        for (int i = tid; i < nh * C; i += NT) {            // gH += W1a^T gP + W1b^T gQ
            const int n = i / C, k = i % C;
            float acc = 0.0f;
#pragma unroll
            for (int d = 0; d < D; ++d) {
                acc = fmaf(W1[d * 2 * LDH + k], fa[n * 2 * D + d], acc);
                acc = fmaf(W1[d * 2 * LDH + LDH + k], fa[n * 2 * D + D + d], acc);
            }
            gH[n * LDH + k] += acc;
        }
        g1.add(nh, fa, 2 * D, Hc, LDH);                        // gW1, gb1
        __syncthreads();
        if (t == 0) break;
        // node pass t-1 -> t (k_node_bwd): H_{t-1}, e_{t-1} in; Hc = H_t holds the pass's outputs
        for (int i = tid; i < nh * LDH; i += NT) Hp[i] = H_all[((int64_t)(t - 1) * N + h0) * LDH + i];
        for (int j = tid; j < ns; j += NT) ec[j] = e_all[(int64_t)(t - 1) * E + s0 + j];
        __syncthreads();
        // M = [mi | mo | h], rows padded to LDH: one (hit, part, 4-float piece) per thread
        for (int i = tid; i < nh * 3 * (LDH / 4); i += NT) {
            const int n = i / (3 * (LDH / 4)), part = (i / (LDH / 4)) % 3, c = i % (LDH / 4);
            float4 m = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if (part == 2) {
                m = *reinterpret_cast<const float4 *>(Hp + n * LDH + 4 * c);
            } else {
                const int32_t *el = part ? oe : ie, *nl = part ? onb : inb;
                for (int k = part ? op[n] : ip[n], k1 = part ? op[n + 1] : ip[n + 1]; k < k1; ++k) {
                    const float w = ec[el[k]];
                    const float4 a = *reinterpret_cast<const float4 *>(Hp + nl[k] * LDH + 4 * c);
                    m.x = fmaf(w, a.x, m.x);
                    m.y = fmaf(w, a.y, m.y);
                    m.z = fmaf(w, a.z, m.z);
                    m.w = fmaf(w, a.w, m.w);
                }
            }
            *reinterpret_cast<float4 *>(Mr + n * 3 * LDH + part * LDH + 4 * c) = m;
        }
        __syncthreads();
        for (int i = tid; i < nh * D; i += NT) {             // q and gr
            const int n = i / D, d = i % D;
            float acc = b3[d];
            const float4 *w = reinterpret_cast<const float4 *>(W3 + d * 3 * LDH);
            const float4 *mrow = reinterpret_cast<const float4 *>(Mr + n * 3 * LDH);
#pragma unroll
            for (int c = 0; c < 3 * LDH / 4; ++c) {
                const float4 a = w[c], b = mrow[c];
                acc = fmaf(a.x, b.x, acc); acc = fmaf(a.y, b.y, acc);
                acc = fmaf(a.z, b.z, acc); acc = fmaf(a.w, b.w, acc);
            }
            qb[i] = tanh_f(acc);
            const float hn = Hc[n * LDH + d];
            fa[n * 2 * D + d] = gH[n * LDH + d] * (1.0f - hn * hn);
        }
        __syncthreads();
        for (int i = tid; i < nh * D; i += NT) {             // gp
            const int n = i / D, k = i % D;
            float acc = 0.0f;
#pragma unroll
            for (int d = 0; d < D; ++d) acc = fmaf(W4[d * D + k], fa[n * 2 * D + d], acc);
            fa[n * 2 * D + D + k] = acc * (1.0f - qb[i] * qb[i]);
        }
        __syncthreads();
        for (int i = tid; i < nh * 3 * C; i += NT) {         // gM -> gmi | gmo | gH_prev (self part)
            const int n = i / (3 * C), k = i % (3 * C);
            float acc = 0.0f;
#pragma unroll
            for (int d = 0; d < D; ++d) acc = fmaf(W3[d * 3 * LDH + (k / C) * LDH + k % C], fa[n * 2 * D + D + d], acc);
            if (k < C) gmio[n * 2 * LDH + k] = acc;
            else if (k < 2 * C) gmio[n * 2 * LDH + LDH + (k - C)] = acc;
            else gHp[n * LDH + (k - 2 * C)] = acc;
        }
        g3.add(nh, fa + D, 2 * D, Mr, 3 * LDH);                // gW3, gb3
        g4.add(nh, fa, 2 * D, qb, D);                          // gW4, gb4
        __syncthreads();
        for (int i = tid; i < nh * (LDH / 4); i += NT) {     // k_agg_bwd_n, 4 columns per thread
            const int n = i / (LDH / 4), c = i % (LDH / 4);
            float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            for (int k = op[n]; k < op[n + 1]; ++k) {
                const float w = ec[oe[k]];
                const float4 a = *reinterpret_cast<const float4 *>(gmio + onb[k] * 2 * LDH + 4 * c);
                acc.x = fmaf(w, a.x, acc.x); acc.y = fmaf(w, a.y, acc.y);
                acc.z = fmaf(w, a.z, acc.z); acc.w = fmaf(w, a.w, acc.w);
            }
            for (int k = ip[n]; k < ip[n + 1]; ++k) {
                const float w = ec[ie[k]];
                const float4 a = *reinterpret_cast<const float4 *>(gmio + inb[k] * 2 * LDH + LDH + 4 * c);
                acc.x = fmaf(w, a.x, acc.x); acc.y = fmaf(w, a.y, acc.y);
                acc.z = fmaf(w, a.z, acc.z); acc.w = fmaf(w, a.w, acc.w);
            }
            float4 *dst = reinterpret_cast<float4 *>(gHp + n * LDH + 4 * c);
            const float4 old = *dst;
            *dst = make_float4(old.x + acc.x, old.y + acc.y, old.z + acc.z, old.w + acc.w);
        }
        __syncthreads();
        float *tmp = gH; gH = gHp; gHp = tmp;
        tmp = Hc; Hc = Hp; Hp = tmp;          // H_{t-1} is the next pass's H_t; ec already holds e_{t-1}
    }
    // input network (k_input_bwd): Hc = H_0
    for (int i = tid; i < nh * D; i += NT) {
        const int n = i / D, d = i % D;
        const float h = Hc[n * LDH + d];
        fa[n * 2 * D + d] = gH[n * LDH + d] * (1.0f - h * h);
    }
    __syncthreads();
    gin.add(nh, fa, 2 * D, Hc + D, LDH);
    // flush: one atomic per gradient element per graph (padding columns of the 4-float pieces skipped)
    g1.flush(rep, [](int l, int kk) { return kk < C ? GL::oW1 + (l % D) * 2 * C + (l / D) * C + kk : -1; }, GL::ob1);
    g3.flush(rep, [](int l, int kk) { return kk % LDH < C ? GL::oW3 + l * 3 * C + (kk / LDH) * C + kk % LDH : -1; },
             GL::ob3);
    g4.flush(rep, [](int l, int kk) { return GL::oW4 + l * D + kk; }, GL::ob4);
    gin.flush(rep, [](int l, int kk) { return kk < F ? GL::oWin + l * F + kk : -1; }, GL::obin);
    {
        constexpr int NS = D + 2, NW = NT / 64;
        __shared__ float red[NW * NS];
        float vals[NS];
#pragma unroll
        for (int i = 0; i < D; ++i) vals[i] = sW2[i];
        vals[D] = sb2;
        vals[D + 1] = spad;
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            float x = vals[i];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
            if ((tid & 63) == 0) red[(tid >> 6) * NS + i] = x;
        }
        __syncthreads();
        if (tid <= D) {
            float x = 0.0f;
#pragma unroll
            for (int w = 0; w < NW; ++w) x += red[w * NS + tid];
            atomicAdd(rep + (tid < D ? GL::oW2 + tid : GL::ob2), x);
        } else if (tid < 2 * D + 1) {
            const int i = tid - D - 1;
            float x = 0.0f;
#pragma unroll
            for (int w = 0; w < NW; ++w) x += red[w * NS + D + 1];
            const float a = tanh_f(b1[i]);
            if (x != 0.0f) atomicAdd(rep + GL::ob1 + i, x * W2[i] * (1.0f - a * a));
        }
    }
}

constexpr size_t kEventBwdLdsMax = 128 * 1024;

template <int F, int D>
int backward_events_t(const gnn_graph_t *g, const gnn_params_t *p, const int32_t *hit_ptr,
                      const int32_t *seg_ptr, int64_t n_graphs, int cap_h, int cap_s, int T,
                      const float *e_all, const float *H_all, const float *grad_out,
                      const gnn_grads_t *gr, char *ws, hipStream_t s)
{
    using GL = GradLayout<F, D>;
    // one row of partial sums per graph (= workgroup), folded in graph order
    const int64_t rows = n_graphs > 0 ? n_graphs : 1;
    float *rp = reinterpret_cast<float *>(ws);
    float *tmp = rp + (size_t)rows * GL::stride;
    hipError_t err = hipMemsetAsync(rp, 0, (size_t)rows * GL::stride * sizeof(float), s);
    if (err != hipSuccess) return fail(-(int)err, "memset of the partial-gradient table failed");
    if (n_graphs > 0) {
        static DevOnce attr_done;
        if (attr_done.need())
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_event_bwd<F, D>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)kEventBwdLdsMax);
        using B = EvBwd<F, D>;
        const size_t lds = B::lds_bytes(cap_h, cap_s);
        GNN_LAUNCH_SH("k_event_bwd", (k_event_bwd<F, D>), (unsigned)n_graphs, kBlock, lds, s, *g, *p, hit_ptr,
                      seg_ptr, T, e_all, H_all, grad_out, rp, cap_h, cap_s);
    }
    return grad_fold<F, D>(rp, rows, tmp, gr, s);
}

// shapes of the one-launch backward: the register-held gradient elements must stay few
#define BWD_EVENT_SHAPES(X_) X_(2, 4) X_(2, 8) X_(2, 16) X_(3, 4) X_(3, 8) X_(3, 16) X_(11, 4) X_(11, 8) X_(11, 16)

// ---- fused BCE loss (value + gradient in one pass over the scores) ----------------------------
// torch.nn.BCELoss semantics (the loss of gnn/estimator.py:57): logs clamped at -100,
// d/de = (e - y) / max(e (1 - e), 1e-12).  Every workgroup sums a fixed set of elements in a fixed
// order and writes one partial; k_bce_final adds the partials in index order: deterministic.
constexpr int kBceBlocks = 1024;          // 4 workgroups per CU (256 left one wave per SIMD: 40 us at 3.2 M segments)

__global__ __launch_bounds__(kBlock) void k_bce(const float *__restrict__ e, const float *__restrict__ y,
                                                int64_t n, float scale, float *__restrict__ grad_e,
                                                float *__restrict__ partial)
{
    __shared__ float red[kBlock];
    float acc = 0.0f;
    for (int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x; j < n; j += (int64_t)gridDim.x * kBlock) {
        const float ej = e[j], yj = y[j];
        const float l1 = fmaxf(logf(ej), -100.0f), l0 = fmaxf(log1pf(-ej), -100.0f);
        acc -= yj * l1 + (1.0f - yj) * l0;
        if (grad_e) grad_e[j] = scale * (ej - yj) / fmaxf(ej * (1.0f - ej), 1e-12f);
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = kBlock / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

// lane l adds partials 16 l .. 16 l + 15 in index order, the 64 lane sums meet in a fixed butterfly
__global__ __launch_bounds__(64) void k_bce_final(const float *__restrict__ partial, int n_partial,
                                                  float scale, float *__restrict__ loss)
{
    static_assert(kBceBlocks == 64 * 16, "one wavefront folds all partials");
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int k = (int)threadIdx.x * 16 + i;
        s += k < n_partial ? partial[k] : 0.0f;
    }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) s += __shfl_xor(s, o, 64);
    if (threadIdx.x == 0) loss[0] = s * scale;
}

#define BWD_FOR_EACH_SHAPE(X_) \
    X_(2, 4) X_(2, 8) X_(2, 16) X_(2, 32) X_(3, 4) X_(3, 8) X_(3, 16) X_(3, 32) X_(3, 64) X_(11, 4) X_(11, 8) \
    X_(11, 16)

}  // namespace

namespace gnn {

int bce_loss(const float *e, const float *y, int64_t n, float scale, float *loss, float *grad_e,
             float *partial, hipStream_t s)
{
    const int64_t need = (n + kBlock - 1) / kBlock;
    const int blocks = (int)(need < 1 ? 1 : need < kBceBlocks ? need : kBceBlocks);
    GNN_LAUNCH("k_bce", k_bce, blocks, kBlock, s, e, y, n, scale, grad_e, partial);
    GNN_LAUNCH("k_bce_final", k_bce_final, 1, 64, s, partial, blocks, scale, loss);
    return 0;
}

int backward_events_supported(int F, int D, int64_t max_hits, int64_t max_segments)
{
#define X_(F_, D_) if (F == F_ && D == D_) return EvBwd<F_, D_>::lds_bytes(max_hits, max_segments) <= kEventBwdLdsMax;
    BWD_EVENT_SHAPES(X_)
#undef X_
    return 0;
}

size_t backward_events_workspace_bytes(int64_t n_graphs, int F, int D)
{
    const int C = F + D;
    const int tot = D * F + D + D * 2 * C + D + D + 1 + D * 3 * C + D + D * D + D;
    const int64_t rows = n_graphs > 0 ? n_graphs : 1;
    return (size_t)(rows + fold_chunks(rows)) * ((tot + 63) & ~63) * sizeof(float) + 256;
}

int backward_events(const gnn_graph_t *g, const gnn_params_t *p, const int32_t *hit_ptr,
                    const int32_t *seg_ptr, int64_t n_graphs, int cap_h, int cap_s, int T,
                    const float *e_all, const float *H_all, const float *grad_out,
                    const gnn_grads_t *gr, void *ws, size_t ws_bytes, hipStream_t s)
{
    if (ws_bytes < backward_events_workspace_bytes(n_graphs, p->F, p->D))
        return fail(GNN_ERR_WORKSPACE, "event backward workspace too small: need %zu bytes",
                    backward_events_workspace_bytes(n_graphs, p->F, p->D));
    if (!backward_events_supported(p->F, p->D, cap_h, cap_s))
        return fail(GNN_ERR_UNSUPPORTED, "events of up to %d hits / %d segments do not fit the one-launch "
                    "backward at input_dim=%d hidden_dim=%d", cap_h, cap_s, p->F, p->D);
    char *base = reinterpret_cast<char *>((reinterpret_cast<uintptr_t>(ws) + 255) & ~(uintptr_t)255);
#define X_(F_, D_) if (p->F == F_ && p->D == D_) return backward_events_t<F_, D_>(g, p, hit_ptr, seg_ptr, n_graphs, cap_h, cap_s, T, e_all, H_all, grad_out, gr, base, s);
    BWD_EVENT_SHAPES(X_)
#undef X_
    return fail(GNN_ERR_UNSUPPORTED, "no one-launch backward for input_dim=%d hidden_dim=%d", p->F, p->D);
}

size_t backward_workspace_bytes(int64_t N, int64_t E, int F, int D)
{
    const int C = F + D;
    return carve_bwd(nullptr, N, E, (C + 3) & ~3, C, D).bytes + 256;
}

int backward(const gnn_graph_t *g, const gnn_params_t *p, int T, const float *e_all,
             const float *H_all, const float *Q_all, const float *grad_out, const gnn_grads_t *gr,
             void *ws, size_t ws_bytes, hipStream_t s)
{
    if (ws_bytes < backward_workspace_bytes(g->n_hits, g->n_segments, p->F, p->D))
        return fail(GNN_ERR_WORKSPACE, "backward workspace too small: need %zu bytes",
                    backward_workspace_bytes(g->n_hits, g->n_segments, p->F, p->D));
    char *base = reinterpret_cast<char *>((reinterpret_cast<uintptr_t>(ws) + 255) & ~(uintptr_t)255);
#define X_(F_, D_) if (p->F == F_ && p->D == D_) return backward_t<F_, D_>(g, p, T, e_all, H_all, Q_all, grad_out, gr, base, s);
    BWD_FOR_EACH_SHAPE(X_)
#undef X_
    return fail(GNN_ERR_UNSUPPORTED, "no backward kernel for input_dim=%d hidden_dim=%d", p->F, p->D);
}

int edge_bwd(const float *H, const gnn_graph_t *g, const gnn_params_t *p, const float *e, const float *ge, float *gH,
             const gnn_grads_t *gr, void *ws, size_t ws_bytes, hipStream_t s)
{
    if (ws_bytes < backward_workspace_bytes(g->n_hits, g->n_segments, p->F, p->D))
        return fail(GNN_ERR_WORKSPACE, "backward workspace too small: need %zu bytes",
                    backward_workspace_bytes(g->n_hits, g->n_segments, p->F, p->D));
    char *base = reinterpret_cast<char *>((reinterpret_cast<uintptr_t>(ws) + 255) & ~(uintptr_t)255);
#define X_(F_, D_) if (p->F == F_ && p->D == D_) return edge_bwd_t<F_, D_>(H, g, p, e, ge, gH, gr, base, s);
    BWD_FOR_EACH_SHAPE(X_)
#undef X_
    return fail(GNN_ERR_UNSUPPORTED, "no backward kernel for input_dim=%d hidden_dim=%d", p->F, p->D);
}

int node_bwd(const float *H, const float *e, const float *Hn, const gnn_graph_t *g, const gnn_params_t *p,
             const float *gHn, float *gH, float *ge, const gnn_grads_t *gr, void *ws, size_t ws_bytes, hipStream_t s)
{
    if (ws_bytes < backward_workspace_bytes(g->n_hits, g->n_segments, p->F, p->D))
        return fail(GNN_ERR_WORKSPACE, "backward workspace too small: need %zu bytes",
                    backward_workspace_bytes(g->n_hits, g->n_segments, p->F, p->D));
    char *base = reinterpret_cast<char *>((reinterpret_cast<uintptr_t>(ws) + 255) & ~(uintptr_t)255);
#define X_(F_, D_) if (p->F == F_ && p->D == D_) return node_bwd_t<F_, D_>(H, e, Hn, g, p, gHn, gH, ge, gr, base, s);
    BWD_FOR_EACH_SHAPE(X_)
#undef X_
    return fail(GNN_ERR_UNSUPPORTED, "no backward kernel for input_dim=%d hidden_dim=%d", p->F, p->D);
}

}  // namespace gnn
