// sell_pipeline.hip - the fused message-passing pipeline on a planned batch (gfx950, wave64).
//
// Reference gnn/model.py:140-156 runs, per iteration, an edge pass (:69-81) and a node pass
// (:113-125) over dense incidence matrices.  Here one kernel per iteration does both.
//
// Algebra (exact in real arithmetic, fp32 rounding differs at the 1e-7 level):
//   W1 [H_s | H_d] + b1           = P[s] + Q[d],   P = W1[:, :C] H + b1,  Q = W1[:, C:] H
//   W3 [mi | mo | H_n] + b3       = sum_in e_j R[s_j] + sum_out e_j S[d_j] + U[n],
//                                    R = W3[:, :C] H,  S = W3[:, C:2C] H,  U = W3[:, 2C:] H + b3
// so every hit publishes two records, PR = [P | R] and QS = [Q | S] (2D floats each; 64 B at
// D = 8), and a hit pulls ONE record per incident segment: the segment's score
// e_j = sigmoid(W2 tanh(P[s] + Q[d]) + b2) is recomputed from that record and the hit's own
// half (never stored), then weights R or S into the hit's accumulator.  The per-segment
// 2C x D contraction of the reference becomes a per-hit one (E/N ~ 10x fewer FMAs), the e
// vector and the H rows never travel, and all gathers are whole, aligned records.
//
// Mapping: 4 lanes per hit ("quad"); lane q of a quad owns dims [q*D/4, (q+1)*D/4) of every
// D-vector and the matching 16-byte (at D=8) chunk of a record, so a quad reads a neighbour's
// record as one coalesced 64-byte access and a wavefront (16 hits = one SELL-16 slice) reads
// its 16 neighbour ids as one 64-byte access.  The D-wide dot product with W2 is finished with
// two DPP quad-permute adds.  After the degree sort of plan.py all 16 hits of a slice have
// the same list length: no divergence, ~3 % padding (padded entries point at the NULL
// record, whose R/S half is zero).
//
// Hit-update MLP tail: lane q computes its D/4 rows of W4 and of the five record matrices;
// its weight rows differ per q, so they cannot be scalar operands: a tiny pack kernel lays
// the weights out per lane role in consumption order and each workgroup copies that table
// (2.4 KB at F=3, D=8) into LDS; the 4 roles read 4 distinct addresses per instruction
// (broadcast, conflict-free).
//
// XCD affinity: workgroups are renumbered so that each of the 8 XCDs walks one contiguous
// eighth of the slices, i.e. whole graphs: a graph's records (1.3 MB at 10k hits) are pulled
// into exactly one XCD's 4 MB L2 and gathered from there.
#include "common.h"

namespace {
using namespace gnn;

constexpr int SLICE = 16;

// ---------------------------------------------------------------------------------------------
// per-lane-role weight table layout (floats)
// ---------------------------------------------------------------------------------------------
template <int F, int D>
struct TL {
    static constexpr int d4 = D / 4, C = F + D;
    static constexpr int o_w2 = 0;                  // [d4]         W2[r]
    static constexpr int o_bin = o_w2 + d4;         // [d4]         bin[r]
    static constexpr int o_Win = o_bin + d4;        // [F][d4]      Win[r][k]
    static constexpr int o_b4 = o_Win + F * d4;     // [d4]         b4[r]
    static constexpr int o_W4 = o_b4 + d4;          // [D][d4]      W4[r][k]
    static constexpr int o_m = o_W4 + D * d4;       // 5 x { [d4] bias, [C][d4] weights }
    static constexpr int m_sz = d4 + C * d4;
    static constexpr int used = o_m + 5 * m_sz;
    static constexpr int stride = ((used + 3) & ~3) + 4;
    static constexpr int total = 4 * stride;
};

template <int F, int D>
__global__ __launch_bounds__(256) void k_pack(gnn_params_t p, float *__restrict__ table,
                                              float *PRa, float *PRb, float *QSa, float *QSb,
                                              float *U, float *Pc, float *Qc, int64_t n_hits)
{
    using L = TL<F, D>;
    constexpr int d4 = L::d4, C = L::C;
    for (int idx = threadIdx.x; idx < L::total; idx += 256) {
        const int q = idx / L::stride, pos = idx % L::stride;
        float v = 0.0f;
        if (pos < L::o_bin) {
            v = p.W2[q * d4 + pos];
        } else if (pos < L::o_Win) {
            v = p.bin[q * d4 + (pos - L::o_bin)];
        } else if (pos < L::o_b4) {
            const int t = pos - L::o_Win, k = t / d4, i = t % d4;
            v = p.Win[(q * d4 + i) * F + k];
        } else if (pos < L::o_W4) {
            v = p.b4[q * d4 + (pos - L::o_b4)];
        } else if (pos < L::o_m) {
            const int t = pos - L::o_W4, k = t / d4, i = t % d4;
            v = p.W4[(q * d4 + i) * D + k];
        } else if (pos < L::used) {
            const int t = pos - L::o_m, m = t / L::m_sz, u = t % L::m_sz;
            if (u < d4) {
                const int r = q * d4 + u;
                v = (m == 0) ? p.b1[r] : (m == 4) ? p.b3[r] : 0.0f;
            } else {
                const int k = (u - d4) / d4, r = q * d4 + (u - d4) % d4;
                switch (m) {
                case 0: v = p.W1[r * 2 * C + k]; break;            // P
                case 1: v = p.W3[r * 3 * C + k]; break;            // R
                case 2: v = p.W1[r * 2 * C + C + k]; break;        // Q
                case 3: v = p.W3[r * 3 * C + C + k]; break;        // S
                default: v = p.W3[r * 3 * C + 2 * C + k]; break;   // U
                }
            }
        }
        table[idx] = v;
    }
    // NULL hit (id n_hits): P = b1, everything else 0.  A padded list entry adds e * 0; a
    // padded segment scores sigmoid(W2 tanh(b1) + b2) (gnn/trainSegmentClassifier.py:83-93).
    for (int t = threadIdx.x; t < 2 * D; t += 256) {
        const int q = t / (2 * d4), w = t % (2 * d4);
        const float pv = (w < d4) ? p.b1[q * d4 + w] : 0.0f;
        PRa[n_hits * 2 * D + t] = pv;
        PRb[n_hits * 2 * D + t] = pv;
        QSa[n_hits * 2 * D + t] = 0.0f;
        QSb[n_hits * 2 * D + t] = 0.0f;
    }
    for (int t = threadIdx.x; t < D; t += 256) {
        U[n_hits * D + t] = 0.0f;
        Pc[n_hits * D + t] = p.b1[t];
        Qc[n_hits * D + t] = 0.0f;
    }
}

// ---------------------------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ float dpp(float v)
{
    return __builtin_bit_cast(
        float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}

// sum over the 4 lanes of a quad, identical bits in all 4 lanes
__device__ __forceinline__ float quad_sum(float v)
{
    v += dpp<0xB1>(v);   // quad_perm [1,0,3,2]
    v += dpp<0x4E>(v);   // quad_perm [2,3,0,1]
    return v;
}

// all-gather inside a quad: out[j*N + i] = loc[i] of lane j
template <int N>
__device__ __forceinline__ void quad_allgather(const float (&loc)[N], float *out)
{
#pragma unroll
    for (int i = 0; i < N; ++i) {
        out[0 * N + i] = dpp<0x00>(loc[i]);
        out[1 * N + i] = dpp<0x55>(loc[i]);
        out[2 * N + i] = dpp<0xAA>(loc[i]);
        out[3 * N + i] = dpp<0xFF>(loc[i]);
    }
}

template <int N>
__device__ __forceinline__ void load_vec(const float *__restrict__ src, float *dst)
{
    if constexpr (N == 1) {
        dst[0] = src[0];
    } else if constexpr (N == 2) {
        const float2 v = *reinterpret_cast<const float2 *>(src);
        dst[0] = v.x; dst[1] = v.y;
    } else {
        static_assert(N % 4 == 0, "vector length");
#pragma unroll
        for (int i = 0; i < N / 4; ++i) {
            const float4 v = reinterpret_cast<const float4 *>(src)[i];
            dst[4 * i] = v.x; dst[4 * i + 1] = v.y; dst[4 * i + 2] = v.z; dst[4 * i + 3] = v.w;
        }
    }
}

template <int N>
__device__ __forceinline__ void store_vec(float *__restrict__ dst, const float *src)
{
    if constexpr (N == 1) {
        dst[0] = src[0];
    } else if constexpr (N == 2) {
        *reinterpret_cast<float2 *>(dst) = make_float2(src[0], src[1]);
    } else {
        static_assert(N % 4 == 0, "vector length");
#pragma unroll
        for (int i = 0; i < N / 4; ++i)
            reinterpret_cast<float4 *>(dst)[i] =
                make_float4(src[4 * i], src[4 * i + 1], src[4 * i + 2], src[4 * i + 3]);
    }
}

// copy the packed weight table global -> LDS (whole workgroup), then barrier
template <int TOTAL>
__device__ __forceinline__ void stage_table(const float *__restrict__ table, float *lds)
{
    static_assert(TOTAL % 4 == 0, "table is float4 granular");
    for (int i = threadIdx.x; i < TOTAL / 4; i += 256)
        reinterpret_cast<float4 *>(lds)[i] = reinterpret_cast<const float4 *>(table)[i];
    __syncthreads();
}

// out[i] = bias[i] + sum_k W[k][i] * in[k]  for this lane's d4 rows; weights from LDS
template <int D4, int KD, int KF>
__device__ __forceinline__ void role_gemv(const float *w, const float *hn, const float *x,
                                          float *out)
{
#pragma unroll
    for (int i = 0; i < D4; ++i) out[i] = w[i];
    w += D4;
#pragma unroll
    for (int k = 0; k < KD; ++k)
#pragma unroll
        for (int i = 0; i < D4; ++i) out[i] = fmaf(w[k * D4 + i], hn[k], out[i]);
#pragma unroll
    for (int k = 0; k < KF; ++k)
#pragma unroll
        for (int i = 0; i < D4; ++i) out[i] = fmaf(w[(KD + k) * D4 + i], x[k], out[i]);
}

// From the new hit features [hn (D) | x (F)] emit this lane's chunk of the records the next
// pass gathers: PR = [P | R], QS = [Q | S], U; or, for the last iteration, compact P and Q.
template <int F, int D, bool LAST>
__device__ __forceinline__ void emit_records(const float *wl, const float *hn, const float *x,
                                             int64_t n, int q, float *__restrict__ PRn,
                                             float *__restrict__ QSn, float *__restrict__ U,
                                             float *__restrict__ Pc, float *__restrict__ Qc)
{
    using L = TL<F, D>;
    constexpr int d4 = L::d4;
    if constexpr (LAST) {
        float pv[d4], qv[d4];
        role_gemv<d4, D, F>(wl + L::o_m + 0 * L::m_sz, hn, x, pv);
        role_gemv<d4, D, F>(wl + L::o_m + 2 * L::m_sz, hn, x, qv);
        store_vec<d4>(Pc + n * D + q * d4, pv);
        store_vec<d4>(Qc + n * D + q * d4, qv);
    } else {
        float pr[2 * d4], qs[2 * d4], u[d4];
        role_gemv<d4, D, F>(wl + L::o_m + 0 * L::m_sz, hn, x, pr);
        role_gemv<d4, D, F>(wl + L::o_m + 1 * L::m_sz, hn, x, pr + d4);
        role_gemv<d4, D, F>(wl + L::o_m + 2 * L::m_sz, hn, x, qs);
        role_gemv<d4, D, F>(wl + L::o_m + 3 * L::m_sz, hn, x, qs + d4);
        role_gemv<d4, D, F>(wl + L::o_m + 4 * L::m_sz, hn, x, u);
        store_vec<2 * d4>(PRn + n * 2 * D + q * 2 * d4, pr);
        store_vec<2 * d4>(QSn + n * 2 * D + q * 2 * d4, qs);
        store_vec<d4>(U + n * D + q * d4, u);
    }
}

// ---------------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------------
// input network (model.py:144-146) + records of iteration 0.  4 lanes per hit.
template <int F, int D, bool LAST>
__global__ __launch_bounds__(256) void k_input4(const float *__restrict__ X,
                                                const float *__restrict__ table,
                                                float *__restrict__ PRn, float *__restrict__ QSn,
                                                float *__restrict__ U, float *__restrict__ Pc,
                                                float *__restrict__ Qc, int64_t n_hits)
{
    using L = TL<F, D>;
    constexpr int d4 = L::d4;
    __shared__ __attribute__((aligned(16))) float lds[L::total];
    stage_table<L::total>(table, lds);
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t n = gid >> 2;
    const int q = threadIdx.x & 3;
    const bool live = n < n_hits;
    const int64_t ne = live ? n : n_hits;          // NULL row is readable
    const float *wl = lds + q * L::stride;
    float x[F];
#pragma unroll
    for (int k = 0; k < F; ++k) x[k] = X[ne * F + k];
    float hl[d4];
#pragma unroll
    for (int i = 0; i < d4; ++i) {
        float a = wl[L::o_bin + i];
#pragma unroll
        for (int k = 0; k < F; ++k) a = fmaf(wl[L::o_Win + k * d4 + i], x[k], a);
        hl[i] = tanh_f(a);
    }
    float hn[D];
    quad_allgather<d4>(hl, hn);
    if (live) emit_records<F, D, LAST>(wl, hn, x, n, q, PRn, QSn, U, Pc, Qc);
}

// one message-passing iteration: edge scores + weighted aggregation + hit update (+ records).
template <int F, int D, bool LAST>
__global__ __launch_bounds__(256) void k_iter(
    const float *__restrict__ X, const float *__restrict__ table, const float *__restrict__ b2p,
    const int32_t *__restrict__ in_off, const int32_t *__restrict__ in_nbr,
    const int32_t *__restrict__ out_off, const int32_t *__restrict__ out_nbr,
    const float *__restrict__ PR, const float *__restrict__ QS, float *__restrict__ U,
    float *__restrict__ PRn, float *__restrict__ QSn, float *__restrict__ Pc,
    float *__restrict__ Qc, int64_t n_hits, int n_slices, int blocks_per_xcd)
{
    using L = TL<F, D>;
    constexpr int d4 = L::d4;
    __shared__ __attribute__((aligned(16))) float lds[L::total];
    stage_table<L::total>(table, lds);

    // XCD-affine renumbering: hardware deals blockIdx round-robin over the 8 XCDs, so
    // blocks with equal (blockIdx & 7) share an L2; give each such group a contiguous range.
    const int vblock = (blockIdx.x & 7) * blocks_per_xcd + (blockIdx.x >> 3);
    const int slice = __builtin_amdgcn_readfirstlane(vblock * 4 + (int)(threadIdx.x >> 6));
    if (slice >= n_slices) return;
    const int lane = threadIdx.x & 63;
    const int q = lane & 3, i16 = lane >> 2;
    const int64_t n = (int64_t)slice * SLICE + i16;
    const bool live = n < n_hits;
    const int64_t ne = live ? n : n_hits;
    const float *wl = lds + q * L::stride;
    const float b2 = b2p[0];

    float w2[d4], Pn[d4], Qn[d4], acc[d4];
#pragma unroll
    for (int i = 0; i < d4; ++i) w2[i] = wl[L::o_w2 + i];
    load_vec<d4>(PR + ne * 2 * D + q * 2 * d4, Pn);       // own P chunk
    load_vec<d4>(QS + ne * 2 * D + q * 2 * d4, Qn);       // own Q chunk
    load_vec<d4>(U + ne * D + q * d4, acc);               // W3[:, 2C:] H_n + b3

    // one neighbour record: score the segment, add its weighted half
    auto pull = [&](const float *rec, const float *own) {
        float part = 0.0f;
#pragma unroll
        for (int i = 0; i < d4; ++i) part = fmaf(w2[i], tanh_f(rec[i] + own[i]), part);
        const float e = sigmoid_f(quad_sum(part) + b2);
#pragma unroll
        for (int i = 0; i < d4; ++i) acc[i] = fmaf(e, rec[d4 + i], acc[i]);
    };
    auto sweep = [&](const int32_t *__restrict__ off, const int32_t *__restrict__ nbr,
                     const float *__restrict__ REC, const float *own) {
        const int base = __builtin_amdgcn_readfirstlane(off[slice]);
        const int len = (__builtin_amdgcn_readfirstlane(off[slice + 1]) - base) >> 4;
        const int32_t *lst = nbr + base + i16;
        // UN independent record gathers in flight per lane (fewer at large D: registers)
        constexpr int UN = d4 <= 2 ? 4 : (d4 <= 4 ? 2 : 1);
        int k = 0;
        for (; k + UN <= len; k += UN) {
            int nb[UN];
            float rec[UN][2 * d4];
#pragma unroll
            for (int j = 0; j < UN; ++j) nb[j] = lst[(k + j) * SLICE];
#pragma unroll
            for (int j = 0; j < UN; ++j)
                load_vec<2 * d4>(REC + (int64_t)nb[j] * 2 * D + q * 2 * d4, rec[j]);
#pragma unroll
            for (int j = 0; j < UN; ++j) pull(rec[j], own);
        }
        for (; k < len; ++k) {
            float rec[2 * d4];
            load_vec<2 * d4>(REC + (int64_t)lst[k * SLICE] * 2 * D + q * 2 * d4, rec);
            pull(rec, own);
        }
    };
    sweep(in_off, in_nbr, PR, Qn);     // segments ending here:   P[start] + Q[n], adds e * R[start]
    sweep(out_off, out_nbr, QS, Pn);   // segments starting here: Q[end] + P[n],   adds e * S[end]

    // hit update: H' = tanh(W4 tanh(acc) + b4)                      (model.py:94-98,125)
    float ql[d4], qa[D];
#pragma unroll
    for (int i = 0; i < d4; ++i) ql[i] = tanh_f(acc[i]);
    quad_allgather<d4>(ql, qa);
    float hl[d4];
#pragma unroll
    for (int i = 0; i < d4; ++i) hl[i] = wl[L::o_b4 + i];
#pragma unroll
    for (int k = 0; k < D; ++k)
#pragma unroll
        for (int i = 0; i < d4; ++i) hl[i] = fmaf(wl[L::o_W4 + k * d4 + i], qa[k], hl[i]);
#pragma unroll
    for (int i = 0; i < d4; ++i) hl[i] = tanh_f(hl[i]);
    float hn[D], x[F];
    quad_allgather<d4>(hl, hn);
#pragma unroll
    for (int k = 0; k < F; ++k) x[k] = X[ne * F + k];               // skip concat (model.py:154)
    if (live) emit_records<F, D, LAST>(wl, hn, x, n, q, PRn, QSn, U, Pc, Qc);
}

// final edge pass (model.py:156): caller's segment order, 4 lanes per segment.
template <int F, int D>
__global__ __launch_bounds__(256) void k_edge4(const int32_t *__restrict__ src,
                                               const int32_t *__restrict__ dst,
                                               const float *__restrict__ Pc,
                                               const float *__restrict__ Qc,
                                               const float *__restrict__ table,
                                               const float *__restrict__ b2p,
                                               float *__restrict__ e, int64_t n_segments,
                                               int blocks_per_xcd)
{
    using L = TL<F, D>;
    constexpr int d4 = L::d4;
    const int64_t vblock = (int64_t)(blockIdx.x & 7) * blocks_per_xcd + (blockIdx.x >> 3);
    const int64_t j = vblock * 64 + (threadIdx.x >> 2);
    if (j >= n_segments) return;
    const int q = threadIdx.x & 3;
    float w2[d4], p[d4], qq[d4];
    load_vec<d4>(table + q * L::stride + L::o_w2, w2);
    load_vec<d4>(Pc + (int64_t)src[j] * D + q * d4, p);
    load_vec<d4>(Qc + (int64_t)dst[j] * D + q * d4, qq);
    float part = 0.0f;
#pragma unroll
    for (int i = 0; i < d4; ++i) part = fmaf(w2[i], tanh_f(p[i] + qq[i]), part);
    const float ev = sigmoid_f(quad_sum(part) + b2p[0]);
    if (q == 0) e[j] = ev;
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
struct Ws {
    float *table, *PRa, *PRb, *QSa, *QSb, *U, *Pc, *Qc;
    size_t bytes;
};

Ws carve(char *b, int64_t n_hits, int table_floats, int D)
{
    Ws w;
    size_t off = 0;
    auto take = [&](size_t nfloat) {
        float *p = reinterpret_cast<float *>(b + off);
        off += align256(nfloat * sizeof(float));
        return p;
    };
    const size_t rec = (size_t)(n_hits + 1) * 2 * D, vec = (size_t)(n_hits + 1) * D;
    w.table = take((size_t)table_floats);
    w.PRa = take(rec); w.PRb = take(rec); w.QSa = take(rec); w.QSb = take(rec);
    w.U = take(vec); w.Pc = take(vec); w.Qc = take(vec);
    w.bytes = off;
    return w;
}

template <int F, int D>
int forward_t(const gnn_plan_t *pl, const gnn_params_t *p, int n_iters, float *e_out, char *ws,
              hipStream_t s)
{
    using L = TL<F, D>;
    const int64_t N = pl->n_hits, E = pl->n_segments;
    Ws w = carve(ws, N, L::total, D);
    GNN_LAUNCH("k_pack", (k_pack<F, D>), 1, 256, s, *p, w.table, w.PRa, w.PRb, w.QSa, w.QSb, w.U,
               w.Pc, w.Qc, N);
    float *PR = w.PRa, *PRn = w.PRb, *QS = w.QSa, *QSn = w.QSb;
    if (N > 0) {
        const unsigned g = (unsigned)((N * 4 + 255) / 256);
        if (n_iters == 0)
            GNN_LAUNCH("k_input4", (k_input4<F, D, true>), g, 256, s, pl->X, w.table, PR, QS, w.U,
                       w.Pc, w.Qc, N);
        else
            GNN_LAUNCH("k_input4", (k_input4<F, D, false>), g, 256, s, pl->X, w.table, PR, QS, w.U,
                       w.Pc, w.Qc, N);
        const int n_slices = (int)pl->n_slices;
        const int bpx = ((n_slices + 3) / 4 + 7) / 8;       // workgroups per XCD group
        for (int t = 0; t < n_iters; ++t) {
            if (t + 1 == n_iters)
                GNN_LAUNCH("k_iter", (k_iter<F, D, true>), 8 * bpx, 256, s, pl->X, w.table, p->b2,
                           pl->in_off, pl->in_nbr, pl->out_off, pl->out_nbr, PR, QS, w.U, PRn, QSn,
                           w.Pc, w.Qc, N, n_slices, bpx);
            else
                GNN_LAUNCH("k_iter", (k_iter<F, D, false>), 8 * bpx, 256, s, pl->X, w.table, p->b2,
                           pl->in_off, pl->in_nbr, pl->out_off, pl->out_nbr, PR, QS, w.U, PRn, QSn,
                           w.Pc, w.Qc, N, n_slices, bpx);
            float *t1 = PR; PR = PRn; PRn = t1;
            float *t2 = QS; QS = QSn; QSn = t2;
        }
    }
    if (E > 0) {
        const int64_t nblk = (E + 63) / 64;
        const int bpx = (int)((nblk + 7) / 8);
        GNN_LAUNCH("k_edge4", (k_edge4<F, D>), 8 * bpx, 256, s, pl->src, pl->dst, w.Pc, w.Qc,
                   w.table, p->b2, e_out, E, bpx);
    }
    return 0;
}

#define SELL_FOR_EACH_SHAPE(X_)                                                          \
    X_(2, 4) X_(2, 8) X_(2, 16) X_(2, 32) X_(3, 4) X_(3, 8) X_(3, 16) X_(3, 32) X_(3, 64) \
    X_(11, 4) X_(11, 8) X_(11, 16)

}  // namespace

namespace gnn {

int sell_shape_supported(int F, int D)
{
#define X_(F_, D_) if (F == F_ && D == D_) return 1;
    SELL_FOR_EACH_SHAPE(X_)
#undef X_
    return 0;
}

size_t sell_workspace_bytes(int64_t n_hits, int64_t n_segments, int F, int D)
{
    (void)n_segments;
#define X_(F_, D_) if (F == F_ && D == D_) return carve(nullptr, n_hits, TL<F_, D_>::total, D).bytes + 256;
    SELL_FOR_EACH_SHAPE(X_)
#undef X_
    return 0;
}

int sell_forward(const gnn_plan_t *pl, const gnn_params_t *p, int n_iters, float *e_out, void *ws,
                 size_t ws_bytes, hipStream_t s)
{
    const size_t need = sell_workspace_bytes(pl->n_hits, pl->n_segments, p->F, p->D);
    if (need == 0) return fail(GNN_ERR_UNSUPPORTED, "no fused kernel for input_dim=%d hidden_dim=%d", p->F, p->D);
    if (!ws || ws_bytes < need) return fail(GNN_ERR_WORKSPACE, "workspace too small: need %zu bytes", need);
    char *base = reinterpret_cast<char *>((reinterpret_cast<uintptr_t>(ws) + 255) & ~(uintptr_t)255);
#define X_(F_, D_) if (p->F == F_ && p->D == D_) return forward_t<F_, D_>(pl, p, n_iters, e_out, base, s);
    SELL_FOR_EACH_SHAPE(X_)
#undef X_
    return fail(GNN_ERR_UNSUPPORTED, "unreachable");
}

}  // namespace gnn
