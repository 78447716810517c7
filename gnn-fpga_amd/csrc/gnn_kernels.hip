// gnn_kernels.hip - SegmentClassifier message-passing forward for gfx950 (MI355X, CDNA4).
//
// What the reference does with dense one-hot incidence matrices and bmm
// (gnn/model.py:69-81,113-125,140-156) is done here in index form, one lane per
// segment (edge pass) or per hit (node pass), wave64, fp32 throughout:
//
//   k_input  H0[n] = [tanh(Win X[n] + bin) | X[n]]                       model.py:144-146
//   k_edge   e[j]  = sigmoid(W2 tanh(P[src j] + Q[dst j]) + b2)          model.py:71-73,45-49
//   k_node   mi/mo = CSR pull segment sums of e[j] * H[nbr]; H' = tanh(W4 tanh(W3 M + b3) + b4)
//                                                                        model.py:114-125,154
//
// The first edge-MLP layer is linear in the concatenated pair [H_src | H_dst], so it is
// split per hit:  W1 [H_s | H_d] + b1 = (W1[:, :C] H_s + b1) + W1[:, C:] H_d = P[s] + Q[d].
// P and Q (D floats each) are produced by the kernel that produces H (k_input / k_node), which
// turns the per-segment 2C x D contraction into a per-hit one (E/N ~ 10x fewer FMAs) and
// shrinks the per-segment gather from 2 x C to 2 x D floats.
//
// Weights are tiny (569 floats at F=3, D=8): every weight address is wave-uniform, so the
// compiler keeps them on the scalar path (s_load / SGPR operands) - no LDS, no VGPR copies.
// No MFMA here by design: at D <= 16 the work is HBM/gather-bound integer-indexed traffic.
#include "common.h"

#include <type_traits>

namespace gnn {

thread_local char g_err[320] = "";

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

struct Profiler {
    bool on = false;
    int cap = 0;
    std::vector<hipEvent_t> ev;     // 2 per record
    std::vector<const char *> name;
    std::vector<int> start;         // index of the event a record starts at
    int n = 0;
    // Inside one library call that launches several kernels back to back (ProfChain), a kernel's
    // interval starts at the event recorded behind the previous kernel: ONE event per kernel
    // boundary, so the intervals of a call tile its time exactly - each is the kernel plus the
    // boundary to the next launch, as in an unprofiled run - instead of two event packets per
    // kernel, whose processing time (~10 us per pair) ended up inside every interval.
    bool chain = false, have_prev = false;
};
Profiler g_prof;

// GNN_DEBUG_SYNC=1: every launch is named on stderr and waited for - the last name printed before a GPU fault is
// the kernel that raised it (a debugging switch; never set in measurements)
static const char *g_dbg_name = nullptr;
static bool debug_sync()
{
    static int on = -1;
    if (on < 0) { const char *e = getenv("GNN_DEBUG_SYNC"); on = (e && e[0] == '1') ? 1 : 0; }
    return on == 1;
}

void prof_pre(const char *name, hipStream_t s)
{
    if (debug_sync()) g_dbg_name = name;
    if (g_prof.on && g_prof.n < g_prof.cap) {
        g_prof.name[g_prof.n] = name;
        if (g_prof.chain && g_prof.have_prev && g_prof.n > 0) {
            g_prof.start[g_prof.n] = 2 * (g_prof.n - 1) + 1;
        } else {
            g_prof.start[g_prof.n] = 2 * g_prof.n;
            (void)hipEventRecord(g_prof.ev[2 * g_prof.n], s);
        }
    }
}
void prof_post(hipStream_t s)
{
    if (debug_sync()) {
        fprintf(stderr, "[gnn] %s\n", g_dbg_name ? g_dbg_name : "?");
        fflush(stderr);
        (void)hipStreamSynchronize(s);
    }
    if (g_prof.on && g_prof.n < g_prof.cap) {
        (void)hipEventRecord(g_prof.ev[2 * g_prof.n + 1], s);
        g_prof.n++;
        g_prof.have_prev = true;
    }
}
ProfChain::ProfChain() : prev(g_prof.chain)
{
    if (!prev) { g_prof.chain = true; g_prof.have_prev = false; }
}
ProfChain::~ProfChain()
{
    if (!prev) g_prof.chain = g_prof.have_prev = false;
}

}  // namespace gnn

namespace {
using namespace gnn;

template <int F, int D>
struct Shape {
    static constexpr int C = F + D;
    static constexpr int LDH = (C + 3) & ~3;
};

// P = W1[:, :C] h + b1, Q = W1[:, C:] h  -> PQ row [P(D) | Q(D)], stored as float4s.
template <int F, int D>
__device__ __forceinline__ void store_pq(const float *h, const float *__restrict__ W1,
                                         const float *__restrict__ b1, float *__restrict__ pq_row)
{
    constexpr int C = F + D;
    float pq[2 * D];
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
    float4 *o = reinterpret_cast<float4 *>(pq_row);
#pragma unroll
    for (int v = 0; v < 2 * D / 4; ++v)
        o[v] = make_float4(pq[4 * v], pq[4 * v + 1], pq[4 * v + 2], pq[4 * v + 3]);
}

template <int N4>
__device__ __forceinline__ void store_row4(float *__restrict__ row, const float *v)
{
    float4 *o = reinterpret_cast<float4 *>(row);
#pragma unroll
    for (int i = 0; i < N4; ++i) o[i] = make_float4(v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]);
}

// ---------------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------------
// input network + skip concat (+ P/Q of the first edge pass).  One hit per lane.
template <int F, int D>
__global__ __launch_bounds__(kBlock) void k_input(const float *__restrict__ X,
                                                  const float *__restrict__ Win,
                                                  const float *__restrict__ bin,
                                                  const float *__restrict__ W1,
                                                  const float *__restrict__ b1,
                                                  float *__restrict__ H, int ldh,
                                                  float *__restrict__ PQ, int64_t n_hits)
{
    constexpr int LDH = Shape<F, D>::LDH;
    const int64_t n = xcd_block() * kBlock + threadIdx.x;
    if (n >= n_hits) return;
    float h[LDH];
#pragma unroll
    for (int k = 0; k < F; ++k) h[D + k] = X[n * F + k];
#pragma unroll
    for (int k = D + F; k < LDH; ++k) h[k] = 0.0f;
#pragma unroll
    for (int d = 0; d < D; ++d) {
        float acc = bin[d];
#pragma unroll
        for (int k = 0; k < F; ++k) acc = fmaf(Win[d * F + k], h[D + k], acc);
        h[d] = tanh_f(acc);
    }
    store_row4<LDH / 4>(H + n * ldh, h);
    if (PQ) store_pq<F, D>(h, W1, b1, PQ + n * 2 * D);
}

// P/Q from an existing H (stand-alone EdgeNetwork entry point).
template <int F, int D>
__global__ __launch_bounds__(kBlock) void k_pq(const float *__restrict__ H, int ldh,
                                               const float *__restrict__ W1,
                                               const float *__restrict__ b1,
                                               float *__restrict__ PQ, int64_t n_hits)
{
    constexpr int C = F + D;
    const int64_t n = xcd_block() * kBlock + threadIdx.x;
    if (n >= n_hits) return;
    float h[C];
#pragma unroll
    for (int k = 0; k < C; ++k) h[k] = H[n * ldh + k];
    store_pq<F, D>(h, W1, b1, PQ + n * 2 * D);
}

// edge pass: one segment per lane; coalesced src/dst reads and e writes, two D-float gathers.
template <int D>
__global__ __launch_bounds__(kBlock) void k_edge(const int32_t *__restrict__ src,
                                                 const int32_t *__restrict__ dst,
                                                 const float *__restrict__ PQ,
                                                 const float *__restrict__ b1,
                                                 const float *__restrict__ W2,
                                                 const float *__restrict__ b2,
                                                 float *__restrict__ e, int64_t n_segments)
{
    const int64_t j = xcd_block() * kBlock + threadIdx.x;
    if (j >= n_segments) return;
    const int s = src[j], d = dst[j];
    float z[D];
    if (s >= 0) {
        const float4 *p = reinterpret_cast<const float4 *>(PQ + (int64_t)s * 2 * D);
        const float4 *q = reinterpret_cast<const float4 *>(PQ + (int64_t)d * 2 * D + D);
#pragma unroll
        for (int v = 0; v < D / 4; ++v) {
            const float4 a = p[v], b = q[v];
            z[4 * v] = a.x + b.x;
            z[4 * v + 1] = a.y + b.y;
            z[4 * v + 2] = a.z + b.z;
            z[4 * v + 3] = a.w + b.w;
        }
    } else {  // padded column: gathered rows are zero -> first layer output is b1
#pragma unroll
        for (int k = 0; k < D; ++k) z[k] = b1[k];
    }
    float acc = b2[0];
#pragma unroll
    for (int k = 0; k < D; ++k) acc = fmaf(W2[k], tanh_f(z[k]), acc);
    e[j] = sigmoid_f(acc);
}

// node pass: one hit per lane; pull-mode segment sums over the two CSRs in ascending segment
// id, fused 3C -> D -> D tanh MLP, skip concat, and P/Q for the next edge pass.
template <int F, int D>
__global__ __launch_bounds__(kBlock) void k_node(
    const float *__restrict__ H, int ldh, const float *__restrict__ e,
    const int32_t *__restrict__ in_ptr, const int32_t *__restrict__ in_eid,
    const int32_t *__restrict__ in_nbr, const int32_t *__restrict__ out_ptr,
    const int32_t *__restrict__ out_eid, const int32_t *__restrict__ out_nbr,
    const float *__restrict__ W3, const float *__restrict__ b3, const float *__restrict__ W4,
    const float *__restrict__ b4, const float *__restrict__ W1, const float *__restrict__ b1,
    float *__restrict__ Hn, int ldhn, float *__restrict__ PQ, float *__restrict__ Qkeep, int64_t n_hits)
{
    constexpr int C = Shape<F, D>::C;
    constexpr int LDH = Shape<F, D>::LDH;
    const int64_t n = xcd_block() * kBlock + threadIdx.x;
    if (n >= n_hits) return;

    float M[3 * LDH];   // [mi | mo | h], each padded to LDH
#pragma unroll
    for (int k = 0; k < 2 * LDH; ++k) M[k] = 0.0f;
    {
        const float4 *hp = reinterpret_cast<const float4 *>(H + n * ldh);
#pragma unroll
        for (int v = 0; v < LDH / 4; ++v) {
            const float4 a = hp[v];
            M[2 * LDH + 4 * v] = a.x;
            M[2 * LDH + 4 * v + 1] = a.y;
            M[2 * LDH + 4 * v + 2] = a.z;
            M[2 * LDH + 4 * v + 3] = a.w;
        }
    }
    // segments ending here: weight e[j], features of the start hit      (model.py:117-118)
    constexpr int U = D <= 16 ? 4 : 1;
    csr_walk<LDH / 4, U>(in_ptr[n], in_ptr[n + 1],
                         [&](int k) { return H + (int64_t)in_nbr[k] * ldh; },
                         [&](int k) { return e[in_eid[k]]; },
                         [&](float w, const float *a) {
#pragma unroll
                             for (int c = 0; c < LDH; ++c) M[c] = fmaf(w, a[c], M[c]);
                         });
    // segments starting here: weight e[j], features of the end hit      (model.py:116,119)
    csr_walk<LDH / 4, U>(out_ptr[n], out_ptr[n + 1],
                         [&](int k) { return H + (int64_t)out_nbr[k] * ldh; },
                         [&](int k) { return e[out_eid[k]]; },
                         [&](float w, const float *a) {
#pragma unroll
                             for (int c = 0; c < LDH; ++c) M[LDH + c] = fmaf(w, a[c], M[LDH + c]);
                         });
    // MLP on M = [mi | mo | h]                                          (model.py:120,94-98)
    float q[D];
#pragma unroll
    for (int d = 0; d < D; ++d) {
        float acc = b3[d];
#pragma unroll
        for (int k = 0; k < C; ++k) acc = fmaf(W3[d * 3 * C + k], M[k], acc);
#pragma unroll
        for (int k = 0; k < C; ++k) acc = fmaf(W3[d * 3 * C + C + k], M[LDH + k], acc);
#pragma unroll
        for (int k = 0; k < C; ++k) acc = fmaf(W3[d * 3 * C + 2 * C + k], M[2 * LDH + k], acc);
        q[d] = tanh_f(acc);
    }
    if (Qkeep) store_row4<D / 4>(Qkeep + n * D, q);         // training: the backward needs no second list walk
    float hn[LDH];
#pragma unroll
    for (int d = 0; d < D; ++d) {
        float acc = b4[d];
#pragma unroll
        for (int k = 0; k < D; ++k) acc = fmaf(W4[d * D + k], q[k], acc);
        hn[d] = tanh_f(acc);
    }
#pragma unroll
    for (int k = D; k < LDH; ++k) hn[k] = M[2 * LDH + k];   // skip concat of X (model.py:154)
    store_row4<LDH / 4>(Hn + n * ldhn, hn);
    if (PQ) store_pq<F, D>(hn, W1, b1, PQ + n * 2 * D);
}

// ---------------------------------------------------------------------------------------------
// node pass for WIDE hidden layers (hidden_dim 32 / 64) at detector size, in two kernels
// ---------------------------------------------------------------------------------------------
// k_node keeps M = [mi | mo | h] (3C = 201 floats at D = 64) in one lane's registers and feeds the MLP
// scalar weight operands: 100 KB of weights per wave through the scalar cache, 0.42 ms per pass at
// 50 k hits.  Here (1) k_node_walkW: 16 lanes per hit walk the two lists (lane p gathers 16-byte
// chunk p of a neighbour's row; list entries and scores one per lane, broadcast in the quad) and
// store mi | mo; (2) k_node_mlpW: the MLP of 256 hits as exact fp32 matrix-core products
// (v_mfma_f32_16x16x4_f32 accumulates in k order from the bias: the same fma chain as k_node's loops)
// from transposed LDS stagings of mi, mo, h, then q, then [H' | x].
template <int SEL>
__device__ __forceinline__ int qb_i(int x) { return __builtin_amdgcn_mov_dpp(x, SEL * 0x55, 0xf, 0xf, true); }
template <int SEL>
__device__ __forceinline__ float qb_f(float x)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), SEL * 0x55, 0xf, 0xf, true));
}

template <int LDH>
__device__ __forceinline__ void row16_walk(int beg, int end, int n, int p, const int32_t *__restrict__ nbr,
                                           const int32_t *__restrict__ eid, const float *__restrict__ e,
                                           const float *__restrict__ H, int ldh, float (*acc)[4])
{
    constexpr int NCH = LDH / 4, CPL = (NCH + 15) / 16;
    const int q = p & 3;
    for (int k = beg; k < end; k += 4) {
        const int kk = k + q;
        const bool ok = kk < end;
        const int nb = ok ? nbr[kk] : n;
        const float ev = ok ? e[eid[kk]] : 0.0f;
        float4 r[4][CPL];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int nbj = j == 0 ? qb_i<0>(nb) : j == 1 ? qb_i<1>(nb) : j == 2 ? qb_i<2>(nb) : qb_i<3>(nb);
            const float4 *row = reinterpret_cast<const float4 *>(H + (int64_t)nbj * ldh);
#pragma unroll
            for (int c = 0; c < CPL; ++c)
                r[j][c] = (p + 16 * c < NCH) ? row[p + 16 * c] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float w = j == 0 ? qb_f<0>(ev) : j == 1 ? qb_f<1>(ev) : j == 2 ? qb_f<2>(ev) : qb_f<3>(ev);
            if (k + j < end) {                              // (row-uniform; the sums are those of k_node)
#pragma unroll
                for (int c = 0; c < CPL; ++c) {
                    acc[c][0] = fmaf(w, r[j][c].x, acc[c][0]);
                    acc[c][1] = fmaf(w, r[j][c].y, acc[c][1]);
                    acc[c][2] = fmaf(w, r[j][c].z, acc[c][2]);
                    acc[c][3] = fmaf(w, r[j][c].w, acc[c][3]);
                }
            }
        }
    }
}

constexpr int kWalkBlock = 1024, kWalkHits = kWalkBlock / 16;
constexpr int64_t kNodeWideMinHits = 2048;
inline unsigned grid_walk(int64_t n)
{
    const unsigned g = (unsigned)((n + kWalkHits - 1) / kWalkHits);
    return g > 8 ? (g + 7) & ~7u : g;
}

template <int F, int D>
__global__ __launch_bounds__(kWalkBlock) void k_node_walkW(
    const float *__restrict__ H, int ldh, const float *__restrict__ e,
    const int32_t *__restrict__ in_ptr, const int32_t *__restrict__ in_eid, const int32_t *__restrict__ in_nbr,
    const int32_t *__restrict__ out_ptr, const int32_t *__restrict__ out_eid, const int32_t *__restrict__ out_nbr,
    float *__restrict__ M, int64_t n_hits)
{
    constexpr int LDH = Shape<F, D>::LDH, NCH = LDH / 4, CPL = (NCH + 15) / 16;
    const int p = threadIdx.x & 15;
    const int64_t n = xcd_block() * kWalkHits + (threadIdx.x >> 4);
    if (n >= n_hits) return;                            // (whole 16-lane rows leave together)
    float mi[CPL][4], mo[CPL][4];
#pragma unroll
    for (int c = 0; c < CPL; ++c)
#pragma unroll
        for (int i = 0; i < 4; ++i) mi[c][i] = mo[c][i] = 0.0f;
    row16_walk<LDH>(in_ptr[n], in_ptr[n + 1], (int)n, p, in_nbr, in_eid, e, H, ldh, mi);       // model.py:117-118
    row16_walk<LDH>(out_ptr[n], out_ptr[n + 1], (int)n, p, out_nbr, out_eid, e, H, ldh, mo);   // model.py:116,119
    float4 *row = reinterpret_cast<float4 *>(M + n * 2 * LDH);
#pragma unroll
    for (int c = 0; c < CPL; ++c)
        if (p + 16 * c < NCH) {
            row[p + 16 * c] = make_float4(mi[c][0], mi[c][1], mi[c][2], mi[c][3]);
            row[NCH + p + 16 * c] = make_float4(mo[c][0], mo[c][1], mo[c][2], mo[c][3]);
        }
}

template <int F, int D>
__global__ __launch_bounds__(kBlock) void k_node_mlpW(
    const float *__restrict__ H, int ldh, const float *__restrict__ M, const float *__restrict__ W3,
    const float *__restrict__ b3, const float *__restrict__ W4, const float *__restrict__ b4,
    const float *__restrict__ W1, const float *__restrict__ b1, float *__restrict__ Hn, int ldhn,
    float *__restrict__ PQ, float *__restrict__ Qkeep, int64_t n_hits)
{
    typedef float f4v __attribute__((ext_vector_type(4)));
    constexpr int C = Shape<F, D>::C, LDH = Shape<F, D>::LDH, RS = kBlock + 4;
    constexpr int RT = D / 16, KS = (C + 3) / 4, ROWS = 4 * KS > D ? 4 * KS : D;
    __shared__ __attribute__((aligned(16))) float lds[ROWS * RS];   // [k][hit]: one operand block at a time
    const int64_t n0 = xcd_block() * kBlock, n = n0 + threadIdx.x;
    const bool active = n < n_hits;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, r16 = lane & 15, g4 = lane >> 4;
    const int hcol = 64 * wv + r16;
    float hp[LDH];
#pragma unroll
    for (int k = 0; k < LDH; ++k) hp[k] = 0.0f;
    if (active) {
        const float4 *row = reinterpret_cast<const float4 *>(H + n * ldh);
#pragma unroll
        for (int v = 0; v < LDH / 4; ++v) {
            const float4 a = row[v];
            hp[4 * v] = a.x; hp[4 * v + 1] = a.y; hp[4 * v + 2] = a.z; hp[4 * v + 3] = a.w;
        }
    }
    // ---- q = tanh(W3 [mi | mo | h] + b3): three k-chunks of C rows            (model.py:120)
    f4v cq[RT][4];
#pragma unroll
    for (int it = 0; it < RT; ++it) {
        const f4v bias = *reinterpret_cast<const f4v *>(b3 + 16 * it + 4 * g4);
#pragma unroll
        for (int ht = 0; ht < 4; ++ht) cq[it][ht] = bias;
    }
#pragma unroll 1
    for (int chunk = 0; chunk < 3; ++chunk) {
        if (chunk) __syncthreads();
        {
            float v[LDH];
            if (chunk < 2) {
#pragma unroll
                for (int k = 0; k < LDH; ++k) v[k] = 0.0f;
                if (active) {
                    const float4 *row = reinterpret_cast<const float4 *>(M + n * 2 * LDH + chunk * LDH);
#pragma unroll
                    for (int u = 0; u < LDH / 4; ++u) {
                        const float4 a = row[u];
                        v[4 * u] = a.x; v[4 * u + 1] = a.y; v[4 * u + 2] = a.z; v[4 * u + 3] = a.w;
                    }
                }
            } else {
#pragma unroll
                for (int k = 0; k < LDH; ++k) v[k] = hp[k];
            }
#pragma unroll
            for (int k = 0; k < 4 * KS; ++k) lds[k * RS + threadIdx.x] = k < C ? v[k < LDH ? k : 0] : 0.0f;
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < RT; ++it) {
            const int i = 16 * it + r16;
#pragma unroll 4
            for (int st = 0; st < KS; ++st) {
                const int kk = 4 * st + g4;
                const float aw = kk < C ? W3[i * 3 * C + chunk * C + kk] : 0.0f;
                const float *bv = lds + kk * RS + hcol;
#pragma unroll
                for (int ht = 0; ht < 4; ++ht) cq[it][ht] = __builtin_amdgcn_mfma_f32_16x16x4f32(aw, bv[16 * ht], cq[it][ht], 0, 0, 0);
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < RT; ++it)
#pragma unroll
        for (int ht = 0; ht < 4; ++ht) {
            const f4v q = {tanh_f(cq[it][ht].x), tanh_f(cq[it][ht].y), tanh_f(cq[it][ht].z), tanh_f(cq[it][ht].w)};
            float *x = lds + (16 * it + 4 * g4) * RS + hcol + 16 * ht;
            x[0] = q.x; x[RS] = q.y; x[2 * RS] = q.z; x[3 * RS] = q.w;
            const int64_t nn = n0 + hcol + 16 * ht;
            if (Qkeep && nn < n_hits) *reinterpret_cast<f4v *>(Qkeep + nn * D + 16 * it + 4 * g4) = q;
        }
    __syncthreads();
    // ---- H' = tanh(W4 q + b4)                                                   (model.py:94-98)
    f4v hn[RT][4];
#pragma unroll
    for (int it = 0; it < RT; ++it) {
        const f4v bias = *reinterpret_cast<const f4v *>(b4 + 16 * it + 4 * g4);
#pragma unroll
        for (int ht = 0; ht < 4; ++ht) hn[it][ht] = bias;
        const int i = 16 * it + r16;
#pragma unroll 4
        for (int st = 0; st < D / 4; ++st) {
            const float aw = W4[i * D + 4 * st + g4];
            const float *bv = lds + (4 * st + g4) * RS + hcol;
#pragma unroll
            for (int ht = 0; ht < 4; ++ht) hn[it][ht] = __builtin_amdgcn_mfma_f32_16x16x4f32(aw, bv[16 * ht], hn[it][ht], 0, 0, 0);
        }
#pragma unroll
        for (int ht = 0; ht < 4; ++ht) {
            hn[it][ht] = f4v{tanh_f(hn[it][ht].x), tanh_f(hn[it][ht].y), tanh_f(hn[it][ht].z), tanh_f(hn[it][ht].w)};
            const int64_t nn = n0 + hcol + 16 * ht;
            if (nn < n_hits) *reinterpret_cast<f4v *>(Hn + nn * ldhn + 16 * it + 4 * g4) = hn[it][ht];
        }
    }
    if (active) {                                        // skip concat of X (and the row's padding), model.py:154
#pragma unroll
        for (int v = D / 4; v < LDH / 4; ++v)
            reinterpret_cast<float4 *>(Hn + n * ldhn)[v] = make_float4(hp[4 * v], hp[4 * v + 1], hp[4 * v + 2], hp[4 * v + 3]);
    }
    if (!PQ) return;                                     // (kernel-uniform)
    // ---- P = W1a [H' | x] + b1, Q = W1b [H' | x] for the next edge pass
    __syncthreads();
#pragma unroll
    for (int it = 0; it < RT; ++it)
#pragma unroll
        for (int ht = 0; ht < 4; ++ht) {
            float *x = lds + (16 * it + 4 * g4) * RS + hcol + 16 * ht;
            x[0] = hn[it][ht].x; x[RS] = hn[it][ht].y; x[2 * RS] = hn[it][ht].z; x[3 * RS] = hn[it][ht].w;
        }
#pragma unroll
    for (int k = D; k < 4 * KS; ++k) lds[k * RS + threadIdx.x] = k < C ? hp[k < LDH ? k : 0] : 0.0f;
    __syncthreads();
#pragma unroll 1
    for (int it = 0; it < RT; ++it) {
        const int i = 16 * it + r16;
        f4v cp[4], cqq[4];
        const f4v bias = *reinterpret_cast<const f4v *>(b1 + 16 * it + 4 * g4);
#pragma unroll
        for (int ht = 0; ht < 4; ++ht) { cp[ht] = bias; cqq[ht] = f4v{0.0f, 0.0f, 0.0f, 0.0f}; }
#pragma unroll 4
        for (int st = 0; st < KS; ++st) {
            const int kk = 4 * st + g4;
            const float ap = kk < C ? W1[i * 2 * C + kk] : 0.0f, aq = kk < C ? W1[i * 2 * C + C + kk] : 0.0f;
            const float *bv = lds + kk * RS + hcol;
#pragma unroll
            for (int ht = 0; ht < 4; ++ht) {
                const float bb = bv[16 * ht];
                cp[ht] = __builtin_amdgcn_mfma_f32_16x16x4f32(ap, bb, cp[ht], 0, 0, 0);
                cqq[ht] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq, bb, cqq[ht], 0, 0, 0);
            }
        }
#pragma unroll
        for (int ht = 0; ht < 4; ++ht) {
            const int64_t nn = n0 + hcol + 16 * ht;
            if (nn < n_hits) {
                *reinterpret_cast<f4v *>(PQ + nn * 2 * D + 16 * it + 4 * g4) = cp[ht];
                *reinterpret_cast<f4v *>(PQ + nn * 2 * D + D + 16 * it + 4 * g4) = cqq[ht];
            }
        }
    }
}

// P / Q rows of 256 hits from their H rows, wide hidden layers: the same products as k_node_mlpW's last
// stage (exact fp32, k ascending from the bias), H staged transposed with coalesced loads.  k_input /
// k_pq feed these products scalar weight operands: 0.13 ms for 50 k hits at D = 64.
template <int F, int D>
__global__ __launch_bounds__(kBlock) void k_pq_mlpW(const float *__restrict__ H, int ldh, const float *__restrict__ W1,
                                                    const float *__restrict__ b1, float *__restrict__ PQ, int64_t n_hits)
{
    typedef float f4v __attribute__((ext_vector_type(4)));
    constexpr int C = Shape<F, D>::C, LDH = Shape<F, D>::LDH, RS = kBlock + 4;
    constexpr int RT = D / 16, KS = (C + 3) / 4;
    static_assert(LDH == 4 * KS, "a padded H row is the k-steps of the products");
    __shared__ __attribute__((aligned(16))) float lds[LDH * RS];
    const int64_t n0 = xcd_block() * kBlock;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, r16 = lane & 15, g4 = lane >> 4;
    const int hcol = 64 * wv + r16;
#pragma unroll 4
    for (int j = threadIdx.x; j < kBlock * (LDH / 4); j += kBlock) {      // consecutive threads, consecutive 16-byte pieces
        const int hit = j / (LDH / 4), c = j % (LDH / 4);
        float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (n0 + hit < n_hits) v = *reinterpret_cast<const float4 *>(H + (n0 + hit) * ldh + 4 * c);
        float *d = lds + (4 * c) * RS + hit;
        d[0] = v.x; d[RS] = v.y; d[2 * RS] = v.z; d[3 * RS] = v.w;
    }
    __syncthreads();
#pragma unroll 1
    for (int it = 0; it < RT; ++it) {
        const int i = 16 * it + r16;
        f4v cp[4], cq[4];
        const f4v bias = *reinterpret_cast<const f4v *>(b1 + 16 * it + 4 * g4);
#pragma unroll
        for (int ht = 0; ht < 4; ++ht) { cp[ht] = bias; cq[ht] = f4v{0.0f, 0.0f, 0.0f, 0.0f}; }
#pragma unroll 4
        for (int st = 0; st < KS; ++st) {
            const int kk = 4 * st + g4;
            const float ap = kk < C ? W1[i * 2 * C + kk] : 0.0f, aq = kk < C ? W1[i * 2 * C + C + kk] : 0.0f;
            const float *bv = lds + kk * RS + hcol;
#pragma unroll
            for (int ht = 0; ht < 4; ++ht) {
                const float bb = bv[16 * ht];
                cp[ht] = __builtin_amdgcn_mfma_f32_16x16x4f32(ap, bb, cp[ht], 0, 0, 0);
                cq[ht] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq, bb, cq[ht], 0, 0, 0);
            }
        }
#pragma unroll
        for (int ht = 0; ht < 4; ++ht) {
            const int64_t nn = n0 + hcol + 16 * ht;
            if (nn < n_hits) {
                *reinterpret_cast<f4v *>(PQ + nn * 2 * D + 16 * it + 4 * g4) = cp[ht];
                *reinterpret_cast<f4v *>(PQ + nn * 2 * D + D + 16 * it + 4 * g4) = cq[ht];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// small events: the WHOLE forward of one graph in one workgroup, one launch for the batch
// ---------------------------------------------------------------------------------------------
// The per-module kernels above cost one launch per pass (2T + 2 launches, ~5 us each on the
// stream): for the reference's muon graphs (tens of hits, gnn/prepareMuonGraphs.py) that is all
// of the time.  Here a workgroup owns one graph of the block-diagonal batch: H (two buffers),
// the P/Q rows and the edge scores live in LDS for all T iterations, hits and segments are dealt
// to the lanes, a barrier separates the passes.  Same arithmetic, in the same order, as
// k_input / k_edge / k_node (the parity tests compare them bit for bit).
// Requires every segment of graph i to join hits of graph i and to lie in
// [seg_ptr[i], seg_ptr[i+1]) (HitGraphBatch.event_layout checks this on the host).
// sizes / LDS offsets (floats, 16-byte aligned) of the ten weight tensors in state_dict order
template <int F, int D>
struct EvWeights {
    static constexpr int C = F + D;
    static constexpr int size(int a)
    {
        return a == 0 ? D * F : a == 2 ? D * 2 * C : a == 5 ? 1 : a == 6 ? D * 3 * C : a == 8 ? D * D : D;
    }
    static constexpr int off(int a)
    {
        int o = 0;
        for (int i = 0; i < a; ++i) o += (size(i) + 3) & ~3;
        return o;
    }
    static constexpr int total = off(10);
};

struct EvW {
    const float *__restrict__ Win, *__restrict__ bin, *__restrict__ W1, *__restrict__ b1,
        *__restrict__ W2, *__restrict__ b2, *__restrict__ W3, *__restrict__ b3, *__restrict__ W4,
        *__restrict__ b4;
};

template <int D>
struct EvCfg {
    // wavefronts per graph: the output rows of every layer are dealt to them
    static constexpr int NWV = D >= 32 ? 16 : D >= 8 ? 8 : 4;
    static constexpr int NT = 64 * NWV;
};

template <int F, int D>
__global__ __launch_bounds__((EvCfg<D>::NT)) void k_event(
    gnn_graph_t g, const float *__restrict__ Win, const float *__restrict__ bin,
    const float *__restrict__ W1, const float *__restrict__ b1, const float *__restrict__ W2,
    const float *__restrict__ b2, const float *__restrict__ W3, const float *__restrict__ b3,
    const float *__restrict__ W4, const float *__restrict__ b4,
    const int32_t *__restrict__ hit_ptr, const int32_t *__restrict__ seg_ptr, int n_iters,
    float *__restrict__ e_out, int cap_hits, int cap_segments, float *__restrict__ e_all,
    float *__restrict__ H_all)
{   // e_all / H_all (training forward): every pass's scores [T+1, E] and hit rows [T+1, N, LDH]
    constexpr int C = Shape<F, D>::C;
    constexpr int LDH = Shape<F, D>::LDH;
    constexpr int NT = EvCfg<D>::NT, NWV = EvCfg<D>::NWV;
    constexpr int RW = D / NWV;          // output rows of a D-row layer per wavefront
    constexpr int RP = 2 * D / NWV;      // rows of the stacked [P; Q] per wavefront
    static_assert(D % NWV == 0 && RW >= 1, "rows are dealt to the wavefronts");
    extern __shared__ __attribute__((aligned(16))) float ev_lds[];
    float *H = ev_lds, *Hn = H + cap_hits * LDH, *PQ = Hn + cap_hits * LDH,
          *qb = PQ + cap_hits * 2 * D, *Mb = qb + cap_hits * D, *es = Mb + cap_hits * 2 * LDH;
    // (hit_ptr == NULL: the batch is ONE graph - a never-seen single event needs no offset arrays on the device)
    const int h0 = hit_ptr ? hit_ptr[blockIdx.x] : 0, nh = hit_ptr ? hit_ptr[blockIdx.x + 1] - h0 : (int)g.n_hits;
    const int s0 = hit_ptr ? seg_ptr[blockIdx.x] : 0, ns = hit_ptr ? seg_ptr[blockIdx.x + 1] - s0 : (int)g.n_segments;
    const float *__restrict__ X = g.X;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // The weights (953 floats at F = 11, D = 8) are copied to LDS once and read from there with
    // broadcast ds_reads: a graph is a handful of wavefronts, nothing hides scalar-load latency
    // (measured 38 us per launch on the s_load path).  A hit is a lane; the output rows of every
    // layer are dealt to the wavefronts (each reads its share of the weights and runs its share
    // of the FMAs), layers meet through LDS.  Inputs stream through in 4-float pieces, so the
    // register footprint does not grow with D.  Per row the arithmetic and its order are those of
    // k_input / k_node, so the scores are bit-identical to the per-module kernels.
    using W = EvWeights<F, D>;
    float *wl = es + ((cap_segments + 3) & ~3);
    {
        const float *srcs[10] = {Win, bin, W1, b1, W2, b2, W3, b3, W4, b4};
#pragma unroll
        for (int a = 0; a < 10; ++a)
            for (int i = threadIdx.x; i < W::size(a); i += NT) wl[W::off(a) + i] = srcs[a][i];
    }
    const EvW p = {wl + W::off(0), wl + W::off(1), wl + W::off(2), wl + W::off(3), wl + W::off(4),
                   wl + W::off(5), wl + W::off(6), wl + W::off(7), wl + W::off(8), wl + W::off(9)};
    // The graph's index arrays too, as LOCAL ids: the pull loops below chase eid -> score and
    // nbr -> feature row per list entry, and from global memory every hop is a dependent
    // round trip.  One coalesced pass here instead.
    int32_t *ip = reinterpret_cast<int32_t *>(wl + W::total), *op = ip + cap_hits + 1,
            *ie = op + cap_hits + 1, *inb = ie + cap_segments, *oe = inb + cap_segments,
            *onb = oe + cap_segments, *sl = onb + cap_segments, *dl = sl + cap_segments;
    const bool raw = g.in_ptr == nullptr;            // no segment lists from the caller: built below, in LDS
    if (nh > 0 && !raw) {
        const int ib = g.in_ptr[h0], ob = g.out_ptr[h0];
        for (int n = threadIdx.x; n <= nh; n += NT) {
            ip[n] = g.in_ptr[h0 + n] - ib;
            op[n] = g.out_ptr[h0 + n] - ob;
        }
        const int ni = g.in_ptr[h0 + nh] - ib, no = g.out_ptr[h0 + nh] - ob;
        for (int k = threadIdx.x; k < ni; k += NT) {
            ie[k] = g.in_eid[ib + k] - s0;
            inb[k] = g.in_nbr[ib + k] - h0;
        }
        for (int k = threadIdx.x; k < no; k += NT) {
            oe[k] = g.out_eid[ob + k] - s0;
            onb[k] = g.out_nbr[ob + k] - h0;
        }
    }
    for (int j = threadIdx.x; j < ns; j += NT) {
        const int sg = g.src[s0 + j];
        sl[j] = sg < 0 ? -1 : sg - h0;
        dl[j] = sg < 0 ? -1 : g.dst[s0 + j] - h0;
    }
    if (raw) {
        // The two lists of a graph nobody has seen before (trigger-style use, gnn/Inference.ipynb cell 3: one event
        // in, scores out), from its (src, dst) alone - what gnn_csr_build does with four launches, here inside the
        // one launch: counts (LDS atomics) -> scan -> slots claimed in arrival order -> every entry's final slot
        // = list base + the number of ids in its list below its own.  Ascending ids whatever the arrival order: the
        // same lists as the caller-built ones, so the scores stay bit-identical to them (and to every other run).
        int *cur_i = reinterpret_cast<int *>(Mb), *cur_o = cur_i + cap_hits, *tmp = reinterpret_cast<int *>(es);
        for (int n = threadIdx.x; n <= nh; n += NT) ip[n] = op[n] = 0;
        __syncthreads();
        for (int j = threadIdx.x; j < ns; j += NT)
            if (sl[j] >= 0) {
                atomicAdd(&ip[dl[j] + 1], 1);
                atomicAdd(&op[sl[j] + 1], 1);
            }
        __syncthreads();
        if (wv < 2) {                                // wavefront 0 scans the in-counts, wavefront 1 the out-counts
            int *ptr = wv ? op : ip;
            const int per = (nh + 64) / 64, b = 1 + lane * per, e = min(b + per, nh + 1);
            int sum = 0;
            for (int n = b; n < e; ++n) sum += ptr[n];
            int incl = sum;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int u = __shfl_up(incl, o, 64);
                if (lane >= o) incl += u;
            }
            int run = incl - sum;
            for (int n = b; n < e; ++n) {
                run += ptr[n];
                ptr[n] = run;
            }
        }
        __syncthreads();
        for (int n = threadIdx.x; n < nh; n += NT) { cur_i[n] = ip[n]; cur_o[n] = op[n]; }
        __syncthreads();
        for (int j = threadIdx.x; j < ns; j += NT)
            if (sl[j] >= 0) tmp[atomicAdd(&cur_i[dl[j]], 1)] = j;
        __syncthreads();
        for (int j = threadIdx.x; j < ns; j += NT)
            if (sl[j] >= 0) {
                const int b = ip[dl[j]], e = ip[dl[j] + 1];
                int r = 0;
                for (int k = b; k < e; ++k) r += tmp[k] < j;
                ie[b + r] = j;
                inb[b + r] = sl[j];
            }
        __syncthreads();
        for (int j = threadIdx.x; j < ns; j += NT)
            if (sl[j] >= 0) tmp[atomicAdd(&cur_o[sl[j]], 1)] = j;
        __syncthreads();
        for (int j = threadIdx.x; j < ns; j += NT)
            if (sl[j] >= 0) {
                const int b = op[sl[j]], e = op[sl[j] + 1];
                int r = 0;
                for (int k = b; k < e; ++k) r += tmp[k] < j;
                oe[b + r] = j;
                onb[b + r] = dl[j];
            }
    }
    // input rows: X into the skip columns of H (k_input), zero padding
    for (int i = threadIdx.x; i < nh * (LDH - D); i += NT) {
        const int n = i / (LDH - D), k = i % (LDH - D);
        H[n * LDH + D + k] = k < F ? X[(int64_t)(h0 + n) * F + k] : 0.0f;
    }
    __syncthreads();

    // acc[r] += sum_j w[r][4c + j] * v[j] over one 4-float piece of the input (k ascending)
    auto fma_piece = [&](float *acc, const float *wrow0, int ldw, int c, const float4 v, auto rows) {
        constexpr int R = decltype(rows)::value;
        const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (4 * c + j < C)
#pragma unroll
                for (int r = 0; r < R; ++r) acc[r] = fmaf(wrow0[r * ldw + 4 * c + j], vv[j], acc[r]);
    };

    // rows [wv*RP, (wv+1)*RP) of [P; Q] for every hit, from the full feature rows in `Hsrc`
    auto pq_rows = [&](const float *Hsrc) {
        const int row0 = wv * RP, d0 = row0 % D;
        const bool isq = row0 >= D;                          // wave-uniform (RP divides D)
        for (int n = lane; n < nh; n += 64) {
            float acc[RP];
#pragma unroll
            for (int r = 0; r < RP; ++r) acc[r] = isq ? 0.0f : p.b1[d0 + r];
#pragma unroll
            for (int c = 0; c < LDH / 4; ++c)
                fma_piece(acc, p.W1 + d0 * 2 * C + (isq ? C : 0), 2 * C, c,
                          *reinterpret_cast<const float4 *>(Hsrc + n * LDH + 4 * c),
                          std::integral_constant<int, RP>{});
#pragma unroll
            for (int r = 0; r < RP; ++r) PQ[n * 2 * D + row0 + r] = acc[r];
        }
    };

    // input network (k_input): rows of tanh(Win x + bin) per wavefront
    for (int n = lane; n < nh; n += 64) {
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            const int d = wv * RW + r;
            float acc = p.bin[d];
#pragma unroll
            for (int k = 0; k < F; ++k) acc = fmaf(p.Win[d * F + k], H[n * LDH + D + k], acc);
            H[n * LDH + d] = tanh_f(acc);
        }
    }
    __syncthreads();
    auto keep_rows = [&](const float *Hsrc, int t) {
        if (H_all)
            for (int i = threadIdx.x; i < nh * LDH; i += NT)
                H_all[((int64_t)t * g.n_hits + h0) * LDH + i] = Hsrc[i];
    };
    keep_rows(H, 0);
    pq_rows(H);
    __syncthreads();

    for (int t = 0;; ++t) {
        const bool last = (t == n_iters);
        // edge pass (k_edge): one segment per lane of the workgroup
        for (int j = threadIdx.x; j < ns; j += NT) {
            const int s = sl[j], d = dl[j];
            float acc = p.b2[0];
#pragma unroll
            for (int v = 0; v < D / 4; ++v) {
                float4 z;
                if (s >= 0) {
                    const float4 a = reinterpret_cast<const float4 *>(PQ + s * 2 * D)[v];
                    const float4 b = reinterpret_cast<const float4 *>(PQ + d * 2 * D + D)[v];
                    z = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
                } else {
                    z = make_float4(p.b1[4 * v], p.b1[4 * v + 1], p.b1[4 * v + 2], p.b1[4 * v + 3]);
                }
                acc = fmaf(p.W2[4 * v], tanh_f(z.x), acc);
                acc = fmaf(p.W2[4 * v + 1], tanh_f(z.y), acc);
                acc = fmaf(p.W2[4 * v + 2], tanh_f(z.z), acc);
                acc = fmaf(p.W2[4 * v + 3], tanh_f(z.w), acc);
            }
            const float e = sigmoid_f(acc);
            if (e_all) e_all[(int64_t)t * g.n_segments + s0 + j] = e;
            if (!last)
                es[j] = e;
            else if (e_out)
                e_out[s0 + j] = e;
        }
        if (last) break;
        __syncthreads();
        // node pass (k_node).  Step 1: the segment sums mi, mo of every hit, one (hit, part, 4-float
        // piece) per thread, into LDS (pull over the local CSR, ascending segment id).
        for (int i = threadIdx.x; i < nh * 2 * (LDH / 4); i += NT) {
            const int n = i / (2 * (LDH / 4)), part = (i / (LDH / 4)) & 1, c = i % (LDH / 4);
            const int32_t *el = part ? oe : ie, *nl = part ? onb : inb;
            float4 m = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            for (int k = part ? op[n] : ip[n], k1 = part ? op[n + 1] : ip[n + 1]; k < k1; ++k) {
                const float w = es[el[k]];
                const float4 a = *reinterpret_cast<const float4 *>(H + nl[k] * LDH + 4 * c);
                m.x = fmaf(w, a.x, m.x);
                m.y = fmaf(w, a.y, m.y);
                m.z = fmaf(w, a.z, m.z);
                m.w = fmaf(w, a.w, m.w);
            }
            *reinterpret_cast<float4 *>(Mb + n * 2 * LDH + part * LDH + 4 * c) = m;
        }
        __syncthreads();
        // Step 2: first layer, rows per wavefront; M = [mi | mo | h] streams through in pieces
        for (int n = lane; n < nh; n += 64) {
            const int d0 = wv * RW;
            float acc[RW];
#pragma unroll
            for (int r = 0; r < RW; ++r) acc[r] = p.b3[d0 + r];
#pragma unroll 1
            for (int part = 0; part < 3; ++part) {
                const float *mrow = part == 2 ? H + n * LDH : Mb + n * 2 * LDH + part * LDH;
#pragma unroll
                for (int c = 0; c < LDH / 4; ++c)
                    fma_piece(acc, p.W3 + d0 * 3 * C + part * C, 3 * C, c,
                              *reinterpret_cast<const float4 *>(mrow + 4 * c), std::integral_constant<int, RW>{});
            }
#pragma unroll
            for (int r = 0; r < RW; ++r) qb[n * D + d0 + r] = tanh_f(acc[r]);
        }
        for (int i = threadIdx.x; i < nh * (LDH - D); i += NT) {   // skip concat of X (model.py:154)
            const int n = i / (LDH - D), k = D + i % (LDH - D);
            Hn[n * LDH + k] = H[n * LDH + k];
        }
        __syncthreads();
        for (int n = lane; n < nh; n += 64) {                // second layer: rows per wavefront
            const int d0 = wv * RW;
            float acc[RW];
#pragma unroll
            for (int r = 0; r < RW; ++r) acc[r] = p.b4[d0 + r];
#pragma unroll
            for (int c = 0; c < D / 4; ++c) {
                const float4 v = *reinterpret_cast<const float4 *>(qb + n * D + 4 * c);
                const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int r = 0; r < RW; ++r) acc[r] = fmaf(p.W4[(d0 + r) * D + 4 * c + j], vv[j], acc[r]);
            }
#pragma unroll
            for (int r = 0; r < RW; ++r) Hn[n * LDH + d0 + r] = tanh_f(acc[r]);
        }
        __syncthreads();
        keep_rows(Hn, t + 1);
        pq_rows(Hn);                                         // nobody reads PQ or H in this step
        __syncthreads();
        float *tmp = H; H = Hn; Hn = tmp;
    }
}

constexpr size_t kEventLdsMax = 128 * 1024;     // of the CU's 160 KB

inline size_t event_lds_bytes(int F, int D, int64_t cap_hits, int64_t cap_segments)
{
    const int ldh = (F + D + 3) & ~3, C = F + D;
    auto a4 = [](int x) { return (x + 3) & ~3; };
    const int wfl = a4(D * F) + a4(D * 2 * C) + a4(1) + a4(D * 3 * C) + a4(D * D) + 5 * a4(D);   // EvWeights::total
    return (size_t)(cap_hits * (4 * ldh + 3 * D) + ((cap_segments + 3) & ~3) + wfl +
                    2 * (cap_hits + 1) + 6 * cap_segments) * sizeof(float);
}

template <int F, int D>
int run_events(const gnn_graph_t *g, const gnn_params_t *p, const int32_t *hit_ptr,
               const int32_t *seg_ptr, int64_t n_graphs, int cap_hits, int cap_segments,
               int n_iters, float *e_out, hipStream_t s, float *e_all = nullptr,
               float *H_all = nullptr)
{
    if (n_graphs <= 0) return 0;
    const size_t lds = event_lds_bytes(F, D, cap_hits, cap_segments);
    static DevOnce attr_done;     // dynamic LDS above 64 KB must be opted into, once per device
    if (attr_done.need())
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_event<F, D>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)kEventLdsMax);
    GNN_LAUNCH_SH("k_event", (k_event<F, D>), (unsigned)n_graphs, EvCfg<D>::NT, lds, s, *g, p->Win, p->bin, p->W1,
                  p->b1, p->W2, p->b2, p->W3, p->b3, p->W4, p->b4, hit_ptr, seg_ptr, n_iters, e_out,
                  cap_hits, cap_segments, e_all, H_all);
    return 0;
}

// bound of |P'|, |Q'| over all hits given |H'| <= 1 and per-feature max |X| (one wavefront)
__global__ __launch_bounds__(64) void k_exp_bound(const float *__restrict__ W1,
                                                 const float *__restrict__ b1,
                                                 const float *__restrict__ xmax, int F, int D,
                                                 float *__restrict__ out)
{
    const int C = F + D;
    float worst = 0.0f;
    for (int i = threadIdx.x; i < 2 * D; i += 64) {       // rows of the P block, then the Q block
        const int row = i % D, half = i / D;
        float acc = half == 0 ? fabsf(b1[row]) : 0.0f;
        for (int k = 0; k < C; ++k) {
            const float w = fabsf(W1[row * 2 * C + half * C + k]);
            acc += k < D ? w : w * xmax[k - D];
        }
        worst = fmaxf(worst, acc);
    }
    for (int o = 32; o > 0; o >>= 1) worst = fmaxf(worst, __shfl_xor(worst, o));
    if (threadIdx.x == 0) out[0] = 2.8853900817779268f * worst;
}

// H (padded rows) -> unpadded [n_hits, C] trace rows (parity tests only).
__global__ __launch_bounds__(kBlock) void k_unpad(const float *__restrict__ H, int ldh, int C,
                                                  float *__restrict__ out, int64_t total)
{
    const int64_t i = xcd_block() * kBlock + threadIdx.x;
    if (i >= total) return;
    out[i] = H[(i / C) * ldh + (i % C)];
}

// ---------------------------------------------------------------------------------------------
// shape dispatch
// ---------------------------------------------------------------------------------------------
#define GNN_FOR_EACH_SHAPE(X_) \
    X_(2, 4) X_(2, 8) X_(2, 16) X_(2, 32) X_(3, 4) X_(3, 8) X_(3, 16) X_(3, 32) X_(3, 64) X_(11, 4) X_(11, 8) \
    X_(11, 16)

template <int F, int D>
int run_input(const float *X, const float *Win, const float *bin, const float *W1,
              const float *b1, float *H, int ldh, float *PQ, int64_t n, hipStream_t s)
{
    if (n <= 0) return 0;
    if constexpr (D >= 32) {
        // wide hidden layers at detector size: the P / Q products on the matrix cores
        if (PQ && n >= kNodeWideMinHits && ldh == Shape<F, D>::LDH && !getenv("GNN_NODE_ONE_LANE")) {
            GNN_LAUNCH("k_input", (k_input<F, D>), grid_for(n), kBlock, s, X, Win, bin, W1, b1, H, ldh,
                       (float *)nullptr, n);
            GNN_LAUNCH("k_pq_mlpW", (k_pq_mlpW<F, D>), grid_for(n), kBlock, s, H, ldh, W1, b1, PQ, n);
            return 0;
        }
    }
    GNN_LAUNCH("k_input", (k_input<F, D>), grid_for(n), kBlock, s, X, Win, bin, W1, b1, H, ldh,
               PQ, n);
    return 0;
}

template <int F, int D>
int run_pq(const float *H, int ldh, const float *W1, const float *b1, float *PQ, int64_t n,
           hipStream_t s)
{
    if (n > 0) GNN_LAUNCH("k_pq", (k_pq<F, D>), grid_for(n), kBlock, s, H, ldh, W1, b1, PQ, n);
    return 0;
}

template <int D>
int run_edge(const int32_t *src, const int32_t *dst, const float *PQ, const float *b1,
             const float *W2, const float *b2, float *e, int64_t n_seg, hipStream_t s)
{
    if (n_seg > 0)
        GNN_LAUNCH("k_edge", (k_edge<D>), grid_for(n_seg), kBlock, s, src, dst, PQ, b1, W2, b2, e,
                   n_seg);
    return 0;
}

template <int F, int D>
int run_node(const float *H, int ldh, const float *e, const gnn_graph_t *g, const float *W3,
             const float *b3, const float *W4, const float *b4, const float *W1, const float *b1,
             float *Hn, int ldhn, float *PQ, hipStream_t s, float *Qkeep = nullptr, float *Mbuf = nullptr)
{
    if (g->n_hits <= 0) return 0;
    if constexpr (D >= 32) {
        // wide hidden layers at detector size: 16-lane list walk + matrix-core MLP (Mbuf: [n_hits, 2 ldh] scratch)
        if (Mbuf && g->n_hits >= kNodeWideMinHits && ldh == Shape<F, D>::LDH && !getenv("GNN_NODE_ONE_LANE")) {
            GNN_LAUNCH("k_node_walkW", (k_node_walkW<F, D>), grid_walk(g->n_hits), kWalkBlock, s, H, ldh, e, g->in_ptr,
                       g->in_eid, g->in_nbr, g->out_ptr, g->out_eid, g->out_nbr, Mbuf, g->n_hits);
            GNN_LAUNCH("k_node_mlpW", (k_node_mlpW<F, D>), grid_for(g->n_hits), kBlock, s, H, ldh, Mbuf, W3, b3, W4, b4,
                       W1, b1, Hn, ldhn, PQ, Qkeep, g->n_hits);
            return 0;
        }
    }
    GNN_LAUNCH("k_node", (k_node<F, D>), grid_for(g->n_hits), kBlock, s, H, ldh, e, g->in_ptr,
               g->in_eid, g->in_nbr, g->out_ptr, g->out_eid, g->out_nbr, W3, b3, W4, b4, W1, b1,
               Hn, ldhn, PQ, Qkeep, g->n_hits);
    return 0;
}

struct Workspace {
    float *Ha, *Hb, *PQ, *e, *M;
    size_t bytes;
};

Workspace carve(void *base, int64_t n_hits, int64_t n_seg, int ldh, int D)
{
    Workspace w;
    size_t off = 0;
    char *b = static_cast<char *>(base);
    const size_t hbytes = align256((size_t)n_hits * ldh * sizeof(float));
    w.Ha = reinterpret_cast<float *>(b + off); off += hbytes;
    w.Hb = reinterpret_cast<float *>(b + off); off += hbytes;
    w.PQ = reinterpret_cast<float *>(b + off); off += align256((size_t)n_hits * 2 * D * sizeof(float));
    w.e = reinterpret_cast<float *>(b + off);  off += align256((size_t)n_seg * sizeof(float));
    w.M = reinterpret_cast<float *>(b + off);             // wide shapes: [mi | mo] between k_node_walkW and k_node_mlpW
    off += align256(D >= 32 ? (size_t)n_hits * 2 * ldh * sizeof(float) : 0);
    w.bytes = off;
    return w;
}

template <int F, int D>
int forward_impl(const gnn_graph_t *g, const gnn_params_t *p, int n_iters, float *e_out,
                 float *e_trace, float *H_trace, void *ws, hipStream_t s, float *H_all = nullptr,
                 float *Q_all = nullptr)
{
    constexpr int C = Shape<F, D>::C;
    constexpr int LDH = Shape<F, D>::LDH;
    const int64_t N = g->n_hits, E = g->n_segments;
    Workspace w = carve(ws, N, E, LDH, D);
    // training forward: every iteration's H is kept (H_all), no ping-pong
    float *H = H_all ? H_all : w.Ha, *Hn = H_all ? H_all + (size_t)N * LDH : w.Hb;
    int rc = run_input<F, D>(g->X, p->Win, p->bin, p->W1, p->b1, H, LDH, w.PQ, N, s);
    if (rc) return rc;
    for (int t = 0; t <= n_iters; ++t) {
        if (H_trace && N > 0)
            GNN_LAUNCH("k_unpad", k_unpad, grid_for(N * C), kBlock, s, H, LDH, C,
                       H_trace + (size_t)t * N * C, N * C);
        const bool last = (t == n_iters);
        float *e_t = e_trace ? e_trace + (size_t)t * E : (last ? e_out : w.e);
        rc = run_edge<D>(g->src, g->dst, w.PQ, p->b1, p->W2, p->b2, e_t, E, s);
        if (rc) return rc;
        if (last) {
            if (e_trace && E > 0 && e_out != e_t) {
                hipError_t err = hipMemcpyAsync(e_out, e_t, (size_t)E * sizeof(float),
                                                hipMemcpyDeviceToDevice, s);
                if (err != hipSuccess) return fail(-(int)err, "copy of final scores failed");
            }
            break;
        }
        rc = run_node<F, D>(H, LDH, e_t, g, p->W3, p->b3, p->W4, p->b4, p->W1, p->b1, Hn, LDH,
                            w.PQ, s, Q_all ? Q_all + (size_t)t * N * D : nullptr, w.M);
        if (rc) return rc;
        if (H_all) {
            H = Hn;
            Hn = H + (size_t)N * LDH;
        } else {
            float *tmp = H; H = Hn; Hn = tmp;
        }
    }
    return 0;
}

}  // namespace

using namespace gnn;

// ---------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------
extern "C" {

int gnn_abi_version(void) { return GNN_ABI_VERSION; }

const char *gnn_last_error(void) { return g_err; }

int gnn_shape_supported(int32_t F, int32_t D)
{
#define X_(F_, D_) if (F == F_ && D == D_) return 1;
    GNN_FOR_EACH_SHAPE(X_)
#undef X_
    return 0;
}

int32_t gnn_h_stride(int32_t F, int32_t D)
{
    return gnn_shape_supported(F, D) ? ((F + D + 3) & ~3) : 0;
}

int gnn_events_supported(int32_t F, int32_t D, int64_t max_hits, int64_t max_segments)
{
    if (!gnn_shape_supported(F, D) || max_hits < 0 || max_segments < 0) return 0;
    return event_lds_bytes(F, D, max_hits, max_segments) <= kEventLdsMax;
}

int gnn_segclf_forward_events(const gnn_graph_t *g, const gnn_params_t *p, const int32_t *hit_ptr,
                              const int32_t *seg_ptr, int64_t n_graphs, int32_t max_hits,
                              int32_t max_segments, int32_t n_iters, float *e_out, void *stream)
{
    if (!g || !p || n_iters < 0 || n_graphs < 0 || max_hits < 0 || max_segments < 0)
        return fail(GNN_ERR_BADARG, "gnn_segclf_forward_events: bad argument");
    if (n_graphs > 1 && (!hit_ptr || !seg_ptr)) return fail(GNN_ERR_BADARG, "gnn_segclf_forward_events: graph offsets missing");
    if (n_graphs == 1 && !hit_ptr != !seg_ptr) return fail(GNN_ERR_BADARG, "gnn_segclf_forward_events: one of hit_ptr / seg_ptr missing");
    if (g->n_segments > 0 && (!e_out || !g->src || !g->dst)) return fail(GNN_ERR_BADARG, "gnn_segclf_forward_events: segment arrays missing");
    if (g->n_hits > 0 && !g->X) return fail(GNN_ERR_BADARG, "gnn_segclf_forward_events: hit features missing");
    // the six list arrays come together or not at all (NULL: the kernel builds them in LDS)
    if (g->in_ptr ? (!g->out_ptr || (g->n_segments > 0 && (!g->in_eid || !g->in_nbr || !g->out_eid || !g->out_nbr)))
                  : (g->out_ptr || g->in_eid || g->in_nbr || g->out_eid || g->out_nbr) != 0)
        return fail(GNN_ERR_BADARG, "gnn_segclf_forward_events: segment lists incomplete (all six arrays, or none)");
    if (!gnn_events_supported(p->F, p->D, max_hits, max_segments))
        return fail(GNN_ERR_UNSUPPORTED, "events of up to %d hits / %d segments do not fit one workgroup's LDS at input_dim=%d hidden_dim=%d",
                    max_hits, max_segments, p->F, p->D);
    hipStream_t s = static_cast<hipStream_t>(stream);
#define X_(F_, D_) if (p->F == F_ && p->D == D_) return run_events<F_, D_>(g, p, hit_ptr, seg_ptr, n_graphs, max_hits, max_segments, n_iters, e_out, s);
    GNN_FOR_EACH_SHAPE(X_)
#undef X_
    return fail(GNN_ERR_UNSUPPORTED, "no kernel for input_dim=%d hidden_dim=%d", p->F, p->D);
}

int gnn_input_fwd(const float *X, const float *Win, const float *bin, float *H, int64_t n_hits,
                  int32_t F, int32_t D, int32_t ldh, void *stream)
{
    if (!X || !Win || !bin || !H || n_hits < 0) return fail(GNN_ERR_BADARG, "gnn_input_fwd: bad argument");
    if (ldh < ((F + D + 3) & ~3) || (ldh & 3)) return fail(GNN_ERR_BADARG, "gnn_input_fwd: ldh must be a multiple of 4 >= C");
    hipStream_t s = static_cast<hipStream_t>(stream);
#define X_(F_, D_) if (F == F_ && D == D_) return run_input<F_, D_>(X, Win, bin, nullptr, nullptr, H, ldh, nullptr, n_hits, s);
    GNN_FOR_EACH_SHAPE(X_)
#undef X_
    return fail(GNN_ERR_UNSUPPORTED, "no kernel for input_dim=%d hidden_dim=%d", F, D);
}

int gnn_edge_fwd(const float *H, int32_t ldh, const int32_t *src, const int32_t *dst,
                 const float *W1, const float *b1, const float *W2, const float *b2, float *e,
                 float *pq_ws, int64_t n_hits, int64_t n_segments, int32_t F, int32_t D,
                 void *stream)
{
    if (!H || !src || !dst || !W1 || !b1 || !W2 || !b2 || !e || !pq_ws || n_hits < 0 || n_segments < 0)
        return fail(GNN_ERR_BADARG, "gnn_edge_fwd: bad argument");
    if (ldh < F + D) return fail(GNN_ERR_BADARG, "gnn_edge_fwd: ldh < C");
    hipStream_t s = static_cast<hipStream_t>(stream);
#define X_(F_, D_)                                                              \
    if (F == F_ && D == D_) {                                                   \
        int rc = run_pq<F_, D_>(H, ldh, W1, b1, pq_ws, n_hits, s);              \
        if (rc) return rc;                                                      \
        return run_edge<D_>(src, dst, pq_ws, b1, W2, b2, e, n_segments, s);     \
    }
    GNN_FOR_EACH_SHAPE(X_)
#undef X_
    return fail(GNN_ERR_UNSUPPORTED, "no kernel for input_dim=%d hidden_dim=%d", F, D);
}

int gnn_node_fwd(const float *H, int32_t ldh, const float *e, const gnn_graph_t *g,
                 const float *W3, const float *b3, const float *W4, const float *b4, float *Hnext,
                 int32_t F, int32_t D, void *stream)
{
    if (!H || !e || !g || !W3 || !b3 || !W4 || !b4 || !Hnext || g->n_hits < 0)
        return fail(GNN_ERR_BADARG, "gnn_node_fwd: bad argument");
    if (g->n_hits > 0 && (!g->in_ptr || !g->out_ptr)) return fail(GNN_ERR_BADARG, "gnn_node_fwd: CSR missing");
    if (ldh < ((F + D + 3) & ~3) || (ldh & 3)) return fail(GNN_ERR_BADARG, "gnn_node_fwd: ldh must be a multiple of 4 >= C");
    if (Hnext == H) return fail(GNN_ERR_BADARG, "gnn_node_fwd: Hnext must not alias H");
    hipStream_t s = static_cast<hipStream_t>(stream);
#define X_(F_, D_) if (F == F_ && D == D_) return run_node<F_, D_>(H, ldh, e, g, W3, b3, W4, b4, nullptr, nullptr, Hnext, ldh, nullptr, s);
    GNN_FOR_EACH_SHAPE(X_)
#undef X_
    return fail(GNN_ERR_UNSUPPORTED, "no kernel for input_dim=%d hidden_dim=%d", F, D);
}

size_t gnn_forward_workspace_bytes(int64_t n_hits, int64_t n_segments, int32_t F, int32_t D)
{
    if (n_hits < 0 || n_segments < 0 || F <= 0 || D <= 0) return 0;
    return carve(nullptr, n_hits, n_segments, (F + D + 3) & ~3, D).bytes + 256;
}

int gnn_segclf_forward(const gnn_graph_t *g, const gnn_params_t *p, int32_t n_iters, float *e_out,
                       float *e_trace, float *H_trace, void *workspace, size_t workspace_bytes,
                       void *stream)
{
    gnn::ProfChain chain_;
    if (!g || !p || n_iters < 0 || g->n_hits < 0 || g->n_segments < 0)
        return fail(GNN_ERR_BADARG, "gnn_segclf_forward: bad argument");
    if (g->n_segments > 0 && (!e_out || !g->src || !g->dst))
        return fail(GNN_ERR_BADARG, "gnn_segclf_forward: segment arrays missing");
    if (g->n_hits > 0 && (!g->X || !g->in_ptr || !g->out_ptr))
        return fail(GNN_ERR_BADARG, "gnn_segclf_forward: hit arrays missing");
    if (!p->Win || !p->bin || !p->W1 || !p->b1 || !p->W2 || !p->b2 || !p->W3 || !p->b3 || !p->W4 || !p->b4)
        return fail(GNN_ERR_BADARG, "gnn_segclf_forward: weight pointer missing");
    if (!gnn_shape_supported(p->F, p->D))
        return fail(GNN_ERR_UNSUPPORTED, "no kernel for input_dim=%d hidden_dim=%d", p->F, p->D);
    if (!workspace || workspace_bytes < gnn_forward_workspace_bytes(g->n_hits, g->n_segments, p->F, p->D))
        return fail(GNN_ERR_WORKSPACE, "workspace too small: need %zu bytes",
                    gnn_forward_workspace_bytes(g->n_hits, g->n_segments, p->F, p->D));
    // 256-byte align the carve base
    void *ws = reinterpret_cast<void *>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);
    hipStream_t s = static_cast<hipStream_t>(stream);
#define X_(F_, D_) if (p->F == F_ && p->D == D_) return forward_impl<F_, D_>(g, p, n_iters, e_out, e_trace, H_trace, ws, s);
    GNN_FOR_EACH_SHAPE(X_)
#undef X_
    return fail(GNN_ERR_UNSUPPORTED, "unreachable");
}

int gnn_segclf_forward_train(const gnn_graph_t *g, const gnn_params_t *p, int32_t n_iters,
                             float *e_all, float *H_all, float *Q_all, void *workspace,
                             size_t workspace_bytes, void *stream)
{
    gnn::ProfChain chain_;
    if (!g || !p || n_iters < 0 || g->n_hits < 0 || g->n_segments < 0)
        return fail(GNN_ERR_BADARG, "gnn_segclf_forward_train: bad argument");
    if ((g->n_segments > 0 && (!e_all || !g->src || !g->dst)) || (g->n_hits > 0 && (!H_all || !g->X || !g->in_ptr || !g->out_ptr)))
        return fail(GNN_ERR_BADARG, "gnn_segclf_forward_train: array missing");
    if (!gnn_shape_supported(p->F, p->D))
        return fail(GNN_ERR_UNSUPPORTED, "no kernel for input_dim=%d hidden_dim=%d", p->F, p->D);
    if (!workspace || workspace_bytes < gnn_forward_workspace_bytes(g->n_hits, g->n_segments, p->F, p->D))
        return fail(GNN_ERR_WORKSPACE, "workspace too small");
    void *ws = reinterpret_cast<void *>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);
    hipStream_t s = static_cast<hipStream_t>(stream);
    // e_trace = e_all makes every edge pass land in its row; the "final scores" copy goes to the
    // last row itself (no-op copy avoided by passing that row as e_out)
    float *last = e_all ? e_all + (size_t)n_iters * g->n_segments : nullptr;
#define X_(F_, D_) if (p->F == F_ && p->D == D_) return forward_impl<F_, D_>(g, p, n_iters, last, e_all, nullptr, ws, s, H_all, Q_all);
    GNN_FOR_EACH_SHAPE(X_)
#undef X_
    return fail(GNN_ERR_UNSUPPORTED, "unreachable");
}

size_t gnn_backward_workspace_bytes(int64_t n_hits, int64_t n_segments, int32_t F, int32_t D)
{
    if (n_hits < 0 || n_segments < 0 || F <= 0 || D <= 0) return 0;
    return backward_workspace_bytes(n_hits, n_segments, F, D);
}

int gnn_segclf_backward(const gnn_graph_t *g, const gnn_params_t *p, int32_t n_iters,
                        const float *e_all, const float *H_all, const float *Q_all,
                        const float *grad_out, const gnn_grads_t *gr, void *workspace,
                        size_t workspace_bytes, void *stream)
{
    gnn::ProfChain chain_;
    if (!g || !p || !gr || n_iters < 0 || g->n_hits < 0 || g->n_segments < 0 || !workspace)
        return fail(GNN_ERR_BADARG, "gnn_segclf_backward: bad argument");
    if ((g->n_segments > 0 && (!e_all || !grad_out)) || (g->n_hits > 0 && !H_all))
        return fail(GNN_ERR_BADARG, "gnn_segclf_backward: saved tensors missing");
    if (!gr->Win || !gr->bin || !gr->W1 || !gr->b1 || !gr->W2 || !gr->b2 || !gr->W3 || !gr->b3 || !gr->W4 || !gr->b4)
        return fail(GNN_ERR_BADARG, "gnn_segclf_backward: gradient pointer missing");
    return backward(g, p, n_iters, e_all, H_all, Q_all, grad_out, gr, workspace, workspace_bytes,
                    static_cast<hipStream_t>(stream));
}

// dense one-hot incidence matrices -> index form, where the matrices live (the reference's input
// contract: [B, N, E] float32 with one 1 per real column, all-zero padded columns;
// gnn/graph.py:28-35, gnn/trainSegmentClassifier.py:66-95).  One lane per (batch, column): the N
// rows of a column are read with the column index fastest across lanes (coalesced).  O(B N E)
// reads - the price of the dense contract, paid once per batch on the device instead of a PCIe
// round trip.  flags[0] |= 1: a column with more than one non-zero, |= 2: a column set in only one
// of Ri / Ro.
__global__ __launch_bounds__(256) void k_dense_to_index(const float *__restrict__ Ri, const float *__restrict__ Ro,
                                                        int64_t B, int64_t N, int64_t E, int32_t *__restrict__ src,
                                                        int32_t *__restrict__ dst, int32_t *flags)
{
    int bad = 0;
    for (int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x; c < B * E; c += (int64_t)gridDim.x * 256) {
        const int64_t b = c / E, j = c % E;
        const float *ri = Ri + b * N * E + j, *ro = Ro + b * N * E + j;
        int ci = 0, co = 0;
        int64_t di = -1, so = -1;
        for (int64_t n = 0; n < N; ++n) {
            if (ri[n * E] != 0.0f) { if (!ci) di = n; ++ci; }
            if (ro[n * E] != 0.0f) { if (!co) so = n; ++co; }
        }
        if (ci > 1 || co > 1) bad |= 1;
        if ((ci == 0) != (co == 0)) bad |= 2;
        const bool real = ci == 1 && co == 1;
        src[c] = real ? (int32_t)(b * N + so) : -1;
        dst[c] = real ? (int32_t)(b * N + di) : -1;
    }
    if (bad) atomicOr(flags, bad);
}

int gnn_dense_to_index(const float *Ri, const float *Ro, int64_t B, int64_t N, int64_t E, int32_t *src, int32_t *dst,
                       int32_t *flags, void *stream)
{
    if (B < 0 || N < 0 || E < 0 || !flags || (B * E > 0 && (!Ri || !Ro || !src || !dst)))
        return fail(GNN_ERR_BADARG, "gnn_dense_to_index: bad argument");
    if (B * N >= (1ll << 31) || B * E >= (1ll << 31))
        return fail(GNN_ERR_UNSUPPORTED, "gnn_dense_to_index: batch too large for 32-bit ids");
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipError_t err = hipMemsetAsync(flags, 0, sizeof(int32_t), s);
    if (err != hipSuccess) return fail(-(int)err, "memset of the flag failed");
    if (B * E > 0) {
        const int64_t g = (B * E + 255) / 256;
        GNN_LAUNCH("k_dense_to_index", k_dense_to_index, (unsigned)(g < 65536 ? g : 65536), 256, s, Ri, Ro, B, N, E, src,
                   dst, flags);
    }
    return 0;
}

static bool grads_complete(const gnn_grads_t *gr)
{
    return gr && gr->Win && gr->bin && gr->W1 && gr->b1 && gr->W2 && gr->b2 && gr->W3 && gr->b3 && gr->W4 && gr->b4;
}

int gnn_edge_bwd(const float *H, int32_t ldh, const gnn_graph_t *g, const gnn_params_t *p, const float *e,
                 const float *grad_e, float *grad_H, const gnn_grads_t *gr, void *workspace,
                 size_t workspace_bytes, void *stream)
{
    if (!g || !p || !workspace || g->n_hits < 0 || g->n_segments < 0 || !grads_complete(gr))
        return fail(GNN_ERR_BADARG, "gnn_edge_bwd: bad argument");
    if (ldh != gnn_h_stride(p->F, p->D)) return fail(GNN_ERR_BADARG, "gnn_edge_bwd: ldh must be gnn_h_stride(F, D)");
    if ((g->n_hits > 0 && (!H || !grad_H)) || (g->n_segments > 0 && (!e || !grad_e)))
        return fail(GNN_ERR_BADARG, "gnn_edge_bwd: tensor missing");
    return edge_bwd(H, g, p, e, grad_e, grad_H, gr, workspace, workspace_bytes, static_cast<hipStream_t>(stream));
}

int gnn_node_bwd(const float *H, int32_t ldh, const float *e, const float *Hnext, const gnn_graph_t *g,
                 const gnn_params_t *p, const float *grad_Hnext, float *grad_H, float *grad_e,
                 const gnn_grads_t *gr, void *workspace, size_t workspace_bytes, void *stream)
{
    if (!g || !p || !workspace || g->n_hits < 0 || g->n_segments < 0 || !grads_complete(gr))
        return fail(GNN_ERR_BADARG, "gnn_node_bwd: bad argument");
    if (ldh != gnn_h_stride(p->F, p->D)) return fail(GNN_ERR_BADARG, "gnn_node_bwd: ldh must be gnn_h_stride(F, D)");
    if ((g->n_hits > 0 && (!H || !Hnext || !grad_Hnext || !grad_H)) || (g->n_segments > 0 && (!e || !grad_e)))
        return fail(GNN_ERR_BADARG, "gnn_node_bwd: tensor missing");
    return node_bwd(H, e, Hnext, g, p, grad_Hnext, grad_H, grad_e, gr, workspace, workspace_bytes,
                    static_cast<hipStream_t>(stream));
}

int gnn_segclf_forward_train_events(const gnn_graph_t *g, const gnn_params_t *p, const int32_t *hit_ptr,
                                    const int32_t *seg_ptr, int64_t n_graphs, int32_t max_hits,
                                    int32_t max_segments, int32_t n_iters, float *e_all, float *H_all,
                                    void *stream)
{
    if (!g || !p || n_iters < 0 || n_graphs < 0 || max_hits < 0 || max_segments < 0)
        return fail(GNN_ERR_BADARG, "gnn_segclf_forward_train_events: bad argument");
    if (n_graphs > 0 && (!hit_ptr || !seg_ptr)) return fail(GNN_ERR_BADARG, "gnn_segclf_forward_train_events: graph offsets missing");
    if (g->n_segments > 0 && (!e_all || !g->src || !g->dst)) return fail(GNN_ERR_BADARG, "gnn_segclf_forward_train_events: segment arrays missing");
    if (g->n_hits > 0 && (!H_all || !g->X || !g->in_ptr || !g->out_ptr)) return fail(GNN_ERR_BADARG, "gnn_segclf_forward_train_events: hit arrays missing");
    if (!gnn_events_supported(p->F, p->D, max_hits, max_segments))
        return fail(GNN_ERR_UNSUPPORTED, "events of up to %d hits / %d segments do not fit one workgroup's LDS at input_dim=%d hidden_dim=%d",
                    max_hits, max_segments, p->F, p->D);
    hipStream_t s = static_cast<hipStream_t>(stream);
#define X_(F_, D_) if (p->F == F_ && p->D == D_) return run_events<F_, D_>(g, p, hit_ptr, seg_ptr, n_graphs, max_hits, max_segments, n_iters, nullptr, s, e_all, H_all);
    GNN_FOR_EACH_SHAPE(X_)
#undef X_
    return fail(GNN_ERR_UNSUPPORTED, "no kernel for input_dim=%d hidden_dim=%d", p->F, p->D);
}

int gnn_events_backward_supported(int32_t F, int32_t D, int64_t max_hits, int64_t max_segments)
{
    if (max_hits < 0 || max_segments < 0) return 0;
    return backward_events_supported(F, D, max_hits, max_segments);
}

size_t gnn_backward_events_workspace_bytes(int64_t n_graphs, int32_t F, int32_t D)
{
    return backward_events_workspace_bytes(n_graphs, F, D);
}

int gnn_segclf_backward_events(const gnn_graph_t *g, const gnn_params_t *p, const int32_t *hit_ptr,
                               const int32_t *seg_ptr, int64_t n_graphs, int32_t max_hits,
                               int32_t max_segments, int32_t n_iters, const float *e_all,
                               const float *H_all, const float *grad_out, const gnn_grads_t *gr,
                               void *workspace, size_t workspace_bytes, void *stream)
{
    if (!g || !p || !gr || n_iters < 0 || n_graphs < 0 || max_hits < 0 || max_segments < 0 || !workspace)
        return fail(GNN_ERR_BADARG, "gnn_segclf_backward_events: bad argument");
    if (n_graphs > 0 && (!hit_ptr || !seg_ptr)) return fail(GNN_ERR_BADARG, "gnn_segclf_backward_events: graph offsets missing");
    if ((g->n_segments > 0 && (!e_all || !grad_out || !g->src || !g->dst)) || (g->n_hits > 0 && (!H_all || !g->in_ptr || !g->out_ptr)))
        return fail(GNN_ERR_BADARG, "gnn_segclf_backward_events: saved tensors or graph arrays missing");
    if (!gr->Win || !gr->bin || !gr->W1 || !gr->b1 || !gr->W2 || !gr->b2 || !gr->W3 || !gr->b3 || !gr->W4 || !gr->b4)
        return fail(GNN_ERR_BADARG, "gnn_segclf_backward_events: gradient pointer missing");
    return backward_events(g, p, hit_ptr, seg_ptr, n_graphs, max_hits, max_segments, n_iters, e_all, H_all,
                           grad_out, gr, workspace, workspace_bytes, static_cast<hipStream_t>(stream));
}

int gnn_bce_loss(const float *e, const float *y, int64_t n, float scale, float *loss_out,
                 float *grad_e, void *workspace, void *stream)
{
    if (n < 0 || !loss_out || !workspace || (n > 0 && (!e || !y)))
        return fail(GNN_ERR_BADARG, "gnn_bce_loss: bad argument");
    return bce_loss(e, y, n, scale, loss_out, grad_e, static_cast<float *>(workspace),
                    static_cast<hipStream_t>(stream));
}

size_t gnn_plan_workspace_bytes(int64_t n_pad, int64_t n_segments, int32_t F, int32_t D)
{
    if (n_pad < 0 || n_segments < 0) return 0;
    return sell_workspace_bytes(n_pad, n_segments, F, D);
}

int gnn_plan_shape_supported(int32_t F, int32_t D) { return sell_shape_supported(F, D); }

int gnn_plan_limits(int32_t F, int32_t D, int32_t *out4)
{
    if (!out4) return fail(GNN_ERR_BADARG, "gnn_plan_limits: null output");
    return sell_limits(F, D, out4);
}

int gnn_segclf_forward_plan(const gnn_plan_t *pl, const gnn_params_t *p, int32_t n_iters,
                            float *e_out, void *workspace, size_t workspace_bytes, void *stream)
{
    if (!pl || !p || n_iters < 0 || pl->n_pad < 0 || pl->n_segments < 0 || pl->n_tiles < 0 ||
        pl->n_chunks < 0 || (pl->n_pad & 15))
        return fail(GNN_ERR_BADARG, "gnn_segclf_forward_plan: bad argument");
    if (pl->iter_lds_records < 0 || pl->edge_lds_rows < 0)
        return fail(GNN_ERR_BADARG, "gnn_segclf_forward_plan: negative LDS size");
    if (!pl->X || !pl->in_off || !pl->out_off || (pl->n_tiles > 0 && !pl->tiles) ||
        (pl->n_segments > 0 && (!pl->src || !pl->dst || !pl->sd16 || !pl->chunks || !e_out)))
        return fail(GNN_ERR_BADARG, "gnn_segclf_forward_plan: plan array missing");
    if (!p->Win || !p->bin || !p->W1 || !p->b1 || !p->W2 || !p->b2 || !p->W3 || !p->b3 || !p->W4 || !p->b4)
        return fail(GNN_ERR_BADARG, "gnn_segclf_forward_plan: weight pointer missing");
    return sell_forward(pl, p, n_iters, e_out, workspace, workspace_bytes, static_cast<hipStream_t>(stream));
}

int gnn_segclf_forward_train_plan(const gnn_plan_t *pl, const gnn_params_t *p, int32_t n_iters,
                                  const int32_t *seg_ptr, const int32_t *tw_src, const int32_t *tw_dst, float *e_all,
                                  float *H_all, float *Q_all, float *e_out, void *workspace, size_t workspace_bytes,
                                  void *stream)
{
    if (!pl || !p || n_iters < 0 || pl->n_pad < 0 || pl->n_segments < 0 || pl->n_tiles < 0 ||
        pl->n_chunks < 0 || (pl->n_pad & 15) || pl->iter_lds_records < 0 || pl->edge_lds_rows < 0)
        return fail(GNN_ERR_BADARG, "gnn_segclf_forward_train_plan: bad argument");
    if (!pl->X || !pl->in_off || !pl->out_off || !pl->in_nbr || !pl->out_nbr || (pl->n_tiles > 0 && !pl->tiles) ||
        (pl->n_segments > 0 && (!pl->src || !pl->dst || !pl->sd16 || !pl->chunks || !e_all)))
        return fail(GNN_ERR_BADARG, "gnn_segclf_forward_train_plan: plan array missing");
    if (pl->n_segments > 0 && ((!tw_src != !tw_dst) || (!e_out && !tw_src)))
        return fail(GNN_ERR_BADARG, "gnn_segclf_forward_train_plan: e_out or the (tw_src, tw_dst) pair is needed");
    if (pl->n_pad > 0 && (!seg_ptr || !H_all || (n_iters > 0 && !Q_all)))
        return fail(GNN_ERR_BADARG, "gnn_segclf_forward_train_plan: output array missing");
    if (!p->Win || !p->bin || !p->W1 || !p->b1 || !p->W2 || !p->b2 || !p->W3 || !p->b3 || !p->W4 || !p->b4)
        return fail(GNN_ERR_BADARG, "gnn_segclf_forward_train_plan: weight pointer missing");
    const int ldh = gnn_h_stride(p->F, p->D);
    if (ldh <= 0) return fail(GNN_ERR_UNSUPPORTED, "no HIP kernel for input_dim=%d hidden_dim=%d", p->F, p->D);
    return sell_forward_train(pl, p, n_iters, seg_ptr, tw_src, tw_dst, e_all, H_all, Q_all, ldh, e_out, workspace,
                              workspace_bytes, static_cast<hipStream_t>(stream));
}

int gnn_exp_product_bound(const gnn_params_t *p, const float *x_absmax, float *bound_out,
                          void *stream)
{
    if (!p || !p->W1 || !p->b1 || !x_absmax || !bound_out || p->F <= 0 || p->D <= 0)
        return fail(GNN_ERR_BADARG, "gnn_exp_product_bound: bad argument");
    hipStream_t s = static_cast<hipStream_t>(stream);
    GNN_LAUNCH("k_exp_bound", k_exp_bound, 1, 64, s, p->W1, p->b1, x_absmax, p->F, p->D, bound_out);
    return 0;
}

int gnn_profile_begin(int32_t capacity)
{
    if (capacity <= 0) return fail(GNN_ERR_BADARG, "gnn_profile_begin: capacity must be positive");
    for (hipEvent_t ev : g_prof.ev) (void)hipEventDestroy(ev);
    g_prof.ev.assign(2 * (size_t)capacity, nullptr);
    g_prof.name.assign((size_t)capacity, nullptr);
    g_prof.start.assign((size_t)capacity, 0);
    g_prof.chain = g_prof.have_prev = false;
    for (auto &ev : g_prof.ev) {
        hipError_t err = hipEventCreate(&ev);
        if (err != hipSuccess) return fail(-(int)err, "hipEventCreate failed");
    }
    g_prof.cap = capacity;
    g_prof.n = 0;
    g_prof.on = true;
    return 0;
}

int gnn_profile_end(const char **names, float *ms, int32_t capacity_out)
{
    g_prof.on = false;
    const int n = g_prof.n;
    for (int i = 0; i < n; ++i) {
        hipError_t err = hipEventSynchronize(g_prof.ev[2 * i + 1]);
        if (err != hipSuccess) return fail(-(int)err, "hipEventSynchronize failed");
        if (i < capacity_out) {
            float t = 0.0f;
            (void)hipEventElapsedTime(&t, g_prof.ev[g_prof.start[i]], g_prof.ev[2 * i + 1]);
            if (ms) ms[i] = t;
            if (names) names[i] = g_prof.name[i];
        }
    }
    for (hipEvent_t ev : g_prof.ev) (void)hipEventDestroy(ev);
    g_prof.ev.clear();
    g_prof.n = 0;
    g_prof.cap = 0;
    return n;
}

}  // extern "C"
