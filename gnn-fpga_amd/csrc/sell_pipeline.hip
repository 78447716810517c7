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
// a 4x4 transpose-add in the quad, so that each lane evaluates the sigmoid of ONE of four
// segments.  After the degree sort of plan.py the 16 hits of a slice have near-equal list
// lengths: no divergence, ~12 % padding at 1000-hit levels (padded entries point at the NULL
// record, whose R/S half is zero).
//
// Hit-update MLP tail: lane q computes its D/4 rows of W4 and of the five record matrices;
// its weight rows differ per q, so they cannot be scalar operands: the weights are laid out
// per lane role in consumption order (table_entry; 2.6 KB at F=3, D=8) and live in LDS; the 4
// roles read 4 distinct addresses per instruction (broadcast, conflict-free).
//
// Kernels: k_iter2 (persistent phase-split iteration kernel: the fast path, see its comment; its
// FIRST variant also runs the input network, so the fast path is T launches of k_iter2 + k_edge),
// k_iter_w (hidden_dim 32 / 64: 16 lanes per hit, hit update on the matrix cores - exact fp32 rows
// and v_mfma_f32_16x16x4_f32 by default, bf16 rows and v_mfma_f32_16x16x32_bf16 with
// GNN_FLAG_BF16_MLP), k_iter (general iteration kernel: any supported shape, global-gather tiles),
// k_input4 / k_input4_x / k_input4_bf (input network + first records where the first iteration is
// not fused), k_edge / k_edge_w (final edge pass), k_pack / k_pack16 / k_pack32 (weight tables).  Activation scales are folded into the weights and an
// optional exp-product mode trades v_exp for a multiply (score4).
//
// LDS-staged windows: plan.py orders hits by (graph, topological level) and cuts them into
// tiles of <= 1280 hits (smaller for small batches), one workgroup each.  All start hits of a tile's incoming segments lie
// in one contiguous id window and all end hits of its outgoing segments in another, so the
// workgroup copies the two windows of records (PR of the previous level, QS of the next: 64 KB
// each at 1000 hits/level, D = 8) into LDS with fully coalesced reads and gathers from LDS
// (window-relative indices).  Measured before this change: random 64-byte gathers from an
// L2-resident table run at the L2 line rate (17 TB/s of 128-byte lines, half of each line
// wasted) and bound the kernel.  Tiles whose windows exceed the LDS budget (irregular graphs)
// gather from global memory instead, with workgroups renumbered so that each XCD walks a
// contiguous range of tiles (whole graphs stay in one XCD's L2).
#include "common.h"

#include <cstdlib>
#include <type_traits>

namespace {
using namespace gnn;

constexpr int SLICE = 16;
constexpr int DESC = 8;   // ints per tile / chunk descriptor (plan.py)

// ---------------------------------------------------------------------------------------------
// per-lane-role weight table layout (floats)
// ---------------------------------------------------------------------------------------------
template <int F, int D>
struct TL {
    static constexpr int d4 = D / 4, C = F + D;
    static constexpr int a4(int x) { return (x + 3) & ~3; }   // segments start 16-byte aligned
    static constexpr int o_w2 = 0;                       // [d4]            W2[r]
    static constexpr int o_in = a4(o_w2 + d4);           // [d4] bin[r], [F][d4] Win[r][k]
    static constexpr int in_sz = d4 + F * d4;
    static constexpr int o_4 = a4(o_in + in_sz);         // [d4] b4[r],  [D][d4] W4[r][k]
    static constexpr int w4_sz = d4 + D * d4;
    static constexpr int o_m = a4(o_4 + w4_sz);          // 5 x { [d4] bias, [C][d4] weights }
    static constexpr int m_sz = d4 + C * d4;
    static constexpr int m_st = a4(m_sz);                // stride between the 5 blocks
    static constexpr int o_b2 = o_m + 5 * m_st;          // [1]  scaled output bias (see k_pack)
    static constexpr int used = o_b2 + 1;
    static constexpr int stride = a4(used) + 4;
    static constexpr int o_flat = 4 * stride;       // [D] scaled W2 in natural order, [1] scaled b2
    static constexpr int total = o_flat + ((D + 1 + 3) & ~3);
};

// Scale folding.  tanh(z) = 1 - 2 r(z'), r(z') = 1 / (1 + 2^z'), z' = 2 log2(e) z, and
// sigmoid(a) = 1 / (1 + 2^a''), a'' = -log2(e) a.  With a = b2 + sum_i w_i tanh(z_i):
//     a'' = -log2(e) (b2 + sum_i w_i) + sum_i (2 log2(e) w_i) r(z'_i)
// so the pack kernel stores P, Q (and their biases) pre-multiplied by 2 log2(e), W2 as
// 2 log2(e) w_i and the bias as -log2(e) (b2 + sum w): per hidden unit the edge MLP costs
// add, v_exp, add, v_rcp, fma - no multiplies by constants in the inner loop.
constexpr float kTwoLog2e = 2.8853900817779268f;
constexpr float kLog2e = 1.4426950408889634f;

// One packed-table entry (weights in per-lane-role consumption order, activation scales folded).
template <int F, int D>
__device__ __forceinline__ float table_entry(const gnn_params_t &p, int idx)
{
    using L = TL<F, D>;
    constexpr int d4 = L::d4, C = L::C;
    const int q = idx / L::stride, pos = idx % L::stride;
    float v = 0.0f;
    if (idx >= L::o_flat) {
        const int t = idx - L::o_flat;
        if (t < D) {
            v = kTwoLog2e * p.W2[t];
        } else if (t == D) {
            float sw = p.b2[0];
            for (int k = 0; k < D; ++k) sw += p.W2[k];
            v = -kLog2e * sw;
        }
    } else if (pos < d4) {
        v = kTwoLog2e * p.W2[q * d4 + pos];
    } else if (pos >= L::o_in && pos < L::o_in + L::in_sz) {
        const int t = pos - L::o_in;
        if (t < d4) {
            v = p.bin[q * d4 + t];
        } else {
            const int k = (t - d4) / d4, i = (t - d4) % d4;
            v = p.Win[(q * d4 + i) * F + k];
        }
    } else if (pos >= L::o_4 && pos < L::o_4 + L::w4_sz) {
        const int t = pos - L::o_4;
        if (t < d4) {
            v = p.b4[q * d4 + t];
        } else {
            const int k = (t - d4) / d4, i = (t - d4) % d4;
            v = p.W4[(q * d4 + i) * D + k];
        }
    } else if (pos >= L::o_m && pos < L::o_b2 && (pos - L::o_m) % L::m_st < L::m_sz) {
        const int t = pos - L::o_m, m = t / L::m_st, u = t % L::m_st;
        if (u < d4) {
            const int r = q * d4 + u;
            v = (m == 0) ? kTwoLog2e * p.b1[r] : (m == 4) ? p.b3[r] : 0.0f;
        } else {
            const int k = (u - d4) / d4, r = q * d4 + (u - d4) % d4;
            switch (m) {
            case 0: v = kTwoLog2e * p.W1[r * 2 * C + k]; break;      // P (scaled)
            case 1: v = p.W3[r * 3 * C + k]; break;                  // R
            case 2: v = kTwoLog2e * p.W1[r * 2 * C + C + k]; break;  // Q (scaled)
            case 3: v = p.W3[r * 3 * C + C + k]; break;              // S
            default: v = p.W3[r * 3 * C + 2 * C + k]; break;         // U
            }
        }
    } else if (pos == L::o_b2) {
        float sw = p.b2[0];
        for (int k = 0; k < D; ++k) sw += p.W2[k];
        v = -kLog2e * sw;
    }
    return v;
}

// NULL hit (id n_hits): P = (scaled) b1, everything else 0.  A padded list entry adds e * 0; a
// padded segment scores sigmoid(W2 tanh(b1) + b2) (gnn/trainSegmentClassifier.py:83-93).
// Exp-product mode stores 2^P', 2^Q' instead of P', Q' (see score4).
template <int F, int D, bool XP>
__device__ __forceinline__ void write_null_rows(const gnn_params_t &p, float *PRa, float *PRb,
                                                float *QSa, float *QSb, float *U, float *Pc,
                                                float *Qc, int64_t n_hits)
{
    constexpr int d4 = D / 4;
    for (int t = threadIdx.x; t < 2 * D; t += 256) {
        const int q = t / (2 * d4), w = t % (2 * d4);
        float pv = (w < d4) ? kTwoLog2e * p.b1[q * d4 + w] : 0.0f;
        if (XP && w < d4) pv = __builtin_amdgcn_exp2f(pv);
        const float qv = (XP && w < d4) ? 1.0f : 0.0f;
        PRa[n_hits * 2 * D + t] = pv;
        PRb[n_hits * 2 * D + t] = pv;
        QSa[n_hits * 2 * D + t] = qv;
        QSb[n_hits * 2 * D + t] = qv;
    }
    for (int t = threadIdx.x; t < D; t += 256) {
        U[n_hits * D + t] = 0.0f;
        const float pb = kTwoLog2e * p.b1[t];
        Pc[n_hits * D + t] = XP ? __builtin_amdgcn_exp2f(pb) : pb;
        Qc[n_hits * D + t] = XP ? 1.0f : 0.0f;
    }
}

// stand-alone pack: for batches without hits, and for the large tables (Cfg::pack_first) that are
// cheaper to build once than in every workgroup of k_input4
template <int F, int D, bool XP>
__global__ __launch_bounds__(256) void k_pack(gnn_params_t p, float *__restrict__ table,
                                              float *PRa, float *PRb, float *QSa, float *QSb,
                                              float *U, float *Pc, float *Qc, int64_t n_hits)
{
    using L = TL<F, D>;
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < L::total; idx += gridDim.x * 256)
        table[idx] = table_entry<F, D>(p, idx);
    if (blockIdx.x == 0) write_null_rows<F, D, XP>(p, PRa, PRb, QSa, QSb, U, Pc, Qc, n_hits);
}

// ---------------------------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ float dpp(float v)
{
    // quad_perm reads only lanes of the own quad (always valid), so no "old" value is needed:
    // mov_dpp (old = undef) saves the v_mov that update_dpp(0, ...) needs to set it up
    return __builtin_bit_cast(
        float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
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

typedef float f2_t __attribute__((ext_vector_type(2)));
typedef float f3_t __attribute__((ext_vector_type(3)));
typedef float f4_t __attribute__((ext_vector_type(4)));

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

// copy n4 float4s global -> LDS with the whole workgroup (no barrier); 8 loads in flight per
// lane before the first LDS write (a plain copy loop waits on every load)
template <int NT>
__device__ __forceinline__ void stage4(const float *__restrict__ g, float *lds, int n4)
{
    const float4 *__restrict__ g4 = reinterpret_cast<const float4 *>(g);
    float4 *l4 = reinterpret_cast<float4 *>(lds);
    int i = threadIdx.x;
    for (; i + 7 * NT < n4; i += 8 * NT) {
        float4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = g4[i + j * NT];
#pragma unroll
        for (int j = 0; j < 8; ++j) l4[i + j * NT] = v[j];
    }
    for (; i < n4; i += NT) l4[i] = g4[i];
}

// per-shape launch configuration and LDS budgets (records), mirrored to plan.py through
// gnn_plan_limits()
template <int F, int D>
struct Cfg {
    static constexpr int NT = (D <= 16) ? 1024 : 256;       // threads per workgroup
    static constexpr int lds_bytes = 160 * 1024;
    static constexpr int table_bytes = TL<F, D>::total * 4;
    // k_iter window: records of 2D floats ([P|R] or [Q|S]); k_edge window: rows of D floats
    // hidden_dim 16 with F <= 4 runs the 16-lanes-per-hit kernel (k_iter_w, exact fp32 matrix-core hit
    // update), which gathers by ABSOLUTE hit id: no LDS windows (window-relative lists) for it either
    static constexpr bool wide16 = (D == 16 && F <= 4);
    static constexpr int it_rec = (D <= 16 && !wide16) ? (lds_bytes - table_bytes - 2048) / (8 * D) : 0;
    static constexpr int ed_rec = (D <= 16) ? (lds_bytes - 2048) / (4 * D) : 0;
    // D <= 16: > one 1000-hit detector level incl. fluctuations (LDS windows).  Wide shapes have no
    // LDS windows; 256-hit tiles = one slice per wave of k_iter_w's 16, and many workgroups per CU
    // for the fp32 k_iter (4 waves each)
    static constexpr int tile_hits = (D <= 16 && !wide16) ? 1280 : 256;
    static constexpr int chunk_segments = 16384;   // > one level pair of a 100k-segment graph
    // Cross-slice prefetch keeps ~25 asm-loaded registers in flight while a slice is processed.
    // That is only legal if the register allocator never spills: a spill of an in-flight
    // register would save garbage.  Enabled for the shapes whose k_iter builds with zero scratch
    // and no AGPR copies (tests/test_abi_and_host.py checks this against the compiler's resource
    // remarks); the others wait for the prefetch immediately (same code, no in-flight window).
    // D >= 32 sits at the 256-register cap: the allocator parks values in AGPRs there.
    static constexpr bool pipelined = (D <= 8 && F <= 3) || (D == 4);
    // shapes whose persistent phase-split kernel (k_iter2) builds with zero scratch / AGPRs
    static constexpr bool iter2 = (D <= 8);
    // big weight tables are packed once by k_pack and copied, not rebuilt per workgroup
    static constexpr bool pack_first = (TL<F, D>::total > 4096);
    // ... and whose first-iteration variant (input network fused in, FIRST) does as well
    static constexpr bool fuse_first = iter2 && F <= 3 && !(F == 2 && D == 8);
};

// out[i] = bias[i] + sum_k W[k][i] * in[k]  for this lane's d4 rows.  The block
// { bias[d4], W[KD+KF][d4] } sits 16-byte aligned in LDS; it is fetched with ds_read_b128
// into registers in chunks of 8 floats before the FMAs consume it (larger chunks cost registers
// the prefetching kernels do not have).
template <int D4, int KD, int KF, bool PK = true>
__device__ __forceinline__ void role_gemv(const float *w, const float *hn, const float *x,
                                          float *out)
{
    constexpr int NW = D4 + (KD + KF) * D4;
    constexpr int CH = 8;
#ifndef GNN_NO_GEMV_PK
    if constexpr (PK && D4 >= 8 && D4 % 4 == 0) {
        // wide layers (D >= 32, where 256 VGPRs are available): rows (i, i+1) of an input k are neighbours in the block, so one
        // v_pk_fma_f32 does two of the lane's D4 FMAs per input (same order per row, same bits);
        // the block is still fetched 16 bytes at a time, 8 floats ahead of their use
        const f4_t *wv = reinterpret_cast<const f4_t *>(__builtin_assume_aligned(w, 16));
        // (compiler barrier: without it the scheduler pulls the LDS reads of several blocks to
        // the front and spills - k_input4 at D = 64 went to 3.4 KB of scratch per lane)
        asm volatile("" ::: "memory");
        f2_t o2[D4 / 2];
#pragma unroll
        for (int v = 0; v < D4 / 4; ++v) {
            const f4_t t = wv[v];
            o2[2 * v] = f2_t{t.x, t.y};
            o2[2 * v + 1] = f2_t{t.z, t.w};
        }
        // software pipeline: the D4 weights of input k + 1 are requested before those of input k
        // are consumed (one wave per SIMD at D = 64: nothing else hides the LDS latency)
        f4_t cur[D4 / 4], nxt[D4 / 4];
#pragma unroll
        for (int v = 0; v < D4 / 4; ++v) cur[v] = wv[D4 / 4 + v];
#pragma unroll
        for (int k = 0; k < KD + KF; ++k) {
            if (k + 1 < KD + KF)
#pragma unroll
                for (int v = 0; v < D4 / 4; ++v) nxt[v] = wv[(D4 + (k + 1) * D4) / 4 + v];
            const float in = k < KD ? hn[k < KD ? k : 0] : x[k >= KD ? k - KD : 0];
            const f2_t in2 = {in, in};
#pragma unroll
            for (int v = 0; v < D4 / 4; ++v) {
                o2[2 * v] = f2_t{cur[v].x, cur[v].y} * in2 + o2[2 * v];
                o2[2 * v + 1] = f2_t{cur[v].z, cur[v].w} * in2 + o2[2 * v + 1];
            }
#pragma unroll
            for (int v = 0; v < D4 / 4; ++v) cur[v] = nxt[v];
        }
#pragma unroll
        for (int i = 0; i < D4 / 2; ++i) {
            out[2 * i] = o2[i].x;
            out[2 * i + 1] = o2[i].y;
        }
        return;
    }
#endif
    const float4 *w4 = reinterpret_cast<const float4 *>(__builtin_assume_aligned(w, 16));
#pragma unroll
    for (int c0 = 0; c0 < NW; c0 += CH) {
        constexpr int dummy = 0; (void)dummy;
        float wr[CH];
#pragma unroll
        for (int v = 0; v < CH / 4; ++v) {
            if (c0 + 4 * v < NW) {
                const float4 t = w4[c0 / 4 + v];
                wr[4 * v] = t.x; wr[4 * v + 1] = t.y; wr[4 * v + 2] = t.z; wr[4 * v + 3] = t.w;
            }
        }
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            const int g = c0 + j;                 // position in the block
            if (g < NW) {
                if (g < D4) {
                    out[g] = wr[j];
                } else {
                    const int k = (g - D4) / D4, i = (g - D4) % D4;
                    out[i] = fmaf(wr[j], k < KD ? hn[k < KD ? k : 0] : x[k >= KD ? k - KD : 0], out[i]);
                }
            }
        }
    }
}

// role_gemv for two rows per lane, whole block requested from LDS up front and consumed by packed
// FMAs (same order, same bits).  For code where every wave of the workgroup runs the same gemv
// at the same time (nothing else to hide the LDS latency behind); costs 4 * ceil(NW / 4) registers.
template <int KD, int KF>
__device__ __forceinline__ void role_gemv_burst(const float *w, const float *hn, const float *x,
                                                float *out)
{
    constexpr int NW = 2 + (KD + KF) * 2, NV = (NW + 3) / 4;   // last vector may be half padding
    const f4_t *wv = reinterpret_cast<const f4_t *>(__builtin_assume_aligned(w, 16));
    f4_t t[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) t[v] = wv[v];
    f2_t acc = {t[0].x, t[0].y};                    // bias
#pragma unroll
    for (int j = 1; j < NW / 2; ++j) {              // pair j holds W[k][0..1], k = j - 1
        const int k = j - 1;
        const float in = k < KD ? hn[k < KD ? k : 0] : x[k >= KD ? k - KD : 0];
        const f2_t wk = (j & 1) ? f2_t{t[j / 2].z, t[j / 2].w} : f2_t{t[j / 2].x, t[j / 2].y};
        acc = wk * f2_t{in, in} + acc;
    }
    out[0] = acc.x;
    out[1] = acc.y;
}

// This lane's chunk of the records the next pass gathers, computed from the new hit features
// [hn (D) | x (F)]: PR = [P | R], QS = [Q | S], U; for the last iteration compact P and Q only.
template <int F, int D, bool LAST, bool XP>
struct Records {
    static constexpr int d4 = D / 4;
    float pr[LAST ? d4 : 2 * d4], qs[LAST ? d4 : 2 * d4], u[d4];

    __device__ __forceinline__ void compute(const float *wl, const float *hn, const float *x)
    {
        using L = TL<F, D>;
        if constexpr (LAST) {
            role_gemv<d4, D, F>(wl + L::o_m + 0 * L::m_st, hn, x, pr);
            role_gemv<d4, D, F>(wl + L::o_m + 2 * L::m_st, hn, x, qs);
        } else {
            role_gemv<d4, D, F>(wl + L::o_m + 0 * L::m_st, hn, x, pr);
            role_gemv<d4, D, F>(wl + L::o_m + 1 * L::m_st, hn, x, pr + d4);
            role_gemv<d4, D, F>(wl + L::o_m + 2 * L::m_st, hn, x, qs);
            role_gemv<d4, D, F>(wl + L::o_m + 3 * L::m_st, hn, x, qs + d4);
            role_gemv<d4, D, F>(wl + L::o_m + 4 * L::m_st, hn, x, u);
        }
        if constexpr (XP) {          // publish 2^P', 2^Q': the per-segment exp becomes a multiply
#pragma unroll
            for (int i = 0; i < d4; ++i) {
                pr[i] = __builtin_amdgcn_exp2f(pr[i]);
                qs[i] = __builtin_amdgcn_exp2f(qs[i]);
            }
        }
    }
    __device__ __forceinline__ void store(int64_t n, int q, float *__restrict__ PRn,
                                          float *__restrict__ QSn, float *__restrict__ U,
                                          float *__restrict__ Pc, float *__restrict__ Qc) const
    {
        if constexpr (LAST) {
            store_vec<d4>(Pc + n * D + q * d4, pr);
            store_vec<d4>(Qc + n * D + q * d4, qs);
        } else {
            store_vec<2 * d4>(PRn + n * 2 * D + q * 2 * d4, pr);
            store_vec<2 * d4>(QSn + n * 2 * D + q * 2 * d4, qs);
            store_vec<d4>(U + n * D + q * d4, u);
        }
    }
};

// Same records, computed and stored piecewise (P|R, then Q|S, then U): shorter live ranges than
// Records for callers that may issue the stores right away (k_iter2: its prefetch has arrived).
template <int F, int D, bool LAST, bool XP, bool PK = true>
__device__ __forceinline__ void emit_to(const float *wl, const float *hn, const float *x,
                                        float *__restrict__ pr_dst, float *__restrict__ qs_dst,
                                        float *__restrict__ u_dst)
{   // pr_dst / qs_dst: this lane's piece of the hit's PR / QS record (LAST: of its Pc / Qc row)
    using L = TL<F, D>;
    constexpr int d4 = D / 4;
    {
        float pr[LAST ? d4 : 2 * d4];
        role_gemv<d4, D, F, PK>(wl + L::o_m + 0 * L::m_st, hn, x, pr);
        if constexpr (XP)
#pragma unroll
            for (int i = 0; i < d4; ++i) pr[i] = __builtin_amdgcn_exp2f(pr[i]);
        if constexpr (!LAST) role_gemv<d4, D, F, PK>(wl + L::o_m + 1 * L::m_st, hn, x, pr + d4);
        store_vec<LAST ? d4 : 2 * d4>(pr_dst, pr);
    }
    {
        float qs[LAST ? d4 : 2 * d4];
        role_gemv<d4, D, F, PK>(wl + L::o_m + 2 * L::m_st, hn, x, qs);
        if constexpr (XP)
#pragma unroll
            for (int i = 0; i < d4; ++i) qs[i] = __builtin_amdgcn_exp2f(qs[i]);
        if constexpr (!LAST) role_gemv<d4, D, F, PK>(wl + L::o_m + 3 * L::m_st, hn, x, qs + d4);
        store_vec<LAST ? d4 : 2 * d4>(qs_dst, qs);
    }
    if constexpr (!LAST) {
        float u[d4];
        role_gemv<d4, D, F, PK>(wl + L::o_m + 4 * L::m_st, hn, x, u);
        store_vec<d4>(u_dst, u);
    }
}

template <int F, int D, bool LAST, bool XP, bool PK = true>
__device__ __forceinline__ void emit_now(const float *wl, const float *hn, const float *x,
                                         int64_t n, int q, float *__restrict__ PRn,
                                         float *__restrict__ QSn, float *__restrict__ U,
                                         float *__restrict__ Pc, float *__restrict__ Qc)
{
    constexpr int d4 = D / 4;
    if constexpr (LAST)
        emit_to<F, D, LAST, XP, PK>(wl, hn, x, Pc + n * D + q * d4, Qc + n * D + q * d4, nullptr);
    else
        emit_to<F, D, LAST, XP, PK>(wl, hn, x, PRn + n * 2 * D + q * 2 * d4, QSn + n * 2 * D + q * 2 * d4,
                                U + n * D + q * d4);
}

// ---------------------------------------------------------------------------------------------
// bf16 matrix-core form of the hit update for wide hidden layers (opt-in, GNN_FLAG_BF16_MLP)
// ---------------------------------------------------------------------------------------------
// At D = 64 the hit update is a [16 hits x 67] x [67 x 384] product per slice: GEMM-shaped, and
// 16x cheaper on v_mfma_f32_16x16x32_bf16 than on the fp32 vector pipe.  Operands are rounded to
// bf16 (weights once, activations - tanh outputs in [-1, 1] and X - per use), accumulation is
// fp32, records stay fp32.  Not bit-compatible with the fp32 path (scores move by ~1e-3, see
// DESIGN.md), hence opt-in.
//
// Orientation: rows = output features (A operand = weights), columns = the 16 hits of the slice
// (B operand = activations).  Lane l then holds, for hit l & 15, the 4 output features
// 16 T + 4 (l >> 4) + r of tile T - i.e. 4 consecutive floats of a record row (one 16-byte store),
// and exactly the features it must supply as B operand of the NEXT product (k-slot j of step s
// <-> feature 16 (2 s + j / 4) + 4 (l >> 4) + j % 4), so the two chained products need no data
// movement in between.  The weights are packed once per forward into A fragments in that k order
// (k_pack16): one 16-byte LDS read per MFMA.  Lane maps checked by tools/mfma_layout_check.hip.
typedef short bf16x8_t __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned short bf16_rne(float f)
{
    const unsigned u = __float_as_uint(f);
    return (unsigned short)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}

// two floats -> two bf16 in one dword, round to nearest even: v_cvt_pk_bf16_f32 (one instruction;
// the integer form above is 4 per element and is kept for the host-order packing kernels)
__device__ __forceinline__ unsigned pack_bf16(float a, float b)
{
    typedef float f2v __attribute__((ext_vector_type(2)));
    typedef __bf16 b2v __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f2v{a, b}, b2v));
}

template <int F, int D>
struct BL {                     // bf16 table layout, in 4-byte words
    static_assert(D % 32 == 0 && F <= 8, "matrix-core path: D = 32 or 64, X in one k-step");
    static constexpr int C = F + D, d4 = D / 4;
    static constexpr int NT1 = D / 16, KS1 = D / 32;            // W4: tiles, k-steps
    static constexpr int KS2 = D / 32 + 1;                      // records: hn steps + the X step
    static constexpr int NT2N = 5 * D / 16, NT2L = 2 * D / 16;  // record tiles (all five / P, Q only)
    static constexpr int o_t4 = 0;                              // [NT1][KS1][64 lanes][8 bf16]
    static constexpr int o_tmn = o_t4 + NT1 * KS1 * 256;        // [NT2N][KS2][64][8]
    static constexpr int o_tml = o_tmn + NT2N * KS2 * 256;      // [NT2L][KS2][64][8]
    static constexpr int o_b4 = o_tml + NT2L * KS2 * 256;       // [D] f32
    static constexpr int o_bmn = o_b4 + D;                      // [5D] f32, output order
    static constexpr int o_bml = o_bmn + 5 * D;                 // [2D] f32
    static constexpr int total = o_bml + 2 * D;
    // what one kernel variant keeps in LDS: [T4 | Tm(variant) | b4 | bm(variant)]
    template <bool LAST> static constexpr int tm_words() { return (LAST ? NT2L : NT2N) * KS2 * 256; }
    template <bool LAST> static constexpr int lds_words() { return NT1 * KS1 * 256 + tm_words<LAST>() + D + (LAST ? 2 : 5) * D; }
    static constexpr int tr_stride = D + 4;                     // transpose scratch row (floats)
};

// row `o` of the record product in OUTPUT order -> (weight row pointer of length C, bias, scale)
// Row order of the exact-fp32 records at D = 64 ("halves"): [P(D) | R(D)] - a hit's own P (or Q) half
// is then 256 contiguous bytes, so reading it touches two lines of the 512-byte row, not four (the
// interleaved order below costs every own-value read the whole row: 2.5 MB per 5000-hit level and
// direction at c5).  A gathering lane reads its P piece and its R piece with one 16-byte load each,
// as before.  bf16 rows and D = 32 keep the interleaved order (one load fetches both pieces there).
template <int D, bool EX>
constexpr bool row_halves() { return EX && D == 64; }

template <int F, int D>
__device__ __forceinline__ float record_weight(const gnn_params_t &p, bool last, int o, int k, bool bias,
                                               bool halves = false)
{
    constexpr int C = F + D, d4 = D / 4;
    int m, d;                                  // m: 0 P, 1 R, 2 Q, 3 S, 4 U (table_entry's blocks)
    if (last) {
        m = o < D ? 0 : 2;
        d = o % D;
    } else if (o >= 4 * D) {
        m = 4;
        d = o - 4 * D;
    } else if (halves) {
        const int row = o / (2 * D), pos = o % (2 * D);
        m = 2 * row + (pos >= D);
        d = pos % D;
    } else {
        // bf16 record rows for k_iter_w, where a hit is 16 lanes and lane p owns dims DL p .. (DL =
        // D / 16): [P(DL) R(DL)] per lane, so ONE load per lane fetches its piece of both halves and
        // the 16 lanes read a neighbour's row as whole, contiguous lines
        constexpr int DL = D / 16;
        const int row = o / (2 * D), pos = o % (2 * D), blk = pos / (2 * DL), w = pos % (2 * DL);
        m = 2 * row + (w >= DL);
        d = blk * DL + w % DL;
    }
    (void)d4;
    if (bias) return m == 0 ? kTwoLog2e * p.b1[d] : m == 4 ? p.b3[d] : 0.0f;
    switch (m) {
    case 0: return kTwoLog2e * p.W1[d * 2 * C + k];
    case 1: return p.W3[d * 3 * C + k];
    case 2: return kTwoLog2e * p.W1[d * 2 * C + C + k];
    case 3: return p.W3[d * 3 * C + C + k];
    default: return p.W3[d * 3 * C + 2 * C + k];
    }
}

template <int F, int D>
__global__ __launch_bounds__(256) void k_pack16(gnn_params_t p, unsigned *__restrict__ t16,
                                                float *PRa, float *PRb, float *QSa, float *QSb,
                                                int64_t n_pad, int xp)
{
    using B = BL<F, D>;
    if (blockIdx.x == 0 && threadIdx.x < 2 * D) {       // bf16 NULL records (cf. write_null_rows)
        constexpr int DL = D / 16;                      // row position: [P(DL) R(DL)] per lane of k_iter_w
        const int t = threadIdx.x, blk = t / (2 * DL), w = t % (2 * DL);
        const bool is_p = w < DL;
        float pv = is_p ? kTwoLog2e * p.b1[blk * DL + w] : 0.0f;
        if (xp && is_p) pv = __builtin_amdgcn_exp2f(pv);
        const float qv = (xp && is_p) ? 1.0f : 0.0f;
        unsigned short *a = reinterpret_cast<unsigned short *>(PRa), *b = reinterpret_cast<unsigned short *>(PRb);
        unsigned short *cq = reinterpret_cast<unsigned short *>(QSa), *dq = reinterpret_cast<unsigned short *>(QSb);
        a[n_pad * 2 * D + t] = b[n_pad * 2 * D + t] = bf16_rne(pv);
        cq[n_pad * 2 * D + t] = dq[n_pad * 2 * D + t] = bf16_rne(qv);
    }
    unsigned short *h = reinterpret_cast<unsigned short *>(t16);
    float *f = reinterpret_cast<float *>(t16);
    for (int i = blockIdx.x * 256 + threadIdx.x; i < 2 * B::o_b4; i += gridDim.x * 256) {
        // halfword i of the fragment area: [tile][step][lane][j]
        const int j = i & 7, l = (i >> 3) & 63, g = l >> 4, frag = i >> 9;
        float v;
        if (i < 2 * B::o_tmn) {                                   // W4
            const int T = frag / B::KS1, st = frag % B::KS1;
            v = p.W4[(16 * T + (l & 15)) * D + 16 * (2 * st + j / 4) + 4 * g + j % 4];
        } else {
            const bool last = i >= 2 * B::o_tml;
            const int fr = frag - (last ? B::o_tml : B::o_tmn) / 256;
            const int T = fr / B::KS2, st = fr % B::KS2, o = 16 * T + (l & 15);
            int k;
            if (st < B::KS2 - 1)
                k = 16 * (2 * st + j / 4) + 4 * g + j % 4;      // an hn input
            else
                k = (8 * g + j < F) ? D + 8 * g + j : -1;       // an X input (or zero padding)
            v = k < 0 ? 0.0f : record_weight<F, D>(p, last, o, k, false);
        }
        h[i] = bf16_rne(v);
    }
    for (int i = blockIdx.x * 256 + threadIdx.x; i < 8 * D; i += gridDim.x * 256) {
        float v;
        if (i < D) v = p.b4[i];
        else if (i < 6 * D) v = record_weight<F, D>(p, false, i - D, 0, true);
        else v = record_weight<F, D>(p, true, i - 6 * D, 0, true);
        f[B::o_b4 + i] = v;
    }
}

// bf16 B fragment of k-step `st` from this lane's fp32 features v[t][r] (t = tile, r = 0..3)
template <int NTILE>
__device__ __forceinline__ bf16x8_t act_frag(const float (*v)[4], int st)
{
    typedef unsigned u4v __attribute__((ext_vector_type(4)));
    const u4v w = {pack_bf16(v[2 * st][0], v[2 * st][1]), pack_bf16(v[2 * st][2], v[2 * st][3]),
                   pack_bf16(v[2 * st + 1][0], v[2 * st + 1][1]), pack_bf16(v[2 * st + 1][2], v[2 * st + 1][3])};
    return __builtin_bit_cast(bf16x8_t, w);
}

template <int F, int D, bool LAST, bool XP>
__device__ __forceinline__ void mfma_records(const bf16x8_t *Tm, const float *bm,
                                             const float (*v)[4], bf16x8_t xb, int lane, int64_t n0,
                                             float *__restrict__ PRn, float *__restrict__ QSn,
                                             float *__restrict__ U, float *__restrict__ Pc,
                                             float *__restrict__ Qc, int T0 = 0, int TS = 1);

template <int F, int D, bool LAST, bool XP>
__device__ __forceinline__ void mfma_tail_scratch(const unsigned *tb, const float *tr, int lane, int64_t n0,
                                                  float *__restrict__ PRn, float *__restrict__ QSn,
                                                  float *__restrict__ U, float *__restrict__ Pc,
                                                  float *__restrict__ Qc, int T0 = 0, int TS = 1);

// Hit update + records of one slice on the matrix cores, from a scratch the caller has filled:
// `tb`: LDS [T4 | Tm | b4 | bm] (BL); tr[hit][0 .. D) = tanh(acc), tr[hit][D .. D + F) = X;
// record tiles T0, T0 + TS, ... only (several waves may share one slice's records)
template <int F, int D, bool LAST, bool XP>
__device__ __forceinline__ void mfma_tail_scratch(const unsigned *tb, const float *tr, int lane, int64_t n0,
                                                  float *__restrict__ PRn, float *__restrict__ QSn,
                                                  float *__restrict__ U, float *__restrict__ Pc,
                                                  float *__restrict__ Qc, int T0, int TS)
{
    using B = BL<F, D>;
    constexpr int NT1 = B::NT1, KS1 = B::KS1;
    typedef float f4v __attribute__((ext_vector_type(4)));
    const bf16x8_t *T4 = reinterpret_cast<const bf16x8_t *>(tb);
    const bf16x8_t *Tm = T4 + NT1 * KS1 * 64;
    const float *b4 = reinterpret_cast<const float *>(tb + NT1 * KS1 * 256 + B::template tm_words<LAST>());
    const float *bm = b4 + D;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // own wave's LDS writes have landed
    const int hit = lane & 15, g = lane >> 4;
    float v[NT1][4];
#pragma unroll
    for (int t = 0; t < NT1; ++t) {
        const f4v r = *reinterpret_cast<const f4v *>(tr + hit * B::tr_stride + 16 * t + 4 * g);
        v[t][0] = r.x; v[t][1] = r.y; v[t][2] = r.z; v[t][3] = r.w;
    }
    bf16x8_t xb;                                                   // the X step's B fragment
#pragma unroll
    for (int j = 0; j < 8; ++j) xb[j] = 0;
    if (g == 0)
#pragma unroll
        for (int j = 0; j < F; ++j) xb[j] = (short)bf16_rne(tr[hit * B::tr_stride + D + j]);
    // 2. hl = tanh(W4 q + b4)
    bf16x8_t qb[KS1];
#pragma unroll
    for (int st = 0; st < KS1; ++st) qb[st] = act_frag<NT1>(v, st);
#pragma unroll
    for (int T = 0; T < NT1; ++T) {
        f4v c = *reinterpret_cast<const f4v *>(b4 + 16 * T + 4 * g);
#pragma unroll
        for (int st = 0; st < KS1; ++st)
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(T4[(T * KS1 + st) * 64 + lane], qb[st], c, 0, 0, 0);
        v[T][0] = tanh_f(c.x); v[T][1] = tanh_f(c.y); v[T][2] = tanh_f(c.z); v[T][3] = tanh_f(c.w);
    }
    mfma_records<F, D, LAST, XP>(Tm, bm, v, xb, lane, n0, PRn, QSn, U, Pc, Qc, T0, TS);
}

// records = Wm [hl | x] + bias, tile by tile, stored as 16-byte pieces of the record rows.
// v: this lane's features of hl in the matrix-core layout (hit = lane & 15, features
// 16 t + 4 (lane >> 4) + r), xb: the X step's B fragment.
template <int F, int D, bool LAST, bool XP>
__device__ __forceinline__ void mfma_records(const bf16x8_t *Tm, const float *bm,
                                             const float (*v)[4], bf16x8_t xb, int lane, int64_t n0,
                                             float *__restrict__ PRn, float *__restrict__ QSn,
                                             float *__restrict__ U, float *__restrict__ Pc,
                                             float *__restrict__ Qc, int T0, int TS)
{
    using B = BL<F, D>;
    constexpr int d4 = D / 4, NT1 = B::NT1, KS1 = B::KS1, KS2 = B::KS2;
    constexpr int NT2 = LAST ? B::NT2L : B::NT2N;
    typedef float f4v __attribute__((ext_vector_type(4)));
    const int hit = lane & 15, g = lane >> 4;
    bf16x8_t hb[KS1];
#pragma unroll
    for (int st = 0; st < KS1; ++st) hb[st] = act_frag<NT1>(v, st);
    const int64_t n = n0 + hit;
    (void)d4;
#pragma unroll
    for (int Tu = 0; Tu < NT2; ++Tu) {
        if (TS != 1 && (Tu % 4) != T0) continue;                   // (TS is 1 or 4; wave-uniform)
        const int T = Tu;
        f4v c = *reinterpret_cast<const f4v *>(bm + 16 * T + 4 * g);
#pragma unroll
        for (int st = 0; st < KS2; ++st)
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Tm[(T * KS2 + st) * 64 + lane],
                                                        st < KS1 ? hb[st < KS1 ? st : 0] : xb, c, 0, 0, 0);
        const int o = 16 * T + 4 * g;                              // first of this lane's 4 outputs
        float *dst;
        bool expo;                                                 // a P / Q position: 2^x in XP mode
        bool expo_half = false;                                    // D = 32: [P P R R] inside one store
        if constexpr (LAST) {
            dst = (o < D ? Pc + n * D + o : Qc + n * D + (o - D));
            expo = true;
        } else if (o >= 4 * D) {
            dst = U + n * D + (o - 4 * D);
            expo = false;
        } else {
            dst = (o < 2 * D ? PRn + n * 2 * D + o : QSn + n * 2 * D + (o - 2 * D));
            constexpr int DL = D / 16;                             // rows: [P(DL) R(DL)] per 2 DL positions
            expo = (o % (2 * DL)) < DL;
            expo_half = DL == 2;
        }
        if (XP && expo) {
            c.x = __builtin_amdgcn_exp2f(c.x); c.y = __builtin_amdgcn_exp2f(c.y);
            if (!expo_half) { c.z = __builtin_amdgcn_exp2f(c.z); c.w = __builtin_amdgcn_exp2f(c.w); }
        }
        if (!LAST && o < 4 * D) {
            // gather records travel as bf16 (row = 2D halfwords, same position order): half the
            // bytes per list step, half the registers per record group
            unsigned *row = reinterpret_cast<unsigned *>(o < 2 * D ? PRn : QSn) + n * D + (o % (2 * D)) / 2;
            *reinterpret_cast<uint2 *>(row) = make_uint2(pack_bf16(c.x, c.y), pack_bf16(c.z, c.w));
        } else {
            *reinterpret_cast<f4v *>(dst) = c;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// the same tail in EXACT fp32 (the default for hidden_dim 32 / 64): v_mfma_f32_16x16x4_f32
// ---------------------------------------------------------------------------------------------
// The fp32 vector form of this tail reads 1600 16-byte weight pieces from LDS per hit and lane
// (0.5 ms of LDS time per iteration at c5 x 8).  v_mfma_f32_16x16x4_f32 is a k-ordered chain of
// fp32 fmas - no operand is rounded - so the same products run on the matrix cores within the fp32
// path's 1e-5: lane l supplies ONE weight (row l & 15, k-slot l >> 4) and ONE activation (hit l & 15,
// k-slot l >> 4) per instruction and receives rows 4 (l >> 4) .. + 3.  With k-step s <-> features
// 16 (s / 4) + 4 g + s % 4 (g = l >> 4) a lane's B operand of step s is output r = s % 4 of tile s / 4
// of the previous product: the chain needs no data movement and no packing at all.  Records are fp32
// rows in k_iter_w's position order ([P(DL) R(DL)] per lane of a hit's 16).
template <int F, int D>
struct BX {                      // fp32 fragment layout, in floats
    static_assert(D % 16 == 0 && F <= 4, "exact matrix-core path: D = 16, 32 or 64, X in one k-step");
    static constexpr int NT1 = D / 16, KS1 = D / 4;             // W4: tiles, k-steps
    static constexpr int KS2 = D / 4 + 1;                       // records: hn steps + the X step
    static constexpr int NT2N = 5 * D / 16, NT2L = 2 * D / 16;
    static constexpr int o_t4 = 0;                              // [NT1][KS1][64 lanes]
    static constexpr int o_tmn = o_t4 + NT1 * KS1 * 64;         // [NT2N][KS2][64]
    static constexpr int o_tml = o_tmn + NT2N * KS2 * 64;       // [NT2L][KS2][64]
    static constexpr int o_b4 = o_tml + NT2L * KS2 * 64;        // [D]
    static constexpr int o_bmn = o_b4 + D;                      // [5D], output order
    static constexpr int o_bml = o_bmn + 5 * D;                 // [2D]
    static constexpr int total = o_bml + 2 * D;
    template <bool LAST> static constexpr int tm_words() { return (LAST ? NT2L : NT2N) * KS2 * 64; }
    template <bool LAST> static constexpr int lds_words() { return NT1 * KS1 * 64 + tm_words<LAST>() + D + (LAST ? 2 : 5) * D; }
    static constexpr int tr_stride = D + 4;
    static constexpr int kidx(int s, int g) { return 16 * (s / 4) + 4 * g + s % 4; }
};

template <int F, int D>
__global__ __launch_bounds__(256) void k_pack32(gnn_params_t p, float *__restrict__ tf,
                                                float *PRa, float *PRb, float *QSa, float *QSb,
                                                int64_t n_pad, int xp)
{
    using B = BX<F, D>;
    constexpr bool HV = row_halves<D, true>();
    if (blockIdx.x == 0 && threadIdx.x < 2 * D) {       // fp32 NULL records in k_iter_w's row order
        constexpr int DL = D / 16;
        const int t = threadIdx.x, blk = t / (2 * DL), w = t % (2 * DL);
        const bool is_p = HV ? t < D : w < DL;
        float pv = is_p ? kTwoLog2e * p.b1[HV ? t : blk * DL + w] : 0.0f;
        if (xp && is_p) pv = __builtin_amdgcn_exp2f(pv);
        const float qv = (xp && is_p) ? 1.0f : 0.0f;
        PRa[n_pad * 2 * D + t] = PRb[n_pad * 2 * D + t] = pv;
        QSa[n_pad * 2 * D + t] = QSb[n_pad * 2 * D + t] = qv;
    }
    for (int i = blockIdx.x * 256 + threadIdx.x; i < B::o_b4; i += gridDim.x * 256) {
        const int l = i & 63, g = l >> 4, frag = i >> 6;
        float v;
        if (i < B::o_tmn) {                                       // W4
            const int T = frag / B::KS1, st = frag % B::KS1;
            v = p.W4[(16 * T + (l & 15)) * D + B::kidx(st, g)];
        } else {
            const bool last = i >= B::o_tml;
            const int fr = frag - (last ? B::o_tml : B::o_tmn) / 64;
            const int T = fr / B::KS2, st = fr % B::KS2, o = 16 * T + (l & 15);
            const int k = st < B::KS1 ? B::kidx(st, g) : (g < F ? D + g : -1);
            v = k < 0 ? 0.0f : record_weight<F, D>(p, last, o, k, false, HV);
        }
        tf[i] = v;
    }
    for (int i = blockIdx.x * 256 + threadIdx.x; i < 8 * D; i += gridDim.x * 256) {
        float v;
        if (i < D) v = p.b4[i];
        else if (i < 6 * D) v = record_weight<F, D>(p, false, i - D, 0, true, HV);
        else v = record_weight<F, D>(p, true, i - 6 * D, 0, true, HV);
        tf[B::o_b4 + i] = v;
    }
}

// records = Wm [hl | x] + bias from this lane's features h[t][r] (feature 16 t + 4 g + r of hit
// lane & 15) and its X slot xb; record tiles T0, T0 + TS, ... only
template <int F, int D, bool LAST, bool XP>
__device__ __forceinline__ void mfma_records_x(const float *Tm, const float *bm, const float (*h)[4], float xb,
                                               int lane, int64_t n0, float *__restrict__ PRn,
                                               float *__restrict__ QSn, float *__restrict__ U,
                                               float *__restrict__ Pc, float *__restrict__ Qc, int T0 = 0, int TS = 1)
{
    using B = BX<F, D>;
    constexpr int KS1 = B::KS1, KS2 = B::KS2, NT2 = LAST ? B::NT2L : B::NT2N;
    typedef float f4v __attribute__((ext_vector_type(4)));
    const int hit = lane & 15, g = lane >> 4;
    const int64_t n = n0 + hit;
#pragma unroll
    for (int T = 0; T < NT2; ++T) {
        if (TS != 1 && (T % 4) != T0) continue;                    // (TS is 1 or 4; wave-uniform)
        f4v c = *reinterpret_cast<const f4v *>(bm + 16 * T + 4 * g);
#pragma unroll
        for (int st = 0; st < KS1; ++st)
            c = __builtin_amdgcn_mfma_f32_16x16x4f32(Tm[(T * KS2 + st) * 64 + lane], h[st / 4][st % 4], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(Tm[(T * KS2 + KS1) * 64 + lane], xb, c, 0, 0, 0);
        const int o = 16 * T + 4 * g;                              // first of this lane's 4 outputs
        float *dst;
        bool expo, expo_half = false;                              // P / Q positions: 2^x in XP mode
        if constexpr (LAST) {
            dst = (o < D ? Pc + n * D + o : Qc + n * D + (o - D));
            expo = true;
        } else if (o >= 4 * D) {
            dst = U + n * D + (o - 4 * D);
            expo = false;
        } else {
            dst = (o < 2 * D ? PRn + n * 2 * D + o : QSn + n * 2 * D + (o - 2 * D));
            constexpr int DL = D / 16;
            constexpr bool HV = row_halves<D, true>();
            expo = HV ? (o % (2 * D)) < D : (o % (2 * DL)) < DL;
            expo_half = !HV && DL == 2;
            if constexpr (DL == 1) {                               // D = 16: [P R P R] inside one store
                if (XP) { c.x = __builtin_amdgcn_exp2f(c.x); c.z = __builtin_amdgcn_exp2f(c.z); }
                expo = false;
            }
        }
        if (XP && expo) {
            c.x = __builtin_amdgcn_exp2f(c.x); c.y = __builtin_amdgcn_exp2f(c.y);
            if (!expo_half) { c.z = __builtin_amdgcn_exp2f(c.z); c.w = __builtin_amdgcn_exp2f(c.w); }
        }
        *reinterpret_cast<f4v *>(dst) = c;
    }
}

// Hit update of a TEAM's slice, stage 1: tr[hit][0 .. D) = q = tanh(acc) -> this wave's share of
// hl = tanh(W4 q + b4) (tiles T0, T0 + 4, ..) into th[hit][16 T + 4 g ..].  The four waves of a team
// split the D / 16 tiles (every wave doing all of them was 64 of 149 MFMAs per wave at D = 64).
template <int F, int D, bool LAST>
__device__ __forceinline__ void mfma_hidden_x(const float *tf, const float *tr, float *th, int lane, int T0)
{
    using B = BX<F, D>;
    constexpr int NT1 = B::NT1, KS1 = B::KS1;
    typedef float f4v __attribute__((ext_vector_type(4)));
    const float *T4 = tf, *b4 = tf + NT1 * KS1 * 64 + B::template tm_words<LAST>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const int hit = lane & 15, g = lane >> 4;
    float v[NT1][4];
#pragma unroll
    for (int t = 0; t < NT1; ++t) {
        const f4v r = *reinterpret_cast<const f4v *>(tr + hit * B::tr_stride + 16 * t + 4 * g);
        v[t][0] = r.x; v[t][1] = r.y; v[t][2] = r.z; v[t][3] = r.w;
    }
#pragma unroll
    for (int T = 0; T < NT1; ++T) {
        if ((T & 3) != T0) continue;                              // (wave-uniform)
        f4v c = *reinterpret_cast<const f4v *>(b4 + 16 * T + 4 * g);
#pragma unroll
        for (int st = 0; st < KS1; ++st)
            c = __builtin_amdgcn_mfma_f32_16x16x4f32(T4[(T * KS1 + st) * 64 + lane], v[st / 4][st % 4], c, 0, 0, 0);
        *reinterpret_cast<f4v *>(th + hit * B::tr_stride + 16 * T + 4 * g) =
            f4v{tanh_f(c.x), tanh_f(c.y), tanh_f(c.z), tanh_f(c.w)};
    }
}

// stage 2 (after a barrier): all of hl from th, X from tr, this wave's quarter of the record tiles
template <int F, int D, bool LAST, bool XP>
__device__ __forceinline__ void mfma_tail_scratch_x(const float *tf, const float *tr, const float *th, int lane,
                                                    int64_t n0, float *__restrict__ PRn, float *__restrict__ QSn,
                                                    float *__restrict__ U, float *__restrict__ Pc,
                                                    float *__restrict__ Qc, int T0, int TS)
{
    using B = BX<F, D>;
    constexpr int NT1 = B::NT1, KS1 = B::KS1;
    typedef float f4v __attribute__((ext_vector_type(4)));
    const float *Tm = tf + NT1 * KS1 * 64;
    const float *bm = Tm + B::template tm_words<LAST>() + D;
    const int hit = lane & 15, g = lane >> 4;
    float h[NT1][4];
#pragma unroll
    for (int t = 0; t < NT1; ++t) {
        const f4v r = *reinterpret_cast<const f4v *>(th + hit * B::tr_stride + 16 * t + 4 * g);
        h[t][0] = r.x; h[t][1] = r.y; h[t][2] = r.z; h[t][3] = r.w;
    }
    const float xb = g < F ? tr[hit * B::tr_stride + D + (g < F ? g : 0)] : 0.0f;
    mfma_records_x<F, D, LAST, XP>(Tm, bm, h, xb, lane, n0, PRn, QSn, U, Pc, Qc, T0, TS);
}

// ---------------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------------
// input network (model.py:144-146) + records of iteration 0.  4 lanes per hit, over the padded
// hit range (dummy hits have X = 0 and are never gathered).
// BF: records on the matrix cores (bf16 operands), H0 itself stays fp32; needs k_pack and
// k_pack16 to have run (weight table / fragments are read from global memory)
template <int F, int D, bool LAST, bool XP>
__global__ __launch_bounds__(256) void k_input4_bf(const float *__restrict__ X,
                                                   const float *__restrict__ table,
                                                   const unsigned *__restrict__ t16,
                                                   float *__restrict__ PRn, float *__restrict__ QSn,
                                                   float *__restrict__ U, float *__restrict__ Pc,
                                                   float *__restrict__ Qc, int64_t n_pad)
{
    using L = TL<F, D>;
    using B = BL<F, D>;
    constexpr int d4 = L::d4, NT1 = B::NT1;
    constexpr int nm = B::template tm_words<LAST>(), nb = (LAST ? 2 : 5) * D;
    typedef float f4v __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) float smem_in[];
    unsigned *tb = reinterpret_cast<unsigned *>(smem_in);           // [Tm | bm]
    for (int i = threadIdx.x; i < nm; i += 256) tb[i] = t16[(LAST ? B::o_tml : B::o_tmn) + i];
    for (int i = threadIdx.x; i < nb; i += 256) tb[nm + i] = t16[(LAST ? B::o_bml : B::o_bmn) + i];
    __syncthreads();
    const bf16x8_t *Tm = reinterpret_cast<const bf16x8_t *>(tb);
    const float *bm = reinterpret_cast<const float *>(tb + nm);
    float *tr = smem_in + nm + nb + (threadIdx.x >> 6) * 16 * B::tr_stride;
    const int lane = threadIdx.x & 63, q = lane & 3;
    for (int64_t n = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 2; n < n_pad;
         n += (int64_t)gridDim.x * 64) {
        float x[F], hl[d4];
#pragma unroll
        for (int k = 0; k < F; ++k) x[k] = X[n * F + k];
        role_gemv<d4, 0, F, false>(table + q * L::stride + L::o_in, x, x, hl);   // in-MLP, fp32
        {   // H0 = tanh(.) and X to the matrix-core lane layout through the wave's scratch
            const int hit = lane >> 2;
#pragma unroll
            for (int i = 0; i < d4; i += 4)
                *reinterpret_cast<f4v *>(tr + hit * B::tr_stride + q * d4 + i) =
                    f4v{tanh_f(hl[i]), tanh_f(hl[i + 1]), tanh_f(hl[i + 2]), tanh_f(hl[i + 3])};
            if (q == 0)
#pragma unroll
                for (int k = 0; k < F; ++k) tr[hit * B::tr_stride + D + k] = x[k];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const int hit = lane & 15, g = lane >> 4;
        float v[NT1][4];
#pragma unroll
        for (int t = 0; t < NT1; ++t) {
            const f4v r = *reinterpret_cast<const f4v *>(tr + hit * B::tr_stride + 16 * t + 4 * g);
            v[t][0] = r.x; v[t][1] = r.y; v[t][2] = r.z; v[t][3] = r.w;
        }
        bf16x8_t xb;
#pragma unroll
        for (int j = 0; j < 8; ++j) xb[j] = 0;
        if (g == 0)
#pragma unroll
            for (int j = 0; j < F; ++j) xb[j] = (short)bf16_rne(tr[hit * B::tr_stride + D + j]);
        // first hit of this wave's 16 (n is this lane's hit in the 4-lanes-per-hit layout)
        const int64_t n0 = n - (lane >> 2);
        mfma_records<F, D, LAST, XP>(Tm, bm, v, xb, lane, n0, PRn, QSn, U, Pc, Qc);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // scratch is rewritten next round
    }
}

// the same with the exact fp32 products (BX / k_pack32).  1024 threads: the 88 KB of fragments allow one
// workgroup per CU, and 4 waves per SIMD are needed to keep dependent MFMA chains apart (256 threads:
// 0.257 ms at c5 x 8)
template <int F, int D, bool LAST, bool XP>
__global__ __launch_bounds__(1024) void k_input4_x(const float *__restrict__ X,
                                                  const float *__restrict__ table,
                                                  const float *__restrict__ tfg,
                                                  float *__restrict__ PRn, float *__restrict__ QSn,
                                                  float *__restrict__ U, float *__restrict__ Pc,
                                                  float *__restrict__ Qc, int64_t n_pad)
{
    using L = TL<F, D>;
    using B = BX<F, D>;
    constexpr int d4 = L::d4, NT1 = B::NT1;
    constexpr int nm = B::template tm_words<LAST>(), nb = (LAST ? 2 : 5) * D;
    typedef float f4v __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) float smem_inx[];
    float *tb = smem_inx;                                           // [Tm | bm]
    for (int i = threadIdx.x; i < nm; i += 1024) tb[i] = tfg[(LAST ? B::o_tml : B::o_tmn) + i];
    for (int i = threadIdx.x; i < nb; i += 1024) tb[nm + i] = tfg[(LAST ? B::o_bml : B::o_bmn) + i];
    __syncthreads();
    float *tr = smem_inx + nm + nb + (threadIdx.x >> 6) * 16 * B::tr_stride;
    const int lane = threadIdx.x & 63, q = lane & 3;
    for (int64_t n = ((int64_t)blockIdx.x * 1024 + threadIdx.x) >> 2; n < n_pad;
         n += (int64_t)gridDim.x * 256) {
        float x[F], hl[d4];
#pragma unroll
        for (int k = 0; k < F; ++k) x[k] = X[n * F + k];
        role_gemv<d4, 0, F, false>(table + q * L::stride + L::o_in, x, x, hl);   // in-MLP, fp32
        {
            const int hit = lane >> 2;
#pragma unroll
            for (int i = 0; i < d4; i += 4)
                *reinterpret_cast<f4v *>(tr + hit * B::tr_stride + q * d4 + i) =
                    f4v{tanh_f(hl[i]), tanh_f(hl[i + 1]), tanh_f(hl[i + 2]), tanh_f(hl[i + 3])};
            if (q == 0)
#pragma unroll
                for (int k = 0; k < F; ++k) tr[hit * B::tr_stride + D + k] = x[k];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const int hit = lane & 15, g = lane >> 4;
        float h[NT1][4];
#pragma unroll
        for (int t = 0; t < NT1; ++t) {
            const f4v r = *reinterpret_cast<const f4v *>(tr + hit * B::tr_stride + 16 * t + 4 * g);
            h[t][0] = r.x; h[t][1] = r.y; h[t][2] = r.z; h[t][3] = r.w;
        }
        const float xb = g < F ? tr[hit * B::tr_stride + D + (g < F ? g : 0)] : 0.0f;
        const int64_t n0 = n - (lane >> 2);
        mfma_records_x<F, D, LAST, XP>(tb, tb + nm, h, xb, lane, n0, PRn, QSn, U, Pc, Qc);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // scratch is rewritten next round
    }
}

template <int F, int D, bool LAST, bool XP, bool TR = false>
__global__ __launch_bounds__(256) void k_input4(const float *__restrict__ X, gnn_params_t p,
                                                float *__restrict__ table,
                                                float *__restrict__ PRn, float *__restrict__ QSn,
                                                float *PRo, float *QSo,
                                                float *__restrict__ U, float *__restrict__ Pc,
                                                float *__restrict__ Qc, int64_t n_pad, float *H0 = nullptr,
                                                int ldh = 0)
{
    using L = TL<F, D>;
    constexpr int d4 = L::d4;
    (void)H0; (void)ldh;
    __shared__ __attribute__((aligned(16))) float lds[L::total];
    // Every workgroup packs the weight table straight from the raw weights (a few L2-resident
    // loads per thread: cheaper than one more kernel boundary); workgroup 0 also publishes it
    // and the NULL rows for the kernels that follow.
    if constexpr (Cfg<F, D>::pack_first) {      // k_pack ran before: 105 KB of table at D = 64
        stage4<256>(table, lds, L::total / 4);
    } else {
        for (int idx = threadIdx.x; idx < L::total; idx += 256) {
            const float v = table_entry<F, D>(p, idx);
            lds[idx] = v;
            if (blockIdx.x == 0) table[idx] = v;
        }
        if (blockIdx.x == 0) write_null_rows<F, D, XP>(p, PRn, PRo, QSn, QSo, U, Pc, Qc, n_pad);
    }
    __syncthreads();
    const int q = threadIdx.x & 3;
    // grid-stride over 64-hit groups: the weight table is staged once per workgroup, not per
    // 64 hits (n_pad % 16 == 0, so whole quads run or exit together)
    for (int64_t n = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 2; n < n_pad;
         n += (int64_t)gridDim.x * 64) {
        int woff = q * L::stride;                   // opaque: keeps LDS weight reads in the loop
        asm volatile("" : "+v"(woff));
        const float *wl = lds + woff;
        float x[F];
#pragma unroll
        for (int k = 0; k < F; ++k) x[k] = X[n * F + k];
        float hl[d4];
        role_gemv<d4, 0, F>(wl + L::o_in, x, x, hl);
#pragma unroll
        for (int i = 0; i < d4; ++i) hl[i] = tanh_f(hl[i]);
        if constexpr (TR) {                 // H_0 row = [tanh(Win x + bin) | x | 0] (model.py:144-146), kept for the backward
            float *hr = H0 + n * ldh;
            store_vec<d4>(hr + q * d4, hl);
            if (q == 0) {
#pragma unroll
                for (int k = 0; k < F; ++k) hr[D + k] = x[k];
                for (int k = D + F; k < ldh; ++k) hr[k] = 0.0f;
            }
        }
        float hn[D];
        quad_allgather<d4>(hl, hn);
        // (scalar-FMA form here: with the packed one the scheduler hoists LDS reads into spills)
        emit_now<F, D, LAST, XP, false>(wl, hn, x, n, q, PRn, QSn, U, Pc, Qc);
    }
}

// r(z') = 1 / (1 + 2^z'): the only transcendental pair of the edge MLP (see scale folding)
#ifdef GNN_ABLATE_TRANS
__device__ __forceinline__ float r_f(float zs) { return fmaf(zs, 0.25f, 0.5f) * zs; }   // timing only
#else
__device__ __forceinline__ float r_f(float zs)
{
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(zs));
}
#endif

// ---- untracked prefetch ------------------------------------------------------------------------
// Loads issued through asm are invisible to the compiler's s_waitcnt insertion, which otherwise
// (vmcnt is one in-order counter) waits for a just-issued prefetch whenever anything older is
// consumed.  Contract kept by k_iter: the asm outputs are the final storage of the loaded
// values, and nothing reads them until a_wait_all() has executed and they have passed a fence;
// both are asm volatile, so they stay ordered after the loads.
template <int OFF>
__device__ __forceinline__ void a_load_i32(int &dst, const int32_t *p)
{
    asm volatile("global_load_dword %0, %1, off offset:%2" : "=v"(dst) : "v"(p), "n"(OFF));
}

// same, address = wave-uniform base (SGPR pair) + 32-bit per-lane byte offset: the lane part of
// an address costs ONE long-lived VGPR instead of a 64-bit pair per array
template <int OFF>
__device__ __forceinline__ void a_load_i32_s(int &dst, const void *sbase, unsigned voff)
{
    asm volatile("global_load_dword %0, %1, %2 offset:%3" : "=v"(dst) : "v"(voff), "s"(sbase), "n"(OFF));
}

// N floats loaded by asm: the asm outputs ARE the storage (no element is copied out before
// arrive), as pieces of 4 / 3 / 2 / 1 dwords.
struct AEmpty {};
template <int N>
struct AVec {
    static constexpr int N4 = N / 4, R = N % 4;
    // only the pieces that exist are members: a struct copy (cur = nxt) must not drag unused
    // vector registers along
    struct V4 { f4_t v[N4 > 0 ? N4 : 1]; };
    [[no_unique_address]] std::conditional_t<(N4 > 0), V4, AEmpty> p4;
    [[no_unique_address]] std::conditional_t<R == 3, f3_t, AEmpty> v3;
    [[no_unique_address]] std::conditional_t<R == 2, f2_t, AEmpty> v2;
    [[no_unique_address]] std::conditional_t<R == 1, float, AEmpty> v1;
    template <int I>
    __device__ __forceinline__ void load4(const float *p)
    {
        if constexpr (I < N4) {
            asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(p4.v[I]) : "v"(p), "n"(16 * I));
            load4<I + 1>(p);
        }
    }
    __device__ __forceinline__ void load(const float *p)
    {
        static_assert(N * 4 < 4096, "immediate offset range");
        load4<0>(p);
        if constexpr (R == 3)
            asm volatile("global_load_dwordx3 %0, %1, off offset:%2" : "=v"(v3) : "v"(p), "n"(16 * N4));
        if constexpr (R == 2)
            asm volatile("global_load_dwordx2 %0, %1, off offset:%2" : "=v"(v2) : "v"(p), "n"(16 * N4));
        if constexpr (R == 1)
            asm volatile("global_load_dword %0, %1, off offset:%2" : "=v"(v1) : "v"(p), "n"(16 * N4));
    }
    template <int I>
    __device__ __forceinline__ void load4_s(const void *sb, unsigned vo)
    {
        if constexpr (I < N4) {
            asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(p4.v[I]) : "v"(vo), "s"(sb), "n"(16 * I));
            load4_s<I + 1>(sb, vo);
        }
    }
    __device__ __forceinline__ void load_s(const void *sb, unsigned vo)   // uniform base + lane offset
    {
        load4_s<0>(sb, vo);
        if constexpr (R == 3)
            asm volatile("global_load_dwordx3 %0, %1, %2 offset:%3" : "=v"(v3) : "v"(vo), "s"(sb), "n"(16 * N4));
        if constexpr (R == 2)
            asm volatile("global_load_dwordx2 %0, %1, %2 offset:%3" : "=v"(v2) : "v"(vo), "s"(sb), "n"(16 * N4));
        if constexpr (R == 1)
            asm volatile("global_load_dword %0, %1, %2 offset:%3" : "=v"(v1) : "v"(vo), "s"(sb), "n"(16 * N4));
    }
    template <int I>
    __device__ __forceinline__ void fence4()
    {
        if constexpr (I < N4) {
            asm volatile("" : "+v"(p4.v[I]));
            fence4<I + 1>();
        }
    }
    __device__ __forceinline__ void fence()
    {
        fence4<0>();
        if constexpr (R == 3) asm volatile("" : "+v"(v3));
        if constexpr (R == 2) asm volatile("" : "+v"(v2));
        if constexpr (R == 1) asm volatile("" : "+v"(v1));
    }
    template <int I>
    __device__ __forceinline__ void get4(float *dst) const
    {
        if constexpr (I < N4) {
            dst[4 * I] = p4.v[I].x; dst[4 * I + 1] = p4.v[I].y; dst[4 * I + 2] = p4.v[I].z; dst[4 * I + 3] = p4.v[I].w;
            get4<I + 1>(dst);
        }
    }
    __device__ __forceinline__ void get(float *dst) const     // only after arrive
    {
        get4<0>(dst);
        if constexpr (R == 3) { dst[4 * N4] = v3.x; dst[4 * N4 + 1] = v3.y; dst[4 * N4 + 2] = v3.z; }
        if constexpr (R == 2) { dst[4 * N4] = v2.x; dst[4 * N4 + 1] = v2.y; }
        if constexpr (R == 1) dst[4 * N4] = v1;
    }
};

__device__ __forceinline__ void a_wait_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

template <typename T, int N>
__device__ __forceinline__ void a_fence(T (&r)[N])
{
#pragma unroll
    for (int i = 0; i < N; ++i) asm volatile("" : "+v"(r[i]));
}

// ---- list sweep ----------------------------------------------------------------------------------
// SELL-16 entries of one list of a 16-hit slice.  The 4 lanes of a quad need the same entry at
// every step, so they load 4 DIFFERENT steps with one instruction (lane q takes step 4c+q of
// chunk c) and broadcast inside the quad with DPP.
template <int J>
__device__ __forceinline__ int quad_bcast_i(int v)
{
    return __builtin_amdgcn_mov_dpp(v, J * 0x55, 0xF, 0xF, true);
}
template <int J>
__device__ __forceinline__ float quad_bcast_f(float v)
{
    return dpp<J * 0x55>(v);
}

constexpr int MAXC = 6;   // chunks (of 4 steps) of a list held in registers; longer lists stream

// The 4 neighbour records of one chunk (this lane's 2*D/4-float piece of each).
template <int D>
struct Recs {
    float r[4][2 * (D / 4)];
    __device__ __forceinline__ void read(int cur, const float *REC, int q)
    {
        constexpr int d4 = D / 4;
        const int nb[4] = {quad_bcast_i<0>(cur), quad_bcast_i<1>(cur), quad_bcast_i<2>(cur),
                           quad_bcast_i<3>(cur)};
#ifdef GNN_ABLATE_LDS
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 2 * d4; ++i) r[j][i] = __int_as_float(nb[j] + i);   // timing only
        (void)REC;
#else
#pragma unroll
        for (int j = 0; j < 4; ++j) load_vec<2 * d4>(REC + (int64_t)nb[j] * 2 * D + q * 2 * d4, r[j]);
#endif
    }
};

// bf16 records (matrix-core path): a lane's piece of a record is D/4 dwords, each two bf16
// ([P pairs | R pairs]); kept packed until scored, so a record group is half the registers of the
// fp32 form and the sweep can hold two groups (the next one in flight while this one is scored).
__device__ __forceinline__ float bf_lo(unsigned w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf_hi(unsigned w) { return __uint_as_float(w & 0xFFFF0000u); }

// ES (training forward): lane q of the quad stores the score of list step k0 + q at eo (when ok)
template <int D4, bool XP, bool ES = false>
__device__ __forceinline__ void score4(const float (*rec)[2 * D4], const float *own,
                                       const float *w2, float b2, int q, float *acc, float *eo = nullptr,
                                       bool ok = false)
{
    float part[4];
#ifndef GNN_NO_PK
    constexpr bool use_pk = (D4 == 2), use_pk_wide = (D4 >= 8 && D4 % 2 == 0);   // D >= 32: 256 VGPRs to work with
#else
    constexpr bool use_pk = false, use_pk_wide = false;   // ablation builds
#endif
    if constexpr (use_pk) {
        // packed fp32 (v_pk_fma_f32 / v_pk_mul_f32: two lanes of math per issue slot): the two
        // hidden units of a record are a register pair, and the partial sums pair up by segment
        const f2_t o2 = {own[0], own[1]}, one2 = {1.0f, 1.0f};
        float r[4][2];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f2_t pj = {rec[j][0], rec[j][1]};
            f2_t a;
            if constexpr (XP) {
                a = pj * o2 + one2;
            } else {
                const f2_t z = pj + o2;
                const f2_t ex = {__builtin_amdgcn_exp2f(z.x), __builtin_amdgcn_exp2f(z.y)};
                a = ex + one2;
            }
            r[j][0] = __builtin_amdgcn_rcpf(a.x);
            r[j][1] = __builtin_amdgcn_rcpf(a.y);
        }
        const f2_t wa = {w2[0], w2[0]}, wb = {w2[1], w2[1]};
        const f2_t p01 = wa * f2_t{r[0][0], r[1][0]} + wb * f2_t{r[0][1], r[1][1]};
        const f2_t p23 = wa * f2_t{r[2][0], r[3][0]} + wb * f2_t{r[2][1], r[3][1]};
        part[0] = p01.x; part[1] = p01.y; part[2] = p23.x; part[3] = p23.y;
    } else if constexpr (use_pk_wide) {
        // wide records (D >= 16): the same packed-fp32 form over dimension pairs
        const f2_t one2 = {1.0f, 1.0f};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f2_t pj = {0.0f, 0.0f};
#pragma unroll
            for (int i = 0; i < D4; i += 2) {
                const f2_t rj = {rec[j][i], rec[j][i + 1]}, o2 = {own[i], own[i + 1]};
                f2_t a;
                if constexpr (XP) {
                    a = rj * o2 + one2;
                } else {
                    const f2_t z = rj + o2;
                    const f2_t ex = {__builtin_amdgcn_exp2f(z.x), __builtin_amdgcn_exp2f(z.y)};
                    a = ex + one2;
                }
                const f2_t r = {__builtin_amdgcn_rcpf(a.x), __builtin_amdgcn_rcpf(a.y)};
                pj = f2_t{w2[i], w2[i + 1]} * r + pj;
            }
            part[j] = pj.x + pj.y;
        }
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            part[j] = 0.0f;
#pragma unroll
            for (int i = 0; i < D4; ++i) {
                const float r = XP ? __builtin_amdgcn_rcpf(fmaf(rec[j][i], own[i], 1.0f))
                                   : r_f(rec[j][i] + own[i]);
                part[j] = fmaf(w2[i], r, part[j]);
            }
        }
    }
    // stage 1 (xor 1): even lanes keep segments {0,2}, odd lanes {1,3}
    const bool odd = q & 1, hi = q & 2;
    const float s0 = odd ? part[0] : part[1], s1 = odd ? part[2] : part[3];   // what I give away
    const float k0 = odd ? part[1] : part[0], k1 = odd ? part[3] : part[2];   // what I keep
    const float t0 = k0 + dpp<0xB1>(s0), t1 = k1 + dpp<0xB1>(s1);
    // stage 2 (xor 2): lanes 0,1 keep the lower segment of their pair, lanes 2,3 the upper
    const float give = hi ? t0 : t1, keep = hi ? t1 : t0;
    const float mine = keep + dpp<0x4E>(give);         // full sum of segment q, in lane q
    const float e = r_f(mine + b2);
    if constexpr (ES)
        if (ok) *eo = e;
    const float e4[4] = {quad_bcast_f<0>(e), quad_bcast_f<1>(e), quad_bcast_f<2>(e),
                         quad_bcast_f<3>(e)};
    if constexpr (use_pk_wide) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f2_t e2 = {e4[j], e4[j]};
#pragma unroll
            for (int i = 0; i < D4; i += 2) {
                const f2_t a2 = f2_t{rec[j][D4 + i], rec[j][D4 + i + 1]} * e2 + f2_t{acc[i], acc[i + 1]};
                acc[i] = a2.x;
                acc[i + 1] = a2.y;
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < D4; ++i) acc[i] = fmaf(e4[j], rec[j][D4 + i], acc[i]);
    }
}

// walk one list: the first MAXC chunks were prefetched into `pre`, the rest (rare) streams from
// `lst` (= nbr + base + 16*q + i16).  REC is the record table (LDS window or global memory);
// `null_idx` is the NULL record's index in REC.
// ES: the scores of the list's segments go to ebase[0 .. len) (entry k of this hit's list; the training
// forward keeps them for the backward, in the order the hit's list has them)
template <int D, bool XP, bool ES = false>
__device__ __forceinline__ void sweep(const int *pre, const int32_t *__restrict__ lst, int len,
                                      int null_idx, const float *REC, int q, const float *own,
                                      const float *w2, float b2, float *acc, float *ebase = nullptr, int elen = 0)
{
    constexpr int d4 = D / 4;
    if (len <= 0) return;
    auto fix = [&](int cur, int rem) {           // steps past the end -> NULL record
        const int lim = rem < 4 ? rem : 4;
        return (q < lim) ? cur : null_idx;
    };
    Recs<D> a;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        if (4 * c < len) {
            a.read(fix(pre[c], len - 4 * c), REC, q);
            score4<d4, XP, ES>(a.r, own, w2, b2, q, acc, ebase + 4 * c + q, 4 * c + q < elen);
        }
    }
    // lists longer than 4*MAXC steps (rare): stream the remaining chunks
    for (int k = 4 * MAXC; k < len; k += 4) {
        a.read(fix(lst[k * SLICE], len - k), REC, q);
        score4<d4, XP, ES>(a.r, own, w2, b2, q, acc, ebase + k + q, k + q < elen);
    }
}

template <int D, int NC, bool XP, bool PIPE>
__device__ __forceinline__ void sweep16(int (&c)[NC], const int32_t *__restrict__ nbr16,
                                        const int32_t *__restrict__ off16, int slice, int i16,
                                        int len, const float *REC, int q, const float *own,
                                        const float *w2, float b2, float *acc)
{
    constexpr int d4 = D / 4;
    if (len <= 0) return;
    int taken = 0;
    auto take_word = [&]() {            // next 8 list steps of the quad
        int w = c[0];
        if (taken >= NC) {              // list longer than the prefetched words (rare): fetch and
                                        // wait here, through asm: no tracked pending load on `w`
            a_load_i32_s<0>(w, nbr16 + __builtin_amdgcn_readfirstlane(off16[slice]) + taken * 4 * SLICE,
                            (unsigned)(q * SLICE + i16) * 4u);
            a_wait_all();
            asm volatile("" : "+v"(w));
        }
#pragma unroll
        for (int i = 0; i + 1 < NC; ++i) c[i] = c[i + 1];
        ++taken;
        return w;
    };
    // the 4 records of one step group; HI = 0: steps held by lanes 0,1 of the quad, 1: lanes 2,3
    // LDS byte address of this lane's piece of record 0; v_mad_u32_u16 then turns a 16-bit window
    // index (either half of a packed word, op_sel) into the piece's address in ONE instruction
    typedef __attribute__((address_space(3))) const float lds_cf;
    const unsigned rec0 = (unsigned)(uintptr_t)(lds_cf *)(REC + q * 2 * d4);
    auto issue = [&](float (*r)[2 * d4], int w, bool hi) {
        const int wa = hi ? quad_bcast_i<2>(w) : quad_bcast_i<0>(w);
        const int wb = hi ? quad_bcast_i<3>(w) : quad_bcast_i<1>(w);
        unsigned ad[4];
        static_assert(8 * D <= 64, "record size must be an inline constant");
        asm("v_mad_u32_u16 %0, %1, %2, %3" : "=v"(ad[0]) : "v"(wa), "n"(8 * D), "v"(rec0));
        asm("v_mad_u32_u16 %0, %1, %2, %3 op_sel:[1,0,0,0]" : "=v"(ad[1]) : "v"(wa), "n"(8 * D), "v"(rec0));
        asm("v_mad_u32_u16 %0, %1, %2, %3" : "=v"(ad[2]) : "v"(wb), "n"(8 * D), "v"(rec0));
        asm("v_mad_u32_u16 %0, %1, %2, %3 op_sel:[1,0,0,0]" : "=v"(ad[3]) : "v"(wb), "n"(8 * D), "v"(rec0));
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if constexpr (d4 == 2) {
                const f4_t v = *(const __attribute__((address_space(3))) f4_t *)(uintptr_t)ad[j];
                r[j][0] = v.x; r[j][1] = v.y; r[j][2] = v.z; r[j][3] = v.w;
            } else {
                const f2_t v = *(const __attribute__((address_space(3))) f2_t *)(uintptr_t)ad[j];
                r[j][0] = v.x; r[j][1] = v.y;
            }
        }
    };
    static_assert(d4 == 1 || d4 == 2, "record piece is 8 or 16 bytes");
    // Software pipeline: the LDS reads of step group g+1 are issued before group g is scored, so
    // the ~100+ cycle (bank-conflicted) LDS latency runs under the 50 VALU instructions of a group
    // instead of in front of them (only 3 sibling waves share the SIMD).
    if constexpr (!PIPE) {              // plain form (where registers are short): read, then score
        for (int k = 0; k < len; k += 8) {
            const int w = take_word();
            float r[4][2 * d4];
            issue(r, w, false);
            score4<d4, XP>(r, own, w2, b2, q, acc);
            if (k + 4 < len) {
                issue(r, w, true);
                score4<d4, XP>(r, own, w2, b2, q, acc);
            }
        }
        return;
    }
    float ra[4][2 * d4], rb[4][2 * d4];
    int w = take_word();
    issue(ra, w, false);
    for (int k = 0; k < len; k += 8) {
        const bool has_hi = k + 4 < len;
        if (has_hi) issue(rb, w, true);
        score4<d4, XP>(ra, own, w2, b2, q, acc);
        if (has_hi) {
            if (k + 8 < len) {
                w = take_word();
                issue(ra, w, false);
            }
            score4<d4, XP>(rb, own, w2, b2, q, acc);
        }
    }
}

// What the TRAINING forward keeps of an iteration besides the next records (gnn_segclf_forward_train_plan; the
// backward kernels read them): the scores e_t of the segments in the order of the hits' IN-lists (= the order of
// the plan-space batch the backward runs on: its segments are sorted by end hit), the node network's hidden layer
// q_t = tanh(W3 [mi | mo | H] + b3) and the new hit rows H_{t+1} = [H' | X | 0] of `ldh` floats.
struct TrainOut {
    const int32_t *seg_ptr;      // [n_pad + 1] first in-segment of every padded hit (CSR pointer of the backward's batch)
    float *e_t, *Q_t, *H_next;
    int ldh;
};

// one message-passing iteration for one tile: edge scores + weighted aggregation + hit update
// (+ records for the next pass).  One workgroup per tile; each wavefront takes 16-hit slices.
template <int F, int D, bool LAST, bool XP, bool TR = false>
__global__ __launch_bounds__((Cfg<F, D>::NT)) void k_iter(
    const float *__restrict__ X, const float *__restrict__ table,
    const int32_t *__restrict__ tiles, const int32_t *__restrict__ in_off,
    const int32_t *__restrict__ in_nbr, const int32_t *__restrict__ out_off,
    const int32_t *__restrict__ out_nbr, const float *__restrict__ PR,
    const float *__restrict__ QS, float *__restrict__ U, float *__restrict__ PRn,
    float *__restrict__ QSn, float *__restrict__ Pc, float *__restrict__ Qc, int64_t n_pad,
    int tiles_per_xcd, int n_tiles, int ablate, TrainOut tro = TrainOut{})
{
    using L = TL<F, D>;
    using G = Cfg<F, D>;
    constexpr int d4 = L::d4, NT = G::NT;
    (void)tro;
    // dynamic LDS: [weight table | record windows]; sized by the host from the plan, so batches
    // of small graphs (small windows) get several workgroups per CU
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *lds = smem, *win = smem + L::total;
    stage4<NT>(table, lds, L::total / 4);

    // XCD-affine renumbering (matters for global-mode tiles only): blockIdx is dealt round-robin
    // over the 8 XCDs, so give each residue class a contiguous range of tiles.
    const int tile = (blockIdx.x & 7) * tiles_per_xcd + (blockIdx.x >> 3);
    if (tile >= n_tiles) return;
    const int32_t *td = tiles + (int64_t)tile * DESC;
    const int s_begin = td[0], s_end = td[1], in_lo = td[2], in_cnt = td[3], out_lo = td[4],
              out_cnt = td[5], mode = td[6];
    const int lane = threadIdx.x & 63;
    const int q = lane & 3, i16 = lane >> 2;

    // Everything a slice needs from global memory (list offsets, index chunks, own records) is
    // requested one slice ahead: index lists stream from HBM exactly once, and a wave has only
    // 3 siblings on its SIMD (LDS-limited occupancy), so loads must be in flight for a whole
    // slice (~10k cycles) rather than a chunk.
    struct Pre {
        int il, ol, ib, ob;          // list lengths and offsets: wave-uniform (SGPRs)
        int cin[MAXC], cout[MAXC];
        AVec<d4> Pn, Qn;
        AVec<d4> acc;
        AVec<F> x;
    };
    auto prefetch = [&](Pre &p, int slice) {
        if (ablate & 8) { p.il = 12; p.ol = 12; p.ib = 0; p.ob = 0;
#pragma unroll
            for (int c = 0; c < MAXC; ++c) { p.cin[c] = 0; p.cout[c] = 0; }
            return; }
        p.ib = __builtin_amdgcn_readfirstlane(in_off[slice]);
        p.il = (__builtin_amdgcn_readfirstlane(in_off[slice + 1]) - p.ib) >> 4;
        p.ob = __builtin_amdgcn_readfirstlane(out_off[slice]);
        p.ol = (__builtin_amdgcn_readfirstlane(out_off[slice + 1]) - p.ob) >> 4;
        const int32_t *li = in_nbr + p.ib + q * SLICE + i16;
        const int32_t *lo = out_nbr + p.ob + q * SLICE + i16;
        // chunk c holds list steps 4c..4c+3 (lane q: step 4c+q); only chunks that exist are read
        // (the last one may over-read up to 3 steps into the next list: plan.py pads the arrays)
#define GNN_PF(C_)                                                              \
        if (4 * C_ < p.il) a_load_i32<C_ * 4 * SLICE * 4>(p.cin[C_], li);           \
        if (4 * C_ < p.ol) a_load_i32<C_ * 4 * SLICE * 4>(p.cout[C_], lo);
        GNN_PF(0) GNN_PF(1) GNN_PF(2) GNN_PF(3) GNN_PF(4) GNN_PF(5)
#undef GNN_PF
        static_assert(MAXC == 6, "prefetch is written out for 6 chunks");
        const int64_t n = (int64_t)slice * SLICE + i16;
        p.Pn.load(PR + n * 2 * D + q * 2 * d4);      // own P chunk
        p.Qn.load(QS + n * 2 * D + q * 2 * d4);      // own Q chunk
        p.acc.load(U + n * D + q * d4);              // W3[:, 2C:] H_n + b3
        p.x.load(X + n * F);                         // skip concat input (model.py:154)
    };
    auto arrive = [&](Pre &p) {      // all prefetched registers of p become readable
        a_wait_all();
        a_fence(p.cin); a_fence(p.cout);
        p.Pn.fence(); p.Qn.fence(); p.acc.fence(); p.x.fence();
    };
    // slices are degree-sorted (heaviest first): deal them to the wavefronts in zig-zag order
    constexpr int NWV = NT / 64;
    const int wv = threadIdx.x >> 6;
    const int rounds = (s_end - s_begin + NWV - 1) / NWV;
    auto slice_of = [&](int r) {
        const int sl = s_begin + r * NWV + ((r & 1) ? NWV - 1 - wv : wv);
        return (r < rounds && sl < s_end) ? sl : -1;
    };
    Pre cur, nxt;
    int slice = slice_of(0);
    if (slice >= 0) prefetch(cur, slice);     // in flight while the windows are staged
    float *winA = win, *winB = win + (in_cnt + 1) * 2 * D;
    if (G::it_rec > 0 && mode && !(ablate & 1)) {
        stage4<NT>(PR + (int64_t)in_lo * 2 * D, winA, in_cnt * 2 * D / 4);
        stage4<NT>(PR + n_pad * 2 * D, winA + in_cnt * 2 * D, 2 * D / 4);     // NULL record
        stage4<NT>(QS + (int64_t)out_lo * 2 * D, winB, out_cnt * 2 * D / 4);
        stage4<NT>(QS + n_pad * 2 * D, winB + out_cnt * 2 * D, 2 * D / 4);
    }
    __syncthreads();
    if (slice >= 0) arrive(cur);
    float w2[d4];
#pragma unroll
    for (int i = 0; i < d4; ++i) w2[i] = lds[q * L::stride + L::o_w2 + i];
    const float b2 = lds[L::o_b2];                      // scaled output bias

    // While `cur` is processed (LDS-mode tiles issue no VMEM instruction there) the loads of
    // `nxt` stay in flight; arrive(nxt) waits for them a whole slice after issue, and this slice's
    // stores are issued after that wait so nothing ever waits on a store.
    for (int r = 0; r < rounds; ++r) {
        const int next = slice_of(r + 1);
        if (next >= 0) {
            prefetch(nxt, next);
            // (the training variant carries more state across the sweep: it waits for its prefetch at
            // once instead of keeping it in flight - the in-flight window is only sound without spills)
            if constexpr (!G::pipelined || TR) arrive(nxt);
        }
        Records<F, D, LAST, XP> rec;
        const int64_t n = (int64_t)slice * SLICE + i16;
        if (slice >= 0) {
            // the weight-table offset is made opaque per iteration: otherwise the compiler hoists
            // every (loop-invariant) LDS weight read out of the slice loop into ~150 VGPRs
            int woff = q * L::stride;
            asm volatile("" : "+v"(woff));
            const float *wl = lds + woff;
            float acc[d4], Pn[d4], Qn[d4], xv[F];
            cur.acc.get(acc); cur.x.get(xv);
            cur.Pn.get(Pn); cur.Qn.get(Qn);
            float *eb = nullptr;          // TR: where the scores of this hit's in-list go, and how many are real
            int en = 0;
            if constexpr (TR) {
                const int s0 = tro.seg_ptr[n];
                en = tro.seg_ptr[n + 1] - s0;
                eb = tro.e_t + s0;
            }
            if (ablate & 2) {
            } else if (G::it_rec > 0 && mode) {
                // segments ending here: P[start] + Q[n], adds e * R[start]; then starting here
                sweep<D, XP, TR>(cur.cin, in_nbr + cur.ib + q * SLICE + i16, cur.il, in_cnt, winA, q, Qn, w2, b2, acc, eb, en);
                sweep<D, XP>(cur.cout, out_nbr + cur.ob + q * SLICE + i16, cur.ol, out_cnt, winB, q, Pn, w2, b2, acc);
            } else {
                sweep<D, XP, TR>(cur.cin, in_nbr + cur.ib + q * SLICE + i16, cur.il, (int)n_pad, PR, q, Qn, w2, b2, acc, eb, en);
                sweep<D, XP>(cur.cout, out_nbr + cur.ob + q * SLICE + i16, cur.ol, (int)n_pad, QS, q, Pn, w2, b2, acc);
            }
            // hit update: H' = tanh(W4 tanh(acc) + b4)                  (model.py:94-98,125)
            float ql[d4], qa[D];
#pragma unroll
            for (int i = 0; i < d4; ++i) ql[i] = tanh_f(acc[i]);
            if constexpr (TR) store_vec<d4>(tro.Q_t + n * D + q * d4, ql);
            quad_allgather<d4>(ql, qa);
            float hl[d4];
            role_gemv<d4, D, 0>(wl + L::o_4, qa, qa, hl);
#pragma unroll
            for (int i = 0; i < d4; ++i) hl[i] = tanh_f(hl[i]);
            if constexpr (TR) {           // H_{t+1} row = [H' | X | 0] (model.py:154)
                float *hr = tro.H_next + n * tro.ldh;
                store_vec<d4>(hr + q * d4, hl);
                if (q == 0) {
#pragma unroll
                    for (int k = 0; k < F; ++k) hr[D + k] = xv[k];
                    for (int k = D + F; k < tro.ldh; ++k) hr[k] = 0.0f;
                }
            }
            float hn[D];
            quad_allgather<d4>(hl, hn);
            if (!(ablate & 4)) rec.compute(wl, hn, xv);
        }
        if constexpr (G::pipelined)
            if (next >= 0) arrive(nxt);
        cur = nxt;
        if (slice >= 0 && !(ablate & 16)) rec.store(n, q, PRn, QSn, U, Pc, Qc);
        slice = next;
    }
}

// ---------------------------------------------------------------------------------------------
// k_iter_w: the iteration kernel for WIDE hidden layers (D = 32 / 64) on bf16 records
// (GNN_FLAG_BF16_MLP; BASELINE configs[4], gnn/MPNN_Seg_ACTS_mu200.ipynb: D = 64, T = 6)
// ---------------------------------------------------------------------------------------------
// What the counters said about the 4-lanes-per-hit form at D = 64 (profiles/r02_c5_a): 420
// registers per lane = ONE wave per SIMD, 320 workgroups for 256 CUs, 38 % L2 hit rate on the record
// gathers (1.8 of the 2 GB gathered per launch came over the fabric), i.e. latency-bound at 4 TB/s.
// Here a hit is 16 lanes (lane p owns dims 4p .. 4p+3): a neighbour's 256-byte bf16 record row
// [P(D) | R(D)] is read as two full 128-byte lines by the 16 lanes (8 bytes each), a record group
// of 4 list steps costs 16 registers instead of 64, the whole sweep fits 128 registers, and a
// 1024-thread workgroup (4 waves per SIMD) shares ONE copy of the 70 KB bf16 weight fragments.
// The W2 dot product is finished by the 4x4 transpose-add inside each quad plus two row rotations
// across the 4 quads of a hit.  Workgroups are persistent: the tables are staged once, then each
// XCD walks its own contiguous range of 256-hit tiles (its 32 CUs work on ~1.6 detector levels at a
// time, whose records fit the XCD's L2), wave w of a workgroup taking slice (w + k) & 15 of the
// k-th tile it visits - no barrier after the staging.  The hit update runs on the matrix cores
// from the wave's own LDS scratch (mfma_tail_scratch), 16 hits at a time.
template <int CTRL>
__device__ __forceinline__ float dpp_row(float v)     // row_ror:n etc.: all 16 lanes of a row are valid sources
{
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}

// DL = D / 16 dims per lane; a lane's piece of a record row is [P(DL) R(DL)] bf16: 16 bytes at
// D = 64, 8 bytes at D = 32 - one load
template <int DL> struct PieceW;
template <> struct PieceW<4> {
    uint4 w;
    __device__ __forceinline__ void load(const void *a) { w = *reinterpret_cast<const uint4 *>(a); }
    __device__ __forceinline__ void first(float *f) const { f[0] = bf_lo(w.x); f[1] = bf_hi(w.x); f[2] = bf_lo(w.y); f[3] = bf_hi(w.y); }
    __device__ __forceinline__ void second(float *f) const { f[0] = bf_lo(w.z); f[1] = bf_hi(w.z); f[2] = bf_lo(w.w); f[3] = bf_hi(w.w); }
};
template <> struct PieceW<1> {         // (no bf16 rows at D = 16: only named by conditional types)
    unsigned w;
    __device__ __forceinline__ void load(const void *a) { w = *reinterpret_cast<const unsigned *>(a); }
    __device__ __forceinline__ void first(float *f) const { f[0] = bf_lo(w); }
    __device__ __forceinline__ void second(float *f) const { f[0] = bf_hi(w); }
};
template <> struct PieceW<2> {
    uint2 w;
    __device__ __forceinline__ void load(const void *a) { w = *reinterpret_cast<const uint2 *>(a); }
    __device__ __forceinline__ void first(float *f) const { f[0] = bf_lo(w.x); f[1] = bf_hi(w.x); }
    __device__ __forceinline__ void second(float *f) const { f[0] = bf_lo(w.y); f[1] = bf_hi(w.y); }
};
// the same pieces of fp32 rows (exact mode): 32 bytes at D = 64, 16 at D = 32
template <int DL> struct PieceX;
template <> struct PieceX<4> {         // D = 64, rows [P(64) | R(64)] (row_halves): the R piece is 256 bytes on
    float4 a, b;
    __device__ __forceinline__ void load(const void *p)
    {
        a = reinterpret_cast<const float4 *>(p)[0];
        b = reinterpret_cast<const float4 *>(p)[16];
    }
    __device__ __forceinline__ void first(float *f) const { f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; }
    __device__ __forceinline__ void second(float *f) const { f[0] = b.x; f[1] = b.y; f[2] = b.z; f[3] = b.w; }
};
template <> struct PieceX<1> {         // D = 16: [P R] per lane, one 8-byte load
    float2 a;
    __device__ __forceinline__ void load(const void *p) { a = *reinterpret_cast<const float2 *>(p); }
    __device__ __forceinline__ void first(float *f) const { f[0] = a.x; }
    __device__ __forceinline__ void second(float *f) const { f[0] = a.y; }
};
template <> struct PieceX<2> {
    float4 a;
    __device__ __forceinline__ void load(const void *p) { a = *reinterpret_cast<const float4 *>(p); }
    __device__ __forceinline__ void first(float *f) const { f[0] = a.x; f[1] = a.y; }
    __device__ __forceinline__ void second(float *f) const { f[0] = a.z; f[1] = a.w; }
};

template <int D, bool EX = false>
struct RecW {                   // one step group (4 list steps): this lane's piece of each of the 4 records
    static constexpr int DL = D / 16;
    static constexpr unsigned row_bytes = (EX ? 8 : 4) * D;      // 2D fp32 or 2D bf16
    // byte offset of lane p's piece inside a row (halves order: the P piece; R follows 4 D bytes on)
    static constexpr unsigned piece_bytes = row_halves<D, EX>() ? 4 * DL : (EX ? 8 : 4) * DL;
    std::conditional_t<EX, PieceX<DL>, PieceW<DL>> r[4];
    // 32-bit byte offsets off the wave-uniform table base (one v_lshl_or / v_mad per address instead
    // of a 64-bit multiply-add; the host checks that the table is below 4 GB)
    __device__ __forceinline__ void read(int cur, const void *__restrict__ REC, unsigned lane_off)
    {
        const int nb[4] = {quad_bcast_i<0>(cur), quad_bcast_i<1>(cur), quad_bcast_i<2>(cur), quad_bcast_i<3>(cur)};
#pragma unroll
        for (int j = 0; j < 4; ++j)
            r[j].load(reinterpret_cast<const char *>(REC) + ((unsigned)nb[j] * row_bytes + lane_off));
    }
};

// score the 4 segments of a group for the 4 hits of this wave pass and add their weighted R / S;
// dimension pairs run on packed fp32 (v_pk_fma_f32 / v_pk_mul_f32: two lanes of math per issue slot)
template <int D, bool XP, bool EX = false>
__device__ __forceinline__ void score_w(const RecW<D, EX> &g, const float *own, const float *w2, float b2, int p,
                                        float *acc)
{
    constexpr int DL = D / 16;
    static_assert(DL == 1 || DL % 2 == 0, "one dimension per lane, or dimension pairs");
    const f2_t one2 = {1.0f, 1.0f};
    float part[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float P[DL];
        g.r[j].first(P);
        if constexpr (DL == 1) {
            const float a = XP ? fmaf(P[0], own[0], 1.0f) : __builtin_amdgcn_exp2f(P[0] + own[0]) + 1.0f;
            part[j] = w2[0] * __builtin_amdgcn_rcpf(a);
        } else {
            f2_t s2 = {0.0f, 0.0f};
#pragma unroll
            for (int i = 0; i + 1 < DL; i += 2) {
                const f2_t pj = {P[i], P[i + 1]}, o2 = {own[i], own[i + 1]};
                f2_t a;
                if constexpr (XP) {
                    a = pj * o2 + one2;
                } else {
                    const f2_t z = pj + o2;
                    a = f2_t{__builtin_amdgcn_exp2f(z.x), __builtin_amdgcn_exp2f(z.y)} + one2;
                }
                const f2_t r = {__builtin_amdgcn_rcpf(a.x), __builtin_amdgcn_rcpf(a.y)};
                s2 = f2_t{w2[i], w2[i + 1]} * r + s2;
            }
            part[j] = s2.x + s2.y;
        }
    }
    // 4x4 transpose-add inside the quad (as score4), then the other three quads of this hit
    const int q = p & 3;
    const bool odd = q & 1, hi = q & 2;
    const float s0 = odd ? part[0] : part[1], s1 = odd ? part[2] : part[3];
    const float k0 = odd ? part[1] : part[0], k1 = odd ? part[3] : part[2];
    const float t0 = k0 + dpp<0xB1>(s0), t1 = k1 + dpp<0xB1>(s1);
    const float give = hi ? t0 : t1, keep = hi ? t1 : t0;
    float mine = keep + dpp<0x4E>(give);               // segment q: sum over this quad's 4 DL dims
    mine += dpp_row<0x124>(mine);                      // row_ror:4
    mine += dpp_row<0x128>(mine);                      // row_ror:8  -> all D dims
    const float e = r_f(mine + b2);
    const float e4[4] = {quad_bcast_f<0>(e), quad_bcast_f<1>(e), quad_bcast_f<2>(e), quad_bcast_f<3>(e)};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float R[DL];
        g.r[j].second(R);
        if constexpr (DL == 1) {
            acc[0] = fmaf(R[0], e4[j], acc[0]);
        } else {
            const f2_t e2 = {e4[j], e4[j]};
#pragma unroll
            for (int i = 0; i + 1 < DL; i += 2) {
                const f2_t a2 = f2_t{R[i], R[i + 1]} * e2 + f2_t{acc[i], acc[i + 1]};
                acc[i] = a2.x;
                acc[i + 1] = a2.y;
            }
        }
    }
}

// walk one hit's list (all 16 lanes of the hit together); `lst` = nbr + the slice's list base
// (wave-uniform), `i16` the hit's index in the slice; two record groups in flight
template <int D, bool XP, bool EX = false>
__device__ __forceinline__ void sweep_w(const int32_t *__restrict__ lst, int i16, int len, int null_idx,
                                        const void *__restrict__ REC, int p, const float *own, const float *w2,
                                        float b2, float *acc)
{
    if (len <= 0) return;
    constexpr int DL = D / 16;
    const int ng = (len + 3) >> 2;
    const unsigned lane_off = RecW<D, EX>::piece_bytes * (unsigned)p;
    const unsigned st_off = (unsigned)((SLICE * (p & 3) + i16) * 4);      // byte offset of step (p & 3)
    auto index_of = [&](int c) {               // lane p reads step 4c + (p & 3); quads broadcast it
        const int cur = *reinterpret_cast<const int32_t *>(reinterpret_cast<const char *>(lst) +
                                                             ((unsigned)(c * 4 * SLICE * 4) + st_off));
        return 4 * c + (p & 3) < len ? cur : null_idx;   // (up to 3 steps past the end: plan.py pads the arrays)
    };
    RecW<D, EX> a, b;
    int ia = index_of(0), ib = ng > 1 ? index_of(1) : 0;
    a.read(ia, REC, lane_off);
    for (int c = 0; c < ng; c += 2) {
        if (c + 1 < ng) b.read(ib, REC, lane_off);
        if (c + 2 < ng) ia = index_of(c + 2);
        score_w<D, XP, EX>(a, own, w2, b2, p, acc);
        if (c + 1 < ng) {
            if (c + 2 < ng) a.read(ia, REC, lane_off);
            if (c + 3 < ng) ib = index_of(c + 3);
            score_w<D, XP, EX>(b, own, w2, b2, p, acc);
        }
    }
}

// this hit's own P (or Q) values: lane p's DL dims of the first half of the hit's row in REC
template <int D, bool EX>
__device__ __forceinline__ void load_own_w(const unsigned *__restrict__ REC, int64_t n, int p, float *out)
{
    constexpr int DL = D / 16;
    if constexpr (row_halves<D, EX>()) {
        load_vec<DL>(reinterpret_cast<const float *>(REC) + n * 2 * D + DL * p, out);
    } else {
        constexpr int WPR = EX ? 2 * D : D, WPL = EX ? 2 * DL : DL;   // 4-byte words per row / per lane piece
        std::conditional_t<EX, PieceX<DL>, PieceW<DL>> w;
        w.load(REC + n * WPR + WPL * p);
        w.first(out);
    }
}

// (hidden_dim 16: registers and LDS allow TWO workgroups per CU - 8 waves per SIMD - and the sweeps
// there are bound by the latency of their dependent loads: 0.147 -> see DESIGN ms per launch at c3 x 32)
template <int F, int D, bool LAST, bool XP, bool EX = false>
__global__ __launch_bounds__(1024, (D == 16 ? 8 : 4)) void k_iter_w(
    const float *__restrict__ X, const float *__restrict__ table, const unsigned *__restrict__ t16,
    const int32_t *__restrict__ tiles, const int32_t *__restrict__ in_off, const int32_t *__restrict__ in_nbr,
    const int32_t *__restrict__ out_off, const int32_t *__restrict__ out_nbr, const unsigned *__restrict__ PR,
    const unsigned *__restrict__ QS, float *__restrict__ U, float *__restrict__ PRn, float *__restrict__ QSn,
    float *__restrict__ Pc, float *__restrict__ Qc, int64_t n_pad, int tiles_per_xcd, int n_tiles, int wmax)
{
    using L = TL<F, D>;
    using B = std::conditional_t<EX, BX<F, D>, BL<F, D>>;      // EX: exact fp32 fragments and fp32 record rows
    static_assert(D % 16 == 0, "16 lanes x D / 16 dims per hit, matrix-core tail");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // (no static LDS: the opt-in for > 64 KB of dynamic LDS is for the whole 160 KB)
    int *grp = reinterpret_cast<int *>(smem + B::template lds_words<LAST>() + (EX ? 12 : 8) * 16 * B::tr_stride);
    unsigned *tb = reinterpret_cast<unsigned *>(smem);
    {
        constexpr int n1 = B::NT1 * B::KS1 * (EX ? 64 : 256), nm = B::template tm_words<LAST>();
        for (int i = threadIdx.x; i < n1; i += 1024) tb[i] = t16[B::o_t4 + i];
        for (int i = threadIdx.x; i < nm; i += 1024) tb[n1 + i] = t16[(LAST ? B::o_tml : B::o_tmn) + i];
        for (int i = threadIdx.x; i < D; i += 1024) tb[n1 + nm + i] = t16[B::o_b4 + i];
        for (int i = threadIdx.x; i < (LAST ? 2 : 5) * D; i += 1024)
            tb[n1 + nm + D + i] = t16[(LAST ? B::o_bml : B::o_bmn) + i];
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int hs = lane >> 4, p = lane & 15;
    // Teams: 4 waves share a slice (wave m of team t sweeps hits 4m .. 4m+3), so only 4 slices per
    // CU = 2048 hits per XCD are in progress at a time.  The team's scratch is double-buffered: one
    // barrier per round.
    const int team = wv >> 2, mem = wv & 3;
    float *scratch = smem + B::template lds_words<LAST>();
    constexpr int DL = D / 16;                         // dims per lane
    float w2[DL];
#pragma unroll
    for (int i = 0; i < DL; ++i) w2[i] = table[(p >> 2) * L::stride + L::o_w2 + DL * (p & 3) + i];
    const float b2 = table[L::o_b2];
    // XCD x (blockIdx & 7) walks its own contiguous eighth of the TILES (whole graphs stay in one L2).
    // GROUPS: consecutive tiles whose start hits of incoming segments (and end hits of outgoing ones)
    // lie in at most `wmax` consecutive records each - what an XCD's L2 holds of ONE record table (a
    // detector level's tiles share both windows, so a group is normally a level).  A workgroup runs ALL
    // in-sweeps of its slices of a group (only the [P|R] window is gathered from), parks the partial
    // sums in U, then all out-sweeps + hit updates (only the [Q|S] window): the two 2.5 MB tables of a
    // 5000-hit level at D = 64 are never live together in the 4 MB L2 (they were: 54 % hit rate,
    // 2.6 GB of traffic per launch for 0.41 GB of records, profiles/r02_c5_final_f32; 1.8 GB with the
    // split, 1.6 GB in k_iter_wx, profiles/r03_c5_f32).  The deal of slice quads over the XCD's workgroups continues round-robin
    // across group boundaries (every workgroup the same number +- 1 over the launch; no workgroup waits
    // for another: the order is for locality only).
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
    const int t_begin = xcd * tiles_per_xcd;
    const int t_end = t_begin + tiles_per_xcd < n_tiles ? t_begin + tiles_per_xcd : n_tiles;
    __syncthreads();                                   // tables staged
    if (t_begin >= t_end) return;                      // (workgroup-uniform)
    int buf = 0, rot = 0;
    for (int t = t_begin; t < t_end;) {
        if (wv == 0) {                                 // the next group: tiles t .. t + cnt - 1
            const int tt = t + lane;
            const bool in = tt < t_end;
            const int32_t *d = tiles + (int64_t)(in ? tt : t) * DESC;
            const int s1 = d[1];
            int ilo = in && d[3] > 0 ? d[2] : 0x7FFFFFFF, ihi = in && d[3] > 0 ? d[2] + d[3] : -1;
            int olo = in && d[5] > 0 ? d[4] : 0x7FFFFFFF, ohi = in && d[5] > 0 ? d[4] + d[5] : -1;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {         // inclusive prefix union of the windows
                const int a = __shfl_up(ilo, o, 64), b = __shfl_up(ihi, o, 64);
                const int c = __shfl_up(olo, o, 64), e = __shfl_up(ohi, o, 64);
                if (lane >= o) {
                    ilo = a < ilo ? a : ilo; ihi = b > ihi ? b : ihi;
                    olo = c < olo ? c : olo; ohi = e > ohi ? e : ohi;
                }
            }
            const bool ok = in && (lane == 0 || ((int64_t)ihi - ilo <= wmax && (int64_t)ohi - olo <= wmax));
            const unsigned long long m = ~__ballot(ok);          // (ok is monotone along the lanes)
            const int cnt = m ? __builtin_ctzll(m) : 64;
            const int send = __shfl(s1, cnt - 1, 64);
            if (lane == 0) { grp[0] = d[0]; grp[1] = send; grp[2] = t + cnt; }
        }
        __syncthreads();
        const int sg0 = grp[0], sg1 = grp[1];
        t = grp[2];
        __syncthreads();                               // (grp is rewritten for the next group)
        // this workgroup's quads of the group (quad q = slices sg0 + 4 q .. + 3, one per team): dealt
        // round-robin from where the previous group's deal stopped, so that no quad straddles two
        // groups (a straddling quad was a round with idle teams in each of them: + 7 % at c5 x 8)
        const int nq = (sg1 - sg0 + 3) >> 2;
        const int q0 = (((local - rot) % per_xcd) + per_xcd) % per_xcd;
        rot = (rot + nq) % per_xcd;
        const int x0 = sg0;
        // (bf16 rows, and fp32 rows narrower than 512 bytes: both tables of a level fit the L2 together and
        // the split only costs - bf16 2.27 -> 2.44 ms at c5 x 8 - so there a hit's two sweeps stay back to
        // back in phase B)
        constexpr bool SPLIT = EX && D >= 64;
        // ---- phase A: segments ENDING at the hit: P[start] with the hit's own Q, adds e R[start]
        for (int q = q0; SPLIT && x0 + 4 * q < sg1; q += per_xcd) {
            const int sl = x0 + 4 * q + team;
            if (sl < sg0 || sl >= sg1) continue;
            const int ib = __builtin_amdgcn_readfirstlane(in_off[sl]);
            const int il = (__builtin_amdgcn_readfirstlane(in_off[sl + 1]) - ib) >> 4;
            if (il <= 0) continue;                     // (wave-uniform; U keeps the start value)
            const int i16 = 4 * mem + hs;
            const int64_t n = (int64_t)sl * SLICE + i16;
            float acc[DL], ownQ[DL];
            load_vec<DL>(U + n * D + DL * p, acc);
            load_own_w<D, EX>(QS, n, p, ownQ);
#ifndef GNN_ABLATE_W_SWEEP
            sweep_w<D, XP, EX>(in_nbr + ib, i16, il, (int)n_pad, PR, p, ownQ, w2, b2, acc);
#endif
            store_vec<DL>(U + n * D + DL * p, acc);    // (re-read by this same lane in phase B)
        }
        // ---- phase B: segments STARTING at the hit, then the hit update and the next records
        for (int q = q0; x0 + 4 * q < sg1; q += per_xcd, buf ^= 1) {   // workgroup-uniform trip count
            const int sl = x0 + 4 * q + team;
            const bool on = sl >= sg0 && sl < sg1;
            float *tr = scratch + (buf * 4 + team) * 16 * B::tr_stride;
            if (on) {
                const int ob = __builtin_amdgcn_readfirstlane(out_off[sl]);
                const int ol = (__builtin_amdgcn_readfirstlane(out_off[sl + 1]) - ob) >> 4;
                const int i16 = 4 * mem + hs;
                const int64_t n = (int64_t)sl * SLICE + i16;
                float acc[DL], ownP[DL];
                load_vec<DL>(U + n * D + DL * p, acc);
                if constexpr (!SPLIT) {
                    const int ib = __builtin_amdgcn_readfirstlane(in_off[sl]);
                    const int il = (__builtin_amdgcn_readfirstlane(in_off[sl + 1]) - ib) >> 4;
                    float ownQ[DL];
                    load_own_w<D, EX>(QS, n, p, ownQ);
                    load_own_w<D, EX>(PR, n, p, ownP);
                    sweep_w<D, XP, EX>(in_nbr + ib, i16, il, (int)n_pad, PR, p, ownQ, w2, b2, acc);
                } else {
                    load_own_w<D, EX>(PR, n, p, ownP);
                }
#ifndef GNN_ABLATE_W_SWEEP
                sweep_w<D, XP, EX>(out_nbr + ob, i16, ol, (int)n_pad, QS, p, ownP, w2, b2, acc);
#endif
#pragma unroll
                for (int i = 0; i < DL; ++i) acc[i] = tanh_f(acc[i]);
                store_vec<DL>(tr + i16 * B::tr_stride + DL * p, acc);
                if (p < F) tr[i16 * B::tr_stride + D + p] = X[n * F + p];
            }
            __syncthreads();                           // the team's 16 hits are in the scratch
            // hit update H' = tanh(W4 tanh(acc) + b4) and this wave's quarter of the record tiles of
            // the next pass (model.py:94-98,125)
            if constexpr (EX) {
                // exact fp32: the team splits the tiles of hl = tanh(W4 q + b4) and meets again in th
                float *th = scratch + (8 + team) * 16 * B::tr_stride;
#ifndef GNN_ABLATE_W_TAIL
                if (on) mfma_hidden_x<F, D, LAST>(smem, tr, th, lane, mem);
#endif
                __syncthreads();
#ifndef GNN_ABLATE_W_TAIL
                if (on)
                    mfma_tail_scratch_x<F, D, LAST, XP>(smem, tr, th, lane, (int64_t)sl * SLICE, PRn, QSn, U, Pc, Qc, mem, 4);
#endif
            } else if (on) {
                mfma_tail_scratch<F, D, LAST, XP>(tb, tr, lane, (int64_t)sl * SLICE, PRn, QSn, U, Pc, Qc, mem, 4);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// k_iter_wx: the exact-fp32 wide iteration (D = 32 / 64) with SWEEP waves and MATRIX-CORE waves
// ---------------------------------------------------------------------------------------------
// What the ablation of k_iter_w<EX> said (c5 x 8, ms per launch): sweeps alone 0.34, hit update alone
// 0.24, both 0.56 - the sum: every round of a workgroup is sweep -> barrier -> W4 product -> barrier ->
// record products for all 16 waves together, so the matrix pipe idles during the sweeps and the
// memory system during the products.  Measured and dropped on the way here (same workload, ms per
// middle launch, the barrier kernel at 0.574): the record tiles deferred into the NEXT sweep of the
// same wave (0.60: in-order issue - a dependent chain of v_mfma_f32_16x16x4_f32 holds its wave for 40
// cycles per instruction, and the four waves of a SIMD reach their chunks together); one-team
// workgroups that drift apart, fragments read from the L2 as A operands (0.67: every MFMA waits for a
// global load); write-through record stores (0.57); all 16 waves sweeping with the update of a slice
// run by one wave of its team in turn (0.576: the updating wave holds its team's next slot back); a
// third record group in flight per sweep wave (no change at D = 32, 180 bytes of scratch at D = 64).
//
// Here the 16 waves of the workgroup (one per CU, fragments in LDS as before) take ROLES: waves 0-11
// are three sweep teams, waves 12-15 - one per SIMD - do nothing but hit updates (0.527).  A team's
// four waves sweep four hits each of a slice, write q = tanh(acc) and X into a slot of a 12-slot ring
// in LDS and publish it (one LDS add per wave); a matrix-core wave takes every fourth slot in sequence,
// reads the 16 hits' rows into registers, hands the slot back, and runs the whole update of the slice
// alone (64 + 340 MFMAs at D = 64; the hidden layer goes from the accumulators straight into the next
// product's B operands: no second LDS exchange).  No workgroup barrier after the table staging: a sweep
// team never waits for a product, a matrix-core wave never for a sweep it does not need.
//   prod[s]  += 1 by each of the 4 waves that filled slot s   -> full at 4 (generation + 1)
//   cons[s]  += 1 by the matrix-core wave once its rows are in registers -> the slot's next generation
//   fin      += 1 by each sweep wave when it has published its last slice
// Every wait is on a counter that only grows and only depends on work with a smaller sequence number,
// and a matrix-core wave that finds fin == 12 BEFORE it finds its slot not yet full knows that no
// further slot comes (fin is read first).  Slots are dealt in sequence i = 3 round + team; a team with
// no slice left in a round publishes an empty slot, so the sequence has no holes.  Waits are bounded
// (2^24 polls: a wave gives up instead of hanging - a bug, not a state the protocol reaches).
// EX = false: the same roles on bf16 records and bf16 fragments (GNN_FLAG_BF16_MLP): there the update is
// cheap on the matrix cores but sat, with its 16 tanh per lane and its stores, behind the round barrier
// of every wave (k_iter_w without its sweeps still took 0.25 of its 0.345 ms per launch at c5 x 8).
template <int F, int D, bool LAST, bool XP, bool EX = true>
__global__ __launch_bounds__(1024, (D == 16 ? 8 : 4)) void k_iter_wx(
    const float *__restrict__ X, const float *__restrict__ table, const unsigned *__restrict__ t16,
    const int32_t *__restrict__ tiles, const int32_t *__restrict__ in_off, const int32_t *__restrict__ in_nbr,
    const int32_t *__restrict__ out_off, const int32_t *__restrict__ out_nbr, const unsigned *__restrict__ PR,
    const unsigned *__restrict__ QS, float *__restrict__ U, float *__restrict__ PRn, float *__restrict__ QSn,
    float *__restrict__ Pc, float *__restrict__ Qc, int64_t n_pad, int tiles_per_xcd, int n_tiles, int wmax)
{
    using L = TL<F, D>;
    using B = std::conditional_t<EX, BX<F, D>, BL<F, D>>;
    constexpr int DL = D / 16, NSLOT = 12, NSW = 12, NMX = 4, NTEAM = NSW / 4;   // slots, sweep waves, matrix-core waves
    constexpr int n1 = B::NT1 * B::KS1 * (EX ? 64 : 256), nm = B::template tm_words<LAST>();
    typedef float f4v __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *ring = smem + B::template lds_words<LAST>();
    int *sync = reinterpret_cast<int *>(ring + NSLOT * 16 * B::tr_stride);   // prod[12] cons[12] meta[12] fin
    int *prod = sync, *cons = sync + NSLOT, *meta = sync + 2 * NSLOT, *fin = sync + 3 * NSLOT, *pha = fin + 1;
    {
        unsigned *tb = reinterpret_cast<unsigned *>(smem);
        for (int i = threadIdx.x; i < n1; i += 1024) tb[i] = t16[B::o_t4 + i];
        for (int i = threadIdx.x; i < nm; i += 1024) tb[n1 + i] = t16[(LAST ? B::o_tml : B::o_tmn) + i];
        for (int i = threadIdx.x; i < D; i += 1024) tb[n1 + nm + i] = t16[B::o_b4 + i];
        for (int i = threadIdx.x; i < (LAST ? 2 : 5) * D; i += 1024)
            tb[n1 + nm + D + i] = t16[(LAST ? B::o_bml : B::o_bmn) + i];
        if (threadIdx.x < 3 * NSLOT + 2) sync[threadIdx.x] = 0;
    }
    __syncthreads();                                   // the only workgroup barrier
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    auto ld = [](const int *w) { return __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
    auto bump = [](int *w) { (void)__hip_atomic_fetch_add(w, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
    constexpr int kSpinLimit = 1 << 24;

    if (wv >= NSW) {
        // ---------------- matrix-core wave: slots c, c + 4, c + 8, ... -------------------------------
        const float *T4 = smem, *Tm = smem + n1, *b4 = smem + n1 + nm, *bm = b4 + D;
        const int hit = lane & 15, g = lane >> 4;
        (void)T4; (void)Tm; (void)b4; (void)bm; (void)hit; (void)g;
        for (int i = wv - NSW;; i += NMX) {
            const int slot = i % NSLOT, need = 4 * (i / NSLOT + 1);
            bool got = false;
            for (int spins = 0; spins < kSpinLimit; ++spins) {
                const int f = ld(fin);                 // BEFORE prod: see the protocol above
                if (ld(prod + slot) >= need) { got = true; break; }
                if (f >= NSW) break;
                __builtin_amdgcn_s_sleep(4);
            }
            if (!got) break;
            const int sl = __builtin_amdgcn_readfirstlane(ld(meta + slot));
            const float *tr = ring + slot * 16 * B::tr_stride;
            if constexpr (!EX) {                       // bf16: the whole tail from the slot, then hand it back
                if (sl >= 0)
                    mfma_tail_scratch<F, D, LAST, XP>(reinterpret_cast<const unsigned *>(smem), tr, lane, (int64_t)sl * SLICE,
                                                      PRn, QSn, U, Pc, Qc, 0, 1);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (lane == 0) bump(cons + slot);
                continue;
            }
            float v[B::NT1][4];
            float xb = 0.0f;
            if (sl >= 0) {
#pragma unroll
                for (int t = 0; t < B::NT1; ++t) {
                    const f4v r = *reinterpret_cast<const f4v *>(tr + hit * B::tr_stride + 16 * t + 4 * g);
                    v[t][0] = r.x; v[t][1] = r.y; v[t][2] = r.z; v[t][3] = r.w;
                }
                xb = g < F ? tr[hit * B::tr_stride + D + (g < F ? g : 0)] : 0.0f;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the rows are in registers
            if (lane == 0) bump(cons + slot);
            if (sl < 0) continue;
            // hl = tanh(W4 q + b4): accumulator r of tile T is exactly B operand (k-step 4 T + r) of the
            // record products (BX::kidx)
            float h[B::NT1][4];
#pragma unroll
            for (int T = 0; T < B::NT1; ++T) {
                f4v c = *reinterpret_cast<const f4v *>(b4 + 16 * T + 4 * g);
#pragma unroll
                for (int st = 0; st < B::KS1; ++st)
                    c = __builtin_amdgcn_mfma_f32_16x16x4f32(T4[(T * B::KS1 + st) * 64 + lane], v[st / 4][st % 4], c, 0, 0, 0);
                h[T][0] = tanh_f(c.x); h[T][1] = tanh_f(c.y); h[T][2] = tanh_f(c.z); h[T][3] = tanh_f(c.w);
            }
            if constexpr (EX)
                mfma_records_x<F, D, LAST, XP>(Tm, bm, h, xb, lane, (int64_t)sl * SLICE, PRn, QSn, U, Pc, Qc, 0, 1);
        }
        return;
    }

    // ---------------- sweep wave ---------------------------------------------------------------------
    const int hs = lane >> 4, p = lane & 15;
    const int team = wv >> 2, mem = wv & 3;
    float w2[DL];
#pragma unroll
    for (int i = 0; i < DL; ++i) w2[i] = table[(p >> 2) * L::stride + L::o_w2 + DL * (p & 3) + i];
    const float b2 = table[L::o_b2];
    // tiles, groups and the deal of slices: as in k_iter_w, with THREE slices per round (one per team);
    // every sweep wave works the group bounds out for itself (no barrier to hand them over)
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
    const int t_begin = xcd * tiles_per_xcd;
    const int t_end = t_begin + tiles_per_xcd < n_tiles ? t_begin + tiles_per_xcd : n_tiles;
    constexpr bool SPLIT = EX && D >= 64;
    int rot = 0, seq = team, ngrp = 0;                 // seq: this team's next slot sequence number
    for (int t = t_begin; t < t_end;) {
        int sg0, sg1;
        {
            const int tt = t + lane;
            const bool in = tt < t_end;
            const int32_t *d = tiles + (int64_t)(in ? tt : t) * DESC;
            const int s1 = d[1];
            int ilo = in && d[3] > 0 ? d[2] : 0x7FFFFFFF, ihi = in && d[3] > 0 ? d[2] + d[3] : -1;
            int olo = in && d[5] > 0 ? d[4] : 0x7FFFFFFF, ohi = in && d[5] > 0 ? d[4] + d[5] : -1;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int a = __shfl_up(ilo, o, 64), b = __shfl_up(ihi, o, 64);
                const int c = __shfl_up(olo, o, 64), e = __shfl_up(ohi, o, 64);
                if (lane >= o) {
                    ilo = a < ilo ? a : ilo; ihi = b > ihi ? b : ihi;
                    olo = c < olo ? c : olo; ohi = e > ohi ? e : ohi;
                }
            }
            const bool ok = in && (lane == 0 || ((int64_t)ihi - ilo <= wmax && (int64_t)ohi - olo <= wmax));
            const unsigned long long m = ~__ballot(ok);
            const int cnt = m ? __builtin_ctzll(m) : 64;
            sg1 = __builtin_amdgcn_readfirstlane(__shfl(s1, cnt - 1, 64));
            sg0 = __builtin_amdgcn_readfirstlane(__shfl(d[0], 0, 64));
            t += cnt;
        }
        const int nq = (sg1 - sg0 + NTEAM - 1) / NTEAM;            // "quads" of three slices here
        const int q0 = (((local - rot) % per_xcd) + per_xcd) % per_xcd;
        rot = (rot + nq) % per_xcd;
        // ---- phase A (D = 64): all in-sweeps of the group, partial sums parked in U
        for (int q = q0; SPLIT && sg0 + NTEAM * q < sg1; q += per_xcd) {
            const int sl = sg0 + NTEAM * q + team;
            if (sl >= sg1) continue;
            const int ib = __builtin_amdgcn_readfirstlane(in_off[sl]);
            const int il = (__builtin_amdgcn_readfirstlane(in_off[sl + 1]) - ib) >> 4;
            if (il <= 0) continue;
            const int i16 = 4 * mem + hs;
            const int64_t n = (int64_t)sl * SLICE + i16;
            float acc[DL], ownQ[DL];
            load_vec<DL>(U + n * D + DL * p, acc);
            load_own_w<D, EX>(QS, n, p, ownQ);
            sweep_w<D, XP, EX>(in_nbr + ib, i16, il, (int)n_pad, PR, p, ownQ, w2, b2, acc);
            store_vec<DL>(U + n * D + DL * p, acc);
        }
        if constexpr (SPLIT) {
            // the sweep waves of the workgroup change phase TOGETHER (an LDS counter, the matrix-core
            // waves are not involved): teams that ran ahead into the other table while their mates still
            // gathered from this one undid the split's effect (L2 hit rate 52 % instead of 67 %)
            ++ngrp;
            if (lane == 0) bump(pha);
            for (int spins = 0; spins < kSpinLimit && ld(pha) < NSW * ngrp; ++spins) __builtin_amdgcn_s_sleep(2);
        }
        // ---- phase B: out-sweeps (and the in-sweeps when the group is not split), then publish
        for (int q = q0; sg0 + NTEAM * q < sg1; q += per_xcd, seq += NTEAM) {
            const int sl = sg0 + NTEAM * q + team;
            const bool on = sl < sg1;
            float acc[DL];
            if (on) {
                const int ob = __builtin_amdgcn_readfirstlane(out_off[sl]);
                const int ol = (__builtin_amdgcn_readfirstlane(out_off[sl + 1]) - ob) >> 4;
                const int i16 = 4 * mem + hs;
                const int64_t n = (int64_t)sl * SLICE + i16;
                float ownP[DL];
                load_vec<DL>(U + n * D + DL * p, acc);
                if constexpr (!SPLIT) {
                    const int ib = __builtin_amdgcn_readfirstlane(in_off[sl]);
                    const int il = (__builtin_amdgcn_readfirstlane(in_off[sl + 1]) - ib) >> 4;
                    float ownQ[DL];
                    load_own_w<D, EX>(QS, n, p, ownQ);
                    load_own_w<D, EX>(PR, n, p, ownP);
                    sweep_w<D, XP, EX>(in_nbr + ib, i16, il, (int)n_pad, PR, p, ownQ, w2, b2, acc);
                } else {
                    load_own_w<D, EX>(PR, n, p, ownP);
                }
                sweep_w<D, XP, EX>(out_nbr + ob, i16, ol, (int)n_pad, QS, p, ownP, w2, b2, acc);
#pragma unroll
                for (int i = 0; i < DL; ++i) acc[i] = tanh_f(acc[i]);
            }
            // the slot of sequence number seq: free once its previous generation has been taken
            const int slot = seq % NSLOT, gen = seq / NSLOT;
            for (int spins = 0; spins < kSpinLimit && ld(cons + slot) < gen; ++spins) __builtin_amdgcn_s_sleep(2);
            float *tr = ring + slot * 16 * B::tr_stride;
            if (on) {
                const int i16 = 4 * mem + hs;
                const int64_t n = (int64_t)sl * SLICE + i16;
                store_vec<DL>(tr + i16 * B::tr_stride + DL * p, acc);
                if (p < F) tr[i16 * B::tr_stride + D + p] = X[n * F + p];
            }
            if (mem == 0 && lane == 0) __hip_atomic_store(meta + slot, on ? sl : -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // this wave's rows have landed in the slot
            if (lane == 0) bump(prod + slot);
        }
        // (a second meeting point here, before the next group's phase A: 6 % less traffic, 2 % more time)
        if constexpr (!SPLIT) {
            // unsplit groups (bf16 rows, D < 64): the sweep waves still enter a GROUP together, so that the
            // workgroup's teams do not spread over several levels' tables
            ++ngrp;
            if (lane == 0) bump(pha);
            for (int spins = 0; spins < kSpinLimit && ld(pha) < NSW * ngrp; ++spins) __builtin_amdgcn_s_sleep(2);
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0) bump(fin);
}

// ---------------------------------------------------------------------------------------------
// exact-fp32 matrix-core products of k_iter2 (D = 8): v_mfma_f32_16x16x4_f32 is a k-ordered fmaf
// chain, so moving the per-hit MLPs there changes no tolerance - and frees the vector pipe.
// ---------------------------------------------------------------------------------------------
// Orientation as in mfma_tail: rows = 16 output positions (A = weights), columns = 16 hits
// (B = activations); lane l supplies ONE input value per k-step for hit l & 15 (input slot
// l >> 4 of that step) and receives output rows 4 (l >> 4) .. + 3.  With the 8 hit features
// living as rows 0..7 of such a result (lane groups 0 and 1), the slot -> input map is
//   step s < 4:  group 0: feature s, group 1: feature 4 + s, group 2: x[s], group 3: x[4 + s]
//   step 4:      group g: x[8 + g]                                   (F = 11 only)
template <int F, int D>
struct MT {
    static_assert(D == 8, "fp32 matrix-core tables are laid out for D = 8");
    static constexpr int C = F + D;
    // F <= 4: the 8 hidden features + F inputs fit 3 k-steps of 4 slots if the last feature of each
    // half (3 and 7) moves to lane group 2 (one ds_bpermute each) - 3 products per record table
    // instead of 4
    static constexpr bool PACK = F <= 4;
    static constexpr int NS = PACK ? 3 : F <= 8 ? 4 : 5;   // k-steps of a record product
    static constexpr int o_h0 = 0;                     // [64]       input network, 1 step (FIRST)
    static constexpr int o_t = 64;                     // 3 x [NS][64]: PR, QS, U  (LAST: P|Q, -, -)
    static constexpr int t_sz = NS * 64;
    static constexpr int o_w4 = o_t + 3 * t_sz;        // [2][64]: slot (s, g) = feature 2 g + s
    static constexpr int o_b = o_w4 + 128;             // biases: h0, T0, T1, T2, W4 (16 each)
    static constexpr int total = o_b + 5 * 16;
    // input index k in [0, C) of slot (s, g), or -1
    static constexpr int slot_k(int s, int g)
    {
        if (PACK) {
            if (g == 0) return s;
            if (g == 1) return 4 + s;
            if (g == 2) return s == 0 ? 3 : s == 1 ? 7 : (0 < F ? D : -1);
            return 1 + s < F ? D + 1 + s : -1;
        }
        if (s < 4) {
            if (g == 0) return s;
            if (g == 1) return 4 + s;
            const int xi = (g == 2) ? s : 4 + s;
            return xi < F ? D + xi : -1;
        }
        return 8 + g < F ? D + 8 + g : -1;
    }
};

template <int F, int D>
constexpr int mt_total()
{
    if constexpr (D == 8) return MT<F, D>::total;
    else return 0;
}

// one entry of the tables above from the raw weights; `last`: T0 holds [P(8) | Q(8)]
template <int F, int D>
__device__ __forceinline__ float mt_entry(const gnn_params_t &p, int i, bool last)
{
    using M = MT<F, D>;
    constexpr int C = F + D;
    if (i < 64) {                                               // input network
        const int row = i & 15, g = i >> 4;
        return (row < 8 && g < F && g < 4) ? p.Win[row * F + g] : 0.0f;
    }
    if (i < M::o_w4) {                                          // record products
        const int j = i - M::o_t, T = j / M::t_sz, st = (j % M::t_sz) >> 6, row = j & 15, g = (j >> 4) & 3;
        const int k = M::slot_k(st, g);
        if (k < 0) return 0.0f;
        if (last) {
            if (T > 0) return 0.0f;
            return kTwoLog2e * p.W1[(row & 7) * 2 * C + (row >> 3) * C + k];
        }
        if (T == 2) return row < 8 ? p.W3[row * 3 * C + 2 * C + k] : 0.0f;          // U
        const int c = row >> 2, w = row & 3, d = 2 * c + (w & 1);                 // [P|R] / [Q|S] chunk order
        return (w < 2) ? kTwoLog2e * p.W1[d * 2 * C + T * C + k] : p.W3[d * 3 * C + T * C + k];
    }
    if (i < M::o_b) {                                           // W4
        const int j = i - M::o_w4, st = j >> 6, row = j & 15, g = (j >> 4) & 3;
        return row < 8 ? p.W4[row * D + 2 * g + st] : 0.0f;
    }
    const int j = i - M::o_b, which = j >> 4, row = j & 15;     // biases
    switch (which) {
    case 0: return row < 8 ? p.bin[row] : 0.0f;
    case 1:
        if (last) return row < 8 ? kTwoLog2e * p.b1[row] : 0.0f;
        return (row & 3) < 2 ? kTwoLog2e * p.b1[2 * (row >> 2) + (row & 1)] : 0.0f;
    case 2: return 0.0f;
    case 3: return (!last && row < 8) ? p.b3[row] : 0.0f;
    default: return row < 8 ? p.b4[row] : 0.0f;
    }
}

// ---------------------------------------------------------------------------------------------
// k_iter2: persistent, phase-split variant of k_iter for batches whose tiles all run in LDS mode
// ---------------------------------------------------------------------------------------------
// k_iter holds both record windows of a tile in LDS (128 KB at 1000 hits/level, D = 8), so a CU
// runs one workgroup whose memory phases (window staging, first prefetch, launch) and compute
// phases simply add up (measured: 0.19 ms VALU issue + 0.24 ms memory = 0.43 ms).  Here ONE
// workgroup per CU walks its tiles (tile = blockIdx.x + k * gridDim.x, fixed trip count, no
// inter-workgroup communication) and splits every tile in two phases:
//   phase A: all in-sweeps of the tile (needs only the PR window, bufA); accumulators of the up to
//            4 slices a wave owns stay in registers;
//   phase B: all out-sweeps (QS window, bufB), hit updates and stores.
// While phase A computes, the tile's QS window is in flight (asm loads into registers, written
// to bufB at the end of the phase); while phase B computes, the NEXT tile's PR window is in
// flight.  So window staging, list prefetch and stores all run under VALU work.
template <int F, int D, bool LAST, bool XP, bool FIRST>
__global__ __launch_bounds__(1024) void k_iter2(
    const float *__restrict__ X, const float *__restrict__ table, gnn_params_t p,
    float *__restrict__ table_out,
    const int32_t *__restrict__ tiles, const int32_t *__restrict__ in_off,
    const int32_t *__restrict__ in_off16, const int32_t *__restrict__ in_nbr16,
    const int32_t *__restrict__ out_off, const int32_t *__restrict__ out_off16,
    const int32_t *__restrict__ out_nbr16, const int32_t *__restrict__ sched_a,
    const int32_t *__restrict__ sched_b, const float *__restrict__ PR,
    const float *__restrict__ QS, float *__restrict__ U, float *__restrict__ PRn,
    float *__restrict__ QSn, float *__restrict__ Pc, float *__restrict__ Qc, int64_t n_pad,
    int n_tiles, int capA, int capB, int xbuf_floats)
{
    using L = TL<F, D>;
    constexpr int d4 = L::d4, NT = 1024, NWV = NT / 64, NC = 3;   // NC words = 24 list steps
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *lds = smem, *bufA = smem + L::total, *bufB = bufA + (int64_t)capA * 2 * D;
    float *xbuf = bufB + (int64_t)capB * 2 * D;      // (end of the window buffers)
    const int tid = threadIdx.x, lane = tid & 63, q = lane & 3, i16 = lane >> 2;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    if constexpr (FIRST) {
        // first iteration: there are no records yet.  The workgroup packs the weight table from
        // the raw weights (workgroup 0 publishes it and the NULL rows for the launches that
        // follow) and computes every record it needs from X (see compute_window).
        for (int idx = tid; idx < L::total; idx += NT) {
            const float v = table_entry<F, D>(p, idx);
            lds[idx] = v;
            if (blockIdx.x == 0) table_out[idx] = v;
        }
        if (blockIdx.x == 0)
            write_null_rows<F, D, XP>(p, PRn, const_cast<float *>(PR), QSn, const_cast<float *>(QS),
                                      U, Pc, Qc, n_pad);
    } else {
        stage4<NT>(table, lds, L::total / 4);        // visible after the first barrier
    }
    float *mt = xbuf + (FIRST ? xbuf_floats : 0);    // D = 8: fp32 matrix-core tables (MT)
    if constexpr (D == 8)
        for (int i = tid; i < MT<F, D>::total; i += NT) mt[i] = mt_entry<F, D>(p, i, LAST);
    // XCD-aware start tile: workgroups are dealt round-robin to the 8 XCDs, so XCD x takes a
    // CONTIGUOUS run of the grid's tiles.  Neighbouring tiles are neighbouring levels of one
    // graph: level l's PR rows are the in-window of tile l+1 and the own-P rows of tile l, its QS
    // rows the out-window of tile l-1 and the own-Q rows of tile l - all in one L2 now.
    int tile = (gridDim.x & 7) ? (int)blockIdx.x
                               : (int)((blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3));
    if (tile >= n_tiles) return;

    struct Desc { int s_begin, s_end, in_lo, in_cnt, out_lo, out_cnt, sbase; };
    auto load_desc = [&](int t) {
        const int32_t *td = tiles + (int64_t)t * DESC;
        Desc d;
        d.s_begin = td[0]; d.s_end = td[1]; d.in_lo = td[2]; d.in_cnt = td[3];
        d.out_lo = td[4]; d.out_cnt = td[5]; d.sbase = td[7];
        return d;
    };
    // which slice this wavefront takes in round r (both phases): the plan's cost-balanced schedule
    auto slice_a = [&](const Desc &d, int r) { return sched_a[d.sbase + r * NWV + wv]; };

    // ---- window staging by LDS-DMA (global_load_lds_dwordx4: no VGPRs, no ds_write) -------------
    // One wave-instruction moves 64 lanes x 16 B = 1 KiB to a wave-uniform LDS address; waves
    // take 1-KiB pieces round-robin.  The last piece may run up to 1 KiB past the window: the
    // buffers are sized in whole pieces plus one (NULL record) and the workspace rows are padded.
    auto stage_issue = [&](const float *src, float *buf, int nrec) {
        const int pieces = (nrec * 2 * D + 255) / 256;
        unsigned lb = (unsigned)lane * 16u;         // opaque: the 64-bit lane address is rebuilt per
        asm volatile("" : "+v"(lb));                // call instead of living in a register pair
        for (int c = wv; c < pieces; c += NWV)
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void *)(
                    reinterpret_cast<const char *>(src + (int64_t)c * 256) + lb),
                (__attribute__((address_space(3))) void *)(buf + c * 256), 16, 0, 0);
    };
    // NULL record of a window (write_null_rows: P = scaled b1, Q = R = S = 0; 2^x of that in
    // exp-product mode), rebuilt from the LDS weight table: no per-lane global address to keep
    // Written by the wave that issued the window's last DMA piece (the piece that may cover the
    // NULL slot): that wave has waited for its own DMA, the others' pieces lie below the slot.
    auto put_null = [&](float *buf, int nrec, auto which) {
        constexpr int M = decltype(which)::value;                      // 0: [P|R], 1: [Q|S]
        const int pieces = (nrec * 2 * D + 255) / 256;
        if (wv == (pieces > 0 ? (pieces - 1) % NWV : 0) && lane < 4) {
            int l = lane;                       // opaque: addresses and constants of this rare
            asm volatile("" : "+v"(l));         // path are built here, not kept in registers
#pragma unroll
            for (int i = 0; i < d4; ++i) {
                float v = (M == 0) ? lds[l * L::stride + L::o_m + i] : 0.0f, z = 0.0f;
                if (XP) v = (M == 0) ? __builtin_amdgcn_exp2f(v) : 1.0f;
                asm volatile("" : "+v"(v), "+v"(z));
                buf[nrec * 2 * D + l * 2 * d4 + i] = v;
                buf[nrec * 2 * D + l * 2 * d4 + d4 + i] = z;
            }
        }
    };
    using WinA = std::integral_constant<int, 0>;
    using WinB = std::integral_constant<int, 1>;
    auto stage_commit = [&](float *buf, int nrec, auto which) {
        a_wait_all();                                                  // DMA pieces have landed
        put_null(buf, nrec, which);
    };

    // ---- FIRST: windows computed from X ------------------------------------------------------------
    // The X rows of a window (cnt * F floats, 12 KB) arrive by LDS-DMA one phase ahead, like the
    // record windows of the later iterations; between the phases every quad turns rows into
    // records: H0 = [tanh(Win x + bin) | x] (model.py:144-146), then [P | R] (M = 0) or [Q | S]
    // (M = 1): at D = 8 as two small fp32 matrix-core products per 16 hits, else with the
    // role_gemv blocks k_input4 / emit_now use.
    // The X rows land in the TAIL of the window buffer they will be turned into (the buffer is idle
    // while they arrive, exactly like the record windows of the later iterations): no buffer of
    // their own.  compute_window turns them into records 256 hits at a time, in ascending order;
    // a record is >= 15 bytes longer than an X row, so the records of one round never reach the X
    // rows of a later round, and inside a round every wave reads its rows before anyone writes.
    auto x_image = [&](float *buf, int cap) { return buf + cap * 2 * D - (cap * F + 63) / 64 * 64; };
    auto xstage_issue = [&](int lo, int cnt, float *buf, int cap) {
        const int pieces = (cnt * F + 63) / 64;                        // 256-byte pieces
        float *xb = x_image(buf, cap);
        unsigned lb = (unsigned)lane * 4u;
        asm volatile("" : "+v"(lb));
        for (int c = wv; c < pieces; c += NWV)
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void *)(
                    reinterpret_cast<const char *>(X + (int64_t)lo * F + c * 64) + lb),
                (__attribute__((address_space(3))) void *)(xb + c * 64), 4, 0, 0);
    };
    auto h0_of = [&](const float *wl, const float *x, float *hn) {
        float hl[d4];
        role_gemv<d4, 0, F>(wl + L::o_in, x, x, hl);
#pragma unroll
        for (int i = 0; i < d4; ++i) hl[i] = tanh_f(hl[i]);
        quad_allgather<d4>(hl, hn);
    };
    auto compute_window = [&](float *buf, int cnt, int cap, auto which) {
        constexpr int M = decltype(which)::value;
        static_assert(2 * D * 4 - F * 4 >= 16, "a record must be longer than an X row (see x_image)");
        const float *xb = x_image(buf, cap);
        const int rounds = (cnt + NT / 4 - 1) / (NT / 4);          // 256 hits per round
        // rounds whose records end below the X image need no barrier between reading and writing
        const int x_off = (int)(xb - buf);
        auto round_fence = [&](int r) {
            if ((r + 1) * (NT / 4) * 2 * D > x_off) __syncthreads();
        };
        // the NULL record sits right behind the last record: when that slot reaches into the X
        // image (a full last round of a window whose records end exactly at the image), the wave
        // that writes it must wait until every wave has read its X rows of the last round
        // (workgroup-uniform condition)
        auto null_fence = [&]() {
            if ((cnt + 1) * 2 * D > x_off) __syncthreads();
        };
        a_wait_all();                                  // this wave's DMA pieces have landed
        __syncthreads();                               // ... and everybody else's
        if constexpr (D == 8) {
            // On the matrix cores, exact fp32 (v_mfma_f32_16x16x4_f32 is a k-ordered fmaf chain):
            // rows = the 16 floats of a record in LDS position order, columns = 16 window hits.
            // Lane l supplies input slot l >> 4 of hit l & 15 per k-step and receives positions
            // 4 (l >> 4) .. + 3 of that hit's record - one 16-byte LDS store.  4 MFMAs per 16 hits
            // instead of ~100 vector instructions per lane.
            typedef float f4v __attribute__((ext_vector_type(4)));
            using MTL = MT<F, D>;
            const float *mb = mt + MTL::o_b;
            const int hit = lane & 15, g = lane >> 4;
            const float a0 = mt[MTL::o_h0 + lane];
            static_assert(MTL::PACK, "the fused first launch exists for F <= 3 only");
            float ar[MTL::NS];
#pragma unroll
            for (int st = 0; st < MTL::NS; ++st) ar[st] = mt[MTL::o_t + MTL::t_sz * M + 64 * st + lane];
            const f4v bias0 = *reinterpret_cast<const f4v *>(mb + 4 * g);
            const f4v biasr = *reinterpret_cast<const f4v *>(mb + 16 + 16 * M + 4 * g);
            for (int r = 0; r < rounds; ++r) {
                const int h = (r * NWV + wv) * 16 + hit;
                const bool live = h < cnt;
                float xs[F];
#pragma unroll
                for (int k = 0; k < F; ++k) xs[k] = live ? xb[h * F + k] : 0.0f;
                round_fence(r);                        // all X rows of this round are in registers
                float b0 = 0.0f;
#pragma unroll
                for (int k = 0; k < F; ++k) b0 = (g == k) ? xs[k] : b0;
                f4v hq = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, bias0, 0, 0, 0);
                const float hh[4] = {tanh_f(hq.x), tanh_f(hq.y), tanh_f(hq.z), tanh_f(hq.w)};
                // features 3 and 7 (held by lane groups 0 and 1) for lane group 2, see MT::slot_k
                const float h3a = __int_as_float(__builtin_amdgcn_ds_bpermute(hit << 2, __float_as_int(hh[3])));
                const float h3b = __int_as_float(__builtin_amdgcn_ds_bpermute((16 + hit) << 2, __float_as_int(hh[3])));
                f4v c = biasr;
#pragma unroll
                for (int st = 0; st < MTL::NS; ++st) {
                    float b = hh[st];
                    if (g == 2) b = st == 0 ? h3a : st == 1 ? h3b : xs[0];
                    if (g == 3) b = 1 + st < F ? xs[1 + st < F ? 1 + st : 0] : 0.0f;
                    c = __builtin_amdgcn_mfma_f32_16x16x4f32(ar[st], b, c, 0, 0, 0);
                }
                if constexpr (XP) {                    // positions 4g, 4g+1 are P (Q) entries
                    c.x = __builtin_amdgcn_exp2f(c.x);
                    c.y = __builtin_amdgcn_exp2f(c.y);
                }
                if (live) *reinterpret_cast<f4v *>(buf + h * 2 * D + 4 * g) = c;
            }
            null_fence();
            put_null(buf, cnt, which);
            return;
        }
        for (int r = 0; r < rounds; ++r) {
            const int h = r * (NT / 4) + (tid >> 2);
            const bool live = h < cnt;
            int woff = q * L::stride;
            asm volatile("" : "+v"(woff));
            const float *wl = lds + woff;
            float x[F], hn[D], rec[2 * d4];
#pragma unroll
            for (int k = 0; k < F; ++k) x[k] = live ? xb[h * F + k] : 0.0f;
            round_fence(r);                            // all X rows of this round are in registers
            if (!live) continue;
            h0_of(wl, x, hn);
            if constexpr (d4 == 2) {
                role_gemv_burst<D, F>(wl + L::o_m + (2 * M) * L::m_st, hn, x, rec);
                role_gemv_burst<D, F>(wl + L::o_m + (2 * M + 1) * L::m_st, hn, x, rec + d4);
            } else {
                role_gemv<d4, D, F>(wl + L::o_m + (2 * M) * L::m_st, hn, x, rec);
                role_gemv<d4, D, F>(wl + L::o_m + (2 * M + 1) * L::m_st, hn, x, rec + d4);
            }
            if constexpr (XP)
#pragma unroll
                for (int i = 0; i < d4; ++i) rec[i] = __builtin_amdgcn_exp2f(rec[i]);
            store_vec<2 * d4>(buf + h * 2 * D + q * 2 * d4, rec);
        }
        null_fence();
        put_null(buf, cnt, which);
    };

    // ---- per-slice prefetch (one slice ahead), split by phase -------------------------------------
    struct PreA0 { int len; int c[NC]; AVec<d4> Q, U; };            // in-list, own Q, acc init
    // wide X rows (F = 11, the muon schema) are not prefetched a slice ahead - two sets of F
    // registers do not fit - but read at the start of the round that uses them (XPRE = false)
    constexpr bool XPRE = F <= 4;
    struct PreB0 { int len; int c[NC]; AVec<d4> P; AVec<XPRE ? F : 0> x; };   // out-list, own P, X row
    struct PreX { int len; int c[NC]; AVec<F> x; };                  // FIRST: own values come from x
    using PreA = std::conditional_t<FIRST, PreX, PreA0>;
    using PreB = std::conditional_t<FIRST, PreX, PreB0>;
    // Addresses are a wave-uniform base (slice-dependent, SGPRs) plus one of four per-lane byte
    // offsets; nothing lane-dependent is 64 bits wide.
    const unsigned lo_list = (unsigned)(q * SLICE + i16) * 4u;                 // packed list word
    const unsigned lo_rec = (unsigned)(i16 * 2 * D + q * 2 * d4) * 4u;         // piece of a record
    const unsigned lo_vec = (unsigned)(i16 * D + q * d4) * 4u;                 // piece of a D-row
    const unsigned lo_x = (unsigned)(i16 * F) * 4u;                            // X row
    auto prefetchA = [&](PreA &p, int slice_) {
        const int slice = __builtin_amdgcn_readfirstlane(slice_);
        p.len = (__builtin_amdgcn_readfirstlane(in_off[slice + 1]) -
                 __builtin_amdgcn_readfirstlane(in_off[slice])) >> 4;
        const int32_t *li = in_nbr16 + __builtin_amdgcn_readfirstlane(in_off16[slice]);
#define GNN_PF(C_) if (8 * C_ < p.len) a_load_i32_s<C_ * 4 * SLICE * 4>(p.c[C_], li, lo_list);
        GNN_PF(0) GNN_PF(1) GNN_PF(2)
#undef GNN_PF
        const int64_t n0 = (int64_t)slice * SLICE;
        if constexpr (FIRST) {
            p.x.load_s(X + n0 * F, lo_x);
        } else {
            p.Q.load_s(QS + n0 * 2 * D, lo_rec);
            p.U.load_s(U + n0 * D, lo_vec);
        }
    };
    auto prefetchB = [&](PreB &p, int slice_) {
        const int slice = __builtin_amdgcn_readfirstlane(slice_);
        p.len = (__builtin_amdgcn_readfirstlane(out_off[slice + 1]) -
                 __builtin_amdgcn_readfirstlane(out_off[slice])) >> 4;
        const int32_t *lo = out_nbr16 + __builtin_amdgcn_readfirstlane(out_off16[slice]);
#define GNN_PF(C_) if (8 * C_ < p.len) a_load_i32_s<C_ * 4 * SLICE * 4>(p.c[C_], lo, lo_list);
        GNN_PF(0) GNN_PF(1) GNN_PF(2)
#undef GNN_PF
        static_assert(NC == 3, "prefetch is written out for 3 words (24 steps)");
        const int64_t n0 = (int64_t)slice * SLICE;
        if constexpr (!FIRST) p.P.load_s(PR + n0 * 2 * D, lo_rec);
        if constexpr (XPRE || FIRST) p.x.load_s(X + n0 * F, lo_x);
    };
    auto arriveA = [&](PreA &p) {
        a_wait_all();
        a_fence(p.c);
        if constexpr (FIRST) {
            p.x.fence();
        } else {
            p.Q.fence();
            p.U.fence();
        }
    };
    auto arriveB = [&](PreB &p) {
        a_wait_all();
        a_fence(p.c);
        if constexpr (!FIRST) p.P.fence();
        if constexpr (XPRE || FIRST) p.x.fence();
    };

    // Partial sums of the (at most MAXR) slices a wavefront owns in a tile stay in registers
    // across the phase barrier.  Round indices are wave-uniform, so a scalar switch selects
    // the slot; the sweep code itself exists once.
    constexpr int MAXR = 5;                    // tile_hits 1280 = 80 slices = 5 rounds of 16 waves
    // (five separate arrays and selects: an indexed 2-D private array is demoted to scratch)
    float ac0[d4], ac1[d4], ac2[d4], ac3[d4], ac4[d4];
    auto acc_put = [&](int r, const float *a) {
#pragma unroll
        for (int i = 0; i < d4; ++i) {
            ac0[i] = (r == 0) ? a[i] : ac0[i];
            ac1[i] = (r == 1) ? a[i] : ac1[i];
            ac2[i] = (r == 2) ? a[i] : ac2[i];
            ac3[i] = (r == 3) ? a[i] : ac3[i];
            ac4[i] = (r >= 4) ? a[i] : ac4[i];
        }
    };
    auto acc_get = [&](int r, float *a) {
#pragma unroll
        for (int i = 0; i < d4; ++i)
            a[i] = (r == 0) ? ac0[i] : (r == 1) ? ac1[i] : (r == 2) ? ac2[i] : (r == 3) ? ac3[i] : ac4[i];
    };
    static_assert(MAXR == 5, "accumulator slots are written out for 5 rounds");
#pragma unroll
    for (int i = 0; i < d4; ++i) ac0[i] = ac1[i] = ac2[i] = ac3[i] = ac4[i] = 0.0f;

    // ---- prologue: first tile's PR window, first slice of phase A ---------------------------------
    Desc d = load_desc(tile);
    PreA a_cur, a_nxt;
    PreB b_cur, b_nxt;
    if constexpr (FIRST) {
        xstage_issue(d.in_lo, d.in_cnt, bufA, capA);
        if (slice_a(d, 0) >= 0) prefetchA(a_cur, slice_a(d, 0));
        compute_window(bufA, d.in_cnt, capA, WinA{});   // barrier inside: table visible
    } else {
        stage_issue(PR + (int64_t)d.in_lo * 2 * D, bufA, d.in_cnt);
        if (slice_a(d, 0) >= 0) prefetchA(a_cur, slice_a(d, 0));
        __syncthreads();                               // weight table visible (put_null reads it)
        stage_commit(bufA, d.in_cnt, WinA{});
    }
    if (slice_a(d, 0) >= 0) arriveA(a_cur);

    float w2[d4];
    for (;;) {
        __syncthreads();                               // bufA (and, first time, the table) visible
#pragma unroll
        for (int i = 0; i < d4; ++i) w2[i] = lds[q * L::stride + L::o_w2 + i];
        const float b2 = lds[L::o_b2];
        const int rounds = (d.s_end - d.s_begin + NWV - 1) / NWV;
        // ================= phase A: in-sweeps; QS window of this tile in flight ==================
        if constexpr (FIRST)
            xstage_issue(d.out_lo, d.out_cnt, bufB, capB);   // bufB is idle during phase A
        else
            stage_issue(QS + (int64_t)d.out_lo * 2 * D, bufB, d.out_cnt);
        // The last round is peeled (LR = true): it requests the first slice of phase B instead of
        // a next in-slice.  Written as one generic lambda so b_cur is DEFINED only there: a
        // conditional assignment inside the loop would keep it live across all rounds.
        auto roundA = [&](int r, auto lr) {
            constexpr bool LR = decltype(lr)::value;
            const int slice = slice_a(d, r);
            const int next = LR ? -1 : slice_a(d, r + 1);
            // first slice of phase B (same wave, same slices as phase A): lists, own P, X row
            if constexpr (LR) {
                if (slice_a(d, 0) >= 0) prefetchB(b_cur, slice_a(d, 0));
            } else {
                if (next >= 0) prefetchA(a_nxt, next);
            }
            float acc[d4];
            if (slice >= 0) {
                float Qn[d4];
                if constexpr (FIRST) {                 // own Q and U from the hit's X row
                    int woff = q * L::stride;
                    asm volatile("" : "+v"(woff));
                    const float *wl = lds + woff;
                    float xo[F], h0[D];
                    a_cur.x.get(xo);
                    h0_of(wl, xo, h0);
                    role_gemv<d4, D, F>(wl + L::o_m + 2 * L::m_st, h0, xo, Qn);
                    if constexpr (XP)
#pragma unroll
                        for (int i = 0; i < d4; ++i) Qn[i] = __builtin_amdgcn_exp2f(Qn[i]);
                    role_gemv<d4, D, F>(wl + L::o_m + 4 * L::m_st, h0, xo, acc);
                } else {
                    a_cur.U.get(acc);
                    a_cur.Q.get(Qn);
                }
                const int len = __builtin_amdgcn_readfirstlane(a_cur.len);
                sweep16<D, NC, XP, !FIRST>(a_cur.c, in_nbr16, in_off16, slice, i16, len, bufA, q, Qn, w2, b2, acc);
            }
            if constexpr (!LR) {
                if (next >= 0) {
                    arriveA(a_nxt);
                    a_cur = a_nxt;
                }
            }
            if (slice >= 0) acc_put(r, acc);
        };
        for (int r = 0; r + 1 < rounds; ++r) roundA(r, std::false_type{});
        roundA(rounds - 1, std::true_type{});
        if constexpr (FIRST)
            compute_window(bufB, d.out_cnt, capB, WinB{});
        else
            stage_commit(bufB, d.out_cnt, WinB{});   // also makes b_cur readable (vmcnt 0)
        if (slice_a(d, 0) >= 0) arriveB(b_cur);
        __syncthreads();                               // bufB visible, bufA free
        // ================= phase B: out-sweeps, hit update, stores; next PR window in flight ======
        const int tnext = tile + gridDim.x;
        Desc dn = d;
        if (tnext < n_tiles) {
            dn = load_desc(tnext);
            if constexpr (FIRST)
                xstage_issue(dn.in_lo, dn.in_cnt, bufA, capA);
            else
                stage_issue(PR + (int64_t)dn.in_lo * 2 * D, bufA, dn.in_cnt);
        }
        auto roundB = [&](int r, auto lr) {
            constexpr bool LR = decltype(lr)::value;
            const int slice = slice_a(d, r);
            const int next = LR ? -1 : slice_a(d, r + 1);
            if constexpr (LR) {      // first slice of the next tile's phase A
                if (tnext < n_tiles && slice_a(dn, 0) >= 0) prefetchA(a_cur, slice_a(dn, 0));
            } else {
                if (next >= 0) prefetchB(b_nxt, next);
            }
            float xv[F], acc[d4];
            if (slice >= 0) {
                float Pn[d4];
                acc_get(r, acc);
                if constexpr (XPRE || FIRST) {
                    b_cur.x.get(xv);
                } else {                                   // plain loads, consumed after the sweep
                    const float *xr = X + ((int64_t)__builtin_amdgcn_readfirstlane(slice) * SLICE + i16) * F;
#pragma unroll
                    for (int k = 0; k < F; ++k) xv[k] = xr[k];
                }
                if constexpr (FIRST) {                 // own P from the hit's X row
                    int woff = q * L::stride;
                    asm volatile("" : "+v"(woff));
                    const float *wl = lds + woff;
                    float h0[D];
                    h0_of(wl, xv, h0);
                    role_gemv<d4, D, F>(wl + L::o_m + 0 * L::m_st, h0, xv, Pn);
                    if constexpr (XP)
#pragma unroll
                        for (int i = 0; i < d4; ++i) Pn[i] = __builtin_amdgcn_exp2f(Pn[i]);
                } else {
                    b_cur.P.get(Pn);
                }
                const int len = __builtin_amdgcn_readfirstlane(b_cur.len);
                sweep16<D, NC, XP, false>(b_cur.c, out_nbr16, out_off16, slice, i16, len, bufB, q, Pn, w2, b2, acc);
            }
            // the next slice's registers arrive here (in flight during the sweep); doing it before
            // the hit update keeps the two register sets from overlapping with the MLP's
            if constexpr (!LR) {
                if (next >= 0) {
                    arriveB(b_nxt);
                    b_cur = b_nxt;
                }
            }
            if constexpr (D == 8) {
                if (slice >= 0) {
                    // Hit update and records on the matrix cores, exact fp32 (MT): the wave's 16
                    // hits are the 16 columns.  tanh(acc) moves from the sweep's lane layout
                    // (lane = hit * 4 + q: dims 2q, 2q+1) to the product's (lane = g * 16 + hit:
                    // slots of group g) with two ds_bpermute, X with F more; everything after
                    // that, stores included, stays in the product's layout.
                    using MTL = MT<F, D>;
                    typedef float f4v __attribute__((ext_vector_type(4)));
                    const float *mb = mt + MTL::o_b;
                    const int hit = lane & 15, g = lane >> 4;
                    const int from = ((hit << 2) | g) << 2;          // byte index of the source lane
                    const float q0 = __int_as_float(__builtin_amdgcn_ds_bpermute(from, __float_as_int(tanh_f(acc[0]))));
                    const float q1 = __int_as_float(__builtin_amdgcn_ds_bpermute(from, __float_as_int(tanh_f(acc[1]))));
                    float xm[F];
#pragma unroll
                    for (int k = 0; k < F; ++k)
                        xm[k] = __int_as_float(__builtin_amdgcn_ds_bpermute(hit << 4, __float_as_int(xv[k])));
                    // H' = tanh(W4 q + b4): slot (s, g) = feature 2 g + s          (model.py:94-98,125)
                    f4v h = *reinterpret_cast<const f4v *>(mb + 64 + 4 * g);
                    h = __builtin_amdgcn_mfma_f32_16x16x4f32(mt[MTL::o_w4 + lane], q0, h, 0, 0, 0);
                    h = __builtin_amdgcn_mfma_f32_16x16x4f32(mt[MTL::o_w4 + 64 + lane], q1, h, 0, 0, 0);
                    const float hh[4] = {tanh_f(h.x), tanh_f(h.y), tanh_f(h.z), tanh_f(h.w)};
                    // this lane's input of step s of the record products (MT::slot_k)
                    float h3a = 0.0f, h3b = 0.0f;       // PACK: features 3 and 7 for lane group 2
                    if constexpr (MTL::PACK) {
                        h3a = __int_as_float(__builtin_amdgcn_ds_bpermute(hit << 2, __float_as_int(hh[3])));
                        h3b = __int_as_float(__builtin_amdgcn_ds_bpermute((16 + hit) << 2, __float_as_int(hh[3])));
                    }
                    auto slot_in = [&](int st) {
                        if constexpr (MTL::PACK) {
                            if (g < 2) return hh[st];
                            if (g == 2) return st == 0 ? h3a : st == 1 ? h3b : (0 < F ? xm[0] : 0.0f);
                            float v = 0.0f;
#pragma unroll
                            for (int k = 1; k < F; ++k) v = (k == 1 + st) ? xm[k] : v;
                            return v;
                        }
                        if (st < 4) {
                            if (g < 2) return hh[st];
                            float v = 0.0f;
#pragma unroll
                            for (int k = 0; k < F; ++k)
                                if (k == st || k == 4 + st) v = (g == (k < 4 ? 2 : 3) && (k & 3) == st) ? xm[k] : v;
                            return v;
                        }
                        float v = 0.0f;
#pragma unroll
                        for (int k = 8; k < F; ++k) v = (g == k - 8) ? xm[k] : v;
                        return v;
                    };
                    float bin_[MTL::NS];
#pragma unroll
                    for (int st = 0; st < MTL::NS; ++st) bin_[st] = slot_in(st);
                    const int64_t n = (int64_t)__builtin_amdgcn_readfirstlane(slice) * SLICE + hit;
                    constexpr int NP = LAST ? 1 : 3;
#pragma unroll
                    for (int T = 0; T < NP; ++T) {
                        f4v c = *reinterpret_cast<const f4v *>(mb + 16 * (1 + T) + 4 * g);
#pragma unroll
                        for (int st = 0; st < MTL::NS; ++st)
                            c = __builtin_amdgcn_mfma_f32_16x16x4f32(mt[MTL::o_t + MTL::t_sz * T + 64 * st + lane],
                                                                     bin_[st], c, 0, 0, 0);
                        if constexpr (LAST) {          // rows 0..7 = P, 8..15 = Q (compact rows)
                            if constexpr (XP) {
                                c.x = __builtin_amdgcn_exp2f(c.x); c.y = __builtin_amdgcn_exp2f(c.y);
                                c.z = __builtin_amdgcn_exp2f(c.z); c.w = __builtin_amdgcn_exp2f(c.w);
                            }
                            *reinterpret_cast<f4v *>((g < 2 ? Pc : Qc) + n * D + 4 * (g & 1)) = c;
                        } else if (T < 2) {            // [P|R] / [Q|S]: this lane's 16-byte chunk g
                            if constexpr (XP) {
                                c.x = __builtin_amdgcn_exp2f(c.x);
                                c.y = __builtin_amdgcn_exp2f(c.y);
                            }
                            *reinterpret_cast<f4v *>((T == 0 ? PRn : QSn) + n * 2 * D + 4 * g) = c;
                        } else if (g < 2) {            // U: rows 0..7
                            *reinterpret_cast<f4v *>(U + n * D + 4 * g) = c;
                        }
                    }
                }
            } else if (slice >= 0) {
                // opaque weight-table offset: keeps the LDS weight reads out of registers
                int woff = q * L::stride;
                asm volatile("" : "+v"(woff));
                const float *wl = lds + woff;
                // hit update: H' = tanh(W4 tanh(acc) + b4)              (model.py:94-98,125)
                float ql[d4], qa[D], hl[d4], hn[D];
#pragma unroll
                for (int i = 0; i < d4; ++i) ql[i] = tanh_f(acc[i]);
                quad_allgather<d4>(ql, qa);
                role_gemv<d4, D, 0>(wl + L::o_4, qa, qa, hl);
#pragma unroll
                for (int i = 0; i < d4; ++i) hl[i] = tanh_f(hl[i]);
                quad_allgather<d4>(hl, hn);
                // stores: uniform row base + 32-bit lane offset (see lo_rec / lo_vec)
                const int64_t n0 = (int64_t)__builtin_amdgcn_readfirstlane(slice) * SLICE;
                auto at = [](float *base, unsigned off) {
                    return reinterpret_cast<float *>(reinterpret_cast<char *>(base) + off);
                };
                if constexpr (LAST)
                    emit_to<F, D, LAST, XP>(wl, hn, xv, at(Pc + n0 * D, lo_vec), at(Qc + n0 * D, lo_vec), nullptr);
                else
                    emit_to<F, D, LAST, XP>(wl, hn, xv, at(PRn + n0 * 2 * D, lo_rec), at(QSn + n0 * 2 * D, lo_rec),
                                            at(U + n0 * D, lo_vec));
            }
        };
        for (int r = 0; r + 1 < rounds; ++r) roundB(r, std::false_type{});
        roundB(rounds - 1, std::true_type{});
        if (tnext >= n_tiles) break;
        if constexpr (FIRST)
            compute_window(bufA, dn.in_cnt, capA, WinA{});
        else
            stage_commit(bufA, dn.in_cnt, WinA{});   // also makes a_cur readable
        if (slice_a(dn, 0) >= 0) arriveA(a_cur);
        tile = tnext;
        d = dn;
    }
}

// final edge pass (model.py:156) for one chunk of the caller's segment order; one lane per
// segment, P rows of the start hits and Q rows of the end hits from LDS windows or global.
template <int F, int D, bool XP>
__global__ __launch_bounds__((Cfg<F, D>::NT)) void k_edge(
    const int32_t *__restrict__ chunks, const int32_t *__restrict__ src,
    const int32_t *__restrict__ dst, const int32_t *__restrict__ sd16,
    const float *__restrict__ Pc, const float *__restrict__ Qc,
    const float *__restrict__ table, float *__restrict__ e, int64_t n_pad, int chunks_per_xcd,
    int n_chunks)
{
    using G = Cfg<F, D>;
    const float *__restrict__ W2 = table + TL<F, D>::o_flat;   // wave-uniform: scalar loads
    constexpr int NT = G::NT;
    extern __shared__ __attribute__((aligned(16))) float win[];    // sized from the plan
    const int chunk = (blockIdx.x & 7) * chunks_per_xcd + (blockIdx.x >> 3);
    if (chunk >= n_chunks) return;
    const int32_t *cd = chunks + (int64_t)chunk * DESC;
    const int e0 = cd[0], e1 = cd[1], s_lo = cd[2], s_cnt = cd[3], d_lo = cd[4], d_cnt = cd[5],
              mode = cd[6];
    float *winA = win, *winB = win + (s_cnt + 1) * D;
    if (G::ed_rec > 0 && mode) {
        stage4<NT>(Pc + (int64_t)s_lo * D, winA, s_cnt * D / 4);
        stage4<NT>(Pc + n_pad * D, winA + s_cnt * D, D / 4);               // NULL row: P = b1
        stage4<NT>(Qc + (int64_t)d_lo * D, winB, d_cnt * D / 4);
        stage4<NT>(Qc + n_pad * D, winB + d_cnt * D, D / 4);               // NULL row: Q = 0
        __syncthreads();
    }
    const float b2 = W2[D];
    auto score = [&](const float *p, const float *qq) {
        float acc = b2;
#pragma unroll
        for (int k = 0; k < D; ++k)
            acc = fmaf(W2[k], XP ? __builtin_amdgcn_rcpf(fmaf(p[k], qq[k], 1.0f)) : r_f(p[k] + qq[k]), acc);
        return r_f(acc);
    };
    if (G::ed_rec > 0 && mode) {
        // LDS mode: a lane's endpoint words are requested 8 at a time (one round trip to memory
        // per 8 segments instead of one per segment), then scored from the LDS windows
        constexpr int B8 = 8;
        for (int j0 = e0 + (int)threadIdx.x; j0 < e1; j0 += B8 * NT) {
            unsigned w[B8];
#pragma unroll
            for (int u = 0; u < B8; ++u) w[u] = (j0 + u * NT < e1) ? (unsigned)sd16[j0 + u * NT] : 0u;
#pragma unroll
            for (int u = 0; u < B8; ++u) {
                if (j0 + u * NT < e1) {
                    float p[D], qq[D];
                    load_vec<D>(winA + (w[u] & 0xFFFFu) * D, p);
                    load_vec<D>(winB + (w[u] >> 16) * D, qq);
                    e[j0 + u * NT] = score(p, qq);
                }
            }
        }
    } else {
        for (int j = e0 + (int)threadIdx.x; j < e1; j += NT) {
            float p[D], qq[D];
            const int s = src[j], d = dst[j];
            load_vec<D>(Pc + (int64_t)s * D, p);
            load_vec<D>(Qc + (int64_t)d * D, qq);
            e[j] = score(p, qq);
        }
    }
}

// The final scores once more, in the order of the plan-space batch the backward runs on (gnn_segclf_forward_train_plan:
// row T of e_all): one lane per segment of THAT batch, endpoints = plan hit ids (-1: padded -> the NULL rows), the same
// arithmetic as k_edge in the same order (the same bits for the same segment).  Its segments are sorted by end hit, so
// the Q rows stream and the P rows stay L2-local - 22 us at c3 x 32 where gathering k_edge's output into that order by
// a 4-byte permutation cost 35.
template <int F, int D, bool XP>
__global__ __launch_bounds__(256) void k_edge_tw(const int32_t *__restrict__ src, const int32_t *__restrict__ dst,
                                                 const float *__restrict__ Pc, const float *__restrict__ Qc,
                                                 const float *__restrict__ table, float *__restrict__ e, int64_t n_pad,
                                                 int64_t n_segments)
{
    const float *__restrict__ W2 = table + TL<F, D>::o_flat;   // wave-uniform: scalar loads
    const int64_t j = xcd_block() * 256 + threadIdx.x;
    if (j >= n_segments) return;
    int s = src[j], d = dst[j];
    if (s < 0) s = d = (int)n_pad;
    float p[D], qq[D];
    load_vec<D>(Pc + (int64_t)s * D, p);
    load_vec<D>(Qc + (int64_t)d * D, qq);
    float acc = W2[D];
#pragma unroll
    for (int k = 0; k < D; ++k)
        acc = fmaf(W2[k], XP ? __builtin_amdgcn_rcpf(fmaf(p[k], qq[k], 1.0f)) : r_f(p[k] + qq[k]), acc);
    e[j] = r_f(acc);
}

// final edge pass for wide hidden layers (no LDS windows): 16 lanes per segment, lane p owns dims
// DL p .. of the start hit's P row and the end hit's Q row - a row is read as whole 128-byte lines
// by the 16 lanes (one lane per segment made every load instruction touch 64 different rows:
// 3.8 TB/s of gathered bytes, 66 % of the wave cycles waiting on issue, profiles/r02_c5_a).  XCD x
// walks its own contiguous eighth of the segments (graphs stay in one L2), four passes in flight.
template <int F, int D, bool XP>
__global__ __launch_bounds__(256) void k_edge_w(const int32_t *__restrict__ src, const int32_t *__restrict__ dst,
                                                const float *__restrict__ Pc, const float *__restrict__ Qc,
                                                const float *__restrict__ table, float *__restrict__ e,
                                                int64_t n_segments)
{
    constexpr int DL = D / 16;
    const float *__restrict__ W2 = table + TL<F, D>::o_flat;
    const int lane = threadIdx.x & 63, hs = lane >> 4, p = lane & 15;
    float w2[DL];
#pragma unroll
    for (int i = 0; i < DL; ++i) w2[i] = W2[DL * p + i];
    const float b2 = W2[D];
    const int64_t per = (n_segments + 7) / 8;
    const int64_t lo = (int64_t)(blockIdx.x & 7) * per;
    const int64_t hi = lo + per < n_segments ? lo + per : n_segments;
    const int64_t wave = (int64_t)(blockIdx.x >> 3) * 4 + (threadIdx.x >> 6), nwaves = (int64_t)(gridDim.x >> 3) * 4;
    auto ends = [&](int64_t j, int &s_, int &d_) {
        s_ = j < hi ? src[j] : 0;
        d_ = j < hi ? dst[j] : 0;
    };
    auto rows = [&](int s_, int d_, float *P, float *Q) {
        load_vec<DL>(Pc + (int64_t)s_ * D + DL * p, P);
        load_vec<DL>(Qc + (int64_t)d_ * D + DL * p, Q);
    };
    auto score = [&](int64_t j, const float *P, const float *Q) {
        float acc = 0.0f;
#pragma unroll
        for (int i = 0; i < DL; ++i)
            acc = fmaf(w2[i], XP ? __builtin_amdgcn_rcpf(fmaf(P[i], Q[i], 1.0f)) : r_f(P[i] + Q[i]), acc);
        acc = quad_sum(acc);
        acc += dpp_row<0x124>(acc);
        acc += dpp_row<0x128>(acc);
        if (p == 0 && j < hi) e[j] = r_f(acc + b2);
    };
    // 16 segments per wave trip (4 passes of 4): all endpoint loads first, then all row loads, then the
    // scores - the kernel waits on two dependent loads per segment and little else
    constexpr int NP = 4;
    for (int64_t b0 = lo + 4 * NP * wave; b0 < hi; b0 += 4 * NP * nwaves) {
        int s_[NP], d_[NP];
        float P[NP][DL], Q[NP][DL];
#pragma unroll
        for (int u = 0; u < NP; ++u) ends(b0 + 4 * u + hs, s_[u], d_[u]);
#pragma unroll
        for (int u = 0; u < NP; ++u) rows(s_[u], d_[u], P[u], Q[u]);
#pragma unroll
        for (int u = 0; u < NP; ++u) score(b0 + 4 * u + hs, P[u], Q[u]);
    }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
struct Ws {
    float *table, *PRa, *PRb, *QSa, *QSb, *U, *Pc, *Qc;
    unsigned *t16;               // bf16 A fragments + biases of the matrix-core hit update (BL)
    size_t bytes;
};

Ws carve(char *b, int64_t n_pad, int table_floats, int D, int t16_words = 0)
{
    Ws w;
    size_t off = 0;
    auto take = [&](size_t nfloat) {
        float *p = reinterpret_cast<float *>(b + off);
        off += align256(nfloat * sizeof(float));
        return p;
    };
    // + 64 rows: the LDS-DMA staging of k_iter2 reads whole 1-KiB pieces past a window's end
    const size_t rec = (size_t)(n_pad + 1 + 64) * 2 * D, vec = (size_t)(n_pad + 1) * D;
    w.table = take((size_t)table_floats);
    w.PRa = take(rec); w.PRb = take(rec); w.QSa = take(rec); w.QSb = take(rec);
    w.U = take(vec); w.Pc = take(vec); w.Qc = take(vec);
    w.t16 = reinterpret_cast<unsigned *>(take((size_t)t16_words));
    w.bytes = off;
    return w;
}

template <int F, int D>
constexpr int t16_words()      // fragment tables of the wide kernels: bf16 (BL) or exact fp32 (BX), one buffer
{
    if constexpr (D % 32 == 0 && F <= 4) return BL<F, D>::total > BX<F, D>::total ? BL<F, D>::total : BX<F, D>::total;
    else if constexpr (D % 32 == 0 && F <= 8) return BL<F, D>::total;
    else if constexpr (D % 16 == 0 && F <= 4) return BX<F, D>::total;      // D = 16: exact fp32 fragments only
    else return 0;
}
template <int F, int D>
constexpr bool can_exact_wide() { return D % 16 == 0 && F <= 4; }
// k_iter_wx (sweep waves + matrix-core waves) from this many padded hits on; below it k_iter_w (round barriers).
// GNN_WIDE_LOCKSTEP=1 / GNN_WIDE_ROLES=1 force one or the other (A / B runs, tests).
constexpr int64_t kRoleSplitMinHits = 32768;

// k_iter_w's group bound: records of one table an XCD's 4 MB L2 can keep while a group's hits stream
// through it (3 MB of rows; GNN_WIDE_WINDOW_KB overrides, 0 = every tile a group of its own,
// 1 << 20 = no grouping effect: experiments)
inline int wide_window_records(int row_bytes)
{
    static const long kb = getenv("GNN_WIDE_WINDOW_KB") ? atol(getenv("GNN_WIDE_WINDOW_KB")) : 3072;
    const long r = kb * 1024 / row_bytes;
    return (int)(r < 0 ? 0 : r > 0x3FFFFFFF ? 0x3FFFFFFF : r);
}

template <int F, int D, bool XP>
int forward_t(const gnn_plan_t *pl, const gnn_params_t *p, int n_iters, float *e_out, char *ws,
              hipStream_t s)
{
    using L = TL<F, D>;
    using G = Cfg<F, D>;
    const int64_t Np = pl->n_pad, E = pl->n_segments;
    Ws w = carve(ws, Np, L::total, D, t16_words<F, D>());
    constexpr bool can_bf = D % 32 == 0 && F <= 8;
    // (k_iter_w addresses record rows and list steps with 32-bit byte offsets)
    const bool bf = can_bf && (p->flags & GNN_FLAG_BF16_MLP) && n_iters > 0 &&
                    (uint64_t)(Np + 2) * D * 4 < (1ull << 32);
    if constexpr (can_bf)
        if (bf && Np > 0)
            GNN_LAUNCH("k_pack16", (k_pack16<F, D>), 64, 256, s, *p, w.t16, w.PRa, w.PRb, w.QSa, w.QSb, Np,
                       XP ? 1 : 0);
    // hidden_dim 16 (F <= 4) takes the 16-lanes-per-hit kernel too - one dim per lane, the hit update (1776
    // multiply-adds per hit on the vector pipe before, with its weights read from LDS) on the fp32
    // matrix-core instruction - where its two 128 KB record windows per 1000-hit level never fitted
    // the LDS of k_iter; GNN_NO_WIDE_EXACT=1 keeps the general kernel
    constexpr bool can_ex = can_exact_wide<F, D>();
    const bool ex = can_ex && !bf && n_iters > 0 && Np > 0 && (uint64_t)(Np + 2) * D * 8 < (1ull << 32) &&
                    !getenv("GNN_NO_WIDE_EXACT");
    if (Np == 0 || G::pack_first || ex)   // no hits (nothing for k_input4 to do), a big table, or k_input4_x reads it
        GNN_LAUNCH("k_pack", (k_pack<F, D, XP>), (L::total + 255) / 256, 256, s, *p, w.table, w.PRa, w.PRb,
                   w.QSa, w.QSb, w.U, w.Pc, w.Qc, Np);
    // wide hidden layers in exact fp32 (the default): 16 lanes per hit over fp32 record rows, hit
    // update on v_mfma_f32_16x16x4_f32 (BX).  k_pack32 runs after k_pack: its NULL rows use k_iter_w's
    // row order and replace the general kernels'.
    if constexpr (can_ex)
        if (ex)
            GNN_LAUNCH("k_pack32", (k_pack32<F, D>), 64, 256, s, *p, reinterpret_cast<float *>(w.t16), w.PRa, w.PRb,
                       w.QSa, w.QSb, Np, XP ? 1 : 0);
    float *PR = w.PRa, *PRn = w.PRb, *QS = w.QSa, *QSn = w.QSb;
    if (Np > 0) {
        const int nt = (int)pl->n_tiles;
        const int tpx = (nt + 7) / 8;
        const size_t it_lds = (size_t)(L::total + (G::it_rec > 0 ? pl->iter_lds_records : 0) * 2 * D + 4) * sizeof(float);
        static DevOnce attr_done;     // dynamic LDS above 64 KB must be opted into, once per device
        if (attr_done.need()) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_iter<F, D, true, XP>), hipFuncAttributeMaxDynamicSharedMemorySize, G::lds_bytes);
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_iter<F, D, false, XP>), hipFuncAttributeMaxDynamicSharedMemorySize, G::lds_bytes);
        }
#ifdef GNN_DIAG
        const char *ab = getenv("GNN_ABLATE");   // timing diagnostics only (results invalid): diag builds
        const int ablate = ab ? atoi(ab) : 0;
#else
        const int ablate = 0;                    // the shipped library never skips work
#endif
        // persistent phase-split kernel when every tile runs in LDS mode and both windows fit LDS;
        // otherwise the general kernel
        bool use2 = false, fuse_first = false;
        size_t it2_lds = 0, it2_lds_first = 0;
        int capA = 0, capB = 0, grid2 = 0, xbuf_floats = 0;
        if constexpr (G::iter2) {
            // window buffers in whole 1-KiB DMA pieces (= 128 / D records) plus one piece that
            // holds the NULL record and absorbs the last piece's overrun
            const int64_t rpp = 128 / D;                       // records per piece
            const int64_t capa = (pl->iter_lds_in + rpp - 1) / rpp * rpp + rpp;
            const int64_t capb = (pl->iter_lds_out + rpp - 1) / rpp * rpp + rpp;
            use2 = !getenv("GNN_NO_ITER2") && nt > 0 && pl->n_lds_tiles == nt &&
                   pl->in_nbr16 && pl->out_nbr16 && pl->in_off16 && pl->out_off16 && pl->sched_a && pl->sched_b;
            capA = (int)capa;
            capB = (int)capb;
            constexpr int mt_floats = mt_total<F, D>();
            it2_lds = (size_t)(L::total + (capa + capb) * 2 * D + 4 + mt_floats) * sizeof(float);
            if (it2_lds > (size_t)G::lds_bytes) use2 = false;
            // first iteration fused with the input network: + one buffer of X rows (256-byte pieces)
            xbuf_floats = 0;     // (the X rows of a window are staged inside the window buffer itself)
            // + the fp32 A fragments / biases of the matrix-core window products (D = 8)
            it2_lds_first = it2_lds;
            // (exp-product mode only: the plain-exp variant of the fused kernel does not fit the
            // register budget without spills, and it is the rarely taken fallback anyway)
            fuse_first = G::fuse_first && XP && use2 && n_iters >= 2 && it2_lds_first <= (size_t)G::lds_bytes &&
                         !getenv("GNN_NO_FUSE_FIRST");
            const int n_cu = device_cus();
            static DevOnce it2_attr;
            if (it2_attr.need()) {
                (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_iter2<F, D, true, XP, false>), hipFuncAttributeMaxDynamicSharedMemorySize, G::lds_bytes);
                (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_iter2<F, D, false, XP, false>), hipFuncAttributeMaxDynamicSharedMemorySize, G::lds_bytes);
                if constexpr (XP && G::fuse_first)
                    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_iter2<F, D, false, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, G::lds_bytes);
            }
            grid2 = nt < n_cu ? nt : n_cu;
        }
        bool input_done = false;
        if constexpr (can_bf) {
            if (bf && G::pack_first) {      // records of iteration 0 on the matrix cores too
                using B = BL<F, D>;
                const size_t lds_in = (size_t)(B::template tm_words<false>() + 5 * D + 4 * 16 * B::tr_stride) * 4;
                static DevOnce in_attr;
                if (in_attr.need())
                    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_input4_bf<F, D, false, XP>), hipFuncAttributeMaxDynamicSharedMemorySize, G::lds_bytes);
                const int64_t g_need = (Np * 4 + 255) / 256;
                GNN_LAUNCH_SH("k_input4", (k_input4_bf<F, D, false, XP>), (unsigned)(g_need < 512 ? g_need : 512), 256,
                              lds_in, s, pl->X, w.table, w.t16, PR, QS, w.U, w.Pc, w.Qc, Np);
                input_done = true;
            }
        }
        if constexpr (can_ex) {
            if (ex) {
                using B = BX<F, D>;
                const size_t lds_in = (size_t)(B::template tm_words<false>() + 5 * D + 16 * 16 * B::tr_stride) * 4;
                static DevOnce inx_attr;
                if (inx_attr.need())
                    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_input4_x<F, D, false, XP>), hipFuncAttributeMaxDynamicSharedMemorySize, G::lds_bytes);
                const int64_t g_need = (Np * 4 + 1023) / 1024;
                const int64_t g_cap = device_cus();
                GNN_LAUNCH_SH("k_input4", (k_input4_x<F, D, false, XP>), (unsigned)(g_need < g_cap ? g_need : g_cap), 1024,
                              lds_in, s, pl->X, w.table, reinterpret_cast<const float *>(w.t16), PR, QS, w.U, w.Pc, w.Qc, Np);
                input_done = true;
            }
        }
        if (!fuse_first && !input_done) {
            const int64_t g_need = (Np * 4 + 255) / 256;
            const unsigned g = (unsigned)(g_need < 4096 ? g_need : 4096);   // 16 workgroups per CU, grid-stride
            if (n_iters == 0)
                GNN_LAUNCH("k_input4", (k_input4<F, D, true, XP>), g, 256, s, pl->X, *p, w.table, PR, QS,
                           PRn, QSn, w.U, w.Pc, w.Qc, Np);
            else
                GNN_LAUNCH("k_input4", (k_input4<F, D, false, XP>), g, 256, s, pl->X, *p, w.table, PR, QS,
                           PRn, QSn, w.U, w.Pc, w.Qc, Np);
        }
        for (int t = 0; t < n_iters; ++t) {
            if constexpr (G::iter2) {
                if (use2) {
#define GNN_IT2(LAST_, FIRST_, LDS_)                                                                   \
    GNN_LAUNCH_SH("k_iter2", (k_iter2<F, D, LAST_, XP, FIRST_>), grid2, 1024, LDS_, s, pl->X, w.table, *p, \
                  w.table, pl->tiles, pl->in_off, pl->in_off16, pl->in_nbr16, pl->out_off,             \
                  pl->out_off16, pl->out_nbr16, pl->sched_a, pl->sched_b, PR, QS, w.U, PRn, QSn, w.Pc,  \
                  w.Qc, Np, nt, capA, capB, xbuf_floats)
                    if (XP && t == 0 && fuse_first) {
                        if constexpr (XP && G::fuse_first) GNN_IT2(false, true, it2_lds_first);
                    } else if (t + 1 == n_iters)
                        GNN_IT2(true, false, it2_lds);
                    else
                        GNN_IT2(false, false, it2_lds);
#undef GNN_IT2
                    float *t1 = PR; PR = PRn; PRn = t1;
                    float *t2 = QS; QS = QSn; QSn = t2;
                    continue;
                }
            }
            bool launched = false;
            if constexpr (can_bf) {
                if (bf) {           // wide hidden layers on bf16 records: k_iter_w (16 lanes per hit)
                    using B = BL<F, D>;
                    static DevOnce bf_attr;
                    if (bf_attr.need()) {
                        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_iter_w<F, D, true, XP>), hipFuncAttributeMaxDynamicSharedMemorySize, G::lds_bytes);
                        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_iter_w<F, D, false, XP>), hipFuncAttributeMaxDynamicSharedMemorySize, G::lds_bytes);
                    }
                    const int ncu = device_cus();
                    const unsigned wgs = (unsigned)(((nt < ncu ? nt : ncu) + 7) / 8 * 8);   // persistent, 8 | grid
                    const size_t trw = (size_t)2 * 4 * 16 * B::tr_stride + 4;  // double-buffered scratch of the 4 teams + the group word
                    const int wmax = wide_window_records(4 * D);
                    const unsigned *PRh = reinterpret_cast<const unsigned *>(PR), *QSh = reinterpret_cast<const unsigned *>(QS);
                    // (small batches - a workgroup gets a handful of slices - keep the barrier kernel: the ring's polls and its
                    // end-of-work detection cost more than they hide there, 1k hits at D = 32: 137 vs 159 us per forward)
                    const bool lockstep_bf = getenv("GNN_WIDE_LOCKSTEP") != nullptr || (Np < kRoleSplitMinHits && !getenv("GNN_WIDE_ROLES"));
                    if (!lockstep_bf) {             // sweep waves + matrix-core waves (k_iter_wx)
                        static DevOnce wxb_attr;
                        if (wxb_attr.need()) {
                            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_iter_wx<F, D, true, XP, false>), hipFuncAttributeMaxDynamicSharedMemorySize, G::lds_bytes);
                            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_iter_wx<F, D, false, XP, false>), hipFuncAttributeMaxDynamicSharedMemorySize, G::lds_bytes);
                        }
                        const size_t ringw = (size_t)12 * 16 * B::tr_stride + 40;
                        if (t + 1 == n_iters)
                            GNN_LAUNCH_SH("k_iter_wx", (k_iter_wx<F, D, true, XP, false>), wgs, 1024,
                                          (B::template lds_words<true>() + ringw) * 4, s, pl->X, w.table, w.t16, pl->tiles,
                                          pl->in_off, pl->in_nbr, pl->out_off, pl->out_nbr, PRh, QSh, w.U, PRn, QSn,
                                          w.Pc, w.Qc, Np, tpx, nt, wmax);
                        else
                            GNN_LAUNCH_SH("k_iter_wx", (k_iter_wx<F, D, false, XP, false>), wgs, 1024,
                                          (B::template lds_words<false>() + ringw) * 4, s, pl->X, w.table, w.t16, pl->tiles,
                                          pl->in_off, pl->in_nbr, pl->out_off, pl->out_nbr, PRh, QSh, w.U, PRn, QSn,
                                          w.Pc, w.Qc, Np, tpx, nt, wmax);
                        float *t1 = PR; PR = PRn; PRn = t1;
                        float *t2 = QS; QS = QSn; QSn = t2;
                        continue;
                    }
                    if (t + 1 == n_iters)
                        GNN_LAUNCH_SH("k_iter_w", (k_iter_w<F, D, true, XP>), wgs, 1024,
                                      (B::template lds_words<true>() + trw) * 4, s, pl->X, w.table, w.t16, pl->tiles,
                                      pl->in_off, pl->in_nbr, pl->out_off, pl->out_nbr, PRh, QSh, w.U, PRn, QSn,
                                      w.Pc, w.Qc, Np, tpx, nt, wmax);
                    else
                        GNN_LAUNCH_SH("k_iter_w", (k_iter_w<F, D, false, XP>), wgs, 1024,
                                      (B::template lds_words<false>() + trw) * 4, s, pl->X, w.table, w.t16, pl->tiles,
                                      pl->in_off, pl->in_nbr, pl->out_off, pl->out_nbr, PRh, QSh, w.U, PRn, QSn,
                                      w.Pc, w.Qc, Np, tpx, nt, wmax);
                    launched = true;
                }
            }
            if constexpr (can_ex) {
                if (ex) {           // the same kernel on fp32 record rows, exact fp32 matrix-core tail
                    using B = BX<F, D>;
                    static DevOnce ex_attr;
                    if (ex_attr.need()) {
                        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_iter_w<F, D, true, XP, true>), hipFuncAttributeMaxDynamicSharedMemorySize, G::lds_bytes);
                        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_iter_w<F, D, false, XP, true>), hipFuncAttributeMaxDynamicSharedMemorySize, G::lds_bytes);
                    }
                    const int ncu = device_cus() * (D == 16 ? 2 : 1);      // D = 16: two workgroups per CU
                    const unsigned wgs = (unsigned)(((nt < ncu ? nt : ncu) + 7) / 8 * 8);
                    if constexpr (D >= 16) {       // sweep waves + matrix-core waves (k_iter_wx)
                        const bool lockstep = getenv("GNN_WIDE_LOCKSTEP") != nullptr || (Np < kRoleSplitMinHits && !getenv("GNN_WIDE_ROLES"));
                        if (!lockstep) {
                            static DevOnce wx_attr;
                            if (wx_attr.need()) {
                                (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_iter_wx<F, D, true, XP>), hipFuncAttributeMaxDynamicSharedMemorySize, G::lds_bytes);
                                (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_iter_wx<F, D, false, XP>), hipFuncAttributeMaxDynamicSharedMemorySize, G::lds_bytes);
                            }
                            const size_t ringw = (size_t)12 * 16 * B::tr_stride + 40;      // 12 slots + the counters (38 words)
                            const unsigned *PRx = reinterpret_cast<const unsigned *>(PR), *QSx = reinterpret_cast<const unsigned *>(QS);
                            const int wmx = wide_window_records(8 * D);
                            if (t + 1 == n_iters)
                                GNN_LAUNCH_SH("k_iter_wx", (k_iter_wx<F, D, true, XP>), wgs, 1024,
                                              (B::template lds_words<true>() + ringw) * 4, s, pl->X, w.table, w.t16, pl->tiles,
                                              pl->in_off, pl->in_nbr, pl->out_off, pl->out_nbr, PRx, QSx, w.U, PRn, QSn,
                                              w.Pc, w.Qc, Np, tpx, nt, wmx);
                            else
                                GNN_LAUNCH_SH("k_iter_wx", (k_iter_wx<F, D, false, XP>), wgs, 1024,
                                              (B::template lds_words<false>() + ringw) * 4, s, pl->X, w.table, w.t16, pl->tiles,
                                              pl->in_off, pl->in_nbr, pl->out_off, pl->out_nbr, PRx, QSx, w.U, PRn, QSn,
                                              w.Pc, w.Qc, Np, tpx, nt, wmx);
                            launched = true;
                            float *t1 = PR; PR = PRn; PRn = t1;
                            float *t2 = QS; QS = QSn; QSn = t2;
                            continue;
                        }
                    }
                    const size_t trw = (size_t)(2 * 4 + 4) * 16 * B::tr_stride + 4;   // double-buffered q scratch + hl scratch + the group word
                    const int wmax = wide_window_records(8 * D);
                    const unsigned *PRh = reinterpret_cast<const unsigned *>(PR), *QSh = reinterpret_cast<const unsigned *>(QS);
                    if (t + 1 == n_iters)
                        GNN_LAUNCH_SH("k_iter_w", (k_iter_w<F, D, true, XP, true>), wgs, 1024,
                                      (B::template lds_words<true>() + trw) * 4, s, pl->X, w.table, w.t16, pl->tiles,
                                      pl->in_off, pl->in_nbr, pl->out_off, pl->out_nbr, PRh, QSh, w.U, PRn, QSn,
                                      w.Pc, w.Qc, Np, tpx, nt, wmax);
                    else
                        GNN_LAUNCH_SH("k_iter_w", (k_iter_w<F, D, false, XP, true>), wgs, 1024,
                                      (B::template lds_words<false>() + trw) * 4, s, pl->X, w.table, w.t16, pl->tiles,
                                      pl->in_off, pl->in_nbr, pl->out_off, pl->out_nbr, PRh, QSh, w.U, PRn, QSn,
                                      w.Pc, w.Qc, Np, tpx, nt, wmax);
                    launched = true;
                }
            }
            if (launched) {
            } else if (t + 1 == n_iters)
                GNN_LAUNCH_SH("k_iter", (k_iter<F, D, true, XP>), 8 * tpx, G::NT, it_lds, s, pl->X, w.table,
                           pl->tiles, pl->in_off, pl->in_nbr, pl->out_off, pl->out_nbr, PR, QS, w.U,
                           PRn, QSn, w.Pc, w.Qc, Np, tpx, nt, ablate);
            else
                GNN_LAUNCH_SH("k_iter", (k_iter<F, D, false, XP>), 8 * tpx, G::NT, it_lds, s, pl->X, w.table,
                           pl->tiles, pl->in_off, pl->in_nbr, pl->out_off, pl->out_nbr, PR, QS, w.U,
                           PRn, QSn, w.Pc, w.Qc, Np, tpx, nt, ablate);
            float *t1 = PR; PR = PRn; PRn = t1;
            float *t2 = QS; QS = QSn; QSn = t2;
        }
    }
    if (E > 0) {
        const int nc = (int)pl->n_chunks;
        const int cpx = (nc + 7) / 8;
        const size_t ed_lds = (size_t)((G::ed_rec > 0 ? pl->edge_lds_rows : 0) * D + 4) * sizeof(float);
        static DevOnce edge_attr;
        if (edge_attr.need())
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_edge<F, D, XP>), hipFuncAttributeMaxDynamicSharedMemorySize, G::lds_bytes);
        if constexpr (G::ed_rec == 0 && D % 16 == 0) {
            // wide rows, no LDS windows: every chunk is in global mode (absolute ids; padded
            // segments point at the NULL rows), so the chunk descriptors are not needed
            const int64_t waves = (E + 15) / 16;                          // 16 segments per wave trip
            int64_t wg = (waves + 3) / 4;
            const int64_t cap = (int64_t)device_cus() * 8;                // 8 workgroups of 256 per CU
            wg = wg < cap ? wg : cap;
            GNN_LAUNCH("k_edge", (k_edge_w<F, D, XP>), (unsigned)((wg + 7) / 8 * 8), 256, s, pl->src, pl->dst, w.Pc, w.Qc,
                       w.table, e_out, E);
        } else {
            GNN_LAUNCH_SH("k_edge", (k_edge<F, D, XP>), 8 * cpx, G::NT, ed_lds, s, pl->chunks, pl->src, pl->dst, pl->sd16, w.Pc,
                       w.Qc, w.table, e_out, Np, cpx, nc);
        }
    }
    return 0;
}

// The TRAINING forward on a planned batch (gnn_segclf_forward_train_plan): the fused tile kernels, keeping what
// the backward needs (TrainOut) - instead of the per-module kernels' k_input + T x (k_pq, k_edge, k_node), whose
// node pass walks both segment lists through the L2 (0.31 ms of a 0.97 ms step at c3 x 32; this: 0.2).
// e_all [(T + 1), E]: rows 0 .. T-1 in the order of the hits' in-lists (segments sorted by end hit, stable),
// valid segments only; row T is NOT written here (the final scores come back in e_out, the plan's segment
// order).  H_all [(T + 1), n_pad, ldh], Q_all [T, n_pad, D].  Shapes on the general tile kernel only (D <= 16
// without the wide route); others: GNN_ERR_UNSUPPORTED (the caller keeps the per-module route).
template <int F, int D, bool XP>
int forward_train_t(const gnn_plan_t *pl, const gnn_params_t *p, int n_iters, const int32_t *seg_ptr,
                    const int32_t *tw_src, const int32_t *tw_dst, float *e_all, float *H_all, float *Q_all, int ldh,
                    float *e_out, char *ws, hipStream_t s)
{
    using L = TL<F, D>;
    using G = Cfg<F, D>;
    if constexpr (D > 16 || G::wide16) {
        return fail(GNN_ERR_UNSUPPORTED, "no fused training forward for input_dim=%d hidden_dim=%d", F, D);
    } else {
        const int64_t Np = pl->n_pad, E = pl->n_segments;
        Ws w = carve(ws, Np, L::total, D, t16_words<F, D>());
        if (Np == 0 || G::pack_first)
            GNN_LAUNCH("k_pack", (k_pack<F, D, XP>), (L::total + 255) / 256, 256, s, *p, w.table, w.PRa, w.PRb,
                       w.QSa, w.QSb, w.U, w.Pc, w.Qc, Np);
        float *PR = w.PRa, *PRn = w.PRb, *QS = w.QSa, *QSn = w.QSb;
        if (Np > 0) {
            const int nt = (int)pl->n_tiles;
            const int tpx = (nt + 7) / 8;
            const size_t it_lds = (size_t)(L::total + (G::it_rec > 0 ? pl->iter_lds_records : 0) * 2 * D + 4) * sizeof(float);
            static DevOnce attr_done;
            if (attr_done.need()) {
                (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_iter<F, D, true, XP, true>), hipFuncAttributeMaxDynamicSharedMemorySize, G::lds_bytes);
                (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_iter<F, D, false, XP, true>), hipFuncAttributeMaxDynamicSharedMemorySize, G::lds_bytes);
            }
            const int64_t g_need = (Np * 4 + 255) / 256;
            const unsigned g = (unsigned)(g_need < 4096 ? g_need : 4096);
            if (n_iters == 0)
                GNN_LAUNCH("k_input4", (k_input4<F, D, true, XP, true>), g, 256, s, pl->X, *p, w.table, PR, QS,
                           PRn, QSn, w.U, w.Pc, w.Qc, Np, H_all, ldh);
            else
                GNN_LAUNCH("k_input4", (k_input4<F, D, false, XP, true>), g, 256, s, pl->X, *p, w.table, PR, QS,
                           PRn, QSn, w.U, w.Pc, w.Qc, Np, H_all, ldh);
            for (int t = 0; t < n_iters; ++t) {
                const TrainOut tro{seg_ptr, e_all + (size_t)t * E, Q_all + (size_t)t * Np * D,
                                   H_all + (size_t)(t + 1) * Np * ldh, ldh};
                if (t + 1 == n_iters)
                    GNN_LAUNCH_SH("k_iter", (k_iter<F, D, true, XP, true>), 8 * tpx, G::NT, it_lds, s, pl->X, w.table,
                                  pl->tiles, pl->in_off, pl->in_nbr, pl->out_off, pl->out_nbr, PR, QS, w.U,
                                  PRn, QSn, w.Pc, w.Qc, Np, tpx, nt, 0, tro);
                else
                    GNN_LAUNCH_SH("k_iter", (k_iter<F, D, false, XP, true>), 8 * tpx, G::NT, it_lds, s, pl->X, w.table,
                                  pl->tiles, pl->in_off, pl->in_nbr, pl->out_off, pl->out_nbr, PR, QS, w.U,
                                  PRn, QSn, w.Pc, w.Qc, Np, tpx, nt, 0, tro);
                float *t1 = PR; PR = PRn; PRn = t1;
                float *t2 = QS; QS = QSn; QSn = t2;
            }
        }
        if (E > 0 && e_out) {
            const int nc = (int)pl->n_chunks;
            const int cpx = (nc + 7) / 8;
            const size_t ed_lds = (size_t)((G::ed_rec > 0 ? pl->edge_lds_rows : 0) * D + 4) * sizeof(float);
            static DevOnce edge_attr;
            if (edge_attr.need())
                (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_edge<F, D, XP>), hipFuncAttributeMaxDynamicSharedMemorySize, G::lds_bytes);
            GNN_LAUNCH_SH("k_edge", (k_edge<F, D, XP>), 8 * cpx, G::NT, ed_lds, s, pl->chunks, pl->src, pl->dst, pl->sd16, w.Pc,
                          w.Qc, w.table, e_out, Np, cpx, nc);
        }
        if (E > 0 && tw_src)      // row T of e_all: the final scores in the backward's own segment order
            GNN_LAUNCH("k_edge_tw", (k_edge_tw<F, D, XP>), grid_for(E), 256, s, tw_src, tw_dst, w.Pc, w.Qc, w.table,
                       e_all + (size_t)n_iters * E, Np, E);
        return 0;
    }
}

#ifdef GNN_QUICK      // compile-time experiments: one shape only
#ifndef GNN_QUICK_F
#define GNN_QUICK_F 3
#endif
#ifndef GNN_QUICK_D
#define GNN_QUICK_D 8
#endif
#define SELL_FOR_EACH_SHAPE(X_) X_(GNN_QUICK_F, GNN_QUICK_D)
#else
#define SELL_FOR_EACH_SHAPE(X_)                                                          \
    X_(2, 4) X_(2, 8) X_(2, 16) X_(2, 32) X_(3, 4) X_(3, 8) X_(3, 16) X_(3, 32) X_(3, 64) \
    X_(11, 4) X_(11, 8) X_(11, 16)
#endif

}  // namespace

namespace gnn {

int sell_shape_supported(int F, int D)
{
#define X_(F_, D_) if (F == F_ && D == D_) return 1;
    SELL_FOR_EACH_SHAPE(X_)
#undef X_
    return 0;
}

int sell_limits(int F, int D, int32_t *out4)
{
#define X_(F_, D_)                                               \
    if (F == F_ && D == D_) {                                    \
        out4[0] = Cfg<F_, D_>::tile_hits;                        \
        out4[1] = Cfg<F_, D_>::it_rec;                           \
        out4[2] = Cfg<F_, D_>::chunk_segments;                   \
        out4[3] = Cfg<F_, D_>::ed_rec;                           \
        return 0;                                                \
    }
    SELL_FOR_EACH_SHAPE(X_)
#undef X_
    return fail(GNN_ERR_UNSUPPORTED, "no fused kernel for input_dim=%d hidden_dim=%d", F, D);
}

size_t sell_workspace_bytes(int64_t n_pad, int64_t n_segments, int F, int D)
{
    (void)n_segments;
#define X_(F_, D_) if (F == F_ && D == D_) return carve(nullptr, n_pad, TL<F_, D_>::total, D, t16_words<F_, D_>()).bytes + 256;
    SELL_FOR_EACH_SHAPE(X_)
#undef X_
    return 0;
}

int sell_forward_train(const gnn_plan_t *pl, const gnn_params_t *p, int n_iters, const int32_t *seg_ptr,
                       const int32_t *tw_src, const int32_t *tw_dst, float *e_all, float *H_all, float *Q_all, int ldh,
                       float *e_out, void *ws, size_t ws_bytes, hipStream_t s)
{
    ProfChain chain_;
    const size_t need = sell_workspace_bytes(pl->n_pad, pl->n_segments, p->F, p->D);
    if (need == 0) return fail(GNN_ERR_UNSUPPORTED, "no fused kernel for input_dim=%d hidden_dim=%d", p->F, p->D);
    if (!ws || ws_bytes < need) return fail(GNN_ERR_WORKSPACE, "workspace too small: need %zu bytes", need);
    char *base = reinterpret_cast<char *>((reinterpret_cast<uintptr_t>(ws) + 255) & ~(uintptr_t)255);
#define X_(F_, D_)                                                                                        \
    if (p->F == F_ && p->D == D_)                                                                        \
        return (p->flags & GNN_FLAG_EXP_PRODUCT)                                                         \
                   ? forward_train_t<F_, D_, true>(pl, p, n_iters, seg_ptr, tw_src, tw_dst, e_all, H_all, Q_all, ldh, e_out, base, s)  \
                   : forward_train_t<F_, D_, false>(pl, p, n_iters, seg_ptr, tw_src, tw_dst, e_all, H_all, Q_all, ldh, e_out, base, s);
    SELL_FOR_EACH_SHAPE(X_)
#undef X_
    return fail(GNN_ERR_UNSUPPORTED, "unreachable");
}

int sell_forward(const gnn_plan_t *pl, const gnn_params_t *p, int n_iters, float *e_out, void *ws,
                 size_t ws_bytes, hipStream_t s)
{
    ProfChain chain_;      // (profiling runs: one event per kernel boundary of this call)
    const size_t need = sell_workspace_bytes(pl->n_pad, pl->n_segments, p->F, p->D);
    if (need == 0) return fail(GNN_ERR_UNSUPPORTED, "no fused kernel for input_dim=%d hidden_dim=%d", p->F, p->D);
    if (!ws || ws_bytes < need) return fail(GNN_ERR_WORKSPACE, "workspace too small: need %zu bytes", need);
    char *base = reinterpret_cast<char *>((reinterpret_cast<uintptr_t>(ws) + 255) & ~(uintptr_t)255);
#define X_(F_, D_)                                                                              \
    if (p->F == F_ && p->D == D_)                                                              \
        return (p->flags & GNN_FLAG_EXP_PRODUCT) ? forward_t<F_, D_, true>(pl, p, n_iters, e_out, base, s) \
                                                 : forward_t<F_, D_, false>(pl, p, n_iters, e_out, base, s);
    SELL_FOR_EACH_SHAPE(X_)
#undef X_
    return fail(GNN_ERR_UNSUPPORTED, "unreachable");
}

}  // namespace gnn
