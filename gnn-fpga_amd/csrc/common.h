// common.h - error reporting, HIP-event profiler and launch macro shared by the kernel files.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <vector>

#include "gnn_hip.h"

namespace gnn {

int fail(int code, const char *fmt, ...);
void prof_pre(const char *name, hipStream_t s);
void prof_post(hipStream_t s);
// gnn_profile_begin/end: the launches of ONE library call are timed from one event per kernel boundary
// (see Profiler::chain); put one at the top of an entry point that launches several kernels
struct ProfChain {
    bool prev;
    ProfChain();
    ~ProfChain();
    ProfChain(const ProfChain &) = delete;
    ProfChain &operator=(const ProfChain &) = delete;
};

#define GNN_LAUNCH(NAME, KERNEL, GRID, BLOCK, STREAM, ...)                                       \
    do {                                                                                         \
        gnn::prof_pre(NAME, STREAM);                                                             \
        hipLaunchKernelGGL(KERNEL, dim3(GRID), dim3(BLOCK), 0, STREAM, __VA_ARGS__);             \
        gnn::prof_post(STREAM);                                                                  \
        hipError_t err_ = hipGetLastError();                                                     \
        if (err_ != hipSuccess)                                                                  \
            return gnn::fail(-(int)err_, "%s launch failed: %s", NAME, hipGetErrorString(err_)); \
    } while (0)

// same with dynamic LDS bytes
#define GNN_LAUNCH_SH(NAME, KERNEL, GRID, BLOCK, SHMEM, STREAM, ...)                             \
    do {                                                                                         \
        gnn::prof_pre(NAME, STREAM);                                                             \
        hipLaunchKernelGGL(KERNEL, dim3(GRID), dim3(BLOCK), SHMEM, STREAM, __VA_ARGS__);         \
        gnn::prof_post(STREAM);                                                                  \
        hipError_t err_ = hipGetLastError();                                                     \
        if (err_ != hipSuccess)                                                                  \
            return gnn::fail(-(int)err_, "%s launch failed: %s", NAME, hipGetErrorString(err_)); \
    } while (0)

constexpr int kBlock = 256;   // 4 waves of 64

// One-time per-DEVICE set-up (hipFuncSetAttribute opt-ins for > 64 KB of dynamic LDS, CU counts):
// a process may drive several devices, and a function attribute set while device 0 was current
// says nothing about device 1.  `static DevOnce once; if (once.need()) { ... }`.
struct DevOnce {
    unsigned long long done = 0;
    bool need()
    {
        int d = 0;
        if (hipGetDevice(&d) != hipSuccess) return true;
        const unsigned long long bit = 1ull << (d & 63);
        if (done & bit) return false;
        done |= bit;
        return true;
    }
};

// CU count of the CURRENT device (cached per device)
inline int device_cus()
{
    static int cus[64] = {0};
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess) return 256;
    int &c = cus[d & 63];
    if (!c) {
        hipDeviceProp_t prop;
        c = (hipGetDeviceProperties(&prop, d) == hipSuccess && prop.multiProcessorCount > 0)
                ? prop.multiProcessorCount : 256;
    }
    return c;
}

// one item per thread; grids beyond 8 workgroups are rounded up to a multiple of 8 so that
// xcd_block() below is a bijection (the surplus workgroups find nothing to do)
inline unsigned grid_for(int64_t n)
{
    const unsigned g = (unsigned)((n + kBlock - 1) / kBlock);
    return g > 8 ? (g + 7) & ~7u : g;
}

// Workgroups are dealt round-robin to the 8 XCDs, each with its own L2.  Items are stored graph
// by graph (hits, segments), and a kernel's gathers stay inside the item's graph: with the plain
// blockIdx order every XCD works on every graph in flight and its L2 thrashes; here XCD x takes
// a CONTIGUOUS eighth of the items, a few graphs at a time.
__device__ __forceinline__ int64_t xcd_block()
{
    const unsigned g = gridDim.x, b = blockIdx.x;
    return (g & 7) ? b : (b & 7) * (g >> 3) + (b >> 3);
}
inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

// tanh(x) = 1 - 2 / (2^(2 log2(e) x) + 1): v_exp_f32 + v_rcp_f32 (1 ulp each), absolute error
// ~1e-7 everywhere (the score tolerance is absolute, 1e-5).  Saturates correctly at +-inf.
__device__ __forceinline__ float tanh_f(float x)
{
    float t = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(t + 1.0f);
}

__device__ __forceinline__ float sigmoid_f(float x)
{
    float t = __builtin_amdgcn_exp2f(x * -1.4426950408889634f);
    return __builtin_amdgcn_rcpf(1.0f + t);
}

// Walk CSR entries [k0, k1) of one hit: `row_of(k)` gives the float row entry k points at (an
// index load), `w_of(k)` its weight (an index load and a dependent gather), `use(w, row)` consumes
// them IN ENTRY ORDER (the sums stay bit-identical to a plain loop).  A plain loop is two
// dependent memory latencies per entry; here the index loads, weights and rows of U entries are
// all in flight together before the first one is consumed.
template <int N4, int U = 4, typename RowOf, typename WOf, typename Use>
__device__ __forceinline__ void csr_walk(int k0, int k1, RowOf row_of, WOf w_of, Use use)
{
    int k = k0;
    for (; k + U <= k1; k += U) {
        const float *rp[U];
        float w[U], r[U][4 * N4];
#pragma unroll
        for (int u = 0; u < U; ++u) rp[u] = row_of(k + u);
#pragma unroll
        for (int u = 0; u < U; ++u) w[u] = w_of(k + u);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float4 *r4 = reinterpret_cast<const float4 *>(rp[u]);
#pragma unroll
            for (int i = 0; i < N4; ++i) {
                const float4 a = r4[i];
                r[u][4 * i] = a.x; r[u][4 * i + 1] = a.y; r[u][4 * i + 2] = a.z; r[u][4 * i + 3] = a.w;
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) use(w[u], r[u]);
    }
    for (; k < k1; ++k) {
        const float *rp = row_of(k);
        const float w = w_of(k);
        float r[4 * N4];
        const float4 *r4 = reinterpret_cast<const float4 *>(rp);
#pragma unroll
        for (int i = 0; i < N4; ++i) {
            const float4 a = r4[i];
            r[4 * i] = a.x; r[4 * i + 1] = a.y; r[4 * i + 2] = a.z; r[4 * i + 3] = a.w;
        }
        use(w, r);
    }
}

// forward of the plan-based pipeline (sell_pipeline.hip)
int sell_forward(const gnn_plan_t *pl, const gnn_params_t *p, int n_iters, float *e_out, void *ws,
                 size_t ws_bytes, hipStream_t s);
int sell_forward_train(const gnn_plan_t *pl, const gnn_params_t *p, int n_iters, const int32_t *seg_ptr,
                       const int32_t *tw_src, const int32_t *tw_dst, float *e_all, float *H_all, float *Q_all, int ldh,
                       float *e_out, void *ws, size_t ws_bytes, hipStream_t s);
size_t sell_workspace_bytes(int64_t n_hits, int64_t n_segments, int F, int D);
int sell_shape_supported(int F, int D);
int sell_limits(int F, int D, int32_t *out4);

// backward.hip
int bce_loss(const float *e, const float *y, int64_t n, float scale, float *loss, float *grad_e,
             float *partial, hipStream_t s);
size_t backward_workspace_bytes(int64_t n_hits, int64_t n_segments, int F, int D);
int backward_events_supported(int F, int D, int64_t max_hits, int64_t max_segments);
size_t backward_events_workspace_bytes(int64_t n_graphs, int F, int D);
int backward_events(const gnn_graph_t *g, const gnn_params_t *p, const int32_t *hit_ptr,
                    const int32_t *seg_ptr, int64_t n_graphs, int cap_h, int cap_s, int T,
                    const float *e_all, const float *H_all, const float *grad_out,
                    const gnn_grads_t *gr, void *ws, size_t ws_bytes, hipStream_t s);
int backward(const gnn_graph_t *g, const gnn_params_t *p, int T, const float *e_all,
             const float *H_all, const float *Q_all, const float *grad_out, const gnn_grads_t *gr,
             void *ws, size_t ws_bytes, hipStream_t s);

int edge_bwd(const float *H, const gnn_graph_t *g, const gnn_params_t *p, const float *e, const float *ge, float *gH,
             const gnn_grads_t *gr, void *ws, size_t ws_bytes, hipStream_t s);
int node_bwd(const float *H, const float *e, const float *Hn, const gnn_graph_t *g, const gnn_params_t *p,
             const float *gHn, float *gH, float *ge, const gnn_grads_t *gr, void *ws, size_t ws_bytes, hipStream_t s);

}  // namespace gnn
