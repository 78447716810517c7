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

inline unsigned grid_for(int64_t n) { return (unsigned)((n + kBlock - 1) / kBlock); }
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

// forward of the plan-based pipeline (sell_pipeline.hip)
int sell_forward(const gnn_plan_t *pl, const gnn_params_t *p, int n_iters, float *e_out, void *ws,
                 size_t ws_bytes, hipStream_t s);
size_t sell_workspace_bytes(int64_t n_hits, int64_t n_segments, int F, int D);
int sell_shape_supported(int F, int D);
int sell_limits(int F, int D, int32_t *out4);

// backward.hip
int bce_loss(const float *e, const float *y, int64_t n, float scale, float *loss, float *grad_e,
             float *partial, hipStream_t s);
size_t backward_workspace_bytes(int64_t n_hits, int64_t n_segments, int F, int D);
int backward(const gnn_graph_t *g, const gnn_params_t *p, int T, const float *e_all,
             const float *H_all, const float *grad_out, const gnn_grads_t *gr, void *ws,
             size_t ws_bytes, hipStream_t s);

}  // namespace gnn
