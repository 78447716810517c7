// Streaming ceiling of this GPU, measured the way MI355X_MICROARCH.md quotes it (float4 copy, 6.29 TB/s
// there): hand-written copy / read-only / 2-reads-1-write kernels over 1 GiB, best of 20, HIP events.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/hbm_copy_probe tools/hbm_copy_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(256) void k_copy(const float4 *__restrict__ a, float4 *__restrict__ b, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) b[i] = a[i];
}
__global__ __launch_bounds__(256) void k_read(const float4 *__restrict__ a, float *out, size_t n)
{
    float s = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float4 v = a[i];
        s += v.x + v.y + v.z + v.w;
    }
    if (s == 123.456f) out[0] = s;
}
__global__ __launch_bounds__(256) void k_triad(const float4 *__restrict__ a, const float4 *__restrict__ b,
                                               float4 *__restrict__ c, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float4 x = a[i], y = b[i];
        c[i] = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
    }
}
int main()
{
    const size_t bytes = 1ull << 30, n = bytes / 16;
    float4 *a, *b, *c;
    float *o;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&c, bytes); hipMalloc(&o, 4);
    hipMemset(a, 1, bytes); hipMemset(b, 2, bytes); hipMemset(c, 0, bytes);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int grid : {2048, 8192, 32768}) {
        float best[3] = {1e9f, 1e9f, 1e9f};
        for (int rep = 0; rep < 20; ++rep) {
            float ms;
            hipEventRecord(e0); k_copy<<<grid, 256>>>(a, b, n); hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1); if (ms < best[0]) best[0] = ms;
            hipEventRecord(e0); k_read<<<grid, 256>>>(a, o, n); hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1); if (ms < best[1]) best[1] = ms;
            hipEventRecord(e0); k_triad<<<grid, 256>>>(a, b, c, n); hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1); if (ms < best[2]) best[2] = ms;
        }
        printf("grid %6d: copy (1R+1W) %.2f TB/s   read %.2f TB/s   2R+1W %.2f TB/s\n", grid,
               2.0 * bytes / best[0] / 1e9, 1.0 * bytes / best[1] / 1e9, 3.0 * bytes / best[2] / 1e9);
    }
    return 0;
}
