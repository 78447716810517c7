// Checks the operand / result lane maps of v_mfma_f32_16x16x32_bf16 that csrc/sell_pipeline.hip's
// bf16 hit-update path relies on (exact small-integer data, asymmetric operands):
//   A fragment: lane l holds A[row = l & 15][k = 8 (l >> 4) + j], j = 0..7
//   B fragment: lane l holds B[k = 8 (l >> 4) + j][col = l & 15]
//   C/D:        lane l, register r holds D[row = 4 (l >> 4) + r][col = l & 15]
// build: hipcc --offload-arch=gfx950 -O2 -o /tmp/mfma_check tools/mfma_layout_check.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __host__ inline uint16_t to_bf16(float f)
{
    union { float f; uint32_t u; } v = {f};
    return (uint16_t)((v.u + 0x7FFF + ((v.u >> 16) & 1)) >> 16);     // round to nearest even
}

__global__ void k(const float *A, const float *B, float *D)   // A [16][32], B [32][16], D [16][16]
{
    const int l = threadIdx.x;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) {
        a[j] = (short)to_bf16(A[(l & 15) * 32 + 8 * (l >> 4) + j]);
        b[j] = (short)to_bf16(B[(8 * (l >> 4) + j) * 16 + (l & 15)]);
    }
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[(4 * (l >> 4) + r) * 16 + (l & 15)] = c[r];
}

int main()
{
    std::vector<float> A(16 * 32), B(32 * 16), D(256), R(256, 0.f);
    for (int i = 0; i < 16; ++i) for (int k = 0; k < 32; ++k) A[i * 32 + k] = (float)((i * 7 + k * 3) % 11 - 5);
    for (int k = 0; k < 32; ++k) for (int j = 0; j < 16; ++j) B[k * 16 + j] = (float)((k * 5 + j * 13) % 9 - 4);
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) for (int k = 0; k < 32; ++k) R[i * 16 + j] += A[i * 32 + k] * B[k * 16 + j];
    float *dA, *dB, *dD;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dD, 1024);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 256; ++i) bad += D[i] != R[i];
    printf("mfma_f32_16x16x32_bf16 layout check: %d mismatches of 256 (D[1][2]=%g ref %g)\n", bad, D[18], R[18]);
    return bad != 0;
}
