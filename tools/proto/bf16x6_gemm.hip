// Prototype (tooling, not product): f32 GEMM C = A(MxK) * B(KxN) through 6 bf16 MFMAs per k16 block
// (3-way truncation split of both operands, products with i+j <= 4), to measure accuracy and raw speed
// against the f32 MFMA.  A is [M][K] row-major, B is [N][K] row-major (k contiguous for both).
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) short short8;
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split3(float x, uint32_t& a1, uint32_t& a2, uint32_t& a3) {
    const uint32_t u = __builtin_bit_cast(uint32_t, x);
    a1 = u & 0xFFFF0000u;
    const float r1 = x - __builtin_bit_cast(float, a1);
    const uint32_t u2 = __builtin_bit_cast(uint32_t, r1);
    a2 = u2 & 0xFFFF0000u;
    const float r2 = r1 - __builtin_bit_cast(float, a2);
    a3 = __builtin_bit_cast(uint32_t, r2) & 0xFFFF0000u;
}

// one wave computes a 32x32 tile; grid (N/32, M/32); operands split on the fly from global (no LDS):
// this measures numerics and the MFMA-side cost, not a tuned pipeline.
template <int MODE>  // 0: bf16x6, 1: f32 mfma
__global__ __launch_bounds__(64) void gemm_kernel(const float* A, const float* B, float* C, int M, int N, int K) {
    const int lane = threadIdx.x;
    const int i = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    if (MODE == 0) {
        for (int k0 = 0; k0 < K; k0 += 16) {
            uint32_t pa[3][4], pb[3][4];
            for (int q = 0; q < 4; ++q) {
                uint32_t lo[3], hi[3];
                split3(A[(long)(m0 + i) * K + k0 + 8 * h + 2 * q], lo[0], lo[1], lo[2]);
                split3(A[(long)(m0 + i) * K + k0 + 8 * h + 2 * q + 1], hi[0], hi[1], hi[2]);
                for (int p = 0; p < 3; ++p) pa[p][q] = (lo[p] >> 16) | hi[p];
                split3(B[(long)(n0 + i) * K + k0 + 8 * h + 2 * q], lo[0], lo[1], lo[2]);
                split3(B[(long)(n0 + i) * K + k0 + 8 * h + 2 * q + 1], hi[0], hi[1], hi[2]);
                for (int p = 0; p < 3; ++p) pb[p][q] = (lo[p] >> 16) | hi[p];
            }
            bf16x8 a[3], b[3];
            for (int p = 0; p < 3; ++p) {
                a[p] = __builtin_bit_cast(bf16x8, *(f32x4*)pa[p]);
                b[p] = __builtin_bit_cast(bf16x8, *(f32x4*)pb[p]);
            }
            // smallest terms first
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc, 0, 0, 0);
        }
    } else {
        for (int k0 = 0; k0 < K; k0 += 2)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[(long)(m0 + i) * K + k0 + h], B[(long)(n0 + i) * K + k0 + h], acc,
                                                       0, 0, 0);
    }
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
        C[(long)(m0 + row) * N + n0 + i] = acc[r];
    }
}

// raw issue-rate probe: registers only
template <int MODE>
__global__ __launch_bounds__(256) void rate_kernel(float* out, int iters) {
    f32x16 acc[4];
    for (int t = 0; t < 4; ++t)
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    bf16x8 a, b;
    for (int q = 0; q < 8; ++q) { a[q] = (__bf16)(threadIdx.x * 0.001f + q); b[q] = (__bf16)(q * 0.5f); }
    float fa = threadIdx.x * 0.001f, fb = 0.5f;
    for (int it = 0; it < iters; ++it) {
        for (int t = 0; t < 4; ++t) {
            if (MODE == 0) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[t], 0, 0, 0);
            else acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc[t], 0, 0, 0);
        }
    }
    float s = 0;
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

extern "C" void proto_gemm(int mode, const float* A, const float* B, float* C, int M, int N, int K, hipStream_t st) {
    dim3 g(N / 32, M / 32);
    if (mode == 0) hipLaunchKernelGGL(gemm_kernel<0>, g, dim3(64), 0, st, A, B, C, M, N, K);
    else hipLaunchKernelGGL(gemm_kernel<1>, g, dim3(64), 0, st, A, B, C, M, N, K);
}
extern "C" void proto_rate(int mode, float* out, int blocks, int iters, hipStream_t st) {
    if (mode == 0) hipLaunchKernelGGL(rate_kernel<0>, dim3(blocks), dim3(256), 0, st, out, iters);
    else hipLaunchKernelGGL(rate_kernel<1>, dim3(blocks), dim3(256), 0, st, out, iters);
}
