// TOOLING (not product): control experiment for the shader-clock question of DESIGN.md section 5.
// The conv kernels' in-kernel stamps (s_memtime ticks / s_memrealtime 100 MHz ticks) read 1.7 - 1.9 GHz under matrix load while
// GRBM_GUI_ACTIVE / duration reads 2.3 - 2.46 GHz.  This program runs loops of known content through THE SAME two stamps:
//   mode 0  pure v_mfma_f32_32x32x16_f16, 4 independent accumulators (the guide's peak loop)
//   mode 1  the same + one ds_read_b128 per MFMA (conv_pl's loop issues ~0.7)
//   mode 2  mode 1 + one 16-byte global load per 3 MFMAs streaming a 1 GiB buffer (HBM traffic next to the matrix pipe)
//   mode 3  no MFMA: dependent v_fma_f32 chain (a light-load reference for the stamps)
// and prints, per mode: clock from the stamps (median over workgroups), TFLOP/s from HIP events, and the two in one line.
// waves per SIMD: 1 (256 workgroups x 256 threads ... one per CU) or 2 (512 workgroups).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void loop_kernel(unsigned long long* stamps, float* sink, const u32x4* stream, long stream_n,
                                                   int iters, int mode) {
    __shared__ u32x4 lds[1024];
    const int tid = threadIdx.x;
    for (int i = tid; i < 1024; i += 256) lds[i] = u32x4{(unsigned)i, 1u, 2u, 3u};
    __syncthreads();
    f16x8 a, b;
    for (int e = 0; e < 8; ++e) {
        a[e] = (_Float16)(0.001f * (tid + e));
        b[e] = (_Float16)(0.002f * (tid - e));
    }
    f32x16 acc[4];
    for (int u = 0; u < 4; ++u)
        for (int r = 0; r < 16; ++r) acc[u][r] = 0.f;
    float chain = (float)tid;
    u32x4 l = u32x4{0, 0, 0, 0}, gacc = u32x4{0, 0, 0, 0};
    long gi = ((long)blockIdx.x * 256 + tid) % stream_n;
    const unsigned long long t0 = __builtin_readcyclecounter();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        if (mode == 3) {
#pragma unroll
            for (int u = 0; u < 64; ++u) chain = chain * 1.0000001f + 0.5f;
            continue;
        }
#pragma unroll
        for (int u = 0; u < 12; ++u) {
            acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[u & 3], 0, 0, 0);
            if (mode >= 1) {
                const u32x4 v = lds[(tid * 4 + u * 67 + it) & 1023];
                l[0] ^= v[0];
                l[1] += v[3];
            }
            if (mode >= 2 && (u % 3) == 0) {
                const u32x4 v = __builtin_nontemporal_load(stream + gi);
                gacc[0] ^= v[0];
                gi += 256L * gridDim.x;
                if (gi >= stream_n) gi -= stream_n;
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    float s = chain + (float)(l[0] + l[1] + gacc[0]);
    for (int u = 0; u < 4; ++u)
        for (int r = 0; r < 16; ++r) s += acc[u][r];
    sink[(long)blockIdx.x * 256 + tid] = s;
    if (tid == 0) {
        stamps[blockIdx.x * 4 + 0] = t0;
        stamps[blockIdx.x * 4 + 1] = t1;
        stamps[blockIdx.x * 4 + 2] = r0;
        stamps[blockIdx.x * 4 + 3] = r1;
    }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

int main(int argc, char** argv) {
    const int seconds_scale = argc > 1 ? atoi(argv[1]) : 1;
    unsigned long long* d_stamps;
    float* d_sink;
    u32x4* d_stream;
    const long stream_n = (1L << 30) / 16;
    CK(hipMalloc(&d_stamps, 1024 * 4 * sizeof(unsigned long long)));
    CK(hipMalloc(&d_sink, 1024 * 256 * sizeof(float)));
    CK(hipMalloc(&d_stream, stream_n * 16));
    CK(hipMemset(d_stream, 1, stream_n * 16));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    printf("mode waves/SIMD  workgroups  iters   event_ms   TFLOP/s(f16 MFMA)  clock_GHz(stamps: median  min  max)  GB/s(mode 2)\n");
    for (int wps = 1; wps <= 2; ++wps)
        for (int mode = 0; mode < 4; ++mode) {
            const int blocks = 256 * wps;
            const int iters = (mode == 3 ? 40000 : 30000) * seconds_scale;
            for (int rep = 0; rep < 2; ++rep) {      // first repetition warms the clocks up
                CK(hipEventRecord(e0, 0));
                hipLaunchKernelGGL(loop_kernel, dim3(blocks), dim3(256), 0, 0, d_stamps, d_sink, d_stream, stream_n, iters, mode);
                CK(hipEventRecord(e1, 0));
                CK(hipDeviceSynchronize());
            }
            float ms = 0.f;
            CK(hipEventElapsedTime(&ms, e0, e1));
            std::vector<unsigned long long> st(blocks * 4);
            CK(hipMemcpy(st.data(), d_stamps, blocks * 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
            std::vector<double> ghz;
            for (int b = 0; b < blocks; ++b) {
                const double ticks = (double)(st[b * 4 + 1] - st[b * 4 + 0]), real = (double)(st[b * 4 + 3] - st[b * 4 + 2]);
                if (real > 0) ghz.push_back(ticks / real * 0.1);     // real-time counter: 100 MHz
            }
            std::sort(ghz.begin(), ghz.end());
            const double flops = mode == 3 ? 0.0 : (double)blocks * 4 * iters * 12 * 2.0 * 32 * 32 * 16;
            const double gbs = mode == 2 ? (double)blocks * 256 * iters * 4 * 16 / (ms * 1e-3) / 1e9 : 0.0;
            printf("%d    %d           %4d     %6d   %8.3f   %10.1f        %.3f  %.3f  %.3f      %.0f\n", mode, wps, blocks, iters, ms,
                   flops / (ms * 1e-3) / 1e12, ghz[ghz.size() / 2], ghz.front(), ghz.back(), gbs);
            fflush(stdout);
        }
    return 0;
}
