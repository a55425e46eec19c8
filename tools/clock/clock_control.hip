// TOOLING (not product): control experiment for the shader-clock question of DESIGN.md section 5.
// The conv kernels' in-kernel stamps (s_memtime ticks / s_memrealtime 100 MHz ticks) read 1.7 - 1.9 GHz under matrix load while
// GRBM_GUI_ACTIVE / duration reads 2.3 - 2.46 GHz.  This program runs loops of KNOWN content through THE SAME two stamps:
//   mode 0  pure v_mfma_f32_32x32x16_f16, 4 independent accumulators, 12 per trip (the guide's peak loop)
//   mode 1  + 8 ds_read_b128 per 12 MFMAs (what conv_pl's 128 x 128 tile issues per slab)
//   mode 2  + 4 LDS-DMA instructions (buffer_load ... lds, 16 B / lane = 1 KiB each) per 12 MFMAs out of an L2-resident window
//           (conv_pl's operand fetch: 16 KiB per slab and workgroup)
//   mode 3  mode 2 with the window = 1 GiB (the same fetch rate straight from HBM: memory-bound reference)
//   mode 4  no MFMA: a dependent v_fma_f32 chain (light-load reference for the stamps)
// per mode and occupancy (1 or 2 waves per SIMD): clock from the stamps (median / min / max over workgroups), f16-MFMA TFLOP/s
// and L2 / HBM -> LDS GB/s from HIP events.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define LDSP(p) ((__attribute__((address_space(3))) void*)(p))
// ds_read_b128 as asm: the compiler neither narrows it to the dwords a dummy consumer uses nor waits for it behind every MFMA
#define RD128(dst, ptr) asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"((unsigned)(unsigned long)(__attribute__((address_space(3))) const void*)(ptr)))

template <int MODE>
__global__ __launch_bounds__(256, 2) void loop_kernel(unsigned long long* stamps, float* sink, const u32x4* window, unsigned win_bytes,
                                                      int iters) {
    __shared__ __attribute__((aligned(1024))) u32x4 lds[2048];      // 32 KiB: 16 KiB read area + 16 KiB DMA landing area
    const int tid = threadIdx.x, wave = tid >> 6;
    for (int i = tid; i < 2048; i += 256) lds[i] = u32x4{(unsigned)i, 1u, 2u, 3u};
    __syncthreads();
    f16x8 a, b;
    for (int e = 0; e < 8; ++e) {
        a[e] = (_Float16)(0.001f * (tid + e));
        b[e] = (_Float16)(0.002f * (tid - e));
    }
    f32x16 c0, c1, c2, c3;
    for (int r = 0; r < 16; ++r) c0[r] = c1[r] = c2[r] = c3[r] = 0.f;
    float chain = (float)tid;
    unsigned lacc = 0;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32x4*>(window), 0, (int)win_bytes, 0x00020000);
    unsigned goff = ((unsigned)blockIdx.x * 256u + (unsigned)tid) * 16u % win_bytes;
    const unsigned gstep = 4096u * 61u;          // co-prime-ish stride through the window, 16-byte aligned
    const unsigned long long t0 = __builtin_readcyclecounter();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 4) {
#pragma unroll
            for (int u = 0; u < 64; ++u) chain = chain * 1.0000001f + 0.5f;
            continue;
        }
        u32x4 rd[8];
        const int rbase = (tid * 3 + it * 7) & 1023;
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
            if (MODE >= 1) RD128(rd[(u * 4 + 0) & 7], lds + ((rbase + u * 64) & 1023));
            if (MODE >= 2 && u == 0) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDSP(lds + 1024 + wave * 256), 16, goff, 0, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDSP(lds + 1024 + wave * 256 + 64), 16, goff, 4096, 0, 0);
            }
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
            if (MODE >= 1) RD128(rd[(u * 4 + 1) & 7], lds + ((rbase + u * 64 + 16) & 1023));
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c2, 0, 0, 0);
            if (MODE >= 1 && u < 2) RD128(rd[(u * 4 + 2) & 7], lds + ((rbase + u * 64 + 32) & 1023));
            if (MODE >= 2 && u == 1) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDSP(lds + 1024 + wave * 256 + 128), 16, goff, 8192, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDSP(lds + 1024 + wave * 256 + 192), 16, goff, 12288, 0, 0);
            }
            c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c3, 0, 0, 0);
            if (MODE >= 1 && u < 2) RD128(rd[(u * 4 + 3) & 7], lds + ((rbase + u * 64 + 48) & 1023));
        }
        if (MODE >= 1) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int k = 0; k < 8; ++k) asm volatile("" ::"v"(rd[k]));      // (the fragments are "used": nothing narrows the reads)
        }
        if (MODE >= 2) {
            goff += gstep;
            if (goff >= win_bytes - 16384u) goff -= (win_bytes - 16384u);
            __builtin_amdgcn_s_waitcnt(0x0f70 | 12);      // vmcnt <= 12: three trips of fetches in flight (expcnt / lgkmcnt untouched)
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    __builtin_amdgcn_s_waitcnt(0);
    float s = chain + (float)lacc + (float)lds[1024 + tid][0];
    for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
    sink[(long)blockIdx.x * 256 + tid] = s;
    if (tid == 0) {
        stamps[blockIdx.x * 4 + 0] = t0;
        stamps[blockIdx.x * 4 + 1] = t1;
        stamps[blockIdx.x * 4 + 2] = r0;
        stamps[blockIdx.x * 4 + 3] = r1;
    }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

template <int MODE>
void launch(int blocks, unsigned long long* st, float* sink, const u32x4* win, unsigned wb, int iters) {
    hipLaunchKernelGGL((loop_kernel<MODE>), dim3(blocks), dim3(256), 0, 0, st, sink, win, wb, iters);
}

int main(int argc, char** argv) {
    const int scale = argc > 1 ? atoi(argv[1]) : 1;
    unsigned long long* d_stamps;
    float* d_sink;
    u32x4* d_win;
    const size_t big = 1UL << 30;
    CK(hipMalloc(&d_stamps, 1024 * 4 * sizeof(unsigned long long)));
    CK(hipMalloc(&d_sink, 1024 * 256 * sizeof(float)));
    CK(hipMalloc(&d_win, big));
    CK(hipMemset(d_win, 1, big));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    printf("mode  waves/SIMD  workgroups  trips    event_ms   TFLOP/s(f16 MFMA)   clock_GHz by the stamps (median  min  max)   fetch GB/s -> LDS\n");
    for (int wps = 1; wps <= 2; ++wps)
        for (int mode = 0; mode < 5; ++mode) {
            const int blocks = 256 * wps;
            const int iters = (mode == 3 ? 40000 : (mode == 4 ? 400000 : 300000)) * scale;
            const unsigned wb = mode == 3 ? (unsigned)big : (2u << 20);
            for (int rep = 0; rep < 2; ++rep) {      // the first repetition brings the clocks up
                CK(hipEventRecord(e0, 0));
                switch (mode) {
                    case 0: launch<0>(blocks, d_stamps, d_sink, d_win, wb, iters); break;
                    case 1: launch<1>(blocks, d_stamps, d_sink, d_win, wb, iters); break;
                    case 2: launch<2>(blocks, d_stamps, d_sink, d_win, wb, iters); break;
                    case 3: launch<2>(blocks, d_stamps, d_sink, d_win, wb, iters); break;
                    default: launch<4>(blocks, d_stamps, d_sink, d_win, wb, iters); break;
                }
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
            const double flops = mode == 4 ? 0.0 : (double)blocks * 4 * iters * 12 * 2.0 * 32 * 32 * 16;
            const double gbs = (mode == 2 || mode == 3) ? (double)blocks * 4 * iters * 4096.0 / (ms * 1e-3) / 1e9 : 0.0;
            printf("%d     %d           %4d        %8d %9.3f   %10.1f          %.3f  %.3f  %.3f      %.0f\n", mode, wps, blocks, iters, ms,
                   flops / (ms * 1e-3) / 1e12, ghz[ghz.size() / 2], ghz.front(), ghz.back(), gbs);
            fflush(stdout);
        }
    return 0;
}
