// TOOLING (not product): calibrates rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 against KNOWN byte counts for the access
// widths the planes kernels use (16 B / lane streams, 8 B / lane epilogue loads and stores, 4 B / lane), so that
// roofline.traffic of the dgrad launches -- 16-byte LDS-DMA operand streams plus 8-byte mask / old-value loads -- is one number
// instead of "between the doubled and the undoubled reading" (VERDICT round 3, weak #6).  Run under
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE  (and, separately, WRITE_SIZE; TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum)
// Each kernel touches BYTES bytes exactly once (1 GiB: past the 256 MiB Infinity Cache).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
constexpr long BYTES = 1L << 30;

__global__ void read16_kernel(const u32x4* p, unsigned* sink) {
    unsigned a = 0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < BYTES / 16; i += (long)gridDim.x * 256) a ^= p[i][0];
    if (a == 0x12345) *sink = a;
}
__global__ void read8_kernel(const u32x2* p, unsigned* sink) {
    unsigned a = 0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < BYTES / 8; i += (long)gridDim.x * 256) a ^= p[i][0];
    if (a == 0x12345) *sink = a;
}
__global__ void read4_kernel(const unsigned* p, unsigned* sink) {
    unsigned a = 0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < BYTES / 4; i += (long)gridDim.x * 256) a ^= p[i];
    if (a == 0x12345) *sink = a;
}
// 8 B / lane with a 16-byte lane pitch (the epilogue pattern of conv_pl: lane = pixel, 4 of its 8 channels): half of every 16 bytes
__global__ void read8_pitch16_kernel(const u32x4* p, unsigned* sink) {
    unsigned a = 0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < BYTES / 16; i += (long)gridDim.x * 256)
        a ^= reinterpret_cast<const u32x2*>(p + i)[0][0];
    if (a == 0x12345) *sink = a;
}
__global__ void lds_dma16_kernel(const u32x4* p, unsigned* sink) {
    __shared__ __attribute__((aligned(1024))) unsigned lds[256 * 4];
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32x4*>(p), 0, 0x7fffffff, 0x00020000);
    for (long base = (long)blockIdx.x * 256; base < BYTES / 16; base += (long)gridDim.x * 256) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(lds + (threadIdx.x >> 6) * 256), 16,
                                                 (unsigned)((base + threadIdx.x) * 16), 0, 0, 0);
    }
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (lds[threadIdx.x] == 0x12345) *sink = 1;
}
__global__ void write16_kernel(u32x4* p) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < BYTES / 16; i += (long)gridDim.x * 256) p[i] = u32x4{1u, 2u, 3u, (unsigned)i};
}
__global__ void write8_kernel(u32x2* p) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < BYTES / 8; i += (long)gridDim.x * 256) p[i] = u32x2{1u, (unsigned)i};
}
__global__ void write8_pitch16_kernel(u32x4* p) {      // both halves of every 16 bytes, by two separate 8-byte stores (hi / lo plane rows)
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < BYTES / 16; i += (long)gridDim.x * 256) {
        reinterpret_cast<u32x2*>(p + i)[0] = u32x2{1u, (unsigned)i};
    }
}
int main() {
    void* buf;
    unsigned* sink;
    if (hipMalloc(&buf, BYTES) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) return 1;
    hipMemset(buf, 1, BYTES);
    hipDeviceSynchronize();
    const dim3 g(4096), b(256);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(read16_kernel, g, b, 0, 0, (const u32x4*)buf, sink);
        hipLaunchKernelGGL(read8_kernel, g, b, 0, 0, (const u32x2*)buf, sink);
        hipLaunchKernelGGL(read4_kernel, g, b, 0, 0, (const unsigned*)buf, sink);
        hipLaunchKernelGGL(read8_pitch16_kernel, g, b, 0, 0, (const u32x4*)buf, sink);
        hipLaunchKernelGGL(lds_dma16_kernel, g, b, 0, 0, (const u32x4*)buf, sink);
        hipLaunchKernelGGL(write16_kernel, g, b, 0, 0, (u32x4*)buf);
        hipLaunchKernelGGL(write8_kernel, g, b, 0, 0, (u32x2*)buf);
        hipLaunchKernelGGL(write8_pitch16_kernel, g, b, 0, 0, (u32x4*)buf);
        hipDeviceSynchronize();
    }
    printf("known bytes per kernel: %ld (read8_pitch16 / write8_pitch16 request half of every 16 bytes: %ld useful)\n", BYTES, BYTES / 2);
    return 0;
}
