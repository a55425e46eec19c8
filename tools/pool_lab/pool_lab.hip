// Tooling (not product code): what bounds the planes pool kernels?  The product kernels of csrc/planes_ops.hip (included as they are)
// and experimental variants on the bench's stem-pool shapes, timed with HIP events; bytes = what the kernel must move.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I action-detection_amd/csrc tools/pool_lab/pool_lab.hip -o tools/pool_lab/pool_lab
#include "../../action-detection_amd/csrc/planes_ops.hip"

#include <cstdio>
#include <vector>

void ssn_set_error(const char*, ...) {}
namespace {

__global__ void fill_kernel(uint32_t* p, size_t n, uint32_t seed, uint32_t mask, uint32_t orv) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        uint32_t x = (uint32_t)i * 2654435761u + seed;
        x ^= x >> 13;
        x *= 0x9E3779B1u;
        p[i] = (x & mask) | orv;
    }
}

// ---- ablations of pl_maxpool_bwd_k3s2_kernel<PAD>: MODE 1 = no stores (results folded into the amax), 2 = no loads (stores only)
template <int PAD, int MODE>
__global__ __launch_bounds__(256) void k3s2_ablate_kernel(PoolArgs p) {
    const int Hb = (p.H + 1) / 2, Wb = (p.W + 1) / 2;
    const uint32_t total = (uint32_t)p.N * (uint32_t)p.G * (uint32_t)Hb * (uint32_t)Wb;
    constexpr int base_off = (PAD + 1) / 2 - 1;
    float vmax = 0.f;
    for (uint32_t idx = blockIdx.x * 256u + threadIdx.x; idx < total; idx += gridDim.x * 256u) {
        const PoolIdx ix = pool_decode(idx, p.dv_wb, p.dv_hb, p.dv_g);
        const int j = (int)ix.w, i = (int)ix.h, g = (int)ix.g, n = (int)ix.n;
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (MODE != 2) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int ho = i + base_off + (t >> 1), wo = j + base_off + (t & 1);
                const bool ok = (unsigned)ho < (unsigned)p.Ho && (unsigned)wo < (unsigned)p.Wo;
                const long oo = ok ? (long)ho * p.Wo + wo : 0;
                const u32x2 am = reinterpret_cast<const u32x2*>(p.argmax)[((long)n * p.G + g) * p.Ho * p.Wo + oo];
                float d[8];
                load8(p.x_hi, p.x_lo, ((long)n * p.x_img_groups + g) * p.Ho * p.Wo + oo, d);
                const u32x4 pm = reinterpret_cast<const u32x4*>(p.mask_hi)[((long)n * p.mask_img_groups + g) * p.Ho * p.Wo + oo];
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += d[e] + (float)((am[e >> 2] >> (8 * (e & 3))) & 0xFFu) + __builtin_bit_cast(float, pm[e >> 1]);
            }
        }
        if (MODE == 1) {
#pragma unroll
            for (int e = 0; e < 8; ++e) vmax = fmaxf(vmax, acc[e]);
            continue;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int h = 2 * i + (q >> 1), w = 2 * j + (q & 1);
            if (h >= p.H || w >= p.W) continue;
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = acc[e] + (float)q;
            const long o = ((long)n * p.y_img_groups + g) * p.H * p.W + (long)h * p.W + w;
            if (p.y_f32) {
                if ((q & 1) == 0) {
                    const long hw = (long)p.H * p.W;
                    float* dst = p.y_f32 + (long)n * p.y_f32_img_stride + (long)(8 * g) * hw + (long)h * p.W + w;
#pragma unroll
                    for (int e = 0; e < 8; ++e) *reinterpret_cast<float2*>(dst + (long)e * hw) = float2{v[e], v[e] + 1.f};
                }
            } else {
                u32x4 hi, lo;
                pl_split8(v, hi, lo);
                reinterpret_cast<u32x4*>(p.y_hi)[o] = hi;
                reinterpret_cast<u32x4*>(p.y_lo)[o] = lo;
            }
        }
    }
    amax_emit(p.y_amax, vmax);
}

// ---- plain streams on the same buffers: what this memory system gives a kernel that reads R bytes and writes W bytes with 16-byte
// accesses per lane, fully coalesced (the roofline of the pool kernels)
__global__ __launch_bounds__(256) void stream_kernel(const u32x4* __restrict__ a, size_t na, u32x4* __restrict__ b, size_t nb) {
    const size_t n = na > nb ? na : nb;
    u32x4 acc = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        if (i < na) {
            const u32x4 v = a[i];
            acc[0] ^= v[0]; acc[1] ^= v[1]; acc[2] ^= v[2]; acc[3] ^= v[3];
        }
        if (i < nb) b[i] = u32x4{(uint32_t)i, acc[1], acc[2], acc[3]};
    }
    if (acc[0] == 0x12345678u && nb) b[0] = acc;
}

struct Timer {
    hipEvent_t s, e;
    Timer() { hipEventCreate(&s); hipEventCreate(&e); }
    template <class F>
    float run(F f, int reps = 20) {
        for (int i = 0; i < 3; ++i) f();
        hipDeviceSynchronize();
        hipEventRecord(s);
        for (int i = 0; i < reps; ++i) f();
        hipEventRecord(e);
        hipEventSynchronize(e);
        float ms;
        hipEventElapsedTime(&ms, s, e);
        return ms / reps;
    }
};

void* dmalloc(size_t bytes, uint32_t seed, uint32_t mask = 0x3BFF3BFFu, uint32_t orv = 0) {
    void* p;
    if (hipMalloc(&p, bytes + 4096) != hipSuccess) { printf("hipMalloc %zu failed\n", bytes); exit(1); }
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, (uint32_t*)p, bytes / 4, seed, mask, orv);
    return p;
}

}  // namespace

int main() {
    Timer tm;
    struct Case { const char* name; int N, C, H, W, pad; bool f32; } cases[] = {
        {"pool1 bwd (64 ch, 112 -> 56, fp32 dx)", 288, 64, 112, 112, 0, true},
        {"pool1 bwd, planes dx", 288, 64, 112, 112, 0, false},
        {"pool2 bwd (192 ch, 56 -> 28)", 288, 192, 56, 56, 0, false},
        {"3c pool bwd (320 ch, 28 -> 14)", 288, 320, 28, 28, 0, false},
        {"4e pool bwd (608 ch, 14 -> 7)", 288, 608, 14, 14, 0, false},
    };
    float* scales = (float*)dmalloc(64, 1);
    {
        float h[4] = {1.f, 1.f, 0.f, 0.f};
        hipMemcpy(scales, h, 16, hipMemcpyHostToDevice);
    }
    for (auto& c : cases) {
        const int Ho = (c.H + 2 * c.pad - 3 + 1) / 2 + 1, Wo = Ho;      // ceil mode (sizes of the bench)
        const int G = c.C / 8;
        const size_t in_px = (size_t)c.N * G * c.H * c.W, out_px = (size_t)c.N * G * Ho * Wo;
        void* dy_hi = dmalloc(out_px * 16, 2);
        void* dy_lo = dmalloc(out_px * 16, 3, 0x03FF03FFu);
        void* am = dmalloc(out_px * 8, 4, 0x07070707u);
        void* mk = dmalloc(out_px * 16, 5);
        void* dx_hi = dmalloc(in_px * 16, 6);
        void* dx_lo = dmalloc(in_px * 16, 7);
        float* aff = (float*)dmalloc(c.C * 4, 8, 0x007FFFFFu, 0x3F000000u);
        PoolArgs a;
        fill_pool(a, dy_hi, dy_lo, G, dx_hi, dx_lo, G, c.N, c.C, c.H, c.W, Ho, Wo, 3, 2, c.pad, scales, scales + 1, scales + 2, "lab");
        a.argmax = (unsigned char*)am;
        a.mask_hi = mk;
        a.aff_scale = aff;
        a.mask_img_groups = G;
        a.relu = 1;      // pooled mask
        if (c.f32) {
            a.y_f32 = (float*)dx_hi;      // (in_px * 16 bytes = half of the fp32 tensor; the other half lands in dx_lo's allocation? no:)
        }
        void* f32buf = nullptr;
        if (c.f32) {
            f32buf = dmalloc(in_px * 32, 9);
            a.y_f32 = (float*)f32buf;
            a.y_f32_img_stride = (long)c.C * c.H * c.W;
            a.y_img_groups = G;
        }
        const size_t rd = out_px * (16 + 16 + 8 + 16), wr = in_px * 32;
        const dim3 grid(grid_for((long)c.N * G * ((c.H + 1) / 2) * ((c.W + 1) / 2)));
        printf("%s: reads %.0f MB, writes %.0f MB\n", c.name, rd / 1e6, wr / 1e6);
        auto rep = [&](const char* what, float ms, size_t bytes) { printf("   %-44s %.4f ms  %.2f TB/s\n", what, ms, bytes / ms / 1e9); };
        rep("product kernel", tm.run([&] { hipLaunchKernelGGL(pl_maxpool_bwd_k3s2_kernel<0>, grid, dim3(256), 0, 0, a); }), rd + wr);
        if (!c.f32) {
            rep("fast kernel (pooled mask)", tm.run([&] { hipLaunchKernelGGL((pl_maxpool_bwd_k3s2_fast_kernel<0, true>), grid, dim3(256), 0, 0, a); }), rd + wr);
            PoolArgs b = a;
            b.mask_hi = nullptr;
            b.relu = 0;
            rep("fast kernel (no mask)", tm.run([&] { hipLaunchKernelGGL((pl_maxpool_bwd_k3s2_fast_kernel<0, false>), grid, dim3(256), 0, 0, b); }), rd - out_px * 16 + wr);
            rep("product kernel (no mask)", tm.run([&] { hipLaunchKernelGGL(pl_maxpool_bwd_k3s2_kernel<0>, grid, dim3(256), 0, 0, b); }), rd - out_px * 16 + wr);
            // forward of the same pool: x = the big tensor (dx buffers), y = the pooled one (dy buffers), argmax written
            PoolArgs f;
            fill_pool(f, dx_hi, dx_lo, G, dy_hi, dy_lo, G, c.N, c.C, c.H, c.W, Ho, Wo, 3, 2, c.pad, scales, scales + 1, scales + 2, "lab");
            f.argmax = (unsigned char*)am;
            const dim3 gf(grid_for((long)c.N * G * Ho * Wo));
            rep("FORWARD general kernel", tm.run([&] { hipLaunchKernelGGL((pl_maxpool_fwd_kernel<3, 2>), gf, dim3(256), 0, 0, f); }), in_px * 32 + out_px * 40);
            rep("FORWARD fast kernel", tm.run([&] { hipLaunchKernelGGL((pl_maxpool_fwd_k3_fast_kernel<2>), gf, dim3(256), 0, 0, f); }), in_px * 32 + out_px * 40);
        }
        rep("loads only", tm.run([&] { hipLaunchKernelGGL((k3s2_ablate_kernel<0, 1>), grid, dim3(256), 0, 0, a); }), rd);
        rep("stores only", tm.run([&] { hipLaunchKernelGGL((k3s2_ablate_kernel<0, 2>), grid, dim3(256), 0, 0, a); }), wr);
        rep("loads + stores, no selection arithmetic", tm.run([&] { hipLaunchKernelGGL((k3s2_ablate_kernel<0, 0>), grid, dim3(256), 0, 0, a); }), rd + wr);
        // plain streams of the same byte counts
        const size_t na = rd / 16, nb = wr / 16;
        void* sa = dmalloc(rd, 11);
        void* sb = c.f32 ? f32buf : dx_hi;      // (write stream: reuse an output allocation; planes: both halves back to back is close enough)
        void* sb2 = c.f32 ? nullptr : dmalloc(wr, 12);
        rep("plain stream: same reads + same writes", tm.run([&] { hipLaunchKernelGGL(stream_kernel, dim3(65536), dim3(256), 0, 0, (const u32x4*)sa, na, (u32x4*)(sb2 ? sb2 : sb), nb); }), rd + wr);
        rep("plain stream: writes only", tm.run([&] { hipLaunchKernelGGL(stream_kernel, dim3(65536), dim3(256), 0, 0, (const u32x4*)sa, (size_t)0, (u32x4*)(sb2 ? sb2 : sb), nb); }), wr);
        rep("plain stream: reads only", tm.run([&] { hipLaunchKernelGGL(stream_kernel, dim3(65536), dim3(256), 0, 0, (const u32x4*)sa, na, (u32x4*)(sb2 ? sb2 : sb), (size_t)0); }), rd);
        for (void* p : {dy_hi, dy_lo, am, mk, dx_hi, dx_lo, (void*)aff, sa}) hipFree(p);
        if (f32buf) hipFree(f32buf);
        if (sb2) hipFree(sb2);
    }
    return 0;
}
