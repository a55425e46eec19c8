// Planes tensors (planes.h): conversion from / to fp32 NCHW, scale bookkeeping, and the HBM-bound layers of the backbone
// in the planes layout (pools, global pool, channel sums) -- the elementwise side of the path behind
// /root/reference/ssn_models.py:266, gfx950.  One thread = one pixel x 8 channels: 16-byte loads / stores per plane.
#include "planes.h"

namespace {

using namespace pl;

// ---- scale bookkeeping ----
// scale[i] <- scale for the next step from amax[i] (kept when the tensor was not written), amax[i] <- 0.
// flag[0] |= 1 when a tensor outgrew the scale it was stored with (values were clamped: the host re-calibrates);
// flag[1] counts the slots whose scale changed (calibration runs until this stays 0).
__global__ __launch_bounds__(256) void scales_update_kernel(float* amax, float* scale, int* flag, int n, int exact) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float a = amax[i], s = scale[i];
    if (a * s >= PL_F16_MAX || a != a) atomicOr(flag, 1);
    float ns = pl_scale_from_amax(a, s);
    if (exact) ns *= (float)(1 << PL_HEADROOM_BITS);   // a tensor measured before it is stored needs no head-room
    // hysteresis: keep the old scale while the stored maximum stays inside [2^10, 2^14) -- at least 2 bits of head-room, at
    // most 3 bits of the low plane's range given away -- so that scales do not flap between steps
    if (!exact && a * s < 16384.f && a * s >= 1024.f) ns = s;
    if (ns != s) atomicAdd(flag + 1, 1);
    scale[i] = ns;
    amax[i] = 0.f;
}

struct CvtArgs {
    const float* x;
    void* hi;
    void* lo;
    const float* scale;
    float* amax;
    long x_img_stride;       // floats between images of x
    int N, C, H, W;          // x dims (C real channels)
    int G;                   // channel groups written (ceil(C' / 8), C' = C or 4C)
    long img_groups;         // groups of the whole planes tensor
    int s2d;                 // space-to-depth: planes channel (c*2+a)*2+b at (h', w') = x[c][2h'+a][2w'+b]
};

__global__ __launch_bounds__(256) void pl_from_f32_kernel(CvtArgs p) {
    const int HWo = p.s2d ? (p.H / 2) * (p.W / 2) : p.H * p.W;
    const long total = (long)p.N * p.G * HWo;
    const float s = *p.scale;
    float vmax = 0.f;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int q = (int)(idx % HWo);
        const long ng = idx / HWo;
        const int g = (int)(ng % p.G), n = (int)(ng / p.G);
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = 8 * g + e;
            float x = 0.f;
            if (!p.s2d) {
                if (c < p.C) x = p.x[(long)n * p.x_img_stride + (long)c * HWo + q];
            } else if (c < 4 * p.C) {
                const int cc = c >> 2, a = (c >> 1) & 1, b = c & 1, Wo = p.W / 2;
                const int ho = q / Wo, wo = q - ho * Wo;
                x = p.x[(long)n * p.x_img_stride + ((long)cc * p.H + 2 * ho + a) * p.W + 2 * wo + b];
            }
            vmax = fmaxf(vmax, fabsf(x));
            v[e] = pl_clamp(x * s);
        }
        u32x4 hi, lo;
        pl_split8(v, hi, lo);
        const long o = (((long)n * p.img_groups + g) * HWo + q);
        reinterpret_cast<u32x4*>(p.hi)[o] = hi;
        reinterpret_cast<u32x4*>(p.lo)[o] = lo;
    }
    amax_emit(p.amax, vmax);
}

__global__ __launch_bounds__(256) void pl_to_f32_kernel(const void* hi, const void* lo, long img_groups, float* y,
                                                       long y_img_stride, int N, int C, int HW, const float* scale) {
    const int G = (C + 7) / 8;
    const long total = (long)N * G * HW;
    const float inv = 1.f / *scale;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int q = (int)(idx % HW);
        const long ng = idx / HW;
        const int g = (int)(ng % G), n = (int)(ng / G);
        const long o = (((long)n * img_groups + g) * HW + q);
        float v[8];
        pl_join8(reinterpret_cast<const u32x4*>(hi)[o], reinterpret_cast<const u32x4*>(lo)[o], v);
#pragma unroll
        for (int e = 0; e < 8; ++e)
            if (8 * g + e < C) y[(long)n * y_img_stride + (long)(8 * g + e) * HW + q] = v[e] * inv;
    }
}

int grid_for(long total) {
    long b = (total + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 65536 ? 65536 : b));
}

}  // namespace

// ------------------------------------------------------------------------------------------ C ABI
// amax / scale: `n` consecutive slots.  exact != 0: scales without head-room (tensors measured before they are stored: the
// caller's frames).  flag: int[2] = {overflow (sticky), slots whose scale changed (cumulative)}.
extern "C" int ssn_pl_scales_update(float* amax, float* scale, int* flag, int n, int exact, hipStream_t stream) {
    SSN_CHECK_ARG(amax && scale && flag && n >= 0, "pl scales update: bad arguments");
    if (n == 0) return SSN_OK;
    hipLaunchKernelGGL(scales_update_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, amax, scale, flag, n, exact);
    SSN_CHECK_LAUNCH("pl_scales_update");
    return SSN_OK;
}

// fp32 NCHW (channel slice: x = first channel, x_img_stride floats between images) -> planes at (hi, lo) = the first
// group of the destination slice.  s2d: the space-to-depth view of the stride-2 stem (4C channels at H/2 x W/2).
// Channels past C (resp. 4C) up to the next multiple of 8 are written as zeros.  amax may be null.
extern "C" int ssn_pl_from_f32(const float* x, long x_img_stride, void* hi, void* lo, int N, int C, int H, int W,
                               long img_groups, int s2d, const float* scale, float* amax, hipStream_t stream) {
    SSN_CHECK_ARG(x && hi && lo && scale && N > 0 && C > 0 && H > 0 && W > 0, "pl from f32: bad arguments");
    SSN_CHECK_ARG(!s2d || (H % 2 == 0 && W % 2 == 0), "pl from f32: space-to-depth needs even sizes");
    CvtArgs a;
    a.x = x;
    a.hi = hi;
    a.lo = lo;
    a.scale = scale;
    a.amax = amax;
    a.x_img_stride = x_img_stride;
    a.N = N;
    a.C = C;
    a.H = H;
    a.W = W;
    a.G = ((s2d ? 4 * C : C) + 7) / 8;
    a.img_groups = img_groups;
    a.s2d = s2d;
    SSN_CHECK_ARG(img_groups >= a.G, "pl from f32: slice wider than its tensor");
    const long total = (long)N * a.G * (s2d ? (H / 2) * (W / 2) : H * W);
    hipLaunchKernelGGL(pl_from_f32_kernel, dim3(grid_for(total)), dim3(256), 0, stream, a);
    SSN_CHECK_LAUNCH("pl_from_f32");
    return SSN_OK;
}

extern "C" int ssn_pl_to_f32(const void* hi, const void* lo, long img_groups, float* y, long y_img_stride, int N, int C,
                             int HW, const float* scale, hipStream_t stream) {
    SSN_CHECK_ARG(hi && lo && y && scale && N > 0 && C > 0 && HW > 0, "pl to f32: bad arguments");
    const long total = (long)N * ((C + 7) / 8) * HW;
    hipLaunchKernelGGL(pl_to_f32_kernel, dim3(grid_for(total)), dim3(256), 0, stream, hi, lo, img_groups, y, y_img_stride, N, C,
                       HW, scale);
    SSN_CHECK_LAUNCH("pl_to_f32");
    return SSN_OK;
}
