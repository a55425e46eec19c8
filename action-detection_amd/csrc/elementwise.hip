// HBM-bound helper kernels of the backbone: BN folding, the fused ReLU+frozen-BN backward mask,
// weight re-layout for dgrad, dropout, and the fused multi-tensor SGD step.
// All are plain coalesced grid-stride kernels (no LDS needed); float4 where the plane allows.
#include <stdarg.h>
#include <stdio.h>

#include "ssn_common.h"

// ---- error string (shared by every translation unit of the library) ----
static thread_local char g_err[512] = "";
void ssn_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* ssn_last_error(void) { return g_err; }
extern "C" int ssn_abi_version(void) { return 9; }

namespace {

// Frozen BatchNorm (eval mode, /root/reference/ssn_models.py:156-174) folded with the conv bias:
//   y = relu(conv(x) * scale + shift),  scale = gamma / sqrt(var + eps),
//   shift = (bias - mean) * scale + beta
__global__ void bn_fold_kernel(const float* bias, const float* gamma, const float* beta, const float* mean,
                               const float* var, float eps, float* scale, float* shift, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float s = gamma[c] / sqrtf(var[c] + eps);
    scale[c] = s;
    shift[c] = ((bias ? bias[c] : 0.f) - mean[c]) * s + beta[c];
}

// g = dy * (y > 0) * scale[c], in place on dy.  Tensors are NCHW channel slices.
__global__ __launch_bounds__(256) void relu_bn_bwd_kernel(float* dy, const float* y, const float* scale, int C,
                                                          int HW, long dy_img_stride, long y_img_stride,
                                                          long total, FastDiv div_chw, FastDiv div_hw, float* amax) {
    float vmax = 0.f;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        uint32_t n, rem, c, hw;
        fd_divmod((uint32_t)idx, div_chw, n, rem);
        fd_divmod(rem, div_hw, c, hw);
        const long di = (long)n * dy_img_stride + rem;
        const long yi = (long)n * y_img_stride + rem;
        const float yv = y[yi];
        const float sc = scale[c];     // NaN: not a ReLU / frozen-BN output (pass-through channel), as in the fused epilogues
        const float o = (sc != sc) ? dy[di] : ((yv > 0.f) ? dy[di] * sc : 0.f);
        dy[di] = o;
        vmax = fmaxf(vmax, fabsf(o));
    }
    amax_emit(amax, vmax);
}

// slot = max(slot, max |x|) over `n` contiguous floats: the amax slot of a tensor that no tracked kernel produced (the
// frames the caller hands in, test inputs).  The slot must hold a valid value (normally 0) on entry.
__global__ __launch_bounds__(256) void tensor_amax_kernel(const float* x, long n, float* slot) {
    float vmax = 0.f;
    const long n4 = n >> 2;
    const bool al = ((uintptr_t)x & 15) == 0;
    if (al) {
        const f32x4* x4 = reinterpret_cast<const f32x4*>(x);
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
            const f32x4 v = x4[i];
            vmax = fmaxf(fmaxf(vmax, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
        }
    }
    for (long i = (al ? n4 * 4 : 0) + (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
        vmax = fmaxf(vmax, fabsf(x[i]));
    amax_emit(slot, vmax);
}

// out[c] = sum over images and pixels of g[n][c][:], in a fixed order (deterministic): the bias gradient of a projection
// whose pooling runs behind it (the weight-gradient kernel's own bias column would sum the POOLED gradient, which
// differs at the image border).  Two passes: workgroup (c, s) sums the images of share s of channel c (lanes along
// the contiguous pixels, no divisions) into part[c][s]; the second pass adds the shares of a channel in order.
__global__ __launch_bounds__(256) void channel_sum_part_kernel(const float* g, float* part, int N, int HW,
                                                               long img_stride, int S) {
    __shared__ float red[4];
    const int c = blockIdx.x, s = blockIdx.y, tid = threadIdx.x;
    const int n0 = (int)((long)N * s / S), n1 = (int)((long)N * (s + 1) / S);
    float a0 = 0.f, a1 = 0.f;
    for (int n = n0; n < n1; ++n) {
        const float* row = g + (long)n * img_stride + (long)c * HW;
        int hw = tid;
        for (; hw + 256 < HW; hw += 512) {
            a0 += row[hw];
            a1 += row[hw + 256];
        }
        if (hw < HW) a0 += row[hw];
    }
    const float v = wave_sum(a0 + a1);
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    if (tid == 0) part[(long)c * S + s] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ __launch_bounds__(256) void channel_sum_final_kernel(const float* part, float* out, int C, int S) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    float t = 0.f;
    for (int s = 0; s < S; ++s) t += part[(long)c * S + s];
    out[c] = t;
}

// Counter-based RNG for dropout: Philox-4x32-10 keyed by (seed), counter = element index / 4.
__device__ __forceinline__ void philox_round(uint32_t& c0, uint32_t& c1, uint32_t& c2, uint32_t& c3, uint32_t k0,
                                             uint32_t k1) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
    const uint32_t hi0 = __umulhi(M0, c0), lo0 = M0 * c0;
    const uint32_t hi1 = __umulhi(M1, c2), lo1 = M1 * c2;
    const uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
    c0 = n0;
    c1 = n1;
    c2 = n2;
    c3 = n3;
}
__device__ __forceinline__ void philox4(uint64_t seed, uint64_t ctr, uint32_t out[4]) {
    uint32_t c0 = (uint32_t)ctr, c1 = (uint32_t)(ctr >> 32), c2 = 0, c3 = 0;
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        philox_round(c0, c1, c2, c3, k0, k1);
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0;
    out[1] = c1;
    out[2] = c2;
    out[3] = c3;
}

// y = x * mask / (1-p); mask bit saved as uint8 (1 = kept).  nn.Dropout semantics
// (/root/reference/ssn_models.py:74); the reference RNG stream itself is not reproducible.
__global__ __launch_bounds__(256) void dropout_fwd_kernel(const float* x, float* y, uint8_t* mask, long total,
                                                          float p, uint64_t seed, const long* counter) {
    if (counter) seed ^= (uint64_t)counter[0] * 0x9E3779B97F4A7C15ull;
    const float inv = 1.f / (1.f - p);
    const long nquad = (total + 3) / 4;
    for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < nquad; q += (long)gridDim.x * 256) {
        uint32_t r[4];
        philox4(seed, (uint64_t)q, r);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const long i = q * 4 + j;
            if (i < total) {
                const float u = (float)(r[j] >> 8) * (1.f / 16777216.f);
                const uint8_t keep = u >= p;
                mask[i] = keep;
                y[i] = keep ? x[i] * inv : 0.f;
            }
        }
    }
}
__global__ void counter_inc_kernel(long* counter) { counter[0] += 1; }
__global__ __launch_bounds__(256) void dropout_bwd_kernel(const float* dy, const uint8_t* mask, float* dx, long total,
                                                          float p) {
    const float inv = 1.f / (1.f - p);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256)
        dx[i] = mask[i] ? dy[i] * inv : 0.f;
}

// torch.optim.SGD semantics as used at /root/reference/ssn_train.py:141-144,252 with the
// per-group lr_mult / decay_mult of ssn_models.py:240-251, over one flat parameter segment:
//   g = grad * grad_scale + wd * w;  buf = momentum * buf + g (buf = g on the first step);  w -= lr * buf
__global__ __launch_bounds__(256) void sgd_kernel(float* w, const float* grad, float* buf, long n, float lr,
                                                  float momentum, float wd, float grad_scale, int first_step, const int* skip) {
    if (skip && *skip) return;      // the step's gradients are not trustworthy (range guard of the planes path): leave w, buf alone
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        float g = grad[i] * grad_scale + wd * w[i];
        float b = first_step ? g : momentum * buf[i] + g;
        buf[i] = b;
        w[i] = w[i] - lr * b;
    }
}

// ---- multi-tensor variants: one launch covers up to MT_MAX small tensors (the per-layer launches of
// bn_fold / sgd are ~4 us each, 200+ per step).  The tensor table travels by value in the kernel arguments, so
// nothing has to be staged in device memory and the launch is hipGraph-capturable.
constexpr int MT_MAX = 48;
struct SgdTable {
    float* w[MT_MAX];
    const float* g[MT_MAX];
    float* buf[MT_MAX];
    int blk0[MT_MAX + 1];   // first block of tensor t (prefix sums; CHUNK elements per block)
    long n[MT_MAX];
    float lr[MT_MAX], wd[MT_MAX];
    int count;
};
constexpr int MT_CHUNK = 4096;
__global__ __launch_bounds__(256) void sgd_multi_kernel(SgdTable t, float momentum, float grad_scale, int first_step,
                                                        const int* skip) {
    if (skip && *skip) return;      // (see sgd_kernel)
    int ti = 0;
    while (ti + 1 < t.count && (int)blockIdx.x >= t.blk0[ti + 1]) ++ti;   // block-uniform linear search
    const long base = (long)((int)blockIdx.x - t.blk0[ti]) * MT_CHUNK;
    float* w = t.w[ti];
    const float* g = t.g[ti];
    float* buf = t.buf[ti];
    const float lr = t.lr[ti], wd = t.wd[ti];
    long end = base + MT_CHUNK;
    if (end > t.n[ti]) end = t.n[ti];
    for (long i = base + threadIdx.x; i < end; i += 256) {
        const float gg = g[i] * grad_scale + wd * w[i];
        const float b = first_step ? gg : momentum * buf[i] + gg;
        buf[i] = b;
        w[i] = w[i] - lr * b;
    }
}

struct FoldTable {
    const float* bias[MT_MAX];
    const float* gamma[MT_MAX];
    const float* beta[MT_MAX];
    const float* mean[MT_MAX];
    const float* var[MT_MAX];
    float* scale[MT_MAX];
    float* shift[MT_MAX];
    int c[MT_MAX];
    float eps[MT_MAX];
    int count;
};
// one block per layer
__global__ __launch_bounds__(256) void bn_fold_multi_kernel(FoldTable t) {
    const int ti = blockIdx.x;
    for (int c = threadIdx.x; c < t.c[ti]; c += 256) {
        const float s = t.gamma[ti][c] / sqrtf(t.var[ti][c] + t.eps[ti]);
        t.scale[ti][c] = s;
        t.shift[ti][c] = ((t.bias[ti] ? t.bias[ti][c] : 0.f) - t.mean[ti][c]) * s + t.beta[ti][c];
    }
}

// sum of squares of a flat segment -> one partial per block (deterministic second stage on host side
// of the ABI: ssn_sumsq reduces partials in a single-block kernel).
__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* x, long n, float* partial) {
    __shared__ float red[4];
    float s = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) s += x[i] * x[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ __launch_bounds__(64) void sumsq_final_kernel(const float* partial, int nb, float* out, int accumulate) {
    float s = 0.f;
    for (int i = threadIdx.x; i < nb; i += 64) s += partial[i];
    s = wave_sum(s);
    if (threadIdx.x == 0) out[0] = accumulate ? out[0] + s : s;
}
__global__ __launch_bounds__(256) void scale_kernel(float* x, long n, const float* coef_dev, float coef) {
    const float c = coef_dev ? coef_dev[0] : coef;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) x[i] *= c;
}

// ---- space-to-depth view of the stem (7x7 / stride 2 / pad 3 on c input channels = 4x4 / stride 1 on 4c channels) ----
// xs[n][(c*2 + a)*2 + b][h'][w'] = x[n][c][2h' + a][2w' + b]; amax emitted (xs is an operand of the split kernels)
__global__ __launch_bounds__(256) void s2d_kernel(const float* x, float* xs, long total, int C, int H2, int W2,
                                                   FastDiv div_chw, FastDiv div_hw, FastDiv div_w, float* amax) {
    // one thread = one horizontal pixel pair of the input (an 8-byte load: W is even, rows start 8-byte aligned when the
    // tensor does) -> one element in each of the two b-planes; idx enumerates (n, c, h, w2)
    float vmax = 0.f;
    const int W = 2 * W2, H = 2 * H2;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        uint32_t n, rem, c, hw, h, w2;
        fd_divmod((uint32_t)idx, div_chw, n, rem);      // div_chw = C * H * W2
        fd_divmod(rem, div_hw, c, hw);                  // div_hw = H * W2
        fd_divmod(hw, div_w, h, w2);                    // div_w = W2
        const float* src = x + (((long)n * C + c) * H + h) * W + 2 * w2;
        const float v0 = src[0], v1 = src[1];
        const long plane = (long)H2 * W2;
        float* dst = xs + ((long)n * 4 * C + (c * 2 + (h & 1)) * 2) * plane + (long)(h >> 1) * W2 + w2;
        dst[0] = v0;
        dst[plane] = v1;
        vmax = fmaxf(vmax, fmaxf(fabsf(v0), fabsf(v1)));
    }
    amax_emit(amax, vmax);
}
// w2[co][(c*2+a)*2+b][r'][s'] = w[co][c][2r'+a-1][2s'+b-1] (0 outside the 7x7 window), and its transpose for the gradient:
// dw[co][c][r][s] = dw2[co][(c*2 + a)*2 + b][r'][s'] with a = (r + 1) & 1, r' = (r + 1) >> 1 (likewise b, s')
__global__ __launch_bounds__(256) void s2d_weight_kernel(const float* w, float* w2, int total, int C, int k) {
    const int k2 = (k + 1) / 2;
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
        int t = idx;
        const int s2 = t % k2; t /= k2;
        const int r2 = t % k2; t /= k2;
        const int c4 = t % (4 * C); t /= 4 * C;
        const int co = t, c = c4 >> 2, a = (c4 >> 1) & 1, b = c4 & 1;
        const int r = 2 * r2 + a - 1, s = 2 * s2 + b - 1;
        w2[idx] = (r >= 0 && r < k && s >= 0 && s < k) ? w[((co * C + c) * k + r) * k + s] : 0.f;
    }
}
__global__ __launch_bounds__(256) void s2d_weight_bwd_kernel(const float* dw2, float* dw, int total, int C, int k) {
    const int k2 = (k + 1) / 2;
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
        int t = idx;
        const int s = t % k; t /= k;
        const int r = t % k; t /= k;
        const int c = t % C; t /= C;
        const int co = t, a = (r + 1) & 1, r2 = (r + 1) >> 1, b = (s + 1) & 1, s2 = (s + 1) >> 1;
        dw[idx] = dw2[((co * 4 * C + (c * 2 + a) * 2 + b) * k2 + r2) * k2 + s2];
    }
}

// dst += src (the flat conv-gradient buffers of the sub-batches of one backward, see bninception.py: chunked execution)
__global__ __launch_bounds__(256) void add_inplace_kernel(float* dst, const float* src, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) dst[i] += src[i];
}

// out[n][c][h][w] = (h < Ho && w < Wo) ? g[n][c][h][w] : 0: the output gradient of an UNPADDED stride-1 convolution laid
// into planes of its input's size, so that its weight gradient is a same-grid problem (see ssn_embed_planes)
__global__ __launch_bounds__(256) void embed_planes_kernel(const float* g, float* out, int C, int Ho, int Wo, long g_img_stride,
                                                           int H, int W, long out_img_stride, long total, FastDiv div_chw,
                                                           FastDiv div_hw, FastDiv div_w) {
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        uint32_t n, rem, c, hw, h, w;
        fd_divmod((uint32_t)idx, div_chw, n, rem);
        fd_divmod(rem, div_hw, c, hw);
        fd_divmod(hw, div_w, h, w);
        float v = 0.f;
        if ((int)h < Ho && (int)w < Wo) v = g[(long)n * g_img_stride + ((long)c * Ho + h) * Wo + w];
        out[(long)n * out_img_stride + rem] = v;
    }
}

inline unsigned grid_for(long total, int cap = 4096) {
    long b = (total + 255) / 256;
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (unsigned)b;
}

}  // namespace

extern "C" int ssn_bn_fold(const float* conv_bias, const float* gamma, const float* beta, const float* mean,
                           const float* var, float eps, float* scale, float* shift, int C, hipStream_t stream) {
    SSN_CHECK_ARG(gamma && beta && mean && var && scale && shift && C > 0, "bn_fold: bad arguments");
    hipLaunchKernelGGL(bn_fold_kernel, dim3((C + 255) / 256), dim3(256), 0, stream, conv_bias, gamma, beta, mean, var,
                       eps, scale, shift, C);
    SSN_CHECK_LAUNCH("bn_fold");
    return SSN_OK;
}

extern "C" int ssn_relu_bn_bwd(float* dy, const float* y, const float* scale, int N, int C, int HW,
                               long dy_img_stride, long y_img_stride, float* dy_amax, hipStream_t stream) {
    SSN_CHECK_ARG(dy && y && scale, "relu_bn_bwd: null pointer");
    const long total = (long)N * C * HW;
    SSN_CHECK_ARG(total < (1l << 31), "relu_bn_bwd: tensor too large");
    hipLaunchKernelGGL(relu_bn_bwd_kernel, dim3(grid_for(total, 8192)), dim3(256), 0, stream, dy, y, scale, C, HW,
                       dy_img_stride, y_img_stride, total, make_fastdiv((uint32_t)(C * HW)),
                       make_fastdiv((uint32_t)HW), dy_amax);
    SSN_CHECK_LAUNCH("relu_bn_bwd");
    return SSN_OK;
}

// ---- [r6] parameter checksum: does a cached derivative of the parameters (packed weights, folded BatchNorm) still belong to them?
// A write through `p.data` / raw pointers is invisible to torch's version counters; the bits are not.  sum over all words of
// word * (2 * position + 1) * tensor odd-multiplier, mod 2^64 (order-free: unsigned atomics), ~45 MB read per BN-Inception check.
struct SsnChecksumEntry {
    const uint32_t* ptr;
    long n;      // 32-bit words
};
namespace {
__global__ __launch_bounds__(256) void param_checksum_kernel(const SsnChecksumEntry* table, unsigned long long* slot) {
    const SsnChecksumEntry e = table[blockIdx.y];
    unsigned long long acc = 0ull;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < e.n; i += (long)gridDim.x * 256)
        acc += (unsigned long long)e.ptr[i] * (unsigned long long)(2 * i + 1);
    acc *= 0x9E3779B97F4A7C15ull * (unsigned long long)(2 * blockIdx.y + 1);
    // wave reduction of the 64-bit partial sums, one atomic per wave
    uint32_t lo = (uint32_t)acc, hi = (uint32_t)(acc >> 32);
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long o = ((unsigned long long)(uint32_t)__shfl_down((int)hi, off) << 32) | (uint32_t)__shfl_down((int)lo, off);
        acc += o;
        lo = (uint32_t)acc;
        hi = (uint32_t)(acc >> 32);
    }
    if ((threadIdx.x & 63) == 0) atomicAdd(slot, acc);
}
// flag |= bit when the fresh sum differs from the expected one; the fresh slot is cleared for the next check
__global__ void checksum_compare_kernel(unsigned long long* fresh, const unsigned long long* expected, int* flag, int bit) {
    if (threadIdx.x == 0) {
        if (*fresh != *expected) atomicOr(flag, bit);
        *fresh = 0ull;
    }
}
}  // namespace

// table: device array of n_entries {pointer, 32-bit word count}; slot: device uint64, must be 0 on entry; adds the checksum of all
// entries.  expected == NULL: just accumulate (recording the reference value).  Otherwise: flag |= bit if slot != *expected, slot <- 0.
extern "C" int ssn_param_checksum(const void* table, int n_entries, unsigned long long* slot, const unsigned long long* expected,
                                  int* flag, int bit, hipStream_t stream) {
    SSN_CHECK_ARG(table && slot && n_entries > 0 && (!expected || flag), "param_checksum: bad arguments");
    hipLaunchKernelGGL(param_checksum_kernel, dim3(16, n_entries), dim3(256), 0, stream, (const SsnChecksumEntry*)table, slot);
    SSN_CHECK_LAUNCH("param_checksum");
    if (expected) {
        hipLaunchKernelGGL(checksum_compare_kernel, dim3(1), dim3(64), 0, stream, slot, expected, flag, bit);
        SSN_CHECK_LAUNCH("checksum_compare");
    }
    return SSN_OK;
}

extern "C" int ssn_tensor_amax(const float* x, long n, float* slot, hipStream_t stream) {
    SSN_CHECK_ARG(x && slot && n >= 0, "tensor_amax: bad arguments");
    if (n == 0) return SSN_OK;
    hipLaunchKernelGGL(tensor_amax_kernel, dim3(grid_for((n + 3) / 4, 2048)), dim3(256), 0, stream, x, n, slot);
    SSN_CHECK_LAUNCH("tensor_amax");
    return SSN_OK;
}

// shares per channel of ssn_channel_sum (workspace: C * ssn_channel_sum_shares(N) floats)
extern "C" int ssn_channel_sum_shares(int N) { return N < 32 ? (N < 1 ? 1 : N) : 32; }
extern "C" int ssn_channel_sum(const float* g, float* out, int N, int C, int HW, long img_stride, void* workspace,
                               size_t ws_bytes, hipStream_t stream) {
    SSN_CHECK_ARG(g && out && workspace && N >= 1 && C >= 1 && HW >= 1, "channel_sum: bad arguments");
    const int S = ssn_channel_sum_shares(N);
    if (ws_bytes < (size_t)C * S * sizeof(float)) {
        ssn_set_error("channel_sum: workspace %zu < %zu bytes", ws_bytes, (size_t)C * S * sizeof(float));
        return SSN_ERR_WORKSPACE;
    }
    hipLaunchKernelGGL(channel_sum_part_kernel, dim3((unsigned)C, (unsigned)S), dim3(256), 0, stream, g,
                       (float*)workspace, N, HW, img_stride, S);
    hipLaunchKernelGGL(channel_sum_final_kernel, dim3((unsigned)((C + 255) / 256)), dim3(256), 0, stream,
                       (const float*)workspace, out, C, S);
    SSN_CHECK_LAUNCH("channel_sum");
    return SSN_OK;
}

extern "C" int ssn_dropout_fwd(const float* x, float* y, unsigned char* mask, long total, float p,
                               unsigned long long seed, long* counter, hipStream_t stream) {
    SSN_CHECK_ARG(x && y && mask && p >= 0.f && p < 1.f, "dropout_fwd: bad arguments");
    hipLaunchKernelGGL(dropout_fwd_kernel, dim3(grid_for((total + 3) / 4)), dim3(256), 0, stream, x, y,
                       (uint8_t*)mask, total, p, (uint64_t)seed, (const long*)counter);
    if (counter) hipLaunchKernelGGL(counter_inc_kernel, dim3(1), dim3(1), 0, stream, counter);
    SSN_CHECK_LAUNCH("dropout_fwd");
    return SSN_OK;
}
extern "C" int ssn_dropout_bwd(const float* dy, const unsigned char* mask, float* dx, long total, float p,
                               hipStream_t stream) {
    SSN_CHECK_ARG(dy && dx && mask, "dropout_bwd: null pointer");
    hipLaunchKernelGGL(dropout_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, stream, dy, (const uint8_t*)mask, dx,
                       total, p);
    SSN_CHECK_LAUNCH("dropout_bwd");
    return SSN_OK;
}

extern "C" int ssn_sgd_step(float* w, const float* grad, float* momentum_buf, long n, float lr, float momentum,
                            float weight_decay, float grad_scale, int first_step, const int* skip_flag, hipStream_t stream) {
    SSN_CHECK_ARG(w && grad && momentum_buf, "sgd_step: null pointer");
    if (n == 0) return SSN_OK;
    hipLaunchKernelGGL(sgd_kernel, dim3(grid_for(n, 2048)), dim3(256), 0, stream, w, grad, momentum_buf, n, lr,
                       momentum, weight_decay, grad_scale, first_step, skip_flag);
    SSN_CHECK_LAUNCH("sgd_step");
    return SSN_OK;
}

// Multi-tensor SGD: the same update as ssn_sgd_step for `count` tensors with per-tensor lr / weight decay
// (host arrays of device pointers); ceil(count / 48) launches.  skip_flag (device int, may be null): when it reads non-zero at
// launch time the update is skipped -- the range guard of the planes path (ssn_pl_range_check) flagged the step's gradients, and
// leaving weights and momentum untouched is what makes the step retryable from a graph replay.
extern "C" int ssn_sgd_step_multi(int count, float* const* w, const float* const* grad, float* const* momentum_buf,
                                  const long* n, const float* lr, const float* weight_decay, float momentum,
                                  float grad_scale, int first_step, const int* skip_flag, hipStream_t stream) {
    SSN_CHECK_ARG(count >= 0 && (count == 0 || (w && grad && momentum_buf && n && lr && weight_decay)),
                  "sgd_step_multi: bad arguments");
    for (int base = 0; base < count; base += MT_MAX) {
        SgdTable t;
        t.count = count - base < MT_MAX ? count - base : MT_MAX;
        int blocks = 0;
        for (int i = 0; i < t.count; ++i) {
            t.w[i] = w[base + i];
            t.g[i] = grad[base + i];
            t.buf[i] = momentum_buf[base + i];
            t.n[i] = n[base + i];
            t.lr[i] = lr[base + i];
            t.wd[i] = weight_decay[base + i];
            t.blk0[i] = blocks;
            blocks += (int)((n[base + i] + MT_CHUNK - 1) / MT_CHUNK);
        }
        t.blk0[t.count] = blocks;
        if (blocks == 0) continue;
        hipLaunchKernelGGL(sgd_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, t, momentum, grad_scale,
                           first_step, skip_flag);
    }
    SSN_CHECK_LAUNCH("sgd_step_multi");
    return SSN_OK;
}

// Multi-tensor ssn_bn_fold: `count` layers in ceil(count / 48) launches (host arrays of device pointers).
extern "C" int ssn_bn_fold_multi(int count, const float* const* conv_bias, const float* const* gamma,
                                 const float* const* beta, const float* const* mean, const float* const* var,
                                 const float* eps, float* const* scale, float* const* shift, const int* channels,
                                 hipStream_t stream) {
    SSN_CHECK_ARG(count >= 0 && (count == 0 || (gamma && beta && mean && var && eps && scale && shift && channels)),
                  "bn_fold_multi: bad arguments");
    for (int base = 0; base < count; base += MT_MAX) {
        FoldTable t;
        t.count = count - base < MT_MAX ? count - base : MT_MAX;
        for (int i = 0; i < t.count; ++i) {
            t.bias[i] = conv_bias ? conv_bias[base + i] : nullptr;
            t.gamma[i] = gamma[base + i];
            t.beta[i] = beta[base + i];
            t.mean[i] = mean[base + i];
            t.var[i] = var[base + i];
            t.scale[i] = scale[base + i];
            t.shift[i] = shift[base + i];
            t.c[i] = channels[base + i];
            t.eps[i] = eps[base + i];
        }
        hipLaunchKernelGGL(bn_fold_multi_kernel, dim3((unsigned)t.count), dim3(256), 0, stream, t);
    }
    SSN_CHECK_LAUNCH("bn_fold_multi");
    return SSN_OK;
}

// out[0] (+)= sum(x^2); workspace: >= 1024 floats
extern "C" int ssn_sumsq(const float* x, long n, float* out, int accumulate, float* workspace, hipStream_t stream) {
    SSN_CHECK_ARG(x && out && workspace, "sumsq: null pointer");
    const unsigned nb = grid_for(n, 1024);
    hipLaunchKernelGGL(sumsq_partial_kernel, dim3(nb), dim3(256), 0, stream, x, n, workspace);
    hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(64), 0, stream, (const float*)workspace, (int)nb, out,
                       accumulate);
    SSN_CHECK_LAUNCH("sumsq");
    return SSN_OK;
}

// x *= coef  (coef read from device memory when coef_dev != nullptr, e.g. a clip coefficient)
// Space-to-depth of the stem input: x [N][C][H][W] (H, W even) -> xs [N][4C][H/2][W/2]; with the weights re-indexed by
// ssn_s2d_weights (k odd, stride 2, pad (k-1)/2 -> (k+1)/2 taps, stride 1, 2 padding pixels in front and 1 behind for k = 7)
// the stem convolution of model_zoo.BNInception (conv1_7x7_s2, reached from /root/reference/ssn_models.py:266) runs on the
// split kernels: y = ssn_conv_x6_fwd_rect(xs, pack(w2), 4x4 taps), dw2 = ssn_conv_wgrad_x6(ksize 4), dw = ssn_s2d_weights_bwd(dw2).
extern "C" int ssn_space_to_depth2(const float* x, float* xs, int N, int C, int H, int W, float* xs_amax, hipStream_t stream) {
    SSN_CHECK_ARG(x && xs && N >= 1 && C >= 1 && H % 2 == 0 && W % 2 == 0, "space_to_depth2: bad arguments");
    const int H2 = H / 2, W2 = W / 2;
    const long total = (long)N * C * H * W2;       // horizontal pixel pairs
    SSN_CHECK_ARG(2 * total < (1l << 31), "space_to_depth2: tensor too large");
    hipLaunchKernelGGL(s2d_kernel, dim3(grid_for(total, 16384)), dim3(256), 0, stream, x, xs, total, C, H2, W2,
                       make_fastdiv((uint32_t)(C * H * W2)), make_fastdiv((uint32_t)(H * W2)), make_fastdiv((uint32_t)W2),
                       xs_amax);
    SSN_CHECK_LAUNCH("space_to_depth2");
    return SSN_OK;
}
extern "C" int ssn_s2d_weights(const float* w, float* w2, int Cout, int C, int k, hipStream_t stream) {
    SSN_CHECK_ARG(w && w2 && Cout >= 1 && C >= 1 && k % 2 == 1, "s2d_weights: bad arguments");
    const int k2 = (k + 1) / 2, total = Cout * 4 * C * k2 * k2;
    hipLaunchKernelGGL(s2d_weight_kernel, dim3(grid_for(total)), dim3(256), 0, stream, w, w2, total, C, k);
    SSN_CHECK_LAUNCH("s2d_weights");
    return SSN_OK;
}
extern "C" int ssn_s2d_weights_bwd(const float* dw2, float* dw, int Cout, int C, int k, hipStream_t stream) {
    SSN_CHECK_ARG(dw2 && dw && Cout >= 1 && C >= 1 && k % 2 == 1, "s2d_weights_bwd: bad arguments");
    const int total = Cout * C * k * k;
    hipLaunchKernelGGL(s2d_weight_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, stream, dw2, dw, total, C, k);
    SSN_CHECK_LAUNCH("s2d_weights_bwd");
    return SSN_OK;
}

extern "C" int ssn_add_inplace(float* dst, const float* src, long n, hipStream_t stream) {
    SSN_CHECK_ARG(dst && src && n >= 0, "add_inplace: bad arguments");
    if (n == 0) return SSN_OK;
    hipLaunchKernelGGL(add_inplace_kernel, dim3(grid_for(n)), dim3(256), 0, stream, dst, src, n);
    SSN_CHECK_LAUNCH("add_inplace");
    return SSN_OK;
}

extern "C" int ssn_embed_planes(const float* g, float* out, int N, int C, int Ho, int Wo, long g_img_stride, int H, int W,
                                long out_img_stride, hipStream_t stream) {
    SSN_CHECK_ARG(g && out && N >= 1 && C >= 1 && Ho >= 1 && Wo >= 1 && H >= Ho && W >= Wo, "embed_planes: bad arguments");
    const long total = (long)N * C * H * W;
    SSN_CHECK_ARG(total < (1l << 32), "embed_planes: tensor too large");
    hipLaunchKernelGGL(embed_planes_kernel, dim3(grid_for(total, 8192)), dim3(256), 0, stream, g, out, C, Ho, Wo, g_img_stride, H, W,
                       out_img_stride, total, make_fastdiv((uint32_t)(C * H * W)), make_fastdiv((uint32_t)(H * W)),
                       make_fastdiv((uint32_t)W));
    SSN_CHECK_LAUNCH("embed_planes");
    return SSN_OK;
}

extern "C" int ssn_scale(float* x, long n, const float* coef_dev, float coef, hipStream_t stream) {
    SSN_CHECK_ARG(x, "scale: null pointer");
    if (n == 0) return SSN_OK;
    hipLaunchKernelGGL(scale_kernel, dim3(grid_for(n, 2048)), dim3(256), 0, stream, x, n, coef_dev, coef);
    SSN_CHECK_LAUNCH("scale");
    return SSN_OK;
}
