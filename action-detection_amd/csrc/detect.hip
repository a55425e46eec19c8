// Detection post-processing of one video on the GPU: the per-video arithmetic of the reference's evaluation script
//   score fusion  softmax(activity)[1:] * exp(completeness)          /root/reference/eval_detection_results.py:91-128
//   top-k over all (proposal, class) pairs                           :113-128 (np.argsort(...)[-top_k:])
//   temporal NMS per class                                           /root/reference/ops/utils.py:56-82
//   location regression                                              /root/reference/eval_detection_results.py:167-178
// which the reference runs in numpy, class by class and video by video.  Box arithmetic is fp64 as in numpy (the
// proposal spans are float64 there and promote everything they touch); the fused scores are fp32 as in numpy.
#include "ssn_common.h"

namespace {

constexpr int DET_MAXN = 2048;   // candidates of one class (sorted and suppressed inside one workgroup)

// combined[p][c] = softmax(act[p][off : off + n_sm])[c + 1 - off] * exp(comp[p][c]),  off = 0 (softmax over all C + 1
// activity scores, then drop the background column) or 1 (softmax over the C class scores only)
__global__ __launch_bounds__(256) void det_scores_kernel(const float* act, const float* comp, float* combined, int P, int C,
                                                         int include_bg) {
    const int p = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (p >= P) return;
    const float* a = act + (long)p * (C + 1);
    const int lo = include_bg ? 0 : 1;
    float mx = -INFINITY;
    for (int c = lo + lane; c <= C; c += 64) mx = fmaxf(mx, a[c]);
    mx = wave_max(mx);
    float sum = 0.f;
    for (int c = lo + lane; c <= C; c += 64) sum += expf(a[c] - mx);
    sum = wave_sum(sum);
    for (int c = lane; c < C; c += 64)
        combined[(long)p * C + c] = (expf(a[c + 1] - mx) / sum) * expf(comp[(long)p * C + c]);
}

// k-th largest of n non-negative floats by radix select on the bit pattern (one workgroup).  out[0] = bits of the
// threshold T (the k-th largest value); every element >= T is kept (ties at T are all kept).
__global__ __launch_bounds__(1024) void det_topk_threshold_kernel(const float* v, long n, long k, uint32_t* out) {
    __shared__ int hist[256];
    __shared__ uint32_t s_prefix, s_mask;
    __shared__ long s_k;
    if (threadIdx.x == 0) {
        s_prefix = 0;
        s_mask = 0;
        s_k = k;
    }
    __syncthreads();
    for (int shift = 24; shift >= 0; shift -= 8) {
        for (int i = threadIdx.x; i < 256; i += 1024) hist[i] = 0;
        __syncthreads();
        const uint32_t prefix = s_prefix, mask = s_mask;
        for (long i = threadIdx.x; i < n; i += 1024) {
            const uint32_t b = __builtin_bit_cast(uint32_t, v[i]);
            if ((b & mask) == prefix) atomicAdd(&hist[(b >> shift) & 255u], 1);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            long need = s_k;
            int d = 255;
            for (; d > 0; --d) {
                if ((long)hist[d] >= need) break;
                need -= hist[d];
            }
            s_k = need;
            s_prefix = prefix | ((uint32_t)d << shift);
            s_mask = mask | (255u << shift);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = s_prefix;
}

struct DetArgs {
    const double* rel_prop;   // [P][2] start, end (normalised)
    const float* combined;    // [P][C]
    const float* reg;         // [P][C][2] or nullptr
    double* dets;             // [C][max_det][5]: start, end, score, loc, dur
    int* counts;              // [C]
    int P, C, max_det;
    uint32_t thr_bits;        // keep candidates with score bits >= thr_bits (0: all)
    const uint32_t* thr_dev;  // ... or read the threshold from device memory
    double nms_thresh;
    int regress;
    int* error;               // set to 1 if a class has more than DET_MAXN candidates
};

__global__ __launch_bounds__(256) void det_nms_kernel(DetArgs a) {
    __shared__ float key[DET_MAXN];
    __shared__ int idx[DET_MAXN];
    __shared__ double t1[DET_MAXN], t2[DET_MAXN];
    __shared__ unsigned char supp[DET_MAXN];
    __shared__ int s_n, s_keep;
    const int c = blockIdx.x, tid = threadIdx.x;
    const uint32_t thr = a.thr_dev ? a.thr_dev[0] : a.thr_bits;
    if (tid == 0) {
        s_n = 0;
        s_keep = 0;
    }
    __syncthreads();
    // candidates of this class (any order: the sort below breaks score ties by proposal index)
    for (int p = tid; p < a.P; p += 256) {
        const float sc = a.combined[(long)p * a.C + c];
        if (__builtin_bit_cast(uint32_t, sc) >= thr) {
            const int slot = atomicAdd(&s_n, 1);
            if (slot < DET_MAXN) {
                key[slot] = sc;
                idx[slot] = p;
            }
        }
    }
    __syncthreads();
    int n = s_n;
    if (n > DET_MAXN) {
        if (tid == 0) *a.error = 1;
        n = DET_MAXN;
    }
    // bitonic sort, descending by score (ties: lower proposal index first)
    int n2 = 1;
    while (n2 < n) n2 <<= 1;
    for (int i = n + tid; i < n2; i += 256) {
        key[i] = -1.f;
        idx[i] = 0x7fffffff;
    }
    __syncthreads();
    for (int k = 2; k <= n2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < n2; i += 256) {
                const int l = i ^ j;
                if (l > i) {
                    const bool desc = (i & k) == 0;
                    const bool before = key[i] > key[l] || (key[i] == key[l] && idx[i] < idx[l]);   // i belongs first
                    if (desc != before) {
                        const float tk = key[i];
                        key[i] = key[l];
                        key[l] = tk;
                        const int ti = idx[i];
                        idx[i] = idx[l];
                        idx[l] = ti;
                    }
                }
            }
            __syncthreads();
        }
    for (int i = tid; i < n; i += 256) {
        t1[i] = a.rel_prop[2 * (long)idx[i]];
        t2[i] = a.rel_prop[2 * (long)idx[i] + 1];
        supp[i] = 0;
    }
    __syncthreads();
    // greedy NMS in score order (ops/utils.py:71-80): box i survives if no surviving higher-scored box overlaps it by
    // more than the threshold; the intersection is NOT clamped at zero there, so neither is it here
    for (int i = 0; i < n; ++i) {
        if (!supp[i]) {          // uniform: supp[i] was settled before the last barrier
            const double a1 = t1[i], a2 = t2[i], da = a2 - a1;
            for (int j = i + 1 + tid; j < n; j += 256) {
                const double tt1 = a1 > t1[j] ? a1 : t1[j];
                const double tt2 = a2 < t2[j] ? a2 : t2[j];
                const double inter = tt2 - tt1;
                const double iou = inter / (da + (t2[j] - t1[j]) - inter);
                if (!(iou <= a.nms_thresh)) supp[j] = 1;
            }
            if (tid == 0) {
                const int o = s_keep++;
                if (o < a.max_det) {
                    double* d = a.dets + ((long)c * a.max_det + o) * 5;
                    const float loc = a.reg ? a.reg[((long)idx[i] * a.C + c) * 2] : 0.f;
                    const float dur = a.reg ? a.reg[((long)idx[i] * a.C + c) * 2 + 1] : 0.f;
                    double s0 = a1, s1 = a2;
                    if (a.regress) {         // eval_detection_results.py:167-178
                        const double center = (a1 + a2) / 2, duration = a2 - a1;
                        const double nc = center + duration * (double)loc;
                        const double nd = duration * exp((double)dur);
                        s0 = fmin(fmax(nc - nd / 2, 0.0), 1.0);
                        s1 = fmin(fmax(nc + nd / 2, 0.0), 1.0);
                    }
                    d[0] = s0;
                    d[1] = s1;
                    d[2] = (double)key[i];
                    d[3] = (double)loc;
                    d[4] = (double)dur;
                }
            }
        }
        __syncthreads();
    }
    if (tid == 0) a.counts[c] = s_keep < a.max_det ? s_keep : a.max_det;
}

}  // namespace

// act [P][C+1], comp [P][C], reg [P][C][2] (or NULL) fp32; rel_prop [P][2] fp64.  combined [P][C] fp32 (scratch + output),
// thr_ws: one uint32 of device scratch, dets [C][max_det][5] fp64, counts [C] int32, error: one int32 (set to 1 when a
// class has more than 2048 candidates).  include_bg: softmax over all C+1 activity scores (the `top_k <= 0` branch,
// :98) instead of over the C class scores (:113); top_k <= 0 keeps every pair.
extern "C" int ssn_detections(const float* act, const float* comp, const float* reg, const double* rel_prop, float* combined,
                              unsigned int* thr_ws, double* dets, int* counts, int* error, int P, int C, int max_det,
                              int top_k, int include_bg, double nms_thresh, int regress, hipStream_t stream) {
    SSN_CHECK_ARG(thr_ws && dets && counts && error, "detections: null pointer");
    SSN_CHECK_ARG(P >= 0 && C >= 1 && max_det >= 1, "detections: bad sizes");
    SSN_CHECK_ARG(P == 0 || (act && comp && rel_prop && combined), "detections: null pointer");
    if (hipMemsetAsync(counts, 0, sizeof(int) * (size_t)C, stream) != hipSuccess ||
        hipMemsetAsync(error, 0, sizeof(int), stream) != hipSuccess) {
        ssn_set_error("detections: memset failed");
        return SSN_ERR_LAUNCH;
    }
    if (P == 0) return SSN_OK;
    hipLaunchKernelGGL(det_scores_kernel, dim3((unsigned)((P + 3) / 4)), dim3(256), 0, stream, act, comp, combined, P, C,
                       include_bg);
    const long n = (long)P * C;
    const bool select = top_k > 0 && (long)top_k < n;
    if (select)
        hipLaunchKernelGGL(det_topk_threshold_kernel, dim3(1), dim3(1024), 0, stream, (const float*)combined, n, (long)top_k,
                           thr_ws);
    DetArgs a;
    a.rel_prop = rel_prop;
    a.combined = combined;
    a.reg = reg;
    a.dets = dets;
    a.counts = counts;
    a.P = P;
    a.C = C;
    a.max_det = max_det;
    a.thr_bits = 0;
    a.thr_dev = select ? thr_ws : nullptr;
    a.nms_thresh = nms_thresh;
    a.regress = regress;
    a.error = error;
    hipLaunchKernelGGL(det_nms_kernel, dim3((unsigned)C), dim3(256), 0, stream, a);
    SSN_CHECK_LAUNCH("detections");
    return SSN_OK;
}
