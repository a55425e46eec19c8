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

// combined[p][c] = softmax(act[p][off : off + n_sm])[c + 1 - off] * exp(comp[p][c]),  off = 0 (include_bg 1: softmax over all
// C + 1 activity scores, then drop the background column) or 1 (include_bg 0: softmax over the C class scores only);
// include_bg 2: NO softmax, the raw class scores act[p][c + 1] (the `--cls_scores` branch without --softmax_before_filter, :135)
__global__ __launch_bounds__(256) void det_scores_kernel(const float* act, const float* comp, float* combined, int P, int C,
                                                         int include_bg) {
    const int p = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (p >= P) return;
    const float* a = act + (long)p * (C + 1);
    if (include_bg == 2) {
        for (int c = lane; c < C; c += 64) combined[(long)p * C + c] = a[c + 1] * expf(comp[(long)p * C + c]);
        return;
    }
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

// Total order on fused scores, the one numpy's sorts use: ascending floats, every NaN after +inf.  The key is the
// usual monotone map of the bit pattern (negative: all bits flipped, non-negative: sign bit set) with all NaNs
// collapsed onto the largest key.  softmax * exp overflows to inf and inf * 0 to NaN on real (diverged) scores; sorting
// on raw floats is then not a total order and a bitonic network may move its padding anywhere.
__device__ __forceinline__ uint32_t det_key(float v) {
    const uint32_t b = __builtin_bit_cast(uint32_t, v);
    if ((b & 0x7fffffffu) > 0x7f800000u) return 0xffffffffu;
    return (b >> 31) ? ~b : (b | 0x80000000u);
}

// Exactly-k selection (np.argsort(combined.ravel())[-top_k:], eval_detection_results.py:113-114) in two numbers:
// out[0] = key T of the k-th largest element (radix select on det_key, one workgroup); elements with a larger key are
// all kept; of the elements whose key equals T, those with flat index >= out[1..2] (a 64-bit cut) are kept -- the
// `need` highest flat indices, which is what a stable ascending argsort leaves in its last k positions.
__global__ __launch_bounds__(1024) void det_topk_threshold_kernel(const float* v, long n, long k, uint32_t* out) {
    __shared__ int hist[256];
    __shared__ uint32_t s_prefix, s_mask;
    __shared__ long s_k, s_ties, s_cut;
    __shared__ int s_cnt;
    __shared__ unsigned char flags[1024];
    if (threadIdx.x == 0) {
        s_prefix = 0;
        s_mask = 0;
        s_k = k;
        s_ties = 0;
        s_cut = 0;
    }
    __syncthreads();
    for (int shift = 24; shift >= 0; shift -= 8) {
        for (int i = threadIdx.x; i < 256; i += 1024) hist[i] = 0;
        __syncthreads();
        const uint32_t prefix = s_prefix, mask = s_mask;
        for (long i = threadIdx.x; i < n; i += 1024) {
            const uint32_t b = det_key(v[i]);
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
            s_ties = hist[d];
            s_prefix = prefix | ((uint32_t)d << shift);
            s_mask = mask | (255u << shift);
        }
        __syncthreads();
    }
    const uint32_t T = s_prefix;
    const long need = s_k;
    if (s_ties > need) {
        // more elements tie at T than are still wanted: walk the array from its end in chunks of 1024 until the chunk
        // that holds the need-th tie, then find it
        long hi = n, cum = 0;
        while (hi > 0) {
            const long base = hi > 1024 ? hi - 1024 : 0;
            if (threadIdx.x == 0) s_cnt = 0;
            __syncthreads();
            const long i = base + threadIdx.x;
            const int tie = i < hi && det_key(v[i]) == T;
            flags[threadIdx.x] = (unsigned char)tie;
            if (tie) atomicAdd(&s_cnt, 1);
            __syncthreads();
            const int cnt = s_cnt;
            if (cum + cnt >= need) {
                if (threadIdx.x == 0) {
                    long c = cum;
                    for (long j = hi - base - 1; j >= 0; --j)
                        if (flags[j] && ++c == need) {
                            s_cut = base + j;
                            break;
                        }
                }
                break;
            }
            cum += cnt;
            hi = base;
            __syncthreads();
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        out[0] = T;
        out[1] = (uint32_t)((unsigned long)s_cut & 0xffffffffu);
        out[2] = (uint32_t)((unsigned long)s_cut >> 32);
    }
}

struct DetArgs {
    const double* rel_prop;   // [P][2] start, end (normalised)
    const float* combined;    // [P][C]
    const float* reg;         // [P][C][2] or nullptr
    double* dets;             // [C][max_det][5]: start, end, score, loc, dur
    int* counts;              // [C]
    int P, C, max_det;
    const uint32_t* sel;      // {threshold key, flat-index cut lo, hi} from the selection kernel, or nullptr: keep all
    double nms_thresh;
    int regress;
    // storage of the candidates of one class when they do not fit in LDS (P > DET_MAXN): [C][cap] each
    uint32_t* g_key;
    int* g_idx;
    unsigned char* g_supp;
    int cap;
};

__device__ __forceinline__ double det_clip01(double x) { return x < 0.0 ? 0.0 : (x > 1.0 ? 1.0 : x); }   // np.clip: NaN stays

// One workgroup per class: collect the class's candidates, sort them (descending score; equal scores: higher proposal
// index first, the order `scores.argsort()[::-1]` gives with a stable sort, ops/utils.py:68), suppress greedily.
// IN_LDS: the candidate arrays live in LDS (up to DET_MAXN per class); otherwise in the caller's workspace.
template <bool IN_LDS>
__global__ __launch_bounds__(256) void det_nms_kernel(DetArgs a) {
    constexpr int LN = IN_LDS ? DET_MAXN : 1;
    __shared__ uint32_t l_key[LN];
    __shared__ int l_idx[LN];
    __shared__ double l_t1[LN], l_t2[LN];
    __shared__ unsigned char l_supp[LN];
    __shared__ int s_n, s_keep;
    const int c = blockIdx.x, tid = threadIdx.x;
    uint32_t* key = IN_LDS ? l_key : a.g_key + (long)c * a.cap;
    int* idx = IN_LDS ? l_idx : a.g_idx + (long)c * a.cap;
    unsigned char* supp = IN_LDS ? l_supp : a.g_supp + (long)c * a.cap;
    const int cap = IN_LDS ? DET_MAXN : a.cap;
    uint32_t thr = 0;
    long cut = 0;
    if (a.sel) {
        thr = a.sel[0];
        cut = (long)(((unsigned long)a.sel[2] << 32) | a.sel[1]);
    }
    if (tid == 0) {
        s_n = 0;
        s_keep = 0;
    }
    __syncthreads();
    // candidates of this class, in any order (the sort's order is total)
    for (int p = tid; p < a.P; p += 256) {
        const long flat = (long)p * a.C + c;
        const uint32_t kk = det_key(a.combined[flat]);
        if (kk > thr || (kk == thr && flat >= cut)) {
            const int slot = atomicAdd(&s_n, 1);
            if (slot < cap) {
                key[slot] = kk;
                idx[slot] = p;
            }
        }
    }
    __syncthreads();
    int n = s_n;
    if (n > cap) n = cap;             // cannot happen: cap >= P (host-checked)
    int n2 = 1;
    while (n2 < n) n2 <<= 1;
    for (int i = n + tid; i < n2; i += 256) {
        key[i] = 0;                   // below every real key (det_key never returns 0 for a non-NaN, NaNs map to the top)
        idx[i] = -1;                  // and behind every real entry on a key tie
    }
    __syncthreads();
    for (int k = 2; k <= n2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < n2; i += 256) {
                const int l = i ^ j;
                if (l > i) {
                    const bool desc = (i & k) == 0;
                    const uint32_t ki = key[i], kl = key[l];
                    const int ii = idx[i], il = idx[l];
                    const bool before = ki > kl || (ki == kl && ii > il);   // i belongs first
                    if (desc != before) {
                        key[i] = kl;
                        key[l] = ki;
                        idx[i] = il;
                        idx[l] = ii;
                    }
                }
            }
            __syncthreads();
        }
    for (int i = tid; i < n; i += 256) {
        int p = idx[i];
        if (p < 0 || p >= a.P) p = 0;          // defensive: never index outside the proposal table
        if (IN_LDS) {
            l_t1[i] = a.rel_prop[2 * (long)p];
            l_t2[i] = a.rel_prop[2 * (long)p + 1];
        }
        idx[i] = p;
        supp[i] = 0;
    }
    __syncthreads();
    // greedy NMS in score order (ops/utils.py:71-80): box i survives if no surviving higher-scored box overlaps it by
    // more than the threshold; the intersection is NOT clamped at zero there, so neither is it here.  A barrier is only
    // needed behind a box that survives (suppressed ones write nothing).
    for (int i = 0; i < n; ++i) {
        if (supp[i]) continue;       // uniform: supp[i] was settled before the last barrier
        const int pi = idx[i];
        const double a1 = IN_LDS ? l_t1[i] : a.rel_prop[2 * (long)pi];
        const double a2 = IN_LDS ? l_t2[i] : a.rel_prop[2 * (long)pi + 1];
        const double da = a2 - a1;
        for (int j = i + 1 + tid; j < n; j += 256) {
            const double b1 = IN_LDS ? l_t1[j] : a.rel_prop[2 * (long)idx[j]];
            const double b2 = IN_LDS ? l_t2[j] : a.rel_prop[2 * (long)idx[j] + 1];
            const double tt1 = a1 > b1 ? a1 : b1;
            const double tt2 = a2 < b2 ? a2 : b2;
            const double inter = tt2 - tt1;
            const double iou = inter / (da + (b2 - b1) - inter);
            if (!(iou <= a.nms_thresh)) supp[j] = 1;
        }
        if (tid == 0) {
            const int o = s_keep++;
            if (o < a.max_det) {
                double* d = a.dets + ((long)c * a.max_det + o) * 5;
                const float loc = a.reg ? a.reg[((long)pi * a.C + c) * 2] : 0.f;
                const float dur = a.reg ? a.reg[((long)pi * a.C + c) * 2 + 1] : 0.f;
                double s0 = a1, s1 = a2;
                if (a.regress) {         // eval_detection_results.py:167-178
                    const double center = (a1 + a2) / 2, duration = a2 - a1;
                    const double nc = center + duration * (double)loc;
                    const double nd = duration * exp((double)dur);
                    s0 = det_clip01(nc - nd / 2);
                    s1 = det_clip01(nc + nd / 2);
                }
                d[0] = s0;
                d[1] = s1;
                d[2] = (double)a.combined[(long)pi * a.C + c];
                d[3] = (double)loc;
                d[4] = (double)dur;
            }
        }
        __syncthreads();
    }
    if (tid == 0) a.counts[c] = s_keep < a.max_det ? s_keep : a.max_det;
}

inline int det_pow2(int n) {
    int n2 = 1;
    while (n2 < n) n2 <<= 1;
    return n2;
}

}  // namespace

// Device scratch ssn_detections needs for a video of P proposals and C classes: the selection words, plus -- when the
// candidates of a class do not fit in LDS (P > 2048) -- key / index / flag arrays of pow2(P) entries per class.
extern "C" size_t ssn_detections_workspace_bytes(int P, int C) {
    size_t b = 16;
    if (P > DET_MAXN) b += (size_t)C * (size_t)det_pow2(P) * 12;     // u32 key + i32 idx + u8 flag (padded to 4)
    return b;
}

// act [P][C+1], comp [P][C], reg [P][C][2] (or NULL) fp32; rel_prop [P][2] fp64.  combined [P][C] fp32 (scratch + output),
// dets [C][max_det][5] fp64, counts [C] int32, workspace: ssn_detections_workspace_bytes(P, C) bytes, 16-byte aligned.
// include_bg: softmax over all C+1 activity scores (the `top_k <= 0` branch, :98) instead of over the C class scores
// (:113); top_k <= 0 keeps every pair.  Non-finite scores are ordered as numpy orders them (NaN above +inf); nothing
// in here can index outside its inputs whatever the scores are.
extern "C" int ssn_detections(const float* act, const float* comp, const float* reg, const double* rel_prop, float* combined,
                              double* dets, int* counts, void* workspace, size_t ws_bytes, int P, int C, int max_det,
                              int top_k, int include_bg, double nms_thresh, int regress, hipStream_t stream) {
    SSN_CHECK_ARG(dets && counts, "detections: null pointer");
    SSN_CHECK_ARG(P >= 0 && C >= 1 && max_det >= 1, "detections: bad sizes");
    SSN_CHECK_ARG((long)P * C < (1L << 40), "detections: too many (proposal, class) pairs");
    SSN_CHECK_ARG(P == 0 || (act && comp && rel_prop && combined && workspace), "detections: null pointer");
    if (hipMemsetAsync(counts, 0, sizeof(int) * (size_t)C, stream) != hipSuccess) {
        ssn_set_error("detections: memset failed");
        return SSN_ERR_LAUNCH;
    }
    if (P == 0) return SSN_OK;
    if (ws_bytes < ssn_detections_workspace_bytes(P, C)) {
        ssn_set_error("detections: workspace of %zu bytes, %zu needed", ws_bytes, ssn_detections_workspace_bytes(P, C));
        return SSN_ERR_WORKSPACE;
    }
    hipLaunchKernelGGL(det_scores_kernel, dim3((unsigned)((P + 3) / 4)), dim3(256), 0, stream, act, comp, combined, P, C,
                       include_bg);
    const long n = (long)P * C;
    const bool select = top_k > 0 && (long)top_k < n;
    uint32_t* sel = (uint32_t*)workspace;
    if (select)
        hipLaunchKernelGGL(det_topk_threshold_kernel, dim3(1), dim3(1024), 0, stream, (const float*)combined, n, (long)top_k,
                           sel);
    DetArgs a;
    a.rel_prop = rel_prop;
    a.combined = combined;
    a.reg = reg;
    a.dets = dets;
    a.counts = counts;
    a.P = P;
    a.C = C;
    a.max_det = max_det;
    a.sel = select ? sel : nullptr;
    a.nms_thresh = nms_thresh;
    a.regress = regress;
    a.g_key = nullptr;
    a.g_idx = nullptr;
    a.g_supp = nullptr;
    a.cap = 0;
    if (P <= DET_MAXN) {
        hipLaunchKernelGGL(det_nms_kernel<true>, dim3((unsigned)C), dim3(256), 0, stream, a);
    } else {
        const size_t cap = (size_t)det_pow2(P);
        char* w = (char*)workspace + 16;
        a.g_key = (uint32_t*)w;
        a.g_idx = (int*)(w + (size_t)C * cap * 4);
        a.g_supp = (unsigned char*)(w + (size_t)C * cap * 8);
        a.cap = (int)cap;
        hipLaunchKernelGGL(det_nms_kernel<false>, dim3((unsigned)C), dim3(256), 0, stream, a);
    }
    SSN_CHECK_LAUNCH("detections");
    return SSN_OK;
}
