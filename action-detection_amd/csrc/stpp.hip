// Structured Temporal Pyramid Pooling (training form) and the re-organised dense-test pooling.
//
// ssn_stpp_fwd / ssn_stpp_bwd replace StructuredTemporalPyramidPooling.forward
// (/root/reference/ops/ssn_ops.py:39-70) -- ~10 tiny slice/mean/cat kernels in the reference
// become one launch each way.  The integer part boundaries (the reference's
// torch.arange + int() truncation, ops/ssn_ops.py:53-55) are computed ONCE on the host by the
// Python mirror and passed in as a table, so segment assignment is index-exact by construction.
//
// ssn_stpp_reorg replaces STPPReorgainzed.forward (/root/reference/ops/ssn_ops.py:109-170): the
// Python loop over proposals x stages x parts becomes one workgroup per proposal.
//
// All are scans with D-contiguous lanes (one float per lane and step: a wave instruction moves 256 consecutive bytes).  At the
// sizes of the path (<= 5 MB per launch) they are latency-bound, 5-25 us each (bench.py: hbm_kernels).
#include "ssn_common.h"

#define SSN_STPP_MAX_PARTS 24

struct SsnStppTable {
    int n_parts;
    int n_seg;
    int act_lo, act_hi;                // course stage [x1, x2) for the stand-alone activity feature
    int lo[SSN_STPP_MAX_PARTS];        // first segment of the part
    int hi[SSN_STPP_MAX_PARTS];        // one past the last segment
    int norm[SSN_STPP_MAX_PARTS];      // stage multiplier the reference divides by (ops/ssn_ops.py:55)
    int col[SSN_STPP_MAX_PARTS];       // -1: unscaled, 0: * scaling[:,0], 1: * scaling[:,1]
};

namespace {

// grid (P, n_parts + 1); the last y index is the activity (course mean) output
__global__ __launch_bounds__(256) void stpp_fwd_kernel(const float* ft, const float* scaling, float* act_ft,
                                                       float* stpp_ft, int D, SsnStppTable t) {
    const int prop = blockIdx.x;
    const int part = blockIdx.y;
    const bool is_act = part == t.n_parts;
    const int lo = is_act ? t.act_lo : t.lo[part];
    const int hi = is_act ? t.act_hi : t.hi[part];
    const float len = (float)(hi - lo);
    const float norm = is_act ? 1.f : (float)t.norm[part];
    const int col = is_act ? -1 : t.col[part];
    const float s = col >= 0 ? scaling[prop * 2 + col] : 1.f;
    const float* src = ft + ((long)prop * t.n_seg) * D;
    float* dst = is_act ? act_ft + (long)prop * D : stpp_ft + ((long)prop * t.n_parts + part) * D;
    for (int d = threadIdx.x; d < D; d += 256) {
        float acc = 0.f;
        for (int seg = lo; seg < hi; ++seg) acc += src[(long)seg * D + d];
        float v = acc / len;
        if (!is_act) {
            v = v / norm;
            if (col >= 0) v = v * s;
        }
        dst[d] = v;
    }
}

// grid (P, n_seg): d_ft[p][seg][:] = sum over parts covering seg (+ activity branch)
__global__ __launch_bounds__(256) void stpp_bwd_kernel(const float* d_act, const float* d_stpp, const float* scaling,
                                                       float* d_ft, int D, SsnStppTable t) {
    const int prop = blockIdx.x;
    const int seg = blockIdx.y;
    float* dst = d_ft + ((long)prop * t.n_seg + seg) * D;
    for (int d = threadIdx.x; d < D; d += 256) {
        float g = 0.f;
        for (int part = 0; part < t.n_parts; ++part) {
            if (seg < t.lo[part] || seg >= t.hi[part]) continue;
            float v = d_stpp[((long)prop * t.n_parts + part) * D + d];
            if (t.col[part] >= 0) v = v * scaling[prop * 2 + t.col[part]];
            v = v / (float)t.norm[part];
            g += v / (float)(t.hi[part] - t.lo[part]);
        }
        if (d_act && seg >= t.act_lo && seg < t.act_hi) g += d_act[(long)prop * D + d] / (float)(t.act_hi - t.act_lo);
        dst[d] = g;
    }
}

// ---- dense-test re-organised pooling -------------------------------------------------------
// scores [T][D] ; one workgroup per proposal.  Part ranges [pl, pr) per (proposal, part) and the
// activity range are computed on the host with the reference's own float arithmetic
// (np.arange + int(), ops/ssn_ops.py:137-147) and passed as int32 tables:
//   ranges[p][part][2]  (pl, pr; pr <= pl means "skip")   act_range[p][2]
// out_act[p][:act_len]  = mean rows [a0, a1) of scores[:, 0:act_len]
// out_comp[p][:comp_len] = sum_part mean rows [pl,pr) of scores[:, comp0 + part*comp_len ...] * s(part)
// out_reg likewise.
__global__ __launch_bounds__(256) void stpp_reorg_kernel(const float* scores, int T, int D, const int* ranges,
                                                         const int* act_range, const float* scaling,
                                                         const int* part_scale_col, int n_parts, int act_len,
                                                         int comp_len, int reg_len, float* out_act, float* out_comp,
                                                         float* out_reg) {
    const int prop = blockIdx.x;
    const int comp0 = act_len;
    const int reg0 = act_len + comp_len * n_parts;
    // activity
    {
        const int a0 = act_range[prop * 2], a1 = act_range[prop * 2 + 1];
        for (int c = threadIdx.x; c < act_len; c += 256) {
            float acc = 0.f;
            for (int r = a0; r < a1; ++r) acc += scores[(long)r * D + c];
            // (a0 < 0: the range starts past the last row -- the reference's mean over an empty slice, NaN)
            out_act[(long)prop * act_len + c] = a0 < 0 ? __builtin_nanf("") : acc / (float)(a1 - a0);
        }
    }
    for (int which = 0; which < 2; ++which) {
        const int len = which == 0 ? comp_len : reg_len;
        const int base = which == 0 ? comp0 : reg0;
        float* out = which == 0 ? out_comp : out_reg;
        if (!out) continue;
        for (int c = threadIdx.x; c < len; c += 256) {
            float tot = 0.f;
            for (int part = 0; part < n_parts; ++part) {
                const int pl = ranges[((long)prop * n_parts + part) * 2];
                const int pr = ranges[((long)prop * n_parts + part) * 2 + 1];
                if (pl < 0) {            // starts past the last row: mean over an empty slice (NaN) in the reference
                    tot += __builtin_nanf("");
                    continue;
                }
                if (pr - pl < 1) continue;
                float acc = 0.f;
                for (int r = pl; r < pr; ++r) acc += scores[(long)r * D + base + part * len + c];
                float v = acc / (float)(pr - pl);
                const int col = part_scale_col[part];
                if (col >= 0) v = v * scaling[prop * 2 + col];
                tot += v;
            }
            out[(long)prop * len + c] = tot;
        }
    }
}

// ---- dense testing (ssn_test.py:78-92) ----
// x [num_crop][T][D] (crop-major rows, what GroupOverSample + view(num_crop, -1, D) give) -> y [T][D] = mean over
// crops.  test_fc is linear, so the reference's fc-then-mean equals mean-then-fc: averaging the 1024-d features
// first runs the folded FC on 10x fewer rows.
template <bool VEC>
__global__ __launch_bounds__(256) void crop_mean_kernel(const float* x, float* y, int num_crop, long TD, float inv) {
    if (VEC) {   // T * D is a multiple of 4: every crop slab starts 16-byte aligned
        const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
        if (i >= TD) return;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int c = 0; c < num_crop; ++c) acc += *reinterpret_cast<const f32x4*>(x + (long)c * TD + i);
        *reinterpret_cast<f32x4*>(y + i) = acc * inv;
    } else {
        const long i = (long)blockIdx.x * 256 + threadIdx.x;
        if (i >= TD) return;
        float acc = 0.f;
        for (int c = 0; c < num_crop; ++c) acc += x[(long)c * TD + i];
        y[i] = acc * inv;
    }
}
// reg [P][C][2] in place: reg[..., k] = reg[..., k] * std[k] + mean[k]   (ssn_test.py:88-90)
__global__ __launch_bounds__(256) void reg_denorm_kernel(float* reg, long n_pairs, float mean0, float std0, float mean1,
                                                         float std1) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_pairs) return;
    float2 v = *reinterpret_cast<float2*>(reg + 2 * i);
    v.x = v.x * std0 + mean0;
    v.y = v.y * std1 + mean1;
    *reinterpret_cast<float2*>(reg + 2 * i) = v;
}

}  // namespace

extern "C" int ssn_crop_mean(const float* x, float* y, int num_crop, int T, int D, hipStream_t stream) {
    SSN_CHECK_ARG(num_crop >= 1 && T >= 0 && D >= 1, "crop_mean: bad arguments");
    const long TD = (long)T * D;
    if (TD == 0) return SSN_OK;      // empty tensors have no storage (NULL) and nothing to do
    SSN_CHECK_ARG(x && y, "crop_mean: null pointer");
    const float inv = 1.0f / (float)num_crop;
    if (TD % 4 == 0)
        hipLaunchKernelGGL(crop_mean_kernel<true>, dim3((unsigned)((TD / 4 + 255) / 256)), dim3(256), 0, stream, x, y,
                           num_crop, TD, inv);
    else
        hipLaunchKernelGGL(crop_mean_kernel<false>, dim3((unsigned)((TD + 255) / 256)), dim3(256), 0, stream, x, y,
                           num_crop, TD, inv);
    SSN_CHECK_LAUNCH("crop_mean");
    return SSN_OK;
}

extern "C" int ssn_reg_denorm(float* reg, long n_pairs, float mean0, float std0, float mean1, float std1,
                              hipStream_t stream) {
    SSN_CHECK_ARG(n_pairs >= 0, "reg_denorm: bad arguments");
    if (n_pairs == 0) return SSN_OK;
    SSN_CHECK_ARG(reg, "reg_denorm: null pointer");
    hipLaunchKernelGGL(reg_denorm_kernel, dim3((unsigned)((n_pairs + 255) / 256)), dim3(256), 0, stream, reg, n_pairs,
                       mean0, std0, mean1, std1);
    SSN_CHECK_LAUNCH("reg_denorm");
    return SSN_OK;
}

extern "C" int ssn_stpp_fwd(const float* ft, const float* scaling, float* act_ft, float* stpp_ft, int P, int D,
                            const SsnStppTable* table, hipStream_t stream) {
    SSN_CHECK_ARG(ft && scaling && act_ft && stpp_ft && table, "stpp_fwd: null pointer");
    SSN_CHECK_ARG(table->n_parts > 0 && table->n_parts <= SSN_STPP_MAX_PARTS, "stpp_fwd: bad part count %d",
                  table->n_parts);
    if (P == 0) return SSN_OK;
    hipLaunchKernelGGL(stpp_fwd_kernel, dim3(P, table->n_parts + 1), dim3(256), 0, stream, ft, scaling, act_ft,
                       stpp_ft, D, *table);
    SSN_CHECK_LAUNCH("stpp_fwd");
    return SSN_OK;
}

extern "C" int ssn_stpp_bwd(const float* d_act, const float* d_stpp, const float* scaling, float* d_ft, int P, int D,
                            const SsnStppTable* table, hipStream_t stream) {
    SSN_CHECK_ARG(d_stpp && scaling && d_ft && table, "stpp_bwd: null pointer");
    SSN_CHECK_ARG(table->n_parts > 0 && table->n_parts <= SSN_STPP_MAX_PARTS, "stpp_bwd: bad part count %d",
                  table->n_parts);
    if (P == 0) return SSN_OK;
    hipLaunchKernelGGL(stpp_bwd_kernel, dim3(P, table->n_seg), dim3(256), 0, stream, d_act, d_stpp, scaling, d_ft, D,
                       *table);
    SSN_CHECK_LAUNCH("stpp_bwd");
    return SSN_OK;
}

extern "C" int ssn_stpp_reorg(const float* scores, int T, int D, const int* ranges, const int* act_range,
                              const float* scaling, const int* part_scale_col, int P, int n_parts, int act_len,
                              int comp_len, int reg_len, float* out_act, float* out_comp, float* out_reg,
                              hipStream_t stream) {
    if (P == 0) return SSN_OK;       // no proposals: the output tensors are empty (NULL storage)
    SSN_CHECK_ARG(scores && ranges && act_range && scaling && part_scale_col && out_act && out_comp,
                  "stpp_reorg: null pointer");
    SSN_CHECK_ARG(D >= act_len + n_parts * (comp_len + (out_reg ? reg_len : 0)), "stpp_reorg: score width %d too small",
                  D);
    hipLaunchKernelGGL(stpp_reorg_kernel, dim3(P), dim3(256), 0, stream, scores, T, D, ranges, act_range, scaling,
                       part_scale_col, n_parts, act_len, comp_len, reg_len, out_act, out_comp, out_reg);
    SSN_CHECK_LAUNCH("stpp_reorg");
    return SSN_OK;
}
