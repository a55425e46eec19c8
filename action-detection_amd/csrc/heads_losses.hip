// Classifier heads and losses of SSN as wavefront-reduction kernels (gfx950, wave64).
//
// * ssn_linear_{fwd,bwd_*}: activity_fc / completeness_fc / regressor_fc / test_fc
//   (/root/reference/ssn_models.py:77-78,87,272-273,283,300).  The matrices are skinny
//   (P x 1024/3072 times 20..1001), so each output element is one wave-wide dot product with
//   float4 loads; weights stay L2-resident.
// * ssn_row_gather / ssn_row_scatter: the prop_type index_select of ssn_models.py:275-289.
// * ssn_ce_loss_*: torch.nn.CrossEntropyLoss (mean) of /root/reference/ssn_train.py:133,210.
// * ssn_completeness_loss_*: CompletenessLoss + OHEMHingeLoss of
//   /root/reference/ops/ssn_ops.py:173-239 -- the reference's per-row Python loops (one D2H sync
//   per element) become one single-workgroup launch each way.
// * ssn_cw_smoothl1_*: ClassWiseRegressionLoss of /root/reference/ops/ssn_ops.py:242-258.
// Reductions are order-fixed (no float atomics) so losses are bit-reproducible run to run.
#include "ssn_common.h"

namespace {

// ---------------------------------------------------------------------------- linear
__global__ __launch_bounds__(256) void linear_fwd_kernel(const float* x, const float* w, const float* b, float* out,
                                                         int R, int O, int D) {
    const long wave = ((long)blockIdx.x * 256 + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (wave >= (long)R * O) return;
    const int r = (int)(wave / O), o = (int)(wave - (long)r * O);
    const float* xp = x + (long)r * D;
    const float* wp = w + (long)o * D;
    float acc = 0.f;
    if ((D & 3) == 0) {
        for (int d = lane * 4; d < D; d += 256) {
            const f32x4 xv = *reinterpret_cast<const f32x4*>(xp + d);
            const f32x4 wv = *reinterpret_cast<const f32x4*>(wp + d);
            acc += xv.x * wv.x + xv.y * wv.y + xv.z * wv.z + xv.w * wv.w;
        }
    } else {
        for (int d = lane; d < D; d += 64) acc += xp[d] * wp[d];
    }
    acc = wave_sum(acc);
    if (lane == 0) out[wave] = acc + (b ? b[o] : 0.f);
}

// dx[r][d] (+)= sum_o dout[r][o] * w[o][d]
__global__ __launch_bounds__(256) void linear_bwd_x_kernel(const float* dout, const float* w, float* dx, int R, int O,
                                                           int D, int accumulate) {
    const int r = blockIdx.y;
    const int d = blockIdx.x * 256 + threadIdx.x;
    if (d >= D) return;
    float acc = 0.f;
    for (int o = 0; o < O; ++o) acc += dout[(long)r * O + o] * w[(long)o * D + d];
    float* dst = dx + (long)r * D + d;
    *dst = accumulate ? *dst + acc : acc;
}
// dw[o][d] = sum_r dout[r][o] * x[r][d]
__global__ __launch_bounds__(256) void linear_bwd_w_kernel(const float* dout, const float* x, float* dw, int R, int O,
                                                           int D) {
    const int o = blockIdx.y;
    const int d = blockIdx.x * 256 + threadIdx.x;
    if (d >= D) return;
    float acc = 0.f;
    for (int r = 0; r < R; ++r) acc += dout[(long)r * O + o] * x[(long)r * D + d];
    dw[(long)o * D + d] = acc;
}
__global__ __launch_bounds__(256) void linear_bwd_b_kernel(const float* dout, float* db, int R, int O) {
    const int o = blockIdx.x * 256 + threadIdx.x;
    if (o >= O) return;
    float acc = 0.f;
    for (int r = 0; r < R; ++r) acc += dout[(long)r * O + o];
    db[o] = acc;
}

// ---------------------------------------------------------------------------- row gather / scatter
__global__ __launch_bounds__(256) void row_gather_kernel(const float* src, const long* index, float* dst, int n_idx,
                                                         int width) {
    const int i = blockIdx.y;
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c < width) dst[(long)i * width + c] = src[index[i] * width + c];
}
__global__ __launch_bounds__(256) void row_scatter_kernel(const float* src, const long* index, float* dst, int n_idx,
                                                          int width) {
    const int i = blockIdx.y;
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c < width) dst[index[i] * width + c] = src[(long)i * width + c];
}
__global__ __launch_bounds__(256) void fill_kernel(float* x, long n, float v) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) x[i] = v;
}

// ---------------------------------------------------------------------------- cross entropy
// single workgroup; rows are distributed over the 4 waves.  ws: [R] lse, then [R] row losses.
__global__ __launch_bounds__(256) void ce_fwd_kernel(const float* logits, const long* target, float* loss, float* lse,
                                                     float* rowloss, int R, int C) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int r = wave; r < R; r += 4) {
        const float* x = logits + (long)r * C;
        float m = -INFINITY;
        for (int c = lane; c < C; c += 64) m = fmaxf(m, x[c]);
        m = wave_max(m);
        float s = 0.f;
        for (int c = lane; c < C; c += 64) s += expf(x[c] - m);
        s = wave_sum(s);
        if (lane == 0) {
            const float l = m + logf(s);
            lse[r] = l;
            rowloss[r] = l - x[target[r]];
        }
    }
    __syncthreads();
    if (wave == 0) {
        float s = 0.f;
        for (int r = lane; r < R; r += 64) s += rowloss[r];
        s = wave_sum(s);
        if (lane == 0) loss[0] = s / (float)R;
    }
}
__global__ __launch_bounds__(256) void ce_bwd_kernel(const float* logits, const long* target, const float* lse,
                                                     const float* gout, float* dlogits, int R, int C) {
    const long total = (long)R * C;
    const float g = gout[0] / (float)R;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int r = (int)(i / C), c = (int)(i - (long)r * C);
        const float p = expf(logits[i] - lse[r]);
        dlogits[i] = (p - (c == (int)target[r] ? 1.f : 0.f)) * g;
    }
}

// ---------------------------------------------------------------------------- completeness (OHEM hinge)
// pred [R][C], labels [R] (1-based class; label-1 == -1 wraps to the last column like the
// reference's Python indexing), rows grouped per video: `group` rows, the first `split` positive.
// coef[R] (out): d(loss * den)/d pred[i][col_i]  (= slope if the row is kept, else 0)
__global__ __launch_bounds__(256) void completeness_fwd_kernel(const float* pred, const long* labels, float* loss,
                                                               float* coef, int R, int C, int group, int split,
                                                               int keep_pos, int keep_neg, float den, float* rowloss) {
    // pass 1: hinge losses
    for (int i = threadIdx.x; i < R; i += 256) {
        const int gi = i % group;
        const float y = gi < split ? 1.f : -1.f;
        int col = (int)labels[i] - 1;
        if (col < 0) col += C;
        const float v = 1.f - y * pred[(long)i * C + col];
        rowloss[i] = fmaxf(0.f, v);
    }
    __syncthreads();
    // pass 2: rank inside the (video, sign) group -> keep the `keep` largest (ties: lower index first)
    for (int i = threadIdx.x; i < R; i += 256) {
        const int v = i / group, gi = i - v * group;
        const bool pos = gi < split;
        const int lo = v * group + (pos ? 0 : split);
        const int hi = v * group + (pos ? split : group);
        const int keep = pos ? keep_pos : keep_neg;
        const float li = rowloss[i];
        int rank = 0;
        for (int j = lo; j < hi; ++j) {
            const float lj = rowloss[j];
            rank += (lj > li) || (lj == li && j < i);
        }
        const bool kept = rank < keep;
        const float slope = (li != 0.f) ? (pos ? -1.f : 1.f) : 0.f;
        coef[i] = kept ? slope : 0.f;
        // stash kept loss (or 0) for the ordered sum
        rowloss[R + i] = kept ? li : 0.f;
    }
    __syncthreads();
    // pass 3: ordered sums (positives first, then negatives), matching the reference's
    // group-by-group accumulation (ops/ssn_ops.py:193-194, 239)
    if (threadIdx.x == 0) {
        float pos_ls = 0.f, neg_ls = 0.f;
        for (int i = 0; i < R; ++i) {
            if ((i % group) < split)
                pos_ls += rowloss[R + i];
            else
                neg_ls += rowloss[R + i];
        }
        loss[0] = pos_ls / den + neg_ls / den;
    }
}
__global__ __launch_bounds__(256) void completeness_bwd_kernel(const long* labels, const float* coef,
                                                               const float* gout, float* dpred, int R, int C,
                                                               float den) {
    const long total = (long)R * C;
    const float g = gout[0] / den;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int r = (int)(i / C), c = (int)(i - (long)r * C);
        int col = (int)labels[r] - 1;
        if (col < 0) col += C;
        dpred[i] = (c == col) ? coef[r] * g : 0.f;
    }
}

// ---------------------------------------------------------------------------- class-wise smooth L1
// pred [n][C][2], labels [n], targets [n][2]; loss = mean_{2n}(smoothl1(pred[i][l_i-1][:] - t_i)) * 2
__global__ __launch_bounds__(256) void cw_smoothl1_fwd_kernel(const float* pred, const long* labels,
                                                              const float* targets, float* loss, float* diff, int n,
                                                              int C) {
    for (int e = threadIdx.x; e < 2 * n; e += 256) {
        const int i = e >> 1, j = e & 1;
        int col = (int)labels[i] - 1;
        if (col < 0) col += C;
        diff[e] = pred[((long)i * C + col) * 2 + j] - targets[e];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float s = 0.f;
        for (int e = 0; e < 2 * n; ++e) {
            const float d = diff[e], a = fabsf(d);
            s += a < 1.f ? 0.5f * d * d : a - 0.5f;
        }
        loss[0] = (s / (float)(2 * n)) * 2.f;
    }
}
__global__ __launch_bounds__(256) void cw_smoothl1_bwd_kernel(const long* labels, const float* diff,
                                                              const float* gout, float* dpred, int n, int C) {
    const long total = (long)n * C * 2;
    const float g = gout[0] * 2.f / (float)(2 * n);
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int j = (int)(idx & 1);
        const long rc = idx >> 1;
        const int i = (int)(rc / C), c = (int)(rc - (long)i * C);
        int col = (int)labels[i] - 1;
        if (col < 0) col += C;
        float v = 0.f;
        if (c == col) {
            const float d = diff[i * 2 + j];
            v = (fabsf(d) < 1.f ? d : (d > 0.f ? 1.f : -1.f)) * g;
        }
        dpred[idx] = v;
    }
}

inline unsigned grid_for(long total, int cap = 2048) {
    long b = (total + 255) / 256;
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (unsigned)b;
}

}  // namespace

extern "C" int ssn_linear_fwd(const float* x, const float* w, const float* b, float* out, int R, int O, int D,
                              hipStream_t stream) {
    SSN_CHECK_ARG(x && w && out, "linear_fwd: null pointer");
    if (R == 0 || O == 0) return SSN_OK;
    const long waves = (long)R * O;
    hipLaunchKernelGGL(linear_fwd_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, stream, x, w, b, out, R, O,
                       D);
    SSN_CHECK_LAUNCH("linear_fwd");
    return SSN_OK;
}
extern "C" int ssn_linear_bwd(const float* dout, const float* x, const float* w, float* dx, float* dw, float* db,
                              int R, int O, int D, int accumulate_dx, hipStream_t stream) {
    SSN_CHECK_ARG(dout && x && w, "linear_bwd: null pointer");
    if (R == 0 || O == 0) return SSN_OK;
    if (dx) hipLaunchKernelGGL(linear_bwd_x_kernel, dim3((D + 255) / 256, R), dim3(256), 0, stream, dout, w, dx, R, O, D,
                               accumulate_dx);
    if (dw) hipLaunchKernelGGL(linear_bwd_w_kernel, dim3((D + 255) / 256, O), dim3(256), 0, stream, dout, x, dw, R, O, D);
    if (db) hipLaunchKernelGGL(linear_bwd_b_kernel, dim3((O + 255) / 256), dim3(256), 0, stream, dout, db, R, O);
    SSN_CHECK_LAUNCH("linear_bwd");
    return SSN_OK;
}

extern "C" int ssn_row_gather(const float* src, const long* index, float* dst, int n_idx, int width,
                              hipStream_t stream) {
    SSN_CHECK_ARG(src && index && dst, "row_gather: null pointer");
    if (n_idx == 0) return SSN_OK;
    hipLaunchKernelGGL(row_gather_kernel, dim3((width + 255) / 256, n_idx), dim3(256), 0, stream, src, index, dst,
                       n_idx, width);
    SSN_CHECK_LAUNCH("row_gather");
    return SSN_OK;
}
// dst[n_rows][width] = 0; dst[index[i]] = src[i]   (indices are unique: they come from nonzero())
extern "C" int ssn_row_scatter(const float* src, const long* index, float* dst, int n_idx, int n_rows, int width,
                               hipStream_t stream) {
    SSN_CHECK_ARG(src && index && dst, "row_scatter: null pointer");
    const long total = (long)n_rows * width;
    hipLaunchKernelGGL(fill_kernel, dim3(grid_for(total)), dim3(256), 0, stream, dst, total, 0.f);
    if (n_idx)
        hipLaunchKernelGGL(row_scatter_kernel, dim3((width + 255) / 256, n_idx), dim3(256), 0, stream, src, index, dst,
                           n_idx, width);
    SSN_CHECK_LAUNCH("row_scatter");
    return SSN_OK;
}

// workspace: 2*R floats (lse then row losses); lse (first R floats) must be kept for the backward
extern "C" int ssn_ce_loss_fwd(const float* logits, const long* target, float* loss, float* workspace, int R, int C,
                               hipStream_t stream) {
    SSN_CHECK_ARG(logits && target && loss && workspace && R > 0, "ce_loss_fwd: bad arguments");
    hipLaunchKernelGGL(ce_fwd_kernel, dim3(1), dim3(256), 0, stream, logits, target, loss, workspace, workspace + R, R,
                       C);
    SSN_CHECK_LAUNCH("ce_loss_fwd");
    return SSN_OK;
}
extern "C" int ssn_ce_loss_bwd(const float* logits, const long* target, const float* lse, const float* gout,
                               float* dlogits, int R, int C, hipStream_t stream) {
    SSN_CHECK_ARG(logits && target && lse && gout && dlogits, "ce_loss_bwd: null pointer");
    hipLaunchKernelGGL(ce_bwd_kernel, dim3(grid_for((long)R * C)), dim3(256), 0, stream, logits, target, lse, gout,
                       dlogits, R, C);
    SSN_CHECK_LAUNCH("ce_loss_bwd");
    return SSN_OK;
}

// workspace: 2*R floats scratch; coef: R floats kept for the backward
extern "C" int ssn_completeness_loss_fwd(const float* pred, const long* labels, float* loss, float* coef,
                                         float* workspace, int R, int C, int group, int split, int keep_pos,
                                         int keep_neg, float den, hipStream_t stream) {
    SSN_CHECK_ARG(pred && labels && loss && coef && workspace, "completeness_fwd: null pointer");
    SSN_CHECK_ARG(group > 0 && split >= 0 && split <= group && R % group == 0,
                  "completeness_fwd: %d rows do not form groups of %d", R, group);
    hipLaunchKernelGGL(completeness_fwd_kernel, dim3(1), dim3(256), 0, stream, pred, labels, loss, coef, R, C, group,
                       split, keep_pos, keep_neg, den, workspace);
    SSN_CHECK_LAUNCH("completeness_fwd");
    return SSN_OK;
}
extern "C" int ssn_completeness_loss_bwd(const long* labels, const float* coef, const float* gout, float* dpred,
                                         int R, int C, float den, hipStream_t stream) {
    SSN_CHECK_ARG(labels && coef && gout && dpred, "completeness_bwd: null pointer");
    hipLaunchKernelGGL(completeness_bwd_kernel, dim3(grid_for((long)R * C)), dim3(256), 0, stream, labels, coef, gout,
                       dpred, R, C, den);
    SSN_CHECK_LAUNCH("completeness_bwd");
    return SSN_OK;
}

// diff: 2*n floats, kept for the backward
extern "C" int ssn_cw_smoothl1_fwd(const float* pred, const long* labels, const float* targets, float* loss,
                                   float* diff, int n, int C, hipStream_t stream) {
    SSN_CHECK_ARG(pred && labels && targets && loss && diff && n > 0, "cw_smoothl1_fwd: bad arguments");
    hipLaunchKernelGGL(cw_smoothl1_fwd_kernel, dim3(1), dim3(256), 0, stream, pred, labels, targets, loss, diff, n, C);
    SSN_CHECK_LAUNCH("cw_smoothl1_fwd");
    return SSN_OK;
}
extern "C" int ssn_cw_smoothl1_bwd(const long* labels, const float* diff, const float* gout, float* dpred, int n,
                                   int C, hipStream_t stream) {
    SSN_CHECK_ARG(labels && diff && gout && dpred, "cw_smoothl1_bwd: null pointer");
    hipLaunchKernelGGL(cw_smoothl1_bwd_kernel, dim3(grid_for((long)n * C * 2)), dim3(256), 0, stream, labels, diff,
                       gout, dpred, n, C);
    SSN_CHECK_LAUNCH("cw_smoothl1_bwd");
    return SSN_OK;
}
