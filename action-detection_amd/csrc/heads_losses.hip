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
__device__ __forceinline__ void ce_fwd_body(const float* logits, const long* target, float* loss, float* lse, float* rowloss, int R,
                                            int C) {
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
__global__ __launch_bounds__(256) void ce_fwd_kernel(const float* logits, const long* target, float* loss, float* lse,
                                                     float* rowloss, int R, int C) {
    ce_fwd_body(logits, target, loss, lse, rowloss, R, C);
}
// elements [first, first + count) of the concatenated gradient outputs of a launch are this body's; g: d loss / d (this loss)
__device__ __forceinline__ void ce_bwd_body(const float* logits, const long* target, const float* lse, float g0, float* dlogits, int R,
                                            int C, long i0, long stride) {
    const long total = (long)R * C;
    const float g = g0 / (float)R;
    for (long i = i0; i < total; i += stride) {
        const int r = (int)(i / C), c = (int)(i - (long)r * C);
        const float p = expf(logits[i] - lse[r]);
        dlogits[i] = (p - (c == (int)target[r] ? 1.f : 0.f)) * g;
    }
}
__global__ __launch_bounds__(256) void ce_bwd_kernel(const float* logits, const long* target, const float* lse,
                                                     const float* gout, float* dlogits, int R, int C) {
    ce_bwd_body(logits, target, lse, gout[0], dlogits, R, C, (long)blockIdx.x * 256 + threadIdx.x, (long)gridDim.x * 256);
}

// ---------------------------------------------------------------------------- completeness (OHEM hinge)
// pred [R][C], labels [R] (1-based class; label-1 == -1 wraps to the last column like the
// reference's Python indexing), rows grouped per video: `group` rows, the first `split` positive.
// coef[R] (out): d(loss * den)/d pred[i][col_i]  (= slope if the row is kept, else 0)
__device__ __forceinline__ void completeness_fwd_body(const float* pred, const long* labels, float* loss, float* coef, int R, int C,
                                                      int group, int split, int keep_pos, int keep_neg, float den, float* rowloss) {
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
__global__ __launch_bounds__(256) void completeness_fwd_kernel(const float* pred, const long* labels, float* loss,
                                                               float* coef, int R, int C, int group, int split,
                                                               int keep_pos, int keep_neg, float den, float* rowloss) {
    completeness_fwd_body(pred, labels, loss, coef, R, C, group, split, keep_pos, keep_neg, den, rowloss);
}
__device__ __forceinline__ void completeness_bwd_body(const long* labels, const float* coef, float g0, float* dpred, int R, int C,
                                                      float den, long i0, long stride) {
    const long total = (long)R * C;
    const float g = g0 / den;
    for (long i = i0; i < total; i += stride) {
        const int r = (int)(i / C), c = (int)(i - (long)r * C);
        int col = (int)labels[r] - 1;
        if (col < 0) col += C;
        dpred[i] = (c == col) ? coef[r] * g : 0.f;
    }
}
__global__ __launch_bounds__(256) void completeness_bwd_kernel(const long* labels, const float* coef,
                                                               const float* gout, float* dpred, int R, int C,
                                                               float den) {
    completeness_bwd_body(labels, coef, gout[0], dpred, R, C, den, (long)blockIdx.x * 256 + threadIdx.x, (long)gridDim.x * 256);
}

// ---------------------------------------------------------------------------- class-wise smooth L1
// pred [n][C][2], labels [n], targets [n][2]; loss = mean_{2n}(smoothl1(pred[i][l_i-1][:] - t_i)) * 2
__device__ __forceinline__ void cw_smoothl1_fwd_body(const float* pred, const long* labels, const float* targets, float* loss,
                                                     float* diff, int n, int C) {
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
__global__ __launch_bounds__(256) void cw_smoothl1_fwd_kernel(const float* pred, const long* labels,
                                                              const float* targets, float* loss, float* diff, int n,
                                                              int C) {
    cw_smoothl1_fwd_body(pred, labels, targets, loss, diff, n, C);
}
__device__ __forceinline__ void cw_smoothl1_bwd_body(const long* labels, const float* diff, float g0, float* dpred, int n, int C,
                                                     long i0, long stride) {
    const long total = (long)n * C * 2;
    const float g = g0 * 2.f / (float)(2 * n);
    for (long idx = i0; idx < total; idx += stride) {
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
__global__ __launch_bounds__(256) void cw_smoothl1_bwd_kernel(const long* labels, const float* diff,
                                                              const float* gout, float* dpred, int n, int C) {
    cw_smoothl1_bwd_body(labels, diff, gout[0], dpred, n, C, (long)blockIdx.x * 256 + threadIdx.x, (long)gridDim.x * 256);
}

// ---------------------------------------------------------------------------- [r6] the training objective in one launch each way
// act + w_comp * comp + w_reg * reg (/root/reference/ssn_train.py:210-214): the three losses above run one after the other in ONE
// workgroup (they are a few hundred numbers), the mix is the last instruction -- instead of three launches + five torch micro-kernels
// forward and three + six backward.  Same bodies, same summation orders: every component is bit-identical to its own launch.
struct TotalLossArgs {
    const float* act_logits; const long* act_target; int Ra, Ca;
    const float* comp_pred; const long* comp_labels; int Rc, Cc, group, split, keep_pos, keep_neg; float den;
    const float* reg_pred; const long* reg_labels; const float* reg_targets; int n_reg, Cr;      // reg_pred == nullptr: no regression
    float w_comp, w_reg;
    float* losses;        // [4]: activity, completeness, regression (0 without), total
    float* lse;           // [Ra]          kept for the backward
    float* coef;          // [Rc]          kept
    float* diff;          // [2 n_reg]     kept
    float* scratch;       // [max(Ra, 2 Rc)]
    const float* gout;    // backward: d objective / d total ([1])
    float* d_act; float* d_comp; float* d_reg;
};
__global__ __launch_bounds__(256) void total_loss_fwd_kernel(TotalLossArgs a) {
    ce_fwd_body(a.act_logits, a.act_target, a.losses, a.lse, a.scratch, a.Ra, a.Ca);
    __syncthreads();
    completeness_fwd_body(a.comp_pred, a.comp_labels, a.losses + 1, a.coef, a.Rc, a.Cc, a.group, a.split, a.keep_pos, a.keep_neg, a.den,
                          a.scratch);
    __syncthreads();
    if (a.reg_pred) cw_smoothl1_fwd_body(a.reg_pred, a.reg_labels, a.reg_targets, a.losses + 2, a.diff, a.n_reg, a.Cr);
    __syncthreads();
    if (threadIdx.x == 0) {
        // (every loss was written by thread 0 of this workgroup -- lane 0 of wave 0 in the CE body: program order, nothing to fence)
        if (!a.reg_pred) a.losses[2] = 0.f;
        const float c = a.losses[1] * a.w_comp;
        const float r = a.reg_pred ? a.losses[2] * a.w_reg : 0.f;
        a.losses[3] = (a.losses[0] + c) + r;      // the reference's association: (act + comp * w) + reg * w
    }
}
__global__ __launch_bounds__(256) void total_loss_bwd_kernel(TotalLossArgs a) {
    const long i0 = (long)blockIdx.x * 256 + threadIdx.x, stride = (long)gridDim.x * 256;
    const float g = a.gout[0];
    ce_bwd_body(a.act_logits, a.act_target, a.lse, g, a.d_act, a.Ra, a.Ca, i0, stride);
    completeness_bwd_body(a.comp_labels, a.coef, g * a.w_comp, a.d_comp, a.Rc, a.Cc, a.den, i0, stride);
    if (a.reg_pred) cw_smoothl1_bwd_body(a.reg_labels, a.diff, g * a.w_reg, a.d_reg, a.n_reg, a.Cr, i0, stride);
}

// the label / target bookkeeping of SSN.train_forward (ssn_models.py:275-289: target[act_indexer], target[comp_indexer],
// target[reg_indexer], reg_target[reg_indexer]) in one launch instead of four index_select
__global__ __launch_bounds__(256) void label_select_kernel(const long* target, const float* reg_target, const long* idx0, int n0,
                                                           const long* idx1, int n1, const long* idx2, int n2, long* out0, long* out1,
                                                           long* out2, float* out_reg) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t < n0) out0[t] = target[idx0[t]];
    if (t < n1) out1[t] = target[idx1[t]];
    if (t < n2) {
        out2[t] = target[idx2[t]];
        out_reg[2 * t] = reg_target[2 * idx2[t]];
        out_reg[2 * t + 1] = reg_target[2 * idx2[t] + 1];
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

// [r6] act + w_comp * comp + w_reg * reg of /root/reference/ssn_train.py:210-214 in one launch: losses[4] = activity, completeness,
// regression, total.  lse [Ra], coef [Rc], diff [2 n_reg]: kept for ssn_total_loss_bwd; scratch: max(2 Ra, 2 Rc) floats.
// reg_pred == nullptr: no regression head (ssn_models.py:288-289).
extern "C" int ssn_total_loss_fwd(const float* act_logits, const long* act_target, int Ra, int Ca, const float* comp_pred,
                                  const long* comp_labels, int Rc, int Cc, int group, int split, int keep_pos, int keep_neg, float den,
                                  const float* reg_pred, const long* reg_labels, const float* reg_targets, int n_reg, int Cr,
                                  float w_comp, float w_reg, float* losses, float* lse, float* coef, float* diff, float* scratch,
                                  hipStream_t stream) {
    SSN_CHECK_ARG(act_logits && act_target && comp_pred && comp_labels && losses && lse && coef && scratch && Ra > 0 && Rc > 0,
                  "total_loss_fwd: bad arguments");
    SSN_CHECK_ARG(group > 0 && split >= 0 && split <= group && Rc % group == 0, "total_loss_fwd: %d rows do not form groups of %d", Rc,
                  group);
    SSN_CHECK_ARG(!reg_pred || (reg_labels && reg_targets && diff && n_reg > 0), "total_loss_fwd: bad regression arguments");
    TotalLossArgs a{};
    a.act_logits = act_logits; a.act_target = act_target; a.Ra = Ra; a.Ca = Ca;
    a.comp_pred = comp_pred; a.comp_labels = comp_labels; a.Rc = Rc; a.Cc = Cc; a.group = group; a.split = split;
    a.keep_pos = keep_pos; a.keep_neg = keep_neg; a.den = den;
    a.reg_pred = reg_pred; a.reg_labels = reg_labels; a.reg_targets = reg_targets; a.n_reg = n_reg; a.Cr = Cr;
    a.w_comp = w_comp; a.w_reg = w_reg; a.losses = losses; a.lse = lse; a.coef = coef; a.diff = diff; a.scratch = scratch;
    hipLaunchKernelGGL(total_loss_fwd_kernel, dim3(1), dim3(256), 0, stream, a);
    SSN_CHECK_LAUNCH("total_loss_fwd");
    return SSN_OK;
}
// gout [1] = d objective / d total; writes d_act [Ra, Ca], d_comp [Rc, Cc], d_reg [n_reg, Cr, 2] (complete tensors, zeros included)
extern "C" int ssn_total_loss_bwd(const float* act_logits, const long* act_target, int Ra, int Ca, const long* comp_labels, int Rc,
                                  int Cc, float den, const long* reg_labels, int n_reg, int Cr, float w_comp, float w_reg,
                                  const float* lse, const float* coef, const float* diff, const float* gout, float* d_act,
                                  float* d_comp, float* d_reg, hipStream_t stream) {
    SSN_CHECK_ARG(act_logits && act_target && comp_labels && lse && coef && gout && d_act && d_comp, "total_loss_bwd: null pointer");
    TotalLossArgs a{};
    a.act_logits = act_logits; a.act_target = act_target; a.Ra = Ra; a.Ca = Ca;
    a.comp_labels = comp_labels; a.Rc = Rc; a.Cc = Cc; a.den = den;
    a.reg_pred = d_reg;      // (non-null = there is a regression head)
    a.reg_labels = reg_labels; a.n_reg = n_reg; a.Cr = Cr; a.w_comp = w_comp; a.w_reg = w_reg;
    a.lse = const_cast<float*>(lse); a.coef = const_cast<float*>(coef); a.diff = const_cast<float*>(diff); a.gout = gout;
    a.d_act = d_act; a.d_comp = d_comp; a.d_reg = d_reg;
    long most = (long)Ra * Ca;
    if ((long)Rc * Cc > most) most = (long)Rc * Cc;
    if (d_reg && (long)n_reg * Cr * 2 > most) most = (long)n_reg * Cr * 2;
    hipLaunchKernelGGL(total_loss_bwd_kernel, dim3(grid_for(most)), dim3(256), 0, stream, a);
    SSN_CHECK_LAUNCH("total_loss_bwd");
    return SSN_OK;
}
// target [P] (int64), reg_target [P, 2] -> target[idx0], target[idx1], target[idx2], reg_target[idx2] (ssn_models.py:275-289); idx2 /
// out2 / out_reg may be null (no regression)
extern "C" int ssn_label_select(const long* target, const float* reg_target, const long* idx0, int n0, const long* idx1, int n1,
                                const long* idx2, int n2, long* out0, long* out1, long* out2, float* out_reg, hipStream_t stream) {
    SSN_CHECK_ARG(target && (n0 == 0 || (idx0 && out0)) && (n1 == 0 || (idx1 && out1)) && (n2 == 0 || (idx2 && out2 && out_reg && reg_target)),
                  "label_select: null pointer");
    int most = n0 > n1 ? n0 : n1;
    if (n2 > most) most = n2;
    if (most == 0) return SSN_OK;
    hipLaunchKernelGGL(label_select_kernel, dim3((most + 255) / 256), dim3(256), 0, stream, target, reg_target, idx0, n0, idx1, n1, idx2, n2,
                       out0, out1, out2, out_reg);
    SSN_CHECK_LAUNCH("label_select");
    return SSN_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------
// The fused head: StructuredTemporalPyramidPooling + the three Linear heads + the prop_type row selection of
// SSN.train_forward (/root/reference/ssn_models.py:268-289, ops/ssn_ops.py:39-70) as ONE launch forward and ONE backward, instead
// of 7 and 16 (stpp, 3 x linear, 3 x row gather | 3 x (fill + row scatter), 3 x (dx, dW, db), stpp backward).  Same arithmetic in
// the same order as the separate kernels above and in stpp.hip (the tests compare the two paths).
#define SSN_STPP_MAX_PARTS 24
struct SsnStppTable {      // (stpp.hip holds the same definition: the table the host builds from the reference's torch.arange + int())
    int n_parts;
    int n_seg;
    int act_lo, act_hi;
    int lo[SSN_STPP_MAX_PARTS];
    int hi[SSN_STPP_MAX_PARTS];
    int norm[SSN_STPP_MAX_PARTS];
    int col[SSN_STPP_MAX_PARTS];
};

namespace {

struct HeadsArgs {
    const float* ft;         // [P * n_seg][D] backbone features (behind the dropout)
    const float* scaling;    // [P][2]
    const float* w[3];       // activity [O0][D], completeness [O1][m D], regression [O2][m D] (null: no regression head)
    const float* b[3];
    const int* pos[3];       // [P]: row of proposal p in head h's gathered output, -1 = not selected by its prop_type
    const long* idx[3];      // [n_h]: proposal of gathered row r (backward)
    float* out[3];           // forward: gathered outputs [n_h][O_h];   backward: their gradients (read)
    float* act_ft;           // [P][D]    forward: written (kept for the backward);  backward: read
    float* stpp_ft;          // [P][m D]
    float* d_ft;             // backward: [P * n_seg][D]
    float* dw[3];            // backward: weight / bias gradients
    float* db[3];
    int O[3], n[3];
    int D, P;
    int wblk0[4];            // backward: first dW block of head h (prefix sums over O_h * chunks_h), [3] = total
    SsnStppTable t;
};

// grid (P, chunks of HEADS_CHUNK outputs): every block pools the features of its proposal into LDS ((1 + m) * D floats <= 64 KiB,
// checked by the host; 36 KB of reads -- cheaper than a second launch) and computes its share of the 81 (C = 20) head outputs;
// chunk 0 also writes the pooled features out for the backward.  (One block per proposal -- 32 blocks -- took 144 us.)
constexpr int HEADS_CHUNK = 8;
__global__ __launch_bounds__(256) void heads_fwd_kernel(HeadsArgs a) {
    __shared__ __attribute__((aligned(16))) float sh[16384];
    const int prop = blockIdx.x, chunk = blockIdx.y, D = a.D, m = a.t.n_parts;
    const float* src = a.ft + ((long)prop * a.t.n_seg) * D;
    constexpr int MAXSEG = 16;      // (the host refuses more segments per proposal)
    for (int d = threadIdx.x; d < D; d += 256) {
        // all segment values of this feature first (independent loads: one memory round trip, not one per segment and part)
        float sv[MAXSEG];
#pragma unroll
        for (int q = 0; q < MAXSEG; ++q) sv[q] = q < a.t.n_seg ? src[(long)q * D + d] : 0.f;
        for (int part = 0; part <= m; ++part) {          // part == m: the activity (course mean) feature
            const bool is_act = part == m;
            const int lo = is_act ? a.t.act_lo : a.t.lo[part], hi = is_act ? a.t.act_hi : a.t.hi[part];
            float acc = 0.f;
#pragma unroll
            for (int q = 0; q < MAXSEG; ++q) acc += (q >= lo && q < hi) ? sv[q] : 0.f;      // (ascending segments, as stpp_fwd_kernel)
            float v = acc / (float)(hi - lo);
            if (!is_act) {
                v = v / (float)a.t.norm[part];
                if (a.t.col[part] >= 0) v = v * a.scaling[prop * 2 + a.t.col[part]];
            }
            if (is_act) {
                if (chunk == 0) a.act_ft[(long)prop * D + d] = v;
                sh[d] = v;
            } else {
                if (chunk == 0) a.stpp_ft[((long)prop * m + part) * D + d] = v;
                sh[(long)(1 + part) * D + d] = v;
            }
        }
    }
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int j = chunk * HEADS_CHUNK + wave; j < (chunk + 1) * HEADS_CHUNK; j += 4) {
        // output j of the concatenated heads -> (head h, row o of its weight)
        int h = 0, o = j;
        while (h < 3 && o >= (a.w[h] ? a.O[h] : 0)) {
            o -= a.w[h] ? a.O[h] : 0;
            ++h;
        }
        if (h >= 3) break;
        const int row = a.pos[h][prop];
        if (row < 0) continue;
        const int Dh = h == 0 ? D : m * D;
        const float* xp = h == 0 ? sh : sh + D;
        {
            const float* wp = a.w[h] + (long)o * Dh;
            float acc = 0.f;
            if ((Dh & 3) == 0) {
                for (int d = lane * 4; d < Dh; d += 256) {
                    const f32x4 xv = *reinterpret_cast<const f32x4*>(xp + d);
                    const f32x4 wv = *reinterpret_cast<const f32x4*>(wp + d);
                    acc += xv.x * wv.x + xv.y * wv.y + xv.z * wv.z + xv.w * wv.w;
                }
            } else {
                for (int d = lane; d < Dh; d += 64) acc += xp[d] * wp[d];
            }
            acc = wave_sum(acc);
            if (lane == 0) a.out[h][(long)row * a.O[h] + o] = acc + (a.b[h] ? a.b[h][o] : 0.f);
        }
    }
}

// blocks [0, P * dchunks): gradient of 256 features of one proposal (heads' dx -> STPP backward);  then wblk0[3] blocks: one (head,
// output row, 256-column chunk) of a weight gradient;  the last 3 blocks: the bias gradients
__global__ __launch_bounds__(256) void heads_bwd_kernel(HeadsArgs a) {
    const int D = a.D, m = a.t.n_parts;
    const int dchunks = (D + 255) / 256;
    int b = blockIdx.x;
    if (b < a.P * dchunks) {
        const int prop = b / dchunks;
        int row[3];
        for (int h = 0; h < 3; ++h) row[h] = a.w[h] ? a.pos[h][prop] : -1;
        for (int d = (b - prop * dchunks) * 256 + threadIdx.x; d < D; d += D) {      // (one feature per thread)
            float da = 0.f;
            if (row[0] >= 0)
                for (int o = 0; o < a.O[0]; ++o) da += a.out[0][(long)row[0] * a.O[0] + o] * a.w[0][(long)o * D + d];
            float ds[SSN_STPP_MAX_PARTS];
            for (int part = 0; part < m; ++part) {
                float acc = 0.f;
                if (row[1] >= 0)
                    for (int o = 0; o < a.O[1]; ++o) acc += a.out[1][(long)row[1] * a.O[1] + o] * a.w[1][(long)o * m * D + (long)part * D + d];
                if (row[2] >= 0) {
                    float acc2 = 0.f;
                    for (int o = 0; o < a.O[2]; ++o) acc2 += a.out[2][(long)row[2] * a.O[2] + o] * a.w[2][(long)o * m * D + (long)part * D + d];
                    acc = acc + acc2;
                }
                ds[part] = acc;
            }
            for (int seg = 0; seg < a.t.n_seg; ++seg) {
                float g = 0.f;
                for (int part = 0; part < m; ++part) {
                    if (seg < a.t.lo[part] || seg >= a.t.hi[part]) continue;
                    float v = ds[part];
                    if (a.t.col[part] >= 0) v = v * a.scaling[prop * 2 + a.t.col[part]];
                    v = v / (float)a.t.norm[part];
                    g += v / (float)(a.t.hi[part] - a.t.lo[part]);
                }
                if (seg >= a.t.act_lo && seg < a.t.act_hi) g += da / (float)(a.t.act_hi - a.t.act_lo);
                a.d_ft[((long)prop * a.t.n_seg + seg) * D + d] = g;
            }
        }
        return;
    }
    int wb = b - a.P * dchunks;
    if (wb < a.wblk0[3]) {
        int h = 0;
        while (h < 2 && wb >= a.wblk0[h + 1]) ++h;
        wb -= a.wblk0[h];
        const int Dh = h == 0 ? D : m * D;
        const int chunks = (Dh + 255) / 256;
        const int o = wb / chunks, d = (wb - o * chunks) * 256 + threadIdx.x;
        if (d >= Dh) return;
        const float* x = h == 0 ? a.act_ft : a.stpp_ft;
        float acc = 0.f;
        int r = 0;
        for (; r + 4 <= a.n[h]; r += 4) {      // four independent (index -> row) chains in flight; summation order unchanged
            long i4[4];
            float g4[4], x4[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                i4[q] = a.idx[h][r + q];
                g4[q] = a.out[h][(long)(r + q) * a.O[h] + o];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) x4[q] = x[i4[q] * Dh + d];
#pragma unroll
            for (int q = 0; q < 4; ++q) acc += g4[q] * x4[q];
        }
        for (; r < a.n[h]; ++r) acc += a.out[h][(long)r * a.O[h] + o] * x[a.idx[h][r] * Dh + d];
        a.dw[h][(long)o * Dh + d] = acc;
        return;
    }
    const int h = wb - a.wblk0[3];
    if (h < 3 && a.w[h] && a.db[h])
        for (int o = threadIdx.x; o < a.O[h]; o += 256) {
            float acc = 0.f;
            for (int r = 0; r < a.n[h]; ++r) acc += a.out[h][(long)r * a.O[h] + o];
            a.db[h][o] = acc;
        }
}

int fill_heads(HeadsArgs& a, const float* ft, const float* scaling, const float* const* w, const float* const* b,
               const int* const* pos, const long* const* idx, float* const* out, const int* O, const int* n, float* act_ft,
               float* stpp_ft, int P, int D, const SsnStppTable* table, const char* what) {
    SSN_CHECK_ARG(ft && scaling && w && b && pos && idx && out && O && n && act_ft && stpp_ft && table, "%s: null pointer", what);
    SSN_CHECK_ARG(P > 0 && D > 0 && table->n_parts >= 1 && table->n_parts <= SSN_STPP_MAX_PARTS && table->n_seg <= 16,
                  "%s: bad shape (at most 24 parts, 16 segments per proposal)", what);
    SSN_CHECK_ARG(w[0] && w[1] && pos[0] && pos[1] && out[0] && out[1], "%s: the activity and completeness heads are required", what);
    a.ft = ft;
    a.scaling = scaling;
    for (int h = 0; h < 3; ++h) {
        a.w[h] = w[h];
        a.b[h] = b[h];
        a.pos[h] = pos[h];
        a.idx[h] = idx[h];
        a.out[h] = out[h];
        a.O[h] = O[h];
        a.n[h] = n[h];
        a.dw[h] = a.db[h] = nullptr;
    }
    a.act_ft = act_ft;
    a.stpp_ft = stpp_ft;
    a.d_ft = nullptr;
    a.D = D;
    a.P = P;
    a.t = *table;
    return SSN_OK;
}

}  // namespace

// w / b / pos / idx / out / O / n: HOST arrays of 3 (activity, completeness, regression; entry 2 of w null = no regression head).
// pos[h]: DEVICE int [P], idx[h]: DEVICE long [n_h].  Forward: out[h] [n_h][O_h] <- the gathered head outputs, act_ft / stpp_ft <-
// the pooled features (kept for the backward).
extern "C" int ssn_heads_fwd(const float* ft, const float* scaling, const float* const* w, const float* const* b,
                             const int* const* pos, const long* const* idx, float* const* out, const int* O, const int* n,
                             float* act_ft, float* stpp_ft, int P, int D, const SsnStppTable* table, hipStream_t stream) {
    HeadsArgs a;
    int rc = fill_heads(a, ft, scaling, w, b, pos, idx, out, O, n, act_ft, stpp_ft, P, D, table, "heads fwd");
    if (rc != SSN_OK) return rc;
    const size_t lds = (size_t)(1 + table->n_parts) * D * sizeof(float);
    SSN_CHECK_ARG(lds <= 65536, "heads fwd: %d parts x %d features do not fit the 64 KiB of LDS this kernel uses", table->n_parts, D);
    int J = 0;
    for (int h = 0; h < 3; ++h) J += w[h] ? O[h] : 0;
    hipLaunchKernelGGL(heads_fwd_kernel, dim3(P, (J + HEADS_CHUNK - 1) / HEADS_CHUNK), dim3(256), 0, stream, a);
    SSN_CHECK_LAUNCH("heads_fwd");
    return SSN_OK;
}
// Backward: dout[h] = gradients of the gathered outputs (read); d_ft [P * n_seg][D], dw[h], db[h] (HOST arrays of 3 device
// pointers; db entries may be null) <- gradients.
extern "C" int ssn_heads_bwd(const float* ft_unused, const float* scaling, const float* const* w, const float* const* b,
                             const int* const* pos, const long* const* idx, float* const* dout, const int* O, const int* n,
                             float* act_ft, float* stpp_ft, int P, int D, const SsnStppTable* table, float* d_ft,
                             float* const* dw, float* const* db, hipStream_t stream) {
    HeadsArgs a;
    int rc = fill_heads(a, ft_unused ? ft_unused : act_ft, scaling, w, b, pos, idx, dout, O, n, act_ft, stpp_ft, P, D, table,
                        "heads bwd");
    if (rc != SSN_OK) return rc;
    SSN_CHECK_ARG(d_ft && dw && db && dw[0] && dw[1] && (!w[2] || dw[2]), "heads bwd: null gradient pointer");
    a.d_ft = d_ft;
    int blocks = 0;
    for (int h = 0; h < 3; ++h) {
        a.dw[h] = dw[h];
        a.db[h] = db[h];
        a.wblk0[h] = blocks;
        if (w[h]) blocks += O[h] * (((h == 0 ? D : table->n_parts * D) + 255) / 256);
    }
    a.wblk0[3] = blocks;
    hipLaunchKernelGGL(heads_bwd_kernel, dim3(P * ((D + 255) / 256) + blocks + 3), dim3(256), 0, stream, a);
    SSN_CHECK_LAUNCH("heads_bwd");
    return SSN_OK;
}
