// Input side of the hot path on the GPU: crop / horizontal flip / channel roll / float conversion / normalisation of
// decoded uint8 frames, i.e. the arithmetic of the reference's transform chain after image decoding and scaling
//   GroupOverSample | (crop + GroupRandomHorizontalFlip)  ->  Stack(roll)  ->  ToTorchFormatTensor(div=False)  ->
//   GroupNormalize          (/root/reference/transforms.py:103-132, 49-64, 256-268, 271-288, 67-80;
//                            wired at ssn_test.py:101-112 and ssn_train.py:106-111)
// which the reference runs image by image in PIL / numpy on the loader workers ("this transpose takes 80% of the
// loading time", transforms.py:286).  HBM-bound: 1 byte read, 4 bytes written per output element.
#include "ssn_common.h"

namespace {

constexpr int MAX_CROPS = 16;
struct CropTable {
    int n;
    int ox[MAX_CROPS], oy[MAX_CROPS], flip[MAX_CROPS];
};

// src [n_img][Hs][Ws][C] uint8 (HWC as decoded);  dst [n_crops][n_img][C][ch][cw] fp32
//   dst[k][i][c][y][x] = (float(px) - mean[c]) / std[c],
//   px = src[i][oy_k + y][ox_k + (flip_k ? cw - 1 - x : x)][roll ? C - 1 - c : c],
//   px = 255 - px when invert_even && flip_k && i is even (flow: the x component changes sign under a flip).
__global__ __launch_bounds__(256) void frames_kernel(const uint8_t* src, float* dst, int n_img, int Hs, int Ws, int C,
                                                     int ch, int cw, CropTable t, int roll, int invert_even,
                                                     const float* mean, const float* stdv, int n_mean, int n_std) {
    const long plane = (long)ch * cw;
    const long total = (long)t.n * n_img * C * plane;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int x = (int)(idx % cw);
        long r = idx / cw;
        const int y = (int)(r % ch);
        r /= ch;
        const int c = (int)(r % C);
        r /= C;
        const int i = (int)(r % n_img);
        const int k = (int)(r / n_img);
        const int sx = t.ox[k] + (t.flip[k] ? cw - 1 - x : x);
        const int sc = roll ? C - 1 - c : c;
        int px = src[(((long)i * Hs + t.oy[k] + y) * Ws + sx) * C + sc];
        if (invert_even && t.flip[k] && (i & 1) == 0) px = 255 - px;
        // GroupNormalize repeats mean / std over the stacked channels: channel (i * C + c) of the stack
        const int sch = i * C + c;
        dst[idx] = ((float)px - mean[sch % n_mean]) / stdv[sch % n_std];
    }
}

// RGBDiff input (SSN._get_diff, /root/reference/ssn_models.py:302-316, keep_rgb = False): every segment holds
// new_length + 1 stacked RGB frames; the network sees the new_length differences of consecutive frames,
//   out[g][x][c][hw] = in[g][x + 1][c][hw] - in[g][x][c][hw],   x < new_length
// i.e. out[g][j] = in[g][j + plane] - in[g][j] with plane = C * HW floats and j < new_length * plane.
__global__ __launch_bounds__(256) void frame_diff_kernel(const float* in, float* out, long per_seg_out, long per_seg_in,
                                                         long plane, long total) {
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const long g = idx / per_seg_out, j = idx - g * per_seg_out;
        const float* src = in + g * per_seg_in + j;
        out[idx] = src[plane] - src[0];
    }
}


// ---------------------------------------------------------------------------------------------------------------------------
// Training augmentation on the GPU: GroupMultiScaleCrop (crop + PIL bilinear resize to the network input) ->
// GroupRandomHorizontalFlip -> Stack(roll) -> ToTorchFormatTensor(div=False) -> GroupNormalize
// (/root/reference/transforms.py:135-206, 49-64, 256-268, 271-288, 67-80; the chain of ssn_train.py:106-111 with
// SSN.get_augmentation(), ssn_models.py:386-395) on decoded uint8 frames, so that loader workers only decode.
//
// The resize restates Pillow's ImagingResample for 8-bit images (what `img.crop(box).resize(size, Image.BILINEAR)` runs): per
// axis, output index xx takes the source window [xmin, xmin + n) around centre (xx + 0.5) * scale with triangle weights of
// support max(scale, 1), normalised, rounded to 22-bit fixed point; the horizontal pass rounds to uint8 before the vertical
// pass reads it.  Coefficients are computed in double with contraction off -- the same IEEE operations in the same order as
// the C library -- so the result is bit-identical to PIL's, not "within an LSB".
constexpr int RS_MAXK = 7;                 // taps per axis: downscaling by up to 3x
constexpr int RS_ENT = 2 + RS_MAXK;        // ints per table entry: first source index, tap count, coefficients
constexpr int RS_PREC = 22;                // Pillow: PRECISION_BITS = 32 - 8 - 2

// one thread per (image, axis, output index): tab[img][out_w + out_h][RS_ENT]
__global__ __launch_bounds__(256) void resize_coeffs_kernel(const int* box, int n_img, int out_w, int out_h, int* tab) {
#pragma clang fp contract(off)
    const int per = out_w + out_h;
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= n_img * per) return;
    const int img = t / per, j = t - img * per;
    const bool vert = j >= out_w;
    const int xx = vert ? j - out_w : j;
    const int inSize = box[img * 4 + (vert ? 3 : 2)], outSize = vert ? out_h : out_w;
    double scale = (double)inSize / (double)outSize;
    double filterscale = scale;
    if (filterscale < 1.0) filterscale = 1.0;
    const double support = 1.0 * filterscale;          // bilinear: support 1
    const double center = 0.0 + (xx + 0.5) * scale;
    const double ss = 1.0 / filterscale;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > inSize) xmax = inSize;
    xmax -= xmin;
    if (xmax > RS_MAXK) xmax = RS_MAXK;                 // (the host refuses such boxes)
    double k[RS_MAXK];
    double ww = 0.0;
    for (int x = 0; x < RS_MAXK; ++x) {
        double w = 0.0;
        if (x < xmax) {
            double a = (x + xmin - center + 0.5) * ss;
            if (a < 0.0) a = -a;
            w = a < 1.0 ? 1.0 - a : 0.0;
        }
        k[x] = w;
        ww += w;
    }
    int* e = tab + ((long)img * per + j) * RS_ENT;
    e[0] = xmin;
    e[1] = xmax;
    for (int x = 0; x < RS_MAXK; ++x) {
        double v = k[x];
        if (ww != 0.0) v /= ww;
        e[2 + x] = v < 0 ? (int)(-0.5 + v * (double)(1 << RS_PREC)) : (int)(0.5 + v * (double)(1 << RS_PREC));
    }
}

__device__ __forceinline__ int rs_clip8(int v) {
    v >>= RS_PREC;
    return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// dst [n_img][C][out_h][out_w] fp32 = normalise(roll(flip(resize(crop(src[i], box[i])))))
__global__ __launch_bounds__(256) void frames_resize_kernel(const uint8_t* src, float* dst, int n_img, int Hs, int Ws, int C,
                                                            int out_h, int out_w, const int* box, const int* flip,
                                                            const int* tab, int roll, int invert_even, const float* mean,
                                                            const float* stdv, int n_mean, int n_std) {
    const long plane = (long)out_h * out_w;
    const long total = (long)n_img * C * plane;
    const int per = out_w + out_h;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int x = (int)(idx % out_w);
        long r = idx / out_w;
        const int y = (int)(r % out_h);
        r /= out_h;
        const int c = (int)(r % C);
        const int i = (int)(r / C);
        const int fl = flip[i];
        const int rx = fl ? out_w - 1 - x : x;                       // FLIP_LEFT_RIGHT is applied to the resized image
        const int sc = roll ? C - 1 - c : c;
        const int x0 = box[i * 4 + 0], y0 = box[i * 4 + 1];
        const int* eh = tab + ((long)i * per + rx) * RS_ENT;
        const int* ev = tab + ((long)i * per + out_w + y) * RS_ENT;
        const int hx = eh[0], hn = eh[1], vy = ev[0], vn = ev[1];
        int acc_v = 1 << (RS_PREC - 1);
        for (int ky = 0; ky < vn; ++ky) {
            const uint8_t* row = src + (((long)i * Hs + y0 + vy + ky) * Ws + x0 + hx) * C + sc;
            int acc_h = 1 << (RS_PREC - 1);
            for (int kx = 0; kx < hn; ++kx) acc_h += (int)row[(long)kx * C] * eh[2 + kx];
            acc_v += rs_clip8(acc_h) * ev[2 + ky];                   // (the horizontal pass is stored as uint8 by Pillow)
        }
        int px = rs_clip8(acc_v);
        if (invert_even && fl && (i & 1) == 0) px = 255 - px;
        const int sch = i * C + c;
        dst[idx] = ((float)px - mean[sch % n_mean]) / stdv[sch % n_std];
    }
}

}  // namespace

extern "C" int ssn_frame_diff(const float* in, float* out, long n_segments, int new_length, int C, int HW,
                              hipStream_t stream) {
    SSN_CHECK_ARG(in && out && n_segments >= 0 && new_length >= 1 && C >= 1 && HW >= 1, "frame_diff: bad arguments");
    const long plane = (long)C * HW;
    const long total = n_segments * new_length * plane;
    if (total == 0) return SSN_OK;
    long blocks = (total + 256 * 8 - 1) / (256 * 8);
    if (blocks > 65535 * 4) blocks = 65535 * 4;
    hipLaunchKernelGGL(frame_diff_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, in, out, new_length * plane,
                       (new_length + 1) * plane, plane, total);
    SSN_CHECK_LAUNCH("frame_diff");
    return SSN_OK;
}

extern "C" int ssn_frames_crop_normalize(const unsigned char* src, float* dst, int n_img, int Hs, int Ws, int C,
                                         int crop_h, int crop_w, int n_crops, const int* off_x, const int* off_y,
                                         const int* flip, int roll, int invert_even, const float* mean, int n_mean,
                                         const float* stdv, int n_std, hipStream_t stream) {
    SSN_CHECK_ARG(src && dst && off_x && off_y && flip && mean && stdv, "frames: null pointer");
    SSN_CHECK_ARG(n_crops >= 1 && n_crops <= MAX_CROPS, "frames: %d crops (max %d)", n_crops, MAX_CROPS);
    SSN_CHECK_ARG(n_mean >= 1 && n_std >= 1 && C >= 1, "frames: bad channel arguments");
    CropTable t;
    t.n = n_crops;
    for (int k = 0; k < n_crops; ++k) {
        SSN_CHECK_ARG(off_x[k] >= 0 && off_y[k] >= 0 && off_x[k] + crop_w <= Ws && off_y[k] + crop_h <= Hs,
                      "frames: crop %d (%d, %d) + %dx%d leaves the %dx%d frame", k, off_x[k], off_y[k], crop_w, crop_h, Ws, Hs);
        t.ox[k] = off_x[k];
        t.oy[k] = off_y[k];
        t.flip[k] = flip[k];
    }
    const long total = (long)n_crops * n_img * C * crop_h * crop_w;
    if (total == 0) return SSN_OK;
    long blocks = (total + 256 * 8 - 1) / (256 * 8);
    if (blocks > 65535 * 4) blocks = 65535 * 4;
    hipLaunchKernelGGL(frames_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, (const uint8_t*)src, dst, n_img, Hs, Ws,
                       C, crop_h, crop_w, t, roll, invert_even, mean, stdv, n_mean, n_std);
    SSN_CHECK_LAUNCH("frames_crop_normalize");
    return SSN_OK;
}

// GroupMultiScaleCrop + GroupRandomHorizontalFlip + Stack + ToTorchFormatTensor + GroupNormalize in two launches (see above).
// src [n_img][Hs][Ws][C] uint8 (decoded frames, HWC); dst [n_img][C][out_h][out_w] fp32.  box: DEVICE int [n_img][4] =
// (x0, y0, crop_w, crop_h) of every image (the reference draws one box per group = per proposal: ssn_dataset.py:347-380), flip:
// DEVICE int [n_img].  The caller validates the boxes (they must lie inside the frame; crop / output <= 3 per axis).
// workspace: ssn_frames_resize_workspace_bytes(n_img, out_h, out_w) bytes of device memory (the coefficient tables).
extern "C" size_t ssn_frames_resize_workspace_bytes(int n_img, int out_h, int out_w) {
    return (size_t)n_img * (size_t)(out_h + out_w) * RS_ENT * sizeof(int);
}
extern "C" int ssn_frames_crop_resize_normalize(const unsigned char* src, float* dst, int n_img, int Hs, int Ws, int C,
                                                int out_h, int out_w, const int* box, const int* flip, int roll,
                                                int invert_even, const float* mean, int n_mean, const float* stdv, int n_std,
                                                void* workspace, size_t workspace_bytes, hipStream_t stream) {
    SSN_CHECK_ARG(src && dst && box && flip && mean && stdv && workspace, "frames resize: null pointer");
    SSN_CHECK_ARG(n_img >= 0 && C >= 1 && out_h >= 1 && out_w >= 1 && n_mean >= 1 && n_std >= 1, "frames resize: bad arguments");
    if (workspace_bytes < ssn_frames_resize_workspace_bytes(n_img, out_h, out_w)) {
        ssn_set_error("frames resize: workspace too small");
        return SSN_ERR_WORKSPACE;
    }
    if (n_img == 0) return SSN_OK;
    int* tab = reinterpret_cast<int*>(workspace);
    const int nt = n_img * (out_w + out_h);
    hipLaunchKernelGGL(resize_coeffs_kernel, dim3((nt + 255) / 256), dim3(256), 0, stream, box, n_img, out_w, out_h, tab);
    const long total = (long)n_img * C * out_h * out_w;
    long blocks = (total + 256 * 4 - 1) / (256 * 4);
    if (blocks > 65535 * 4) blocks = 65535 * 4;
    hipLaunchKernelGGL(frames_resize_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, (const uint8_t*)src, dst, n_img, Hs,
                       Ws, C, out_h, out_w, box, flip, tab, roll, invert_even, mean, stdv, n_mean, n_std);
    SSN_CHECK_LAUNCH("frames_crop_resize_normalize");
    return SSN_OK;
}
