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
