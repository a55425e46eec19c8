// Rectangular-tap forward convolutions on the split-operand (f16 x 3, "x6" family) kernel of conv_x6_kernel.h, gfx950.
//
// BN-Inception only has square 1x1 / 3x3 taps (conv_x6.hip).  The Inception-v3 backbone the reference's tester runs on
// ActivityNet (/root/reference/ssn_models.py:133-139, BASELINE.json configs[4]) also has 5x5, 1x7, 7x1, 1x3 and 3x1
// layers with per-axis padding; they are the same kernel template instantiated for KH x KW taps (the slab is still 16
// channels x one tap, taps innermost, rows of KW), forward only (dense testing runs no backward).  A separate
// translation unit so that the instantiations compile in parallel with conv_x6.hip.
#include "conv_epilogue.h"
#include "ssn_common.h"
#include "conv_x6_kernel.h"

namespace {

using namespace x6;

template <int KH, int KW, int WM, int WN, int TM, int TN>
int launch_rect_cfg(X6Args& a, hipStream_t stream) {
    constexpr int BM = WM * TM * 32;
    constexpr int BN = WN * TN * 32;
    a.hw_real = a.Ho * a.Wo;
    a.n_mtiles = (a.M + BM - 1) / BM;
    a.div_mt = make_fastdiv((uint32_t)a.n_mtiles);
    // 16-byte activation loads under the same conditions as the square kernels (same-size stride-1 convolution, the largest
    // tap displacement inside the readable guard in front of x; planes that are not a multiple of 4 pixels -- 35 x 35, 17 x 17
    // -- enumerated padded)
    if (a.x_guard >= 256 && a.Ho == a.H && a.Wo == a.W && x6_reach_bytes(a.pad_h, a.pad_w, KH, KW, a.W) <= 256) {
        x6_pad_enumeration(a);
        a.n_ptiles = (a.P + BN - 1) / BN;
        const unsigned nblk = (unsigned)a.n_ptiles * (unsigned)a.n_mtiles;
        hipLaunchKernelGGL((conv_x6_kernel<KH, KW, 1, MODE_FWD, true, 1, WM, WN, TM, TN>), dim3(nblk), dim3(256), 0, stream, a);
        SSN_CHECK_LAUNCH("conv_x6_rect (wide)");
        return SSN_OK;
    }
    a.n_ptiles = (a.P + BN - 1) / BN;
    const unsigned nblk = (unsigned)a.n_ptiles * (unsigned)a.n_mtiles;
    hipLaunchKernelGGL((conv_x6_kernel<KH, KW, 1, MODE_FWD, false, 1, WM, WN, TM, TN>), dim3(nblk), dim3(256), 0, stream, a);
    SSN_CHECK_LAUNCH("conv_x6_rect");
    return SSN_OK;
}

// tiles (rows x pixels), ids as in conv_x6.hip: 1 64x128, 3 64x64, 2 96x128 (1x4 waves), 5 128x128 (1x4 waves), 6 64x128 (1x4 waves)
template <int KH, int KW>
int launch_rect(X6Args& a, int cfg, hipStream_t stream) {
    if (cfg < 0) {
        cfg = (a.M % 128 == 0 || a.M > 256) ? 5 : ((a.M % 96 == 0 || (a.M > 128 && a.M <= 192)) ? 2 : (a.M <= 64 ? 6 : 5));
        // a small problem does not fill the 512 workgroup slots with 128-pixel tiles: take 64 x 64 ones
        const long wg = ((a.P + 127) / 128) * ((a.M + 127) / 128);
        if (wg < 384) cfg = 3;
    }
    switch (cfg) {
        case 1: return launch_rect_cfg<KH, KW, 2, 2, 1, 2>(a, stream);
        case 3: return launch_rect_cfg<KH, KW, 2, 2, 1, 1>(a, stream);
        case 2: return launch_rect_cfg<KH, KW, 1, 4, 3, 1>(a, stream);
        case 6: return launch_rect_cfg<KH, KW, 1, 4, 2, 1>(a, stream);
        default: return launch_rect_cfg<KH, KW, 1, 4, 4, 1>(a, stream);
    }
}

}  // namespace

extern "C" long ssn_conv_x6_dgrad_s2_packed_floats(int Cout, int Cin) {
    long tot = 0;
    for (int cls = 0; cls < 4; ++cls) tot += x6_packed_dwords_kk(Cout, Cin, (1 + (cls >> 1)) * (1 + (cls & 1)), 1);
    return tot;
}

// Data gradient of a 3x3 / stride-2 / pad-1 convolution (even H, W) on the split kernel: four stride-1 launches,
// one per parity class of the input pixel (see ssn_conv_x6_pack_dgrad_s2), each a (1 + a) x (1 + b)-tap gather over
// dy whose results are stored at the class's pixels of dx.  No tap is multiplied that does not contribute (the
// exact-f32 kernel of conv_igemm.hip gets there with a parity-ordered pixel enumeration; this is the same idea on
// the 2.65x faster matrix path).  accumulate / mask_y / mask_scale as ssn_conv_x6_dgrad.
extern "C" int ssn_conv_x6_dgrad_s2(const float* dy, const float* wt_packed, float* dx, int N, int Cout, int Ho, int Wo,
                                    long dy_img_stride, int Cin, int H, int W, long dx_img_stride, int accumulate,
                                    const float* mask_y, long mask_img_stride, const float* mask_scale,
                                    int dy_guard_bytes, int tile_cfg, const float* dy_amax, float* dx_amax, int pad,
                                    hipStream_t stream) {
    SSN_CHECK_ARG(dy && wt_packed && dx, "conv x6 dgrad s2: null pointer");
    SSN_CHECK_ARG(dy_amax, "conv x6 dgrad s2: the source tensor's amax slot is required");
    SSN_CHECK_ARG((pad == 1 && H % 2 == 0 && W % 2 == 0 && Ho == H / 2 && Wo == W / 2) ||
                      (pad == 0 && H >= 3 && W >= 3 && Ho == (H - 3) / 2 + 1 && Wo == (W - 3) / 2 + 1),
                  "conv x6 dgrad s2: needs a 3x3 / stride-2 convolution with pad 1 and even input size, or pad 0 (%dx%d -> %dx%d, pad %d)",
                  H, W, Ho, Wo, pad);
    // pad 1: the two-tap classes are the ODD input rows / columns (2u + 1 <- dy[u], dy[u + 1]), every class grid is dy's.
    // pad 0: the two-tap classes are the EVEN ones (2u <- dy[u - 1], dy[u]: one padding row in front of the gather), and
    //        the class grids cover the input rows / columns of their parity (input rows a 3x3 window never reaches --
    //        the last one of an even-sized input -- get the zero they are owed from taps that fall off dy).
    long off = 0;
    for (int cls = 0; cls < 4; ++cls) {
        const int ca = cls >> 1, cb = cls & 1, kh = 1 + ca, kw = 1 + cb;
        const int sub_a = pad ? ca : 1 - ca, sub_b = pad ? cb : 1 - cb;     // parity of the class's input rows / columns
        const int gh = pad ? Ho : (H - sub_a + 1) / 2, gw = pad ? Wo : (W - sub_b + 1) / 2;
        X6Args a;
        a.x = dy;
        a.ap = (const uint32_t*)wt_packed + off;
        a.y = dx;
        a.x_amax = dy_amax;
        a.y_amax = dx_amax;
        a.x_amax2 = nullptr;
        a.y_amax2 = nullptr;
        a.scale = nullptr;
        a.shift = nullptr;
        a.N = N;
        a.C = Cout;
        a.H = Ho;
        a.W = Wo;
        a.x_img_stride = dy_img_stride;
        a.M = Cin;
        a.Ho = gh;          // enumerated grid = the class's pixels (u, v)
        a.Wo = gw;
        a.y_img_stride = dx_img_stride;
        a.P = N * gh * gw;
        a.pad_h = pad ? 0 : ca;
        a.pad_w = pad ? 0 : cb;
        a.relu = 0;
        a.raw_from = a.row_split = 0x7fffffff;
        a.row_gap = a.k_split = a.k_gap = 0;
        a.accumulate = accumulate;
        a.mask_y = mask_scale ? mask_y : nullptr;
        a.mask_scale = mask_y ? mask_scale : nullptr;
        a.mask_img_stride = mask_img_stride;
        a.ngroups = (Cout + 15) / 16;
        a.x_guard = dy_guard_bytes;
        a.trace = nullptr;
        a.dbg = 0;
        a.sub_a = sub_a;
        a.sub_b = sub_b;
        a.sub_W = W;
        a.sub_HW = H * W;
        a.div_hw = make_fastdiv((uint32_t)(gh * gw));
        a.div_w = make_fastdiv((uint32_t)gw);
        const long xb = ((long)(N - 1) * dy_img_stride + (long)Cout * Ho * Wo) * 4;
        const long ab = x6_row_dwords_kk(Cout, Cin, kh * kw, 1) * 4;   // packed rows; the amax tail follows
        const long yb = ((long)(N - 1) * dx_img_stride + (long)Cin * H * W) * 4;
        const long mb = a.mask_y ? ((long)(N - 1) * mask_img_stride + (long)Cin * H * W) * 4 : 0;
        SSN_CHECK_ARG(xb < (1l << 31) && ab < (1l << 31) && yb < (1l << 31) && mb < (1l << 31),
                      "conv x6 dgrad s2: operand larger than 2 GiB (buffer addressing)");
        a.x_bytes = (uint32_t)xb;
        a.a_bytes = (uint32_t)ab;
        a.y_bytes = (uint32_t)yb;
        a.mask_bytes = (uint32_t)mb;
        int rc;
        if (cls == 0) rc = launch_rect<1, 1>(a, tile_cfg, stream);
        else if (cls == 1) rc = launch_rect<1, 2>(a, tile_cfg, stream);
        else if (cls == 2) rc = launch_rect<2, 1>(a, tile_cfg, stream);
        else rc = launch_rect<2, 2>(a, tile_cfg, stream);
        if (rc != SSN_OK) return rc;
        off += ab / 4 + ATAIL;
    }
    return SSN_OK;
}

static int dispatch_rect(X6Args& a, int kh, int kw, int tile_cfg, hipStream_t stream) {
    if (kh == 4 && kw == 4) return launch_rect<4, 4>(a, tile_cfg, stream);
    if (kh == 5 && kw == 5) return launch_rect<5, 5>(a, tile_cfg, stream);
    if (kh == 1 && kw == 7) return launch_rect<1, 7>(a, tile_cfg, stream);
    if (kh == 7 && kw == 1) return launch_rect<7, 1>(a, tile_cfg, stream);
    if (kh == 1 && kw == 3) return launch_rect<1, 3>(a, tile_cfg, stream);
    if (kh == 3 && kw == 1) return launch_rect<3, 1>(a, tile_cfg, stream);
    ssn_set_error("conv x6 rect: %dx%d taps have no kernel", kh, kw);
    return SSN_ERR_ARG;
}

// Data gradient of a stride-1, same-size layer with kh x kw taps (the layers ssn_conv_x6_fwd_rect runs forward): the
// forward correlation of dy with the transposed, tap-reversed weight (ssn_conv_x6_pack_dgrad_rect), i.e. the SAME kernel
// instantiations as the forward pass.  accumulate / mask_y / mask_scale as ssn_conv_x6_dgrad.
extern "C" int ssn_conv_x6_dgrad_rect(const float* dy, const float* wt_packed, float* dx, int N, int Cout, int H, int W,
                                      long dy_img_stride, int Cin, long dx_img_stride, int kh, int kw, int pad_h, int pad_w,
                                      int accumulate, const float* mask_y, long mask_img_stride, const float* mask_scale,
                                      int dy_guard_bytes, int tile_cfg, const float* dy_amax, float* dx_amax,
                                      hipStream_t stream) {
    SSN_CHECK_ARG(dy && wt_packed && dx, "conv x6 dgrad rect: null pointer");
    SSN_CHECK_ARG(dy_amax, "conv x6 dgrad rect: the source tensor's amax slot is required");
    SSN_CHECK_ARG(2 * pad_h == kh - 1 && 2 * pad_w == kw - 1, "conv x6 dgrad rect: same-size layers only (%dx%d taps, pad %d,%d)",
                  kh, kw, pad_h, pad_w);
    X6Args a;
    a.x = dy;
    a.ap = (const uint32_t*)wt_packed;
    a.y = dx;
    a.x_amax = dy_amax;
    a.y_amax = dx_amax;
    a.x_amax2 = nullptr;
    a.y_amax2 = nullptr;
    a.scale = nullptr;
    a.shift = nullptr;
    a.N = N;
    a.C = Cout;
    a.H = H;
    a.W = W;
    a.x_img_stride = dy_img_stride;
    a.M = Cin;
    a.Ho = H;
    a.Wo = W;
    a.y_img_stride = dx_img_stride;
    a.P = N * H * W;
    a.pad_h = kh - 1 - pad_h;
    a.pad_w = kw - 1 - pad_w;
    a.relu = 0;
    a.raw_from = a.row_split = 0x7fffffff;
    a.row_gap = a.k_split = a.k_gap = 0;
    a.accumulate = accumulate;
    a.mask_y = mask_scale ? mask_y : nullptr;
    a.mask_scale = mask_y ? mask_scale : nullptr;
    a.mask_img_stride = mask_img_stride;
    a.ngroups = (Cout + 15) / 16;
    a.x_guard = dy_guard_bytes;
    a.trace = nullptr;
    a.dbg = 0;
    a.sub_a = a.sub_b = a.sub_W = a.sub_HW = 0;
    a.div_hw = make_fastdiv((uint32_t)(H * W));
    a.div_w = make_fastdiv((uint32_t)W);
    const long xb = ((long)(N - 1) * dy_img_stride + (long)Cout * H * W) * 4;
    const long ab = x6_row_dwords_kk(Cout, Cin, kh * kw, 1) * 4;
    const long yb = ((long)(N - 1) * dx_img_stride + (long)Cin * H * W) * 4;
    const long mb = a.mask_y ? ((long)(N - 1) * mask_img_stride + (long)Cin * H * W) * 4 : 0;
    SSN_CHECK_ARG(xb < (1l << 31) && ab < (1l << 31) && yb < (1l << 31) && mb < (1l << 31) && (long)N * H * W < (1l << 31),
                  "conv x6 dgrad rect: operand larger than 2 GiB (buffer addressing)");
    a.x_bytes = (uint32_t)xb;
    a.a_bytes = (uint32_t)ab;
    a.y_bytes = (uint32_t)yb;
    a.mask_bytes = (uint32_t)mb;
    return dispatch_rect(a, kh, kw, tile_cfg, stream);
}

extern "C" long ssn_conv_x6_packed_floats_rect(int Cout, int Cin, int kh, int kw) {
    return x6_packed_dwords_kk(Cout, Cin, kh * kw, 0);
}

// y = relu?(scale * conv(x, w) + shift), stride 1, taps kh x kw in {5x5, 1x7, 7x1, 1x3, 3x1}, per-axis padding.
// w_packed from ssn_conv_x6_pack_weights_rect.  Other arguments as ssn_conv_x6_fwd.
extern "C" int ssn_conv_x6_fwd_rect(const float* x, const float* w_packed, const float* scale, const float* shift, float* y,
                                    int N, int Cin, int H, int W, long x_img_stride, int Cout, int Ho, int Wo,
                                    long y_img_stride, int kh, int kw, int pad_h, int pad_w, int relu, int x_guard_bytes,
                                    int tile_cfg, const float* x_amax, float* y_amax, hipStream_t stream) {
    SSN_CHECK_ARG(x && w_packed && y, "conv x6 rect: null pointer");
    SSN_CHECK_ARG(x_amax, "conv x6 rect: the source tensor's amax slot is required");
    // (an output grid SHORTER than H + 2 pad - k + 1 = less padding behind the image than in front of it: the space-to-depth
    //  form of the 7x7 / stride-2 stem is a 4x4 convolution with 2 padding pixels in front and 1 behind)
    SSN_CHECK_ARG(Ho <= H + 2 * pad_h - kh + 1 && Wo <= W + 2 * pad_w - kw + 1 && Ho >= H + pad_h - kh + 1 && Wo >= W + pad_w - kw + 1,
                  "conv x6 rect: output %dx%d does not match", Ho, Wo);
    X6Args a;
    a.x = x;
    a.ap = (const uint32_t*)w_packed;
    a.y = y;
    a.x_amax = x_amax;
    a.y_amax = y_amax;
    a.x_amax2 = nullptr;
    a.y_amax2 = nullptr;
    a.scale = scale;
    a.shift = shift;
    a.N = N;
    a.C = Cin;
    a.H = H;
    a.W = W;
    a.x_img_stride = x_img_stride;
    a.M = Cout;
    a.Ho = Ho;
    a.Wo = Wo;
    a.y_img_stride = y_img_stride;
    a.P = N * Ho * Wo;
    a.pad_h = pad_h;
    a.pad_w = pad_w;
    a.relu = relu;
    a.raw_from = a.row_split = 0x7fffffff;
    a.row_gap = a.k_split = a.k_gap = 0;
    a.accumulate = 0;
    a.mask_y = nullptr;
    a.mask_scale = nullptr;
    a.mask_img_stride = 0;
    a.ngroups = (Cin + 15) / 16;
    a.x_guard = x_guard_bytes;
    a.trace = nullptr;
    a.dbg = 0;
    a.sub_a = a.sub_b = a.sub_W = a.sub_HW = 0;
    a.div_hw = make_fastdiv((uint32_t)(Ho * Wo));
    a.div_w = make_fastdiv((uint32_t)Wo);
    const long xb = ((long)(N - 1) * x_img_stride + (long)Cin * H * W) * 4;
    const long ab = x6_row_dwords_kk(Cout, Cin, kh * kw, 0) * 4;
    const long yb = ((long)(N - 1) * y_img_stride + (long)Cout * Ho * Wo) * 4;
    SSN_CHECK_ARG(xb < (1l << 31) && ab < (1l << 31) && yb < (1l << 31) && (long)N * Ho * Wo < (1l << 31),
                  "conv x6 rect: operand larger than 2 GiB (buffer addressing)");
    a.x_bytes = (uint32_t)xb;
    a.a_bytes = (uint32_t)ab;
    a.y_bytes = (uint32_t)yb;
    a.mask_bytes = 0;
    return dispatch_rect(a, kh, kw, tile_cfg, stream);
}
