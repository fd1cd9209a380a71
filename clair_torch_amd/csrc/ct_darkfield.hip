// ct_darkfield.hip -- dark-field conditional blur and its uncertainty term, gfx950.
//
// Reference: conditional_gaussian_blur(images, dark, threshold = 0.05, kernel_size = 3, differentiable = True)
// (clair_torch/common/general_functions.py:440-486), called by compute_hdr_image (inference/hdr_merge.py:76-92) and
// linearize_dataset_generator (inference/linearization.py:73-92) on every batch when a dark-field dataset matches:
//
//     m   = sigmoid(50 (D - 0.05))                              soft mask from the dark field D, per pixel
//     xb  = m * blur3x3(x) + (1 - m) * x                        blur = torchvision GaussianBlur(3, sigma = 1): separable
//                                                               [e^-1/2, 1, e^-1/2] / sum, reflect padding
// and the reference then REBINDS `images` to xb: everything downstream (weights, ICRF, merge, and the autograd gradient
// "with respect to images") acts on xb, so the image uncertainty is applied per pixel to xb -- the blur does not couple
// neighbouring pixels in the variance.  The dark field's own uncertainty enters through the mask only, again per pixel:
//     d out / d D = (d out / d xb) * (blur(x) - x) * 50 m (1 - m)
// and, the matched dark fields coming one per frame (datasets/base.py:225-255), its variance term is a sum of squares
// over the frames like the image term (hdr_merge.py:117-126, linearization.py:108-116).  Both terms therefore share the
// per-sample gradient g_n = d out / d xb_n:
//     var = sum_n g_n^2 sigma_n^2 + sum_n g_n^2 (dterm_n sigma_D,n)^2 = sum_n (g_n sigma_eff,n)^2,
//     dterm = (blur(x) - x) 50 m (1 - m),   sigma_eff = sqrt(sigma^2 + (dterm sigma_D)^2)
// so this file needs ONE kernel: it writes xb and sigma_eff as float32 stacks, and the existing merge / linearize
// kernels (float32 pixels + explicit uncertainty stack) do the rest unchanged.
//
// Parity status: UNPINNED.  The blur's arithmetic lives in torchvision, which the reference does not vendor and this
// image lacks; oracle/eager_torch.py restates its published algorithm and the tests compare against that restatement.
//
// Row bands (multi-GPU): the 3x3 stencil needs one row above and below the band.  The caller passes them in
// `halo` ((B, C, 2, W): row above, row below, in the stack's element type); at the top / bottom of the GLOBAL image the
// reflect padding applies instead and that halo row is ignored.  Roofline: HBM (a few planes per frame, each read once
// from HBM and re-read from L2 by the neighbouring rows' threads); written for clarity, one element per thread.
#include <algorithm>
#include "ct_device.hpp"

namespace ct {

struct DarkArgs {
    const void *stack;       // (B, C, H_tile, W) T
    const void *halo;        // (B, C, 2, W) T or NULL
    const float *std_stack;  // explicit sigma (B, C, H_tile, W) or NULL
    const float *dark;       // (Bd, C, H_tile, W), Bd = 1 or B
    const float *dark_std;   // same shape
    float *xb_out;           // (B, C, H_tile, W)
    float *std_out;          // (B, C, H_tile, W)
    int64_t image_stride, dark_stride;  // elements between frames (dark_stride = 0 for one shared dark field)
    int32_t batch, channels, h_tile, width, h_global, row_offset;
    NormConst norm;
    int32_t std_mode;
    float std_value, threshold, alpha;
    float k0, k1;            // the two distinct taps of the normalised 1-D kernel: k0 (centre), k1 (sides)
};

template <typename T>
__global__ __launch_bounds__(kBlock) void dark_blur_kernel(const DarkArgs a)
{
    const int64_t plane = (int64_t)a.h_tile * a.width;
    const int64_t per_image = plane * a.channels;
    const int64_t total = per_image * a.batch;
    const T *stack = static_cast<const T *>(a.stack);
    const T *halo = static_cast<const T *>(a.halo);
    for (int64_t idx = blockIdx.x * (int64_t)kBlock + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * kBlock) {
        const int n = (int)(idx / per_image);
        const int64_t q = idx - (int64_t)n * per_image;
        const int c = (int)(q / plane);
        const int64_t r = q - (int64_t)c * plane;
        const int h = (int)(r / a.width), w = (int)(r - (int64_t)h * a.width);
        const T *img = stack + (int64_t)n * a.image_stride + (int64_t)c * plane;
        // column taps with reflect padding (index -1 -> 1, W -> W - 2)
        const int wl = w > 0 ? w - 1 : 1, wr = w + 1 < a.width ? w + 1 : a.width - 2;
        // a 3-tap row at local row hh, or from the halo / the reflected row at the band's edges
        auto row_blur = [&](int hh) -> float {
            const T *row;
            if (hh < 0) {
                if (a.row_offset == 0)
                    row = img + (int64_t)1 * a.width;  // global top: reflect -1 -> 1
                else
                    row = halo + (((int64_t)n * a.channels + c) * 2 + 0) * a.width;
            } else if (hh >= a.h_tile) {
                if (a.row_offset + a.h_tile == a.h_global)
                    row = img + (int64_t)(a.h_tile - 2) * a.width;  // global bottom: reflect H -> H - 2
                else
                    row = halo + (((int64_t)n * a.channels + c) * 2 + 1) * a.width;
            } else {
                row = img + (int64_t)hh * a.width;
            }
            const float l = to_pixel<T>(row[wl], a.norm), m = to_pixel<T>(row[w], a.norm), rr = to_pixel<T>(row[wr], a.norm);
            return (a.k1 * l + a.k0 * m) + a.k1 * rr;
        };
        const float x = to_pixel<T>(img[r], a.norm);
        const float blurred = (a.k1 * row_blur(h - 1) + a.k0 * row_blur(h)) + a.k1 * row_blur(h + 1);
        const int64_t dq = (int64_t)n * a.dark_stride + q;
        const float d = a.dark[dq];
        const float m = 1.0f / (1.0f + expf(-(d - a.threshold) * a.alpha));  // torch.sigmoid
        const float xb = m * blurred + (1.0f - m) * x;
        a.xb_out[idx] = xb;
        if (a.std_out) {
            float sg = 0.0f;
            if (a.std_mode == CT_STD_EXPLICIT) sg = a.std_stack[(int64_t)n * a.image_stride + q];
            if (a.std_mode == CT_STD_CONSTANT) sg = a.std_value;
            if (a.std_mode == CT_STD_MULTIPLIER) sg = a.std_value * x;  // the dataset derives sigma from the RAW image
            const float dterm = (blurred - x) * (a.alpha * m * (1.0f - m));
            const float ds = dterm * a.dark_std[dq];
            a.std_out[idx] = sqrtf(sg * sg + ds * ds);
        }
    }
}

}  // namespace ct

extern "C" int ct_norm_constants(float max_code, float *hi, float *lo);

extern "C" int ct_dark_field_blur(const void *stack_dev, int32_t dtype, float max_code, int32_t batch,
                                  const ct_geometry *geom, const void *halo_dev, const float *std_dev, int32_t std_mode,
                                  float std_value, const float *dark_dev, const float *dark_std_dev, int32_t dark_batch,
                                  float threshold, float alpha, float *xb_out_dev, float *std_out_dev, void *stream)
{
    using namespace ct;
    if (!stack_dev || !geom || !dark_dev || !xb_out_dev || batch <= 0) return CT_ERR_INVALID_ARGUMENT;
    if (geom->channels <= 0 || geom->h_tile <= 0 || geom->width < 2 || geom->h_global < 2 || geom->h_global < geom->h_tile ||
        geom->row_offset < 0 || geom->row_offset + geom->h_tile > geom->h_global)
        return CT_ERR_INVALID_ARGUMENT;
    if (geom->layout != CT_LAYOUT_NCHW) return CT_ERR_UNSUPPORTED;
    if (dark_batch != 1 && dark_batch != batch) return CT_ERR_INVALID_ARGUMENT;
    // One SHARED dark field for several frames with its uncertainty propagated: the reference's autograd on a (1, C, H, W)
    // mask_map sums the gradient over the frames BEFORE squaring, (sum_n g_n)^2 sigma_D^2, which a per-frame effective
    // sigma cannot express (it would give sum_n (g_n sigma_D)^2).  Not built; refuse rather than return the other quantity.
    if (dark_batch == 1 && batch > 1 && dark_std_dev && std_out_dev) return CT_ERR_UNSUPPORTED;
    if (std_mode < CT_STD_NONE || std_mode > CT_STD_EXPLICIT) return CT_ERR_INVALID_ARGUMENT;
    if (std_mode == CT_STD_EXPLICIT && !std_dev) return CT_ERR_INVALID_ARGUMENT;
    if (std_out_dev && !dark_std_dev) return CT_ERR_INVALID_ARGUMENT;
    // a band that does not touch the global top / bottom needs its neighbours' rows; so does a one-row band's reflection
    const bool top = geom->row_offset == 0, bottom = geom->row_offset + geom->h_tile == geom->h_global;
    if ((!top || !bottom) && !halo_dev) return CT_ERR_INVALID_ARGUMENT;
    if ((top || bottom) && geom->h_tile < 2 && !(top && bottom)) return CT_ERR_UNSUPPORTED;
    const int64_t plane = geom->h_tile * geom->width;
    if (geom->image_stride < plane * geom->channels) return CT_ERR_INVALID_ARGUMENT;
    DarkArgs a{};
    a.stack = stack_dev;
    a.halo = halo_dev;
    a.std_stack = std_dev;
    a.dark = dark_dev;
    a.dark_std = dark_std_dev;
    a.xb_out = xb_out_dev;
    a.std_out = std_out_dev;
    a.image_stride = geom->image_stride;
    a.dark_stride = dark_batch == 1 ? 0 : plane * geom->channels;
    a.batch = batch;
    a.channels = geom->channels;
    a.h_tile = (int32_t)geom->h_tile;
    a.width = (int32_t)geom->width;
    a.h_global = (int32_t)geom->h_global;
    a.row_offset = (int32_t)geom->row_offset;
    a.std_mode = std_mode;
    a.std_value = std_value;
    a.threshold = threshold;
    a.alpha = alpha;
    // torchvision _get_gaussian_kernel1d(3, 1.0): pdf = exp(-0.5 x^2), x = -1, 0, 1; normalised in float32
    const float side = expf(-0.5f), sum = (side + 1.0f) + side;
    a.k0 = 1.0f / sum;
    a.k1 = side / sum;
    a.norm = NormConst{1.0f, 0.0f};
    const int64_t total = plane * geom->channels * batch;
    const int grid = (int)std::min<int64_t>((total + kBlock - 1) / kBlock, (int64_t)compute_units() * 16);
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (dtype) {
        case CT_DTYPE_U8:
            if (ct_norm_constants(max_code, &a.norm.hi, &a.norm.lo) != CT_OK) return CT_ERR_UNSUPPORTED;
            hipLaunchKernelGGL(dark_blur_kernel<uint8_t>, dim3(grid), dim3(kBlock), 0, s, a);
            break;
        case CT_DTYPE_U16:
            if (ct_norm_constants(max_code, &a.norm.hi, &a.norm.lo) != CT_OK) return CT_ERR_UNSUPPORTED;
            hipLaunchKernelGGL(dark_blur_kernel<uint16_t>, dim3(grid), dim3(kBlock), 0, s, a);
            break;
        case CT_DTYPE_F32: hipLaunchKernelGGL(dark_blur_kernel<float>, dim3(grid), dim3(kBlock), 0, s, a); break;
        default: return CT_ERR_UNSUPPORTED;
    }
    return hipGetLastError() == hipSuccess ? CT_OK : CT_ERR_LAUNCH;
}
