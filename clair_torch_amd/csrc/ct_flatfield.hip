// ct_flatfield.hip -- flat-field correction epilogues, gfx950.
//
//  * compute_hdr_image, clair_torch/inference/hdr_merge.py:131-153:
//      M_c = flat_field_mean(flat, 1.0)  (common/general_functions.py:182-210, whole image)
//      corrected = mean / (flat + 1e-6) * M_c  (flatfield_correction, general_functions.py:214-238)
//      var += (d sum(corrected)/d flat * flat_std)^2, the gradient flowing both directly and through M_c
//  * linearize_dataset_generator, clair_torch/inference/linearization.py:48-57,118-130: same correction per frame,
//    M_c a constant (no term through the mean), all float32.
//
// Two tiny kernels: per-channel sums (sum flat, sum value/(flat+eps); float64 atomics, additive over row bands so
// ranks all-reduce them), then an elementwise apply.  Both are HBM-bound on a few image planes -- negligible beside
// the stack pass -- so they are written for clarity, with 16-byte accesses where alignment allows.
#include <algorithm>
#include "ct_device.hpp"

namespace ct {

// VEC = 4: four consecutive elements per thread and iteration through 16/32-byte loads (plane and pointers 4-element
// aligned), two iterations in flight; VEC = 1: any alignment.
template <typename VT, int VEC>
__global__ __launch_bounds__(kBlock) void flatfield_sums_kernel(const VT *value, const float *flat, int64_t plane,
                                                                double *sums)
{
    const int c = blockIdx.y;
    double sf = 0.0, sv = 0.0;
    const float *fl = flat + c * plane;
    const VT *vl = value ? value + c * plane : nullptr;
    struct alignas(sizeof(VT) * VEC) VPack { VT v[VEC]; };
    struct alignas(sizeof(float) * VEC) FPack { float v[VEC]; };
    const int64_t groups = plane / VEC, step = (int64_t)gridDim.x * kBlock;
#pragma unroll 2
    for (int64_t g = blockIdx.x * (int64_t)kBlock + threadIdx.x; g < groups; g += step) {
        const FPack f = *reinterpret_cast<const FPack *>(fl + g * VEC);
        VPack v;
        if (vl) v = *reinterpret_cast<const VPack *>(vl + g * VEC);
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            sf += (double)f.v[e];
            if (vl) sv += (double)v.v[e] / (double)(f.v[e] + 1e-6f);
        }
    }
    // wave reduce (64 lanes), workgroup reduce through LDS, then ONE pair of global atomics per workgroup: all
    // workgroups of a channel hit the same two addresses, and same-address float64 atomics serialise in L2
    for (int off = 32; off > 0; off >>= 1) {
        sf += __shfl_down(sf, off, 64);
        sv += __shfl_down(sv, off, 64);
    }
    __shared__ double part[2 * (kBlock / 64)];
    if ((threadIdx.x & 63) == 0) {
        part[2 * (threadIdx.x >> 6)] = sf;
        part[2 * (threadIdx.x >> 6) + 1] = sv;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double tf = 0.0, tv = 0.0;
        for (int w = 0; w < kBlock / 64; ++w) {
            tf += part[2 * w];
            tv += part[2 * w + 1];
        }
        atomicAdd(&sums[2 * c], tf);
        if (value) atomicAdd(&sums[2 * c + 1], tv);
    }
}

// value: (F, C, plane) VT in/out; var_or_std: (F, C, plane) float32 in/out or NULL
//   IN_IS_VAR : var_or_std holds the variance on entry (merge) or the std (linearize); a std is always written
//   through    : (C) float64 = [sum value/(flat+eps)] / P_global, or NULL when the mean is a constant
template <typename VT, bool IN_IS_VAR>
__global__ __launch_bounds__(kBlock) void flatfield_apply_kernel(VT *value, float *var_or_std, const float *flat,
                                                                 const float *flat_std, const float *flat_mean,
                                                                 const double *through, int channels, int64_t plane,
                                                                 int64_t n_frames)
{
    const int c = blockIdx.y;
    const float M = flat_mean[c];
    const double thr = through ? through[c] : 0.0;
    for (int64_t k = blockIdx.x * (int64_t)kBlock + threadIdx.x; k < plane; k += (int64_t)gridDim.x * kBlock) {
        const float den = flat[c * plane + k] + 1e-6f;
        const float fs = flat_std ? flat_std[c * plane + k] : 0.0f;
        for (int64_t f = 0; f < n_frames; ++f) {
            const int64_t q = (f * channels + c) * plane + k;
            const VT v = value[q];
            float grad;
            if constexpr (sizeof(VT) == 8) {
                value[q] = v / (double)den * (double)M;
                grad = (float)(-(double)v * (double)M / ((double)den * (double)den) + thr);
            } else {
                value[q] = (v / den) * M;
                grad = -(M * v) / (den * den);
            }
            if (var_or_std) {
                float var = var_or_std[q];
                if constexpr (!IN_IS_VAR) var = var * var;
                if (flat_std) {
                    const float gs = grad * fs;
                    var = var + gs * gs;
                }
                var_or_std[q] = sqrtf(var);
            }
        }
    }
}

static int grid_x(int64_t plane)
{
    // per channel: what the device holds at once, so the grid-stride loops run as full rounds of equal work
    const int64_t g = (plane + kBlock - 1) / kBlock, cap = resident_workgroups(0, kBlock);
    return (int)(g > cap ? cap : (g < 1 ? 1 : g));
}

}  // namespace ct

extern "C" int ct_flatfield_sums(const void *value_dev, int32_t value_is_f64, const float *flat_dev, int32_t channels,
                                 int64_t plane, double *sums_dev, void *stream)
{
    using namespace ct;
    if (!flat_dev || !sums_dev || channels <= 0 || plane <= 0) return CT_ERR_INVALID_ARGUMENT;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t vbytes = value_is_f64 ? 32 : 16;
    const bool vec = plane % 4 == 0 && reinterpret_cast<uintptr_t>(flat_dev) % 16 == 0 &&
                     (!value_dev || reinterpret_cast<uintptr_t>(value_dev) % vbytes == 0);
    dim3 grid(std::min(grid_x(vec ? plane / 4 : plane), 512), channels);  // fewer, longer workgroups: fewer atomics
#define CT_FF_SUMS(VT, VEC)                                                                                           \
    hipLaunchKernelGGL((flatfield_sums_kernel<VT, VEC>), grid, dim3(kBlock), 0, s, static_cast<const VT *>(value_dev), \
                       flat_dev, plane, sums_dev)
    if (value_is_f64) {
        if (vec) CT_FF_SUMS(double, 4); else CT_FF_SUMS(double, 1);
    } else {
        if (vec) CT_FF_SUMS(float, 4); else CT_FF_SUMS(float, 1);
    }
#undef CT_FF_SUMS
    return hipGetLastError() == hipSuccess ? CT_OK : CT_ERR_LAUNCH;
}

extern "C" int ct_flatfield_apply(void *value_dev, int32_t value_is_f64, int64_t n_frames, float *var_or_std_dev,
                                  int32_t input_is_variance, const float *flat_dev, const float *flat_std_dev,
                                  const float *flat_mean_dev, const double *through_mean_dev, int32_t channels,
                                  int64_t plane, void *stream)
{
    using namespace ct;
    if (!value_dev || !flat_dev || !flat_mean_dev || channels <= 0 || plane <= 0 || n_frames <= 0)
        return CT_ERR_INVALID_ARGUMENT;
    hipStream_t s = static_cast<hipStream_t>(stream);
    dim3 grid(grid_x(plane), channels);
#define CT_FF_LAUNCH(VT, ISVAR)                                                                                        \
    hipLaunchKernelGGL((flatfield_apply_kernel<VT, ISVAR>), grid, dim3(kBlock), 0, s, static_cast<VT *>(value_dev),     \
                       var_or_std_dev, flat_dev, flat_std_dev, flat_mean_dev, through_mean_dev, channels, plane, n_frames)
    if (value_is_f64) {
        if (input_is_variance) CT_FF_LAUNCH(double, true); else CT_FF_LAUNCH(double, false);
    } else {
        if (input_is_variance) CT_FF_LAUNCH(float, true); else CT_FF_LAUNCH(float, false);
    }
#undef CT_FF_LAUNCH
    return hipGetLastError() == hipSuccess ? CT_OK : CT_ERR_LAUNCH;
}
