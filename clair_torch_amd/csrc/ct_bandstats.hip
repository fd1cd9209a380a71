// ct_bandstats.hip -- per-channel statistics of one merged row band (gfx950).
//
// BASELINE configuration C5 ("32-exposure 8192x8192x3 stack tile-sharded across 8 x MI355X, merge + uncertainty, RCCL
// gather of per-tile stats") gathers, per rank, min / max / sum of the merged mean and of its std per channel.  With
// torch reductions that is six passes over the band (1.3 ms for the whole 8192^2 image on one GPU, a quarter of the
// step); here it is ONE pass: every workgroup reduces a slice of one channel plane to six numbers, a second tiny kernel
// folds the slices.  No atomics: the result is deterministic, and min / max / sum combine over bands on the host side
// of the gather exactly as the torch version's did.
//
// Roofline: HBM, 12 B per element read (float64 mean + float32 std), nothing written.
#include <algorithm>
#include "ct_device.hpp"

namespace ct {

constexpr int kStatSlices = 512;  // workgroups per channel: 1536 for C = 3, six per CU

struct BandStat {
    double mn_lo, mn_hi, mn_sum, sd_lo, sd_hi, sd_sum;
};

__device__ __forceinline__ void fold(BandStat &a, const BandStat &b)
{
    a.mn_lo = fmin(a.mn_lo, b.mn_lo);
    a.mn_hi = fmax(a.mn_hi, b.mn_hi);
    a.mn_sum += b.mn_sum;
    a.sd_lo = fmin(a.sd_lo, b.sd_lo);
    a.sd_hi = fmax(a.sd_hi, b.sd_hi);
    a.sd_sum += b.sd_sum;
}

__device__ __forceinline__ BandStat shuffle_down(const BandStat &s, int off)
{
    BandStat r;
    r.mn_lo = __shfl_down(s.mn_lo, off, 64);
    r.mn_hi = __shfl_down(s.mn_hi, off, 64);
    r.mn_sum = __shfl_down(s.mn_sum, off, 64);
    r.sd_lo = __shfl_down(s.sd_lo, off, 64);
    r.sd_hi = __shfl_down(s.sd_hi, off, 64);
    r.sd_sum = __shfl_down(s.sd_sum, off, 64);
    return r;
}

// workgroup -> one BandStat in part[] (thread 0 returns true and holds it)
__device__ __forceinline__ bool block_fold(BandStat &s)
{
    for (int off = 32; off > 0; off >>= 1) {
        const BandStat o = shuffle_down(s, off);
        fold(s, o);
    }
    __shared__ BandStat part[kBlock / 64];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x != 0) return false;
    for (int w = 1; w < kBlock / 64; ++w) fold(s, part[w]);
    return true;
}

// grid (kStatSlices, C): slice x of channel y.  VEC = 2: 16-byte loads of the mean, 8-byte loads of the std.
template <int VEC>
__global__ __launch_bounds__(kBlock) void band_stats_kernel(const double *mean, const float *sd, int64_t plane, BandStat *partial)
{
    const int c = blockIdx.y;
    const double *m = mean + c * plane;
    const float *s = sd ? sd + c * plane : nullptr;
    BandStat acc{INFINITY, -INFINITY, 0.0, INFINITY, -INFINITY, 0.0};
    float s_lo = INFINITY, s_hi = -INFINITY;  // float32 min / max of float32 data: exact
    struct alignas(8 * VEC) MPack { double v[VEC]; };
    struct alignas(4 * VEC) SPack { float v[VEC]; };
    const int64_t groups = plane / VEC, step = (int64_t)gridDim.x * kBlock;
#pragma unroll 4
    for (int64_t g = blockIdx.x * (int64_t)kBlock + threadIdx.x; g < groups; g += step) {
        const MPack mv = *reinterpret_cast<const MPack *>(m + g * VEC);
        SPack sv;
        if (s) sv = *reinterpret_cast<const SPack *>(s + g * VEC);
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            acc.mn_lo = fmin(acc.mn_lo, mv.v[e]);
            acc.mn_hi = fmax(acc.mn_hi, mv.v[e]);
            acc.mn_sum += mv.v[e];
            if (s) {
                s_lo = fminf(s_lo, sv.v[e]);
                s_hi = fmaxf(s_hi, sv.v[e]);
                acc.sd_sum += (double)sv.v[e];
            }
        }
    }
    acc.sd_lo = s_lo;
    acc.sd_hi = s_hi;
    if (block_fold(acc)) partial[(int64_t)c * gridDim.x + blockIdx.x] = acc;
}

// grid (C): folds the slices of a channel; out (6, C) float64 rows: min mean, max mean, sum mean, min std, max std, sum std
__global__ __launch_bounds__(kBlock) void band_stats_fold_kernel(const BandStat *partial, int slices, int channels, int has_std,
                                                                 double *out)
{
    const int c = blockIdx.x;
    BandStat acc{INFINITY, -INFINITY, 0.0, INFINITY, -INFINITY, 0.0};
    for (int k = threadIdx.x; k < slices; k += kBlock) fold(acc, partial[(int64_t)c * slices + k]);
    if (!block_fold(acc)) return;
    out[0 * channels + c] = acc.mn_lo;
    out[1 * channels + c] = acc.mn_hi;
    out[2 * channels + c] = acc.mn_sum;
    out[3 * channels + c] = has_std ? acc.sd_lo : 0.0;
    out[4 * channels + c] = has_std ? acc.sd_hi : 0.0;
    out[5 * channels + c] = has_std ? acc.sd_sum : 0.0;
}

}  // namespace ct

extern "C" int64_t ct_band_stats_workspace(int32_t channels)
{
    return channels > 0 ? (int64_t)channels * ct::kStatSlices * (int64_t)sizeof(ct::BandStat) : 0;
}

extern "C" int ct_band_stats(const double *mean_dev, const float *std_dev, int32_t channels, int64_t plane,
                             void *workspace_dev, int64_t workspace_bytes, double *out_dev, void *stream)
{
    using namespace ct;
    if (!mean_dev || !out_dev || channels <= 0 || plane <= 0) return CT_ERR_INVALID_ARGUMENT;
    if (!workspace_dev || workspace_bytes < ct_band_stats_workspace(channels) ||
        reinterpret_cast<uintptr_t>(workspace_dev) % 8 != 0)
        return CT_ERR_INVALID_ARGUMENT;
    hipStream_t s = static_cast<hipStream_t>(stream);
    BandStat *partial = static_cast<BandStat *>(workspace_dev);
    const bool vec = plane % 2 == 0 && reinterpret_cast<uintptr_t>(mean_dev) % 16 == 0 &&
                     (std_dev == nullptr || reinterpret_cast<uintptr_t>(std_dev) % 8 == 0);
    const int64_t groups = vec ? plane / 2 : plane;
    const int slices = (int)std::min<int64_t>(kStatSlices, (groups + kBlock - 1) / kBlock);
    if (vec)
        hipLaunchKernelGGL(band_stats_kernel<2>, dim3(slices, channels), dim3(kBlock), 0, s, mean_dev, std_dev, plane, partial);
    else
        hipLaunchKernelGGL(band_stats_kernel<1>, dim3(slices, channels), dim3(kBlock), 0, s, mean_dev, std_dev, plane, partial);
    hipLaunchKernelGGL(band_stats_fold_kernel, dim3(channels), dim3(kBlock), 0, s, partial, slices, channels, std_dev ? 1 : 0, out_dev);
    return hipGetLastError() == hipSuccess ? CT_OK : CT_ERR_LAUNCH;
}
