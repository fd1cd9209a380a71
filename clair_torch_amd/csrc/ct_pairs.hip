// ct_pairs.hip -- per-pixel exposure-pair linearity residual: spatial sums (forward) and LUT gradient (backward).
//
// Replaces the interior of one train_icrf step (clair_torch/training/icrf_training.py:105-133) and of
// measure_linearity (clair_torch/inference/measure_linearity.py:44-72): get_pairwise_valid_pixel_mask
// (common/general_functions.py:276-312), combined_gaussian_pair_weights (training/losses.py:208-235),
// the model forward, pixelwise_linearity_loss (losses.py:13-67) and compute_spatial_linearity_loss ->
// weighted_mean_and_std over (H, W) (losses.py:70-108, general_functions.py:118-178).  The reference
// materialises five (P, C, H, W) float64 tensors (P = O(N^2) pairs); here nothing pair-sized ever leaves the CU.
//
// Structure (all kernels): a workgroup walks tiles of TP pixels of ONE channel plane.  Phase 1 linearizes all N
// samples of the tile once and parks (f(x), Gaussian weight -- -inf when the sample is outside [lo, hi] --[, LUT
// coordinate, linearized std]) in LDS; phase 2 evaluates the pairs out of LDS.  Integer stacks at full range with a
// whole-step LINEAR curve and no uncertainties are staged in the CODE DOMAIN (code_domain_sample: interval by one
// round-down FMA on the float code, f = g[i] + slope (code - i step), mask as a code interval: ~12 instead of ~33
// instructions per staged sample); everything else keeps the reference's float32 operation order.
//   forward       : thread <-> pair(s); each thread walks the tile's columns and keeps its pairs' sums in registers
//                   (float32 inside a tile, float64 across tiles), one float64 atomic per sum per workgroup at the end.
//   backward      : wavefront <-> sample i, lane <-> column; each pair is evaluated once from its first sample, the
//                   partner's share goes to a (sample, column) float64 accumulator in LDS (ds_add_f64), and after a
//                   barrier every (sample, column) is scattered into a (C, L) float64 histogram in LDS, flushed with
//                   float64 global atomics at the end.  Pair constants come from a global table through scalar loads.
//                   The uncertainty-weighted loss (weights depending on the LUT through err) uses the same kernel with the
//                   linearized stds staged as a fourth per-column array.
// Grids are sized to whole rounds of what the device holds at once (resident_workgroups, ct_device.hpp).
//
// Arithmetic: the reference computes the residual in float64 because the ratio is float64.  Here
// diff = I_i - I_j * r is formed with two float32 FMAs against r = r_hi + r_lo, which is exact to ~1 ulp of the
// (small) difference, so float32 carries the residual to ~1e-7 relative; sums are float64 across tiles.
//
// Roofline: float32 VALU (P pair evaluations per N loaded samples per pixel; P >> N), not HBM.
#include <algorithm>
#include <cstdlib>
#include "ct_device.hpp"

namespace ct {

#ifndef CT_ABLATE_PAIR_PHASE
#define CT_ABLATE_PAIR_PHASE 0  // 1 (tools/pairs_bench only): skip the pair loops, time staging + scatter + barriers alone
#endif

struct PairArgs {
    const void *stack;
    const float *std_stack;
    const float *lut;
    const int32_t *i_idx, *j_idx;
    const double *ratio;
    double *sums;          // (P, C, 5)
    const double *center;  // (P, C) or NULL: sum 2 accumulates (v - center)^2 w m (second pass of the std)
    // backward only
    const int32_t *part_off, *part_sample, *part_pair;  // CSR partner lists per sample
    const double *coef;                                 // (P, C)
    const double *smean;                                // (P, C) spatial means (uncertainty-weighted backward)
    double *lut_grad;                                   // (C, L) float64, +=
    const int32_t *first_g;                             // workspace: N + 1 offsets of the i-side entries per sample, then C flags
    const void *table_g;                                // workspace: (C, P) OnceEntry, grouped by sample i
    const void *lane_table_g;                           // workspace: (C, band, 64) LaneEntry or NULL (lane <-> sample backward)
    int32_t lane_band;                                  // max j - i the caller promises (0: generic backward only)
    int32_t lane_lds_bytes;                             // dynamic LDS size of the lane kernel's launch
    int64_t image_stride;
    TileMap tile;
    uint32_t plane_local;
    int32_t n_images, n_pairs, channels, n_points;
    int32_t tp;           // pixels per tile (power of two, divides the workgroup size)
    int32_t tp_shift;     // log2(tp)
    int32_t vec;          // 1: planes are 4-element aligned -> vectorised, prefetching staging with permuted columns
    int32_t val_offset;   // backward: byte offset of the per-tile arrays in LDS (after LUT, histogram[, entries, splits])
    int32_t row_pitch;    // LDS row pitch in entries (tp + pad)
    int32_t pair_begin;   // first pair handled by this launch (forward)
    NormConst norm;
    float lower, upper, neg_scale_log2e, std_value;
    int32_t use_relative, use_unc_weight;
    // Code-domain staging (integer stacks at full range, LINEAR, no uncertainties, LUT step a whole number of codes;
    // fill_code_domain): everything a sample needs is formed from the raw code as a float -- no normalisation, no
    // float LUT coordinate, no floor / float -> int chain.
    int32_t code_domain;
    float code_rcp;        // 1 / step rounded up: interval = floor(code / step) by one round-down FMA (ct_device.hpp)
    float code_step, code_inv_step;
    float code_lo, code_hi;          // smallest / largest code whose normalised value lies in [lower, upper]
    float code_dk_mul, code_dk_add;  // Gaussian weight = exp2(-dk^2), dk = code * mul + add
};

template <typename T>
__device__ __forceinline__ float load_pixel(const void *base, int64_t idx, NormConst nc)
{
    return to_pixel<T>(static_cast<const T *>(base)[idx], nc);
}

// Where channel plane c of pixel `pixel` (index inside the tile's plane) lives in one image, and the distance to the next
// pixel of the same plane: planar stacks c * plane + pixel / 1, interleaved (H, W, C) stacks pixel * C + c' / C with
// c' = c (RGB order) or C - 1 - c (BGR, what OpenCV decodes to; general_functions.py:315-335 is folded in here).
__device__ __forceinline__ uint32_t sample_index(const PairArgs &a, int c, uint32_t pixel, uint32_t &pixel_stride)
{
    if (a.tile.layout == CT_LAYOUT_NCHW) {
        pixel_stride = 1u;
        return (uint32_t)c * a.plane_local + pixel;
    }
    pixel_stride = (uint32_t)a.channels;
    return pixel * (uint32_t)a.channels + (uint32_t)(a.tile.layout == CT_LAYOUT_NHWC_BGR ? a.channels - 1 - c : c);
}

// element e of a packed raw vector as a float: one conversion straight from the packed dword (no unpacking)
__device__ __forceinline__ float raw_code_as_float(uint32_t v, int e) { return (float)((v >> (8 * e)) & 0xffu); }  // v_cvt_f32_ubyteN
__device__ __forceinline__ float raw_code_as_float(uint2 v, int e)
{
    const uint32_t dw = e < 2 ? v.x : v.y;
    float r;
    if (e & 1)
        asm("v_cvt_f32_u32_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(r) : "v"(dw));
    else
        asm("v_cvt_f32_u32_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0" : "=v"(r) : "v"(dw));
    return r;
}
__device__ __forceinline__ float raw_code_as_float(float4, int) { return 0.0f; }  // never used: float stacks are not code-domain

// One sample of the code-domain staging: t = 1.5 * 2^23 + interval (floor_index_bits), px = the code as a float.
//   f = g[i] + slope * (px - i * step)   offset exact, ONE rounding in the FMA: closer to the exact interpolant than the
//                                        reference's own float32 order g0 * (1 - w) + g1 * w (three roundings), and equal
//                                        to it bit for bit when step == 1
//   gauss = exp2(-(px * mul + add)^2), -inf outside [code_lo, code_hi] (the reference's mask on the normalised value,
//           translated to codes on the host with the reference's own float32 division)
//   coordinate (backward) = i + (px - i * step) / step
template <bool WANT_COORD>
__device__ __forceinline__ float2 code_domain_sample(const PairArgs &a, const char *lut_lds, uint32_t row_constant, float t,
                                                     float magic, float px, float &coord)
{
    const float2 g = *reinterpret_cast<const float2 *>(lut_lds + lds_entry_address(t, row_constant));
    const float i0f = t - magic;                                // exact
    const float off = __builtin_fmaf(i0f, -a.code_step, px);    // exact: an integer below step
    const float lin = __builtin_fmaf(g.y, off, g.x);
    const float dk = __builtin_fmaf(px, a.code_dk_mul, a.code_dk_add);
    const float gw = __builtin_amdgcn_exp2f(-dk * dk);
    const bool valid = __builtin_amdgcn_fmed3f(px, a.code_lo, a.code_hi) == px;
    if constexpr (WANT_COORD) coord = __builtin_fmaf(off, a.code_inv_step, i0f);
    return make_float2(lin, valid ? gw : -INFINITY);
}

// Phase 1 shared by both kernels: linearize every sample of the tile into LDS.
//   val[n*pitch + px] = (f(x), gauss(x))   gauss = -inf when x is outside [lo, hi]: the pair weight g_i + g_j is then
//                                          -inf as well and max(., 0) applies the pair mask in one instruction
//   aux[n*pitch + px] = LUT coordinate s (backward) or linearized std |f'(x) sigma| (forward, STD != none)
// The tile width is a power of two that divides the workgroup size (pick_tile), so a thread keeps one pixel column
// for the whole tile and walks the samples: pixel index, global index and LUT row are computed once per tile.
template <typename T, int INTERP, int STD, bool WANT_COORD>
__device__ __forceinline__ void stage_tile(const PairArgs &a, const char *lut_lds, float2 *val, float *aux, int c,
                                           uint32_t pix0, int npix, int block, float *aux2 = nullptr)
{
    constexpr bool kRanged = sizeof(T) != 4;
    constexpr int kEntry = lut_entry_bytes(INTERP);
    const int N = a.n_images, L = a.n_points;
    const float top = INTERP == CT_INTERP_NONE ? 1.0f : (float)(L - 1);
    const int px = (int)threadIdx.x & (a.tp - 1);
    const int n0 = (int)threadIdx.x >> a.tp_shift, nstep = block >> a.tp_shift;  // block = workgroup size (constant)
    const bool inb = px < npix;
    const uint32_t ql = (uint32_t)c * a.plane_local + pix0 + (uint32_t)px;
    const uint32_t qg = ql + (uint32_t)c * a.tile.chan_skip + a.tile.base;  // TileMap::locate with the channel known
    const char *row = lut_lds + (inb ? lut_row<INTERP>(qg, c, a.channels) : 0) * L * kEntry;
    // samples are walked in batches of kBatch with all of a batch's HBM loads issued before the first is used
    constexpr int kBatch = 4;
    uint32_t pstride;
    const uint32_t qm = sample_index(a, c, pix0 + (uint32_t)px, pstride);  // == ql for planar stacks
    const T *src = static_cast<const T *>(a.stack) + qm;
    if constexpr (sizeof(T) != 4 && INTERP == CT_INTERP_LINEAR && STD == CT_STD_NONE) {
        if (a.code_domain) {
            const uint32_t rowc = lds_row_constant((inb ? lut_row<INTERP>(qg, c, a.channels) : 0) * L * kEntry);
            float magic = kFloorMagic;
            asm volatile("" : "+v"(magic));
            for (int nb = n0; nb < N; nb += kBatch * nstep) {
                T raw[kBatch];
#pragma unroll
                for (int k = 0; k < kBatch; ++k) {
                    const int n = nb + k * nstep;
                    raw[k] = (inb && n < N) ? src[(int64_t)n * a.image_stride] : T(0);
                }
#pragma unroll
                for (int k = 0; k < kBatch; ++k) {
                    const int n = nb + k * nstep;
                    if (n >= N) break;
                    float2 v = make_float2(0.0f, -INFINITY);
                    float ax = 0.0f;
                    if (inb) {
                        const float pxv[1] = {(float)raw[k]};
                        float tf[1];
                        floor_index_bits<1>(pxv, a.code_rcp, magic, tf);
                        v = code_domain_sample<WANT_COORD>(a, lut_lds, rowc, tf[0], magic, pxv[0], ax);
                    }
                    val[n * a.row_pitch + px] = v;
                    if constexpr (WANT_COORD) aux[n * a.row_pitch + px] = ax;
                }
            }
            return;
        }
    }
    for (int nb = n0; nb < N; nb += kBatch * nstep) {
        T raw[kBatch];
        float sraw[kBatch];
#pragma unroll
        for (int k = 0; k < kBatch; ++k) {
            const int n = nb + k * nstep;
            const bool live = inb && n < N;
            raw[k] = live ? src[(int64_t)n * a.image_stride] : T(0);
            if constexpr (STD == CT_STD_EXPLICIT) sraw[k] = live ? a.std_stack[(int64_t)n * a.image_stride + qm] : 0.0f;
        }
#pragma unroll
        for (int k = 0; k < kBatch; ++k) {
            const int n = nb + k * nstep;
            if (n >= N) break;
            float2 v = make_float2(0.0f, -INFINITY);
            float ax = 0.0f, ax2 = 0.0f;
            if (inb) {
                const float x = to_pixel<T>(raw[k], a.norm);
                float dfdx;
                const float lin = icrf_sample<INTERP, true, kRanged>(x, row, top, dfdx);
                const float d = x - 0.5f;
                const float gw = __builtin_amdgcn_exp2f((d * d) * a.neg_scale_log2e);
                const bool valid = x >= a.lower && x <= a.upper;
                v = make_float2(lin, valid ? gw : -INFINITY);
                float lsd = 0.0f;
                if constexpr (STD != CT_STD_NONE) {
                    float sigma = a.std_value;
                    if constexpr (STD == CT_STD_MULTIPLIER) sigma = x * a.std_value;
                    if constexpr (STD == CT_STD_EXPLICIT) sigma = sraw[k];
                    lsd = fabsf(dfdx * sigma);  // icrf_training.py:117-126: |grads * stds|
                }
                if constexpr (WANT_COORD) {
                    ax = fminf(fmaxf(x * top, 0.0f), top);
                    ax2 = lsd;
                } else {
                    ax = lsd;
                }
            }
            val[n * a.row_pitch + px] = v;
            if constexpr (WANT_COORD || STD != CT_STD_NONE) aux[n * a.row_pitch + px] = ax;
            if constexpr (WANT_COORD && STD != CT_STD_NONE) aux2[n * a.row_pitch + px] = ax2;
        }
    }
}

// ---- vectorised, prefetching variant of phase 1 -----------------------------------------------------------
// Used when every image plane is 4-element aligned (host: PairArgs::vec).  A thread owns FOUR consecutive pixels of one
// sample per pass and fetches them with one 4/8/16-byte load; the loads of the NEXT tile are issued before the pair
// phase of the current one (issue) and consumed after it (commit), so HBM latency hides behind the pair arithmetic.
// Tile columns are permuted so the LDS stores stay conflict-free: pixel 4*g + e lives in column e*(tp/4) + g.
// Consumers treat columns as opaque; pixel_of_column() recovers the pixel where it matters (LUT row, bounds).
template <typename T> struct RawVec;
template <> struct RawVec<uint8_t> { using type = uint32_t; };
template <> struct RawVec<uint16_t> { using type = uint2; };
template <> struct RawVec<float> { using type = float4; };

__device__ __forceinline__ uint8_t raw_elem(uint32_t v, int e) { return (uint8_t)(v >> (8 * e)); }
__device__ __forceinline__ uint16_t raw_elem(uint2 v, int e) { return (uint16_t)((e < 2 ? v.x : v.y) >> (16 * (e & 1))); }
__device__ __forceinline__ float raw_elem(float4 v, int e) { return e == 0 ? v.x : (e == 1 ? v.y : (e == 2 ? v.z : v.w)); }

// four strided elements gathered into the packed form the vector load would have delivered (interleaved stacks)
__device__ __forceinline__ uint32_t pack_raw(uint8_t a, uint8_t b, uint8_t c, uint8_t d)
{
    return (uint32_t)a | ((uint32_t)b << 8) | ((uint32_t)c << 16) | ((uint32_t)d << 24);
}
__device__ __forceinline__ uint2 pack_raw(uint16_t a, uint16_t b, uint16_t c, uint16_t d)
{
    return make_uint2((uint32_t)a | ((uint32_t)b << 16), (uint32_t)c | ((uint32_t)d << 16));
}
__device__ __forceinline__ float4 pack_raw(float a, float b, float c, float d) { return make_float4(a, b, c, d); }

__device__ __forceinline__ int pixel_of_column(const PairArgs &a, int col)
{
    if (!a.vec) return col;
    const int gshift = a.tp_shift - 2;
    return ((col & ((1 << gshift) - 1)) << 2) + (col >> gshift);
}

// Workgroup barrier that orders LDS traffic only: __syncthreads() also drains vmcnt, which would stall on the
// prefetched HBM loads that are meant to stay in flight across the pair phase.
__device__ __forceinline__ void lds_barrier()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

template <typename T, int INTERP, int STD, bool WANT_COORD, int KB>
struct VecStager {
    using Vec = typename RawVec<T>::type;
    Vec raw[KB];
    float4 sraw[KB];

    __device__ __forceinline__ void issue(const PairArgs &a, int c, uint32_t pix0, int npix, int block)
    {
        const int gshift = a.tp_shift - 2;
        const int pg = (int)threadIdx.x & ((1 << gshift) - 1);
        const int n0 = (int)threadIdx.x >> gshift, nstep = block >> gshift;
        const bool inb = 4 * pg < npix;
        if (a.tile.layout != CT_LAYOUT_NCHW) {
            // Interleaved stack: the group's four pixels of plane c are C elements apart.  Four element loads rebuild the
            // packed vector; the other C - 1 planes' share of the same cache lines is read by those planes' workgroups
            // (channel-major grid: the launch streams the stack C times, out of HBM when it exceeds the caches -- the
            // pair phase, not the staging, bounds these kernels).
            uint32_t ps;
            const uint32_t qm = sample_index(a, c, pix0 + 4u * (uint32_t)pg, ps);
            const T *src = static_cast<const T *>(a.stack) + qm;
#pragma unroll
            for (int k = 0; k < KB; ++k) {
                const int n = n0 + k * nstep;
                if (inb && n < a.n_images) {
                    const T *p = src + (int64_t)n * a.image_stride;
                    raw[k] = pack_raw(p[0], p[ps], p[2u * ps], p[3u * ps]);
                    if constexpr (STD == CT_STD_EXPLICIT) {
                        const float *q = a.std_stack + (int64_t)n * a.image_stride + qm;
                        sraw[k] = make_float4(q[0], q[ps], q[2u * ps], q[3u * ps]);
                    }
                }
            }
            return;
        }
        const uint32_t ql = (uint32_t)c * a.plane_local + pix0 + 4u * (uint32_t)pg;
        const T *src = static_cast<const T *>(a.stack) + ql;
#pragma unroll
        for (int k = 0; k < KB; ++k) {
            const int n = n0 + k * nstep;
            if (inb && n < a.n_images) {
                raw[k] = *reinterpret_cast<const Vec *>(src + (int64_t)n * a.image_stride);
                if constexpr (STD == CT_STD_EXPLICIT)
                    sraw[k] = *reinterpret_cast<const float4 *>(a.std_stack + (int64_t)n * a.image_stride + ql);
            }
        }
    }

    __device__ __forceinline__ void commit(const PairArgs &a, const char *lut_lds, float2 *val, float *aux, float *aux2,
                                           int c, uint32_t pix0, int npix, int block)
    {
        constexpr bool kRanged = sizeof(T) != 4;
        constexpr int kEntry = lut_entry_bytes(INTERP);
        const int N = a.n_images, L = a.n_points;
        const float top = INTERP == CT_INTERP_NONE ? 1.0f : (float)(L - 1);
        const int gshift = a.tp_shift - 2, G = 1 << gshift;
        const int pg = (int)threadIdx.x & (G - 1);
        const int n0 = (int)threadIdx.x >> gshift, nstep = block >> gshift;
        const bool inb = 4 * pg < npix;
        const uint32_t ql = (uint32_t)c * a.plane_local + pix0 + 4u * (uint32_t)pg;
        const uint32_t qg = ql + (uint32_t)c * a.tile.chan_skip + a.tile.base;
        int row_off[4];
        {
            int r = lut_row<INTERP>(qg, c, a.channels);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                row_off[e] = r * L * kEntry;
                if constexpr (INTERP != CT_INTERP_LOOKUP) r = r + 1 == a.channels ? 0 : r + 1;
            }
        }
        if constexpr (sizeof(T) != 4 && INTERP == CT_INTERP_LINEAR && STD == CT_STD_NONE) {
            if (a.code_domain) {
                uint32_t rowc[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) rowc[e] = lds_row_constant(row_off[e]);
                float magic = kFloorMagic;
                asm volatile("" : "+v"(magic));
#pragma unroll
                for (int k = 0; k < KB; ++k) {
                    const int n = n0 + k * nstep;
                    if (n >= N) break;
                    float pxv[4], tf[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) pxv[e] = raw_code_as_float(raw[k], e);
                    floor_index_bits<4>(pxv, a.code_rcp, magic, tf);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float2 v = make_float2(0.0f, -INFINITY);
                        float ax = 0.0f;
                        if (inb) v = code_domain_sample<WANT_COORD>(a, lut_lds, rowc[e], tf[e], magic, pxv[e], ax);
                        const int at = n * a.row_pitch + e * G + pg;
                        val[at] = v;
                        if constexpr (WANT_COORD) aux[at] = ax;
                    }
                }
                return;
            }
        }
#pragma unroll
        for (int k = 0; k < KB; ++k) {
            const int n = n0 + k * nstep;
            if (n >= N) break;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float2 v = make_float2(0.0f, -INFINITY);
                float ax = 0.0f, ax2 = 0.0f;
                if (inb) {
                    const float x = to_pixel<T>(raw_elem(raw[k], e), a.norm);
                    float dfdx;
                    const float lin = icrf_sample<INTERP, true, kRanged>(x, lut_lds + row_off[e], top, dfdx);
                    const float d = x - 0.5f;
                    const float gw = __builtin_amdgcn_exp2f((d * d) * a.neg_scale_log2e);
                    const bool valid = x >= a.lower && x <= a.upper;
                    v = make_float2(lin, valid ? gw : -INFINITY);
                    float lsd = 0.0f;
                    if constexpr (STD != CT_STD_NONE) {
                        float sigma = a.std_value;
                        if constexpr (STD == CT_STD_MULTIPLIER) sigma = x * a.std_value;
                        if constexpr (STD == CT_STD_EXPLICIT) sigma = raw_elem(sraw[k], e);
                        lsd = fabsf(dfdx * sigma);  // icrf_training.py:117-126: |grads * stds|
                    }
                    if constexpr (WANT_COORD) {
                        ax = fminf(fmaxf(x * top, 0.0f), top);
                        ax2 = lsd;
                    } else {
                        ax = lsd;
                    }
                }
                const int at = n * a.row_pitch + e * G + pg;
                val[at] = v;
                if constexpr (WANT_COORD || STD != CT_STD_NONE) aux[at] = ax;
                if constexpr (WANT_COORD && STD != CT_STD_NONE) aux2[at] = ax2;
            }
        }
    }
};

// sign(d) in {-1, 0, +1} without compares: d * 2^127 is >= 2 in magnitude for every normal d, the median clamps it.
__device__ __forceinline__ float sign_of(float d)
{
    return __builtin_amdgcn_fmed3f(d * 0x1p127f, -1.0f, 1.0f);
}

// ---- forward -------------------------------------------------------------------------------------
// LEVEL 0: sums 0,1 (training loss).  LEVEL 1: all five sums (measure_linearity: std, error, count).
#ifndef CT_FWD_KERNEL_ATTR
#define CT_FWD_KERNEL_ATTR __attribute__((amdgpu_waves_per_eu(4, 4)))
#endif
template <typename T, int INTERP, int STD, int PPT, int LEVEL, bool REL>
__global__ __launch_bounds__(kBlock) CT_FWD_KERNEL_ATTR void pair_fwd_kernel(const PairArgs a)
{
    extern __shared__ __align__(16) char lds[];
    constexpr int kEntry = lut_entry_bytes(INTERP);
    constexpr int NS = LEVEL == 0 ? 2 : 5;
    const int C = a.channels, L = a.n_points, N = a.n_images;
    const int lut_bytes = INTERP == CT_INTERP_NONE ? 0 : ((C * L * kEntry + 15) & ~15);
    float2 *val = reinterpret_cast<float2 *>(lds + lut_bytes);
    float *aux = reinterpret_cast<float *>(val + (size_t)N * a.row_pitch);
    if (INTERP == CT_INTERP_LINEAR && a.code_domain)
        stage_lut_slope(lds, a.lut, C, L, a.code_step);
    else
        stage_lut<INTERP>(lds, a.lut, C, L);

    // my pairs
    int bi[PPT], bj[PPT];
    float rhi[PPT], rlo[PPT];
    bool live[PPT];
#pragma unroll
    for (int s = 0; s < PPT; ++s) {
        const int p = a.pair_begin + s * kBlock + (int)threadIdx.x;
        live[s] = p < a.n_pairs;
        const int pp = live[s] ? p : 0;
        bi[s] = a.i_idx[pp] * a.row_pitch;
        bj[s] = a.j_idx[pp] * a.row_pitch;
        const double r = a.ratio[pp];
        rhi[s] = (float)r;
        rlo[s] = (float)(r - (double)rhi[s]);
    }
    double acc[PPT][NS];
#pragma unroll
    for (int s = 0; s < PPT; ++s)
#pragma unroll
        for (int k = 0; k < NS; ++k) acc[s][k] = 0.0;

    const int c = (int)(blockIdx.x / (gridDim.x / (uint32_t)C));  // channel-major: a channel's workgroups are consecutive
    float cen[PPT];
#pragma unroll
    for (int s = 0; s < PPT; ++s) {
        const int p = a.pair_begin + s * kBlock + (int)threadIdx.x;
        cen[s] = (LEVEL == 1 && a.center && live[s]) ? (float)a.center[(int64_t)p * C + c] : 0.0f;
    }
    const uint32_t tiles = (a.plane_local + a.tp - 1) / a.tp;
    const uint32_t gstep = gridDim.x / C;
    // v_sqrt_f32 (1 ulp) instead of sqrtf's correctly rounded expansion (~12 instructions) for the residual's error:
    // the tolerance of these statistics is 1e-5; the uncertainty weight is switched by a 0/1 factor, not a branch
    const float unc_on = a.use_unc_weight ? 1.0f : 0.0f;
    VecStager<T, INTERP, STD, false, 8> stager;
    const uint32_t t_first = blockIdx.x % gstep;
    if (a.vec && t_first < tiles)
        stager.issue(a, c, t_first * a.tp, (int)min((uint32_t)a.tp, a.plane_local - t_first * a.tp), kBlock);
    for (uint32_t t = t_first; t < tiles; t += gstep) {
        const uint32_t pix0 = t * a.tp;
        const int npix = (int)min((uint32_t)a.tp, a.plane_local - pix0);
        lds_barrier();  // previous tile's readers are done (also orders stage_lut before first use)
        if (a.vec) {
            stager.commit(a, lds, val, aux, nullptr, c, pix0, npix, kBlock);
            const uint32_t tn = t + gstep;  // next tile's loads fly during this tile's pair phase
            if (tn < tiles) stager.issue(a, c, tn * a.tp, (int)min((uint32_t)a.tp, a.plane_local - tn * a.tp), kBlock);
        } else {
            stage_tile<T, INTERP, STD, false>(a, lds, val, aux, c, pix0, npix, kBlock);
        }
        lds_barrier();
        const int ncol = a.vec ? a.tp : npix;
#pragma unroll
        for (int s = 0; s < PPT; ++s) {
            if (!live[s]) continue;
            float f[NS];
#pragma unroll
            for (int k = 0; k < NS; ++k) f[k] = 0.0f;
#ifdef CT_ABLATE_FWD_ONE_READ  // tools/pairs_bench only: both operands from one LDS read (wrong sums, timing only)
            const float2 *vi = val + bi[s], *vj = vi;
#else
            const float2 *vi = val + bi[s], *vj = val + bj[s];
#endif
            const float *xi = aux + bi[s], *xj = aux + bj[s];
            // partial tiles: padding columns carry weight -inf and contribute nothing (their order is permuted when a.vec)
#ifndef CT_FWD_UNROLL
#define CT_FWD_UNROLL 8
#endif
#pragma unroll CT_FWD_UNROLL
            for (int px = 0; px < (CT_ABLATE_PAIR_PHASE ? 0 : ncol); ++px) {
                const float2 A = vi[px], Bv = vj[px];
                // expected = I_j * r, diff = I_i - expected (losses.py:41-43), compensated in float32
                const float d1 = __builtin_fmaf(-Bv.x, rhi[s], A.x);
                float diff = __builtin_fmaf(-Bv.x, rlo[s], d1);
                float inv_es = 0.0f;  // 1 / (expected + 1e-6), losses.py:45-47
                if constexpr (REL) {
                    inv_es = __builtin_amdgcn_rcpf(__builtin_fmaf(Bv.x, rhi[s], 1e-6f));
                    diff *= inv_es;
                }
                float wt = A.y + Bv.y;  // Gaussian pair weight (losses.py:231-234); -inf unless both samples are valid
                float err = 0.0f;
                if constexpr (STD != CT_STD_NONE) {
                    const float si = xi[px], sj = xj[px];
                    if constexpr (REL) {  // losses.py:52-59
                        const float ijs = fmaxf(Bv.x, 1e-6f);
                        const float t1 = si * inv_es;
                        const float t2 = (A.x * sj) * inv_es * __builtin_amdgcn_rcpf(ijs);
                        err = __builtin_amdgcn_sqrtf(__builtin_fmaf(t1, t1, __builtin_fmaf(t2, t2, 1e-6f)));
                    } else {  // losses.py:61
                        const float rs = rhi[s] * sj;
                        err = __builtin_amdgcn_sqrtf(__builtin_fmaf(si, si, rs * rs));
                    }
                    wt = __builtin_fmaf(unc_on, __builtin_amdgcn_rcpf(err + 1e-6f), wt);  // losses.py:96 (unc_on is 0 or 1)
                }
                const float wm = fmaxf(wt, 0.0f);
                f[0] += wm;
                f[1] = __builtin_fmaf(fabsf(diff), wm, f[1]);
                if constexpr (LEVEL == 1) {
                    const bool m = wt >= 0.0f;
                    const float dvc = fabsf(diff) - cen[s];
                    f[2] = __builtin_fmaf(dvc * dvc, wm, f[2]);
                    f[3] += m ? err : 0.0f;
                    f[4] += m ? 1.0f : 0.0f;
                }
            }
#pragma unroll
            for (int k = 0; k < NS; ++k) acc[s][k] += (double)f[k];
        }
    }
#pragma unroll
    for (int s = 0; s < PPT; ++s) {
        if (!live[s]) continue;
        const int p = a.pair_begin + s * kBlock + (int)threadIdx.x;
        double *o = a.sums + ((int64_t)p * C + c) * 5;
#pragma unroll
        for (int k = 0; k < NS; ++k)
            if (acc[s][k] != 0.0) atomicAdd(&o[k], acc[s][k]);
    }
}

// ---- backward ------------------------------------------------------------------------------------
// coef[p][c] = dL/d(spatial mean_pc) / max(sum w m, 1e-8).  Without uncertainty weighting the weights do not depend
// on the LUT and d mean / d I = w m / den * d v / d I.  With it (STD != CT_STD_NONE: weights w = 1/(err + 1e-6) + gauss,
// losses.py:93-100) the weights depend on the LUT through err when the loss is relative:
//   d mean = sum m [w dv + (v - mean) dw] / D.
// Every pair is evaluated ONCE, from its first sample (wavefront <-> sample i, lane <-> tile column): the shared part
// of the evaluation (residual, 1/es, mask, sign, err) serves both sides.  The i-side term stays in a register; the
// j-side term goes to a per-tile (sample, column) float64 accumulator in LDS with ds_add_f64 on lane-linear addresses
// (~9 cycles per wave-instruction; the float32 LDS atomic costs ~190 for ANY address pattern on gfx950, which is why
// an earlier float32 attempt at this design lost to evaluating every pair from both sides; tools/lds_atomic_rates.hip).
// After a barrier every (sample, column) entry is scattered into the (C, L) float64 histogram.
// 1024 threads (16 waves) per workgroup: at N = 64 the per-tile arrays take ~100 KB of LDS, so one workgroup fits a
// CU and it has to bring all the wavefronts the SIMDs get.
constexpr int kBwdBlock = 1024;

// Grid of the tile-walking pair kernels: every channel gets as many workgroups as the device holds at once, so the
// launch runs as exactly C full rounds of equal work (measured on the backward: 7.7 ms against 9.6 ms with a grid
// of 1026 on 256 one-slot units).
static int workgroups_per_channel(size_t lds_bytes, int block, uint32_t tiles)
{
    return (int)std::min<uint32_t>(tiles, (uint32_t)resident_workgroups(lds_bytes, block));
}

// The same with the residency the runtime reports for the actual kernel (registers count too: the forward kernel's
// LDS would admit four workgroups per CU where its VGPRs admit three, and a grid of four per CU then runs as one full
// round plus a one-third-occupied second one).
template <typename KernelT>
static int workgroups_per_channel(KernelT kernel, size_t lds_bytes, int block, uint32_t tiles)
{
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, block, lds_bytes) != hipSuccess || per_cu < 1)
        return workgroups_per_channel(lds_bytes, block, tiles);
    return (int)std::min<uint32_t>(tiles, (uint32_t)(compute_units() * per_cu));
}

// Entry of the pair-once backward's global partner table: 32 bytes, read with one s_load_dwordx8 (uniform index,
// constant address space) so every field arrives in an SGPR.
struct OnceEntry {
    int row;         // byte offset of the partner's row in val[] and gacc[]
    float rhi, rlo;  // exposure ratio split in two floats
    float cf;        // upstream coefficient of the pair for this channel (i-side factor)
    float cfr;       // -cf * r (j-side factor)
    float sm;        // spatial mean of the pair for this channel (uncertainty-weighted backward)
    int pad[2];
};
typedef int32_t Words8 __attribute__((ext_vector_type(8)));
typedef const Words8 __attribute__((address_space(4))) *ConstWords;
__device__ __forceinline__ OnceEntry load_entry(ConstWords table, int k)
{
    const Words8 w = table[k];
    OnceEntry pe;
    pe.row = w.s0;
    pe.rhi = __int_as_float(w.s1);
    pe.rlo = __int_as_float(w.s2);
    pe.cf = __int_as_float(w.s3);
    pe.cfr = __int_as_float(w.s4);
    pe.sm = __int_as_float(w.s5);
    return pe;
}

// One pair, both sides.  Gi accumulates the own (i) side; gj is the partner's (j) side term.
//   u_i = sign(q) w m dq/dI_i with q the (relative) residual = sign(diff) w m / |es|; sign(diff) * t is formed as
//   clamp(diff * 2^127, -t, t): exact whenever |diff| * 2^127 >= t, and 0 at diff == 0.
//   u_j = u_i (I_i + eps) / es; times -r cf it is the j-side term.
template <bool REL>
__device__ __forceinline__ void once_term(const OnceEntry &pe, float2 own, float2 oth, float &Gi, float &gj)
{
    const float Ii = own.x, Ij = oth.x;
    const float d1 = __builtin_fmaf(-Ij, pe.rhi, Ii);
    const float diff = __builtin_fmaf(-Ij, pe.rlo, d1);
    const float wm = fmaxf(own.y + oth.y, 0.0f);  // -inf encoded mask
    float ti = wm, qj = 1.0f;
    if constexpr (REL) {
        const float inv_es = __builtin_amdgcn_rcpf(__builtin_fmaf(Ij, pe.rhi, 1e-6f));
        ti = wm * fabsf(inv_es);
        qj = (Ii + 1e-6f) * inv_es;
    }
    const float ui = __builtin_amdgcn_fmed3f(diff * 0x1p127f, -ti, ti);
    const float uj = REL ? ui * qj : ui;
    Gi = __builtin_fmaf(pe.cf, ui, Gi);
    gj = pe.cfr * uj;
}

// The same for the uncertainty-weighted loss: w = gauss + 1/(err + 1e-6), err from the linearized stds s_i, s_j
// (losses.py:50-63); relative loss: d err / d I_i goes through t2 only, d err / d I_j through (e + eps) in both terms
// and through clamp(I_j, eps) in t2.  Both sides share 1/es, 1/clamp(I_j), err, 1/err and the (v - mean) dw factor.
template <bool REL>
__device__ __forceinline__ void once_term_unc(const OnceEntry &pe, float2 own, float own_sd, float2 oth, float oth_sd,
                                              float &Gi, float &gj)
{
    const float Ii = own.x, Ij = oth.x, si = own_sd, sj = oth_sd;
    const float d1 = __builtin_fmaf(-Ij, pe.rhi, Ii);
    const float diff = __builtin_fmaf(-Ij, pe.rlo, d1);
    const float ws = own.y + oth.y;  // Gaussian pair weight, -inf unless both samples are valid
    float wt = fmaxf(ws, 0.0f);      // finite even when masked
    float dvi, dvj, ei = 0.0f, ej = 0.0f;
    if constexpr (REL) {
        const float inv_es = __builtin_amdgcn_rcpf(__builtin_fmaf(Ij, pe.rhi, 1e-6f));
        const float q = diff * inv_es;
        const float sg = sign_of(q);
        dvi = sg * inv_es;
        dvj = -sg * pe.rhi * (Ii + 1e-6f) * inv_es * inv_es;
        const float inv_ijs = __builtin_amdgcn_rcpf(fmaxf(Ij, 1e-6f));
        const float t1 = si * inv_es, u2 = sj * inv_es * inv_ijs, t2 = Ii * u2;  // losses.py:55-57
        const float err = __builtin_amdgcn_sqrtf(__builtin_fmaf(t1, t1, __builtin_fmaf(t2, t2, 1e-6f)));
        const float uw = __builtin_amdgcn_rcpf(err + 1e-6f);
        wt += uw;
        const float rinv = pe.rhi * inv_es;
        const float derr_i = t2 * u2;
        const float derr_j = -(t1 * t1 * rinv + t2 * t2 * (rinv + (Ij >= 1e-6f ? inv_ijs : 0.0f)));
        const float c1 = (fabsf(q) - pe.sm) * (-uw * uw) * __builtin_amdgcn_rcpf(err);  // (v - mean) dw/derr / err
        ei = c1 * derr_i;
        ej = c1 * derr_j;
    } else {  // losses.py:61: the error does not depend on the LUT
        const float sg = sign_of(diff);
        dvi = sg;
        dvj = -sg * pe.rhi;
        const float rs = pe.rhi * sj;
        wt += __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(__builtin_fmaf(si, si, rs * rs)) + 1e-6f);
    }
    const float cfm = ws >= 0.0f ? pe.cf : 0.0f;
    Gi = __builtin_fmaf(cfm, __builtin_fmaf(wt, dvi, ei), Gi);
    gj = cfm * __builtin_fmaf(wt, dvj, ej);
}

// Fills the workspace of the pair-once backward: first[N + 1], active[C] (does the channel have any non-zero upstream
// coefficient -- loss[c].backward() of the reference's per-channel loop touches one channel at a time, and the main
// kernel's workgroups of the other channels return at once) and, per channel, the i-side partner entries grouped by
// sample (one workgroup per channel).  The entries are wavefront-uniform in the main kernel, which therefore reads
// them with scalar loads (SGPR operands, no LDS traffic, no VALU moves).
__global__ __launch_bounds__(256) void pair_entries_kernel(const PairArgs a, int32_t *first, OnceEntry *table, float4 *lane_table)
{
    __shared__ int sfirst[1025];
    const int N = a.n_images, C = a.channels, c = blockIdx.x;
    for (int n = threadIdx.x; n < N; n += blockDim.x) {  // i-side entry count of sample n ...
        int cnt = 0;
        for (int e = a.part_off[n]; e < a.part_off[n + 1]; ++e) cnt += a.part_pair[e] >= 0 ? 1 : 0;
        sfirst[n + 1] = cnt;
    }
    __syncthreads();
    if (threadIdx.x == 0) {  // ... turned into offsets (N <= 1024: a serial scan over LDS is a few microseconds)
        int run = 0;
        sfirst[0] = 0;
        for (int n = 1; n <= N; ++n) {
            run += sfirst[n];
            sfirst[n] = run;
        }
    }
    __syncthreads();
    if (c == 0)
        for (int n = threadIdx.x; n <= N; n += blockDim.x) first[n] = sfirst[n];
    OnceEntry *tab = table + (size_t)c * a.n_pairs;
    // lane <-> sample backward: entry (d - 1, i) of the channel's (band, 64) table holds pair (i, i + d); absent pairs
    // stay zero and contribute nothing.  Any pair outside 0 < j - i <= band (or listed twice) invalidates the table and
    // the generic kernel runs instead.
    float4 *lt = lane_table ? lane_table + (size_t)c * a.lane_band * 64 : nullptr;
    if (lt) {
        for (int k = threadIdx.x; k < a.lane_band * 64; k += blockDim.x) lt[k] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        __syncthreads();
    }
    int nonzero = 0, bad = 0;
    for (int n = threadIdx.x; n < N; n += blockDim.x) {
        int at = sfirst[n];
        unsigned long long seen = 0;
        for (int e = a.part_off[n]; e < a.part_off[n + 1]; ++e) {
            const int p = a.part_pair[e];
            if (p < 0) continue;
            const double r = a.ratio[p];
            OnceEntry pe{};
            pe.row = a.part_sample[e] * a.row_pitch * 8;
            pe.rhi = (float)r;
            pe.rlo = (float)(r - (double)pe.rhi);
            pe.cf = (float)a.coef[(int64_t)p * C + c];
            pe.cfr = -pe.cf * pe.rhi;
            pe.sm = a.smean ? (float)a.smean[(int64_t)p * C + c] : 0.0f;
            nonzero |= pe.cf != 0.0f ? 1 : 0;
            tab[at++] = pe;
            if (lt) {
                const int d = a.part_sample[e] - n;
                if (N > 64 || d < 1 || d > a.lane_band || ((seen >> d) & 1ull)) {
                    bad = 1;
                } else {
                    seen |= 1ull << d;
                    lt[(d - 1) * 64 + n] = make_float4(pe.rhi, pe.rlo, pe.cf, pe.cfr);
                }
            }
        }
    }
    const int any = __syncthreads_or(nonzero);
    const int any_bad = __syncthreads_or(bad);
    if (threadIdx.x == 0) {
        first[N + 1 + c] = any;
        first[N + 1 + C + c] = (lt && !any_bad) ? 1 : 0;  // 1: the lane kernel handles this channel, the generic one returns
    }
}

// dL/dI of one (sample, pixel) spread over the LUT entries its interpolation read (transpose of the sampler):
// s = LUT coordinate of the sample, hrow = the pixel's row of the float64 histogram in LDS.
template <int INTERP>
__device__ __forceinline__ void scatter_lut_grad(double *hrow, float s, float Gk, int L)
{
    if constexpr (INTERP == CT_INTERP_LOOKUP) {
        atomicAdd(&hrow[(int)rintf(s)], (double)Gk);
    } else {
        const float fl = floorf(s);
        const int i0 = (int)fl;
        const float tt = s - fl;
        if constexpr (INTERP == CT_INTERP_LINEAR) {
            const int i1 = i0 + 1 < L ? i0 + 1 : L - 1;
            atomicAdd(&hrow[i0], (double)(Gk * (1.0f - tt)));
            atomicAdd(&hrow[i1], (double)(Gk * tt));
        } else {
            const float t2 = tt * tt, t3 = t2 * tt;
            const float w0 = -0.5f * t3 + t2 - 0.5f * tt, w1 = 1.5f * t3 - 2.5f * t2 + 1.0f;
            const float w2 = -1.5f * t3 + 2.0f * t2 + 0.5f * tt, w3 = 0.5f * t3 - 0.5f * t2;
            const int im = i0 > 0 ? i0 - 1 : 0, i1 = i0 + 1 < L ? i0 + 1 : L - 1, i2 = i0 + 2 < L ? i0 + 2 : L - 1;
            atomicAdd(&hrow[im], (double)(Gk * w0));
            atomicAdd(&hrow[i0], (double)(Gk * w1));
            atomicAdd(&hrow[i1], (double)(Gk * w2));
            atomicAdd(&hrow[i2], (double)(Gk * w3));
        }
    }
}

template <typename T, int INTERP, bool REL, int STD>
__global__ __launch_bounds__(kBwdBlock) void pair_bwd_once_kernel(const PairArgs a)
{
    extern __shared__ __align__(16) char lds[];
    constexpr int kEntry = lut_entry_bytes(INTERP);
    const int C = a.channels, L = a.n_points, N = a.n_images;
    const int lut_bytes = INTERP == CT_INTERP_NONE ? 0 : ((C * L * kEntry + 15) & ~15);
    double *hist64 = reinterpret_cast<double *>(lds + lut_bytes);
    double *gacc = reinterpret_cast<double *>(lds + a.val_offset);         // (N, row_pitch) dL/dI accumulators
    float2 *val = reinterpret_cast<float2 *>(gacc + (size_t)N * a.row_pitch);
    constexpr bool kUnc = STD != CT_STD_NONE;
    float *aux = reinterpret_cast<float *>(val + (size_t)N * a.row_pitch);
    float *lsdv = aux + (size_t)N * a.row_pitch;  // linearized std per (sample, column), uncertainty weighting only
    int *colrow = reinterpret_cast<int *>(lsdv + (kUnc ? (size_t)N * a.row_pitch : 0));  // histogram row offset per column
    const int c = (int)(blockIdx.x / (gridDim.x / (uint32_t)C));  // channel-major: a channel's workgroups are consecutive
    // constant address space + uniform index = scalar loads (s_load_dwordx4 into SGPRs); the tables were written by
    // the preceding launch and are read-only here
    typedef const int32_t __attribute__((address_space(4))) *ConstInts;
    ConstInts first = (ConstInts)(uintptr_t)a.first_g;
    ConstWords ent = (ConstWords)(uintptr_t)(static_cast<const OnceEntry *>(a.table_g) + (size_t)c * a.n_pairs);
    if (first[N + 1 + c] == 0) return;  // no upstream gradient for this channel (uniform: the whole workgroup leaves)
    if (first[N + 1 + C + c] != 0) return;  // the lane <-> sample kernel of this launch sequence handles the channel
    if (INTERP == CT_INTERP_LINEAR && a.code_domain)
        stage_lut_slope(lds, a.lut, C, L, a.code_step);
    else
        stage_lut<INTERP>(lds, a.lut, C, L);
    for (int k = threadIdx.x; k < C * L; k += blockDim.x) hist64[k] = 0.0;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int nwaves = kBwdBlock >> 6;
    const uint32_t tiles = (a.plane_local + a.tp - 1) / a.tp;
    const uint32_t gstep = gridDim.x / C;
    VecStager<T, INTERP, STD, true, 4> stager;
    const uint32_t t_first = blockIdx.x % gstep;
    if (a.vec && t_first < tiles)
        stager.issue(a, c, t_first * a.tp, (int)min((uint32_t)a.tp, a.plane_local - t_first * a.tp), kBwdBlock);
    const int col = min(lane, a.tp - 1);
    for (uint32_t t = t_first; t < tiles; t += gstep) {
        const uint32_t pix0 = t * a.tp;
        const int npix = (int)min((uint32_t)a.tp, a.plane_local - pix0);
        lds_barrier();  // the previous tile's scatter is done
        if (a.vec) {
            stager.commit(a, lds, val, aux, lsdv, c, pix0, npix, kBwdBlock);
            const uint32_t tn = t + gstep;
            if (tn < tiles) stager.issue(a, c, tn * a.tp, (int)min((uint32_t)a.tp, a.plane_local - tn * a.tp), kBwdBlock);
        } else {
            stage_tile<T, INTERP, STD, true>(a, lds, val, aux, c, pix0, npix, kBwdBlock, lsdv);
        }
        for (int k = threadIdx.x; k < N * a.row_pitch; k += blockDim.x) gacc[k] = 0.0;
        if ((int)threadIdx.x < a.tp) {  // the LUT row of a column depends on the pixel only (models/base.py:173-176)
            const uint32_t px = (uint32_t)pixel_of_column(a, (int)threadIdx.x);
            const uint32_t qg = (uint32_t)c * (a.plane_local + a.tile.chan_skip) + a.tile.base + pix0 + px;
            colrow[threadIdx.x] = lut_row<INTERP>(qg, c, C) * L;
        }
        lds_barrier();
        {   // uniform control flow throughout: with 32-column tiles the upper half-wave works on a masked copy
            const char *valb = reinterpret_cast<const char *>(val + col);
            const char *lsdb = reinterpret_cast<const char *>(lsdv + col);
            char *gaccb = reinterpret_cast<char *>(gacc + col);
            for (int n = wave; n < N; n += nwaves) {
                float2 own = val[n * a.row_pitch + col];
                const float own_sd = kUnc ? lsdv[n * a.row_pitch + col] : 0.0f;
                if (lane >= a.tp) own.y = -INFINITY;
                float Gi = 0.0f;
                const int e0 = first[n], e1 = first[n + 1];
                // groups of kGroup partners: all LDS reads first, then the arithmetic, then the atomics -- the compiler
                // cannot hoist reads over the ds_add_f64 of the previous partner, so the order is spelled out
                constexpr int kGroup = 4;
                int e = e0;
                for (; e + kGroup <= e1; e += kGroup) {
                    OnceEntry pe[kGroup];
                    float2 oth[kGroup];
                    float osd[kGroup], gj[kGroup];
#pragma unroll
                    for (int u = 0; u < kGroup; ++u) pe[u] = load_entry(ent, e + u);
#pragma unroll
                    for (int u = 0; u < kGroup; ++u) {
                        oth[u] = *reinterpret_cast<const float2 *>(valb + pe[u].row);
                        if constexpr (kUnc) osd[u] = *reinterpret_cast<const float *>(lsdb + (pe[u].row >> 1));
                    }
#pragma unroll
                    for (int u = 0; u < kGroup; ++u) {
                        if constexpr (kUnc)
                            once_term_unc<REL>(pe[u], own, own_sd, oth[u], osd[u], Gi, gj[u]);
                        else
                            once_term<REL>(pe[u], own, oth[u], Gi, gj[u]);
                    }
#pragma unroll
                    for (int u = 0; u < kGroup; ++u)
                        atomicAdd(reinterpret_cast<double *>(gaccb + pe[u].row), (double)gj[u]);
                }
                for (; e < e1; ++e) {
                    const OnceEntry pe = load_entry(ent, e);
                    const float2 oth = *reinterpret_cast<const float2 *>(valb + pe.row);
                    float gj;
                    if constexpr (kUnc)
                        once_term_unc<REL>(pe, own, own_sd, oth, *reinterpret_cast<const float *>(lsdb + (pe.row >> 1)), Gi, gj);
                    else
                        once_term<REL>(pe, own, oth, Gi, gj);
                    atomicAdd(reinterpret_cast<double *>(gaccb + pe.row), (double)gj);
                }
                atomicAdd(&gacc[n * a.row_pitch + col], (double)Gi);
            }
        }
        lds_barrier();
        // columns beyond the plane's end carry weight -inf, so their accumulators are exactly zero: no bounds test
        for (int k = threadIdx.x; k < N * a.tp; k += blockDim.x) {
            const int n = k >> a.tp_shift, colk = k & (a.tp - 1);
            const float Gk = (float)gacc[n * a.row_pitch + colk];
            if (Gk != 0.0f) scatter_lut_grad<INTERP>(hist64 + colrow[colk], aux[n * a.row_pitch + colk], Gk, L);
        }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < C * L; k += blockDim.x)
        if (hist64[k] != 0.0) atomicAdd(&a.lut_grad[k], hist64[k]);
}

// ---- backward, lane <-> sample variant -------------------------------------------------------------------
// For pair lists that form a band (every pair (i, j) has 0 < j - i <= band, N <= 64; get_valid_exposure_pairs with a
// ratio limit on a geometric exposure series is one): a wavefront takes a tile COLUMN, lane n holds sample n of that
// pixel, and the band is walked from the far end: at step d lane i evaluates pair (i, i + d) against the partner row
// i + d of the staged tile (one conflict-free ds_read_b64, prefetched a step ahead).  The partner's share of the
// gradient travels in a running register R that rotates by one lane per step (v_add_f32_dpp wave_ror:1): after step d,
// R in lane i is the sum destined for lane i + d, and one more rotation after d = 1 delivers it.  So there is no
// (sample, column) accumulator array, no LDS atomic and no third barrier in the pair phase, the per-pair constants are
// per-lane VGPR operands (one 16-byte LDS read per step, shared by kLaneCols columns) instead of SGPR operands, and the
// loop is 14 VALU instructions per lane-step (the generic kernel: 16 + address arithmetic + a float64 conversion, a
// third of them at the slower SGPR-operand rate).  At N = 64, L = 256 the workgroup needs 62 KB + band KB of LDS, so two
// 512-thread workgroups share a CU and one stages while the other computes.  Measured on C3: 5.40 ms against 7.71 ms.
// Rotation (not shift) keeps the bookkeeping consistent modulo 64; wrapped or absent pairs have all-zero constants.
constexpr int kLaneBlock = 512;
#ifndef CT_LANE_COLS
#define CT_LANE_COLS 4  // measured on C3: 4 columns 5.40 ms, 2: 5.78, 8: 5.80 (128 VGPRs, spills)
#endif
constexpr int kLaneCols = CT_LANE_COLS;

__device__ __forceinline__ float lane_rotate_up(float v)  // lane i receives the value of lane (i - 1) mod 64
{
    // old = the value itself: every lane has a source under a rotation, and a tied operand spares the compiler a v_mov
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x13C /* wave_ror:1 */, 0xf, 0xf, false));
}
__device__ __forceinline__ float lane_rotate_up_add(float v, float add)  // rotate_up(v) + add as ONE v_add_f32_dpp
{
    // old = 0 with bound_ctrl lets LLVM's DPP combiner fold the move into the consuming VOP2 add
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x13C, 0xf, 0xf, true)) + add;
}

template <typename T, int INTERP, bool REL>
__global__ __launch_bounds__(kLaneBlock) __attribute__((amdgpu_waves_per_eu(4, 4))) void pair_bwd_lane_kernel(const PairArgs a)
{
    extern __shared__ __align__(16) char lds[];
    constexpr int kEntry = lut_entry_bytes(INTERP);
    const int C = a.channels, L = a.n_points, N = a.n_images, band = a.lane_band;
    const int lut_bytes = INTERP == CT_INTERP_NONE ? 0 : ((C * L * kEntry + 15) & ~15);
    double *hist64 = reinterpret_cast<double *>(lds + lut_bytes);
    float4 *ktab = reinterpret_cast<float4 *>(lds + a.val_offset);  // (band, 64) pair constants of this channel
    float2 *val = reinterpret_cast<float2 *>(ktab + (size_t)band * 64);
    float *aux = reinterpret_cast<float *>(val + (size_t)N * a.row_pitch);
    int *colrow = reinterpret_cast<int *>(aux + (size_t)N * a.row_pitch);
    const int c = (int)(blockIdx.x / (gridDim.x / (uint32_t)C));
    typedef const int32_t __attribute__((address_space(4))) *ConstInts;
    ConstInts first = (ConstInts)(uintptr_t)a.first_g;
    if (first[N + 1 + c] == 0 || first[N + 1 + C + c] == 0) return;  // no gradient / the generic kernel has the channel
    if (INTERP == CT_INTERP_LINEAR && a.code_domain)
        stage_lut_slope(lds, a.lut, C, L, a.code_step);
    else
        stage_lut<INTERP>(lds, a.lut, C, L);
    for (int k = threadIdx.x; k < C * L; k += blockDim.x) hist64[k] = 0.0;
    {
        const float4 *src = static_cast<const float4 *>(a.lane_table_g) + (size_t)c * band * 64;
        for (int k = threadIdx.x; k < band * 64; k += blockDim.x) ktab[k] = src[k];
        // The pair phase reads val rows up to 63 + band; rows >= N alias the LUT coordinates, row padding and whatever
        // follows.  Those partners only ever meet all-zero constants, but 0 * NaN is NaN: everything the staging does not
        // rewrite each tile is zeroed once here (LUT coordinates are finite by construction).
        float *tile_words = reinterpret_cast<float *>(val);
        const int n_words = (int)((a.lane_lds_bytes - ((const char *)val - lds)) >> 2);
        for (int k = threadIdx.x; k < n_words; k += blockDim.x) tile_words[k] = 0.0f;
    }
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int nwaves = kLaneBlock >> 6;
    const uint32_t tiles = (a.plane_local + a.tp - 1) / a.tp;
    const uint32_t gstep = gridDim.x / C;
    VecStager<T, INTERP, CT_STD_NONE, true, 2> stager;
    const uint32_t t_first = blockIdx.x % gstep;
    if (a.vec && t_first < tiles)
        stager.issue(a, c, t_first * a.tp, (int)min((uint32_t)a.tp, a.plane_local - t_first * a.tp), kLaneBlock);
    const int own_row = min(lane, N - 1) * a.row_pitch;
    for (uint32_t t = t_first; t < tiles; t += gstep) {
        const uint32_t pix0 = t * a.tp;
        const int npix = (int)min((uint32_t)a.tp, a.plane_local - pix0);
        lds_barrier();  // the previous tile's pair phase has read val / aux / colrow
        if (a.vec) {
            stager.commit(a, lds, val, aux, nullptr, c, pix0, npix, kLaneBlock);
            const uint32_t tn = t + gstep;
            if (tn < tiles) stager.issue(a, c, tn * a.tp, (int)min((uint32_t)a.tp, a.plane_local - tn * a.tp), kLaneBlock);
        } else {
            stage_tile<T, INTERP, CT_STD_NONE, true>(a, lds, val, aux, c, pix0, npix, kLaneBlock);
        }
        if ((int)threadIdx.x < a.tp) {
            const uint32_t px = (uint32_t)pixel_of_column(a, (int)threadIdx.x);
            const uint32_t qg = (uint32_t)c * (a.plane_local + a.tile.chan_skip) + a.tile.base + pix0 + px;
            colrow[threadIdx.x] = lut_row<INTERP>(qg, c, C) * L;
        }
        lds_barrier();
        for (int col0 = wave * kLaneCols; col0 < a.tp; col0 += nwaves * kLaneCols) {
            float2 own[kLaneCols], oth[kLaneCols];
            float R[kLaneCols], Gi[kLaneCols];
            // partner of step d = row lane + d of the staged tile, read straight from LDS (conflict-free: the row pitch
            // is odd); rows N .. N + band - 1 alias the LUT-coordinate array behind val[] -- finite garbage that only
            // ever meets all-zero constants
            const float2 *partner = val + (lane + band) * a.row_pitch + col0;
#pragma unroll
            for (int k = 0; k < kLaneCols; ++k) {
                own[k] = val[own_row + col0 + k];
                oth[k] = partner[k];
                R[k] = 0.0f;
                Gi[k] = 0.0f;
            }
            // two steps per trip so that the prefetched partners alternate between two register sets without copies
            // (band is even: the host rounds it up and the extra step has all-zero constants)
            auto step = [&](int d, const float2 (&cur)[kLaneCols], float2 (&nxt)[kLaneCols]) {
                const float4 K = ktab[(d - 1) * 64 + lane];
                OnceEntry pe;
                pe.rhi = K.x;
                pe.rlo = K.y;
                pe.cf = K.z;
                pe.cfr = K.w;
                partner -= a.row_pitch;
#pragma unroll
                for (int k = 0; k < kLaneCols; ++k) nxt[k] = partner[k];  // step d - 1 (after d = 1: the own row, unused)
#pragma unroll
                for (int k = 0; k < kLaneCols; ++k) {
                    float gj;
                    once_term<REL>(pe, own[k], cur[k], Gi[k], gj);
                    R[k] = lane_rotate_up_add(R[k], gj);
                }
            };
            float2 alt[kLaneCols];
            for (int d = CT_ABLATE_PAIR_PHASE ? 0 : band; d >= 2; d -= 2) {
                step(d, oth, alt);
                step(d - 1, alt, oth);
            }
            // padding columns and rows >= N carry weight -inf or zero constants: their totals are exactly zero
#pragma unroll
            for (int k = 0; k < kLaneCols; ++k) {
                const float Gk = lane_rotate_up_add(R[k], Gi[k]);
                if (Gk != 0.0f) scatter_lut_grad<INTERP>(hist64 + colrow[col0 + k], aux[own_row + col0 + k], Gk, L);
            }
        }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < C * L; k += blockDim.x)
        if (hist64[k] != 0.0) atomicAdd(&a.lut_grad[k], hist64[k]);
}

// ---- host side -------------------------------------------------------------------------------------
static int pick_tile(int n_images, size_t fixed_bytes, int bytes_per_entry, int want)
{
    // largest tile (power of two, 32 <= tp <= want) whose staging fits beside the fixed LDS part in ~144 KiB
    const size_t budget = 144 * 1024;
    for (int tp = want; tp >= 32; tp /= 2) {
        const size_t need = fixed_bytes + (size_t)n_images * (tp + 1) * bytes_per_entry;
        if (need <= budget) return tp;
    }
    return 0;
}

// The vectorised staging needs every plane of every image to start on a 4-element boundary (and the base pointers
// aligned accordingly), and at most `max_passes` samples per thread.
template <typename T>
static int vec_ok(const PairArgs &a, int block, int max_passes)
{
    const int nstep = block / (a.tp / 4);
    // interleaved stacks are gathered element by element: only the plane size matters (groups of four pixels)
    const bool aligned = a.plane_local % 4 == 0 &&
                         (a.tile.layout != CT_LAYOUT_NCHW ||
                          (a.image_stride % 4 == 0 && reinterpret_cast<uintptr_t>(a.stack) % (4 * sizeof(T)) == 0 &&
                           (a.std_stack == nullptr || reinterpret_cast<uintptr_t>(a.std_stack) % 16 == 0)));
    return aligned && (a.n_images + nstep - 1) / nstep <= max_passes ? 1 : 0;
}

template <typename T, int INTERP, int STD, int PPT, int LEVEL, bool REL>
static int fwd_launch_one(const PairArgs &a, size_t lds, hipStream_t s)
{
    auto kernel = pair_fwd_kernel<T, INTERP, STD, PPT, LEVEL, REL>;
    const uint32_t tiles = (a.plane_local + a.tp - 1) / a.tp;
    const int grid = workgroups_per_channel(kernel, lds, kBlock, tiles) * a.channels;
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(kBlock), lds, s, a);
    return hipGetLastError() == hipSuccess ? CT_OK : CT_ERR_LAUNCH;
}

template <typename T, int INTERP, int STD, int PPT>
static int fwd_launch_level(const PairArgs &a, size_t lds, int level, hipStream_t s)
{
    if (level == 0)
        return a.use_relative ? fwd_launch_one<T, INTERP, STD, PPT, 0, true>(a, lds, s) : fwd_launch_one<T, INTERP, STD, PPT, 0, false>(a, lds, s);
    return a.use_relative ? fwd_launch_one<T, INTERP, STD, PPT, 1, true>(a, lds, s) : fwd_launch_one<T, INTERP, STD, PPT, 1, false>(a, lds, s);
}

template <typename T, int INTERP, int STD>
static int fwd_launch(PairArgs a, int level, hipStream_t s)
{
    const size_t lut_bytes = INTERP == CT_INTERP_NONE ? 0 : (((size_t)a.channels * a.n_points * lut_entry_bytes(INTERP) + 15) & ~(size_t)15);
    const int entry = STD == CT_STD_NONE ? 8 : 12;
    // 64-pixel tiles: ~40 KB of LDS at N = 64 -> four workgroups (16 waves) per CU hide the LDS latency of phase 2
    const int tp = pick_tile(a.n_images, lut_bytes, entry, 64);
    if (tp == 0) return CT_ERR_TOO_LARGE;
    a.tp = tp;
    a.tp_shift = tp == 64 ? 6 : 5;
    a.row_pitch = tp + 1;
    a.vec = vec_ok<T>(a, kBlock, 8);
    const size_t lds = lut_bytes + (size_t)a.n_images * a.row_pitch * entry;
    // pairs are walked in chunks of 4 * 256 per launch
    for (int begin = 0; begin < a.n_pairs; begin += 4 * kBlock) {
        a.pair_begin = begin;
        // always four pair slots per thread (slots past the end of the list are skipped): one instantiation instead of
        // three keeps the build time of this file in check
        const int rc = fwd_launch_level<T, INTERP, STD, 4>(a, lds, level, s);
        if (rc != CT_OK) return rc;
    }
    return CT_OK;
}

// CT_PAIRS_MINIMAL (tools/pairs_bench.hip only): instantiate just uint16 / LINEAR / no std so the harness builds fast.
template <typename T, int INTERP>
static int fwd_dispatch_std(const PairArgs &a, int std_mode, int level, hipStream_t s)
{
#ifdef CT_PAIRS_MINIMAL
    return std_mode == CT_STD_NONE ? fwd_launch<T, INTERP, CT_STD_NONE>(a, level, s) : CT_ERR_UNSUPPORTED;
#endif
    switch (std_mode) {
        case CT_STD_NONE: return fwd_launch<T, INTERP, CT_STD_NONE>(a, level, s);
        case CT_STD_CONSTANT: return fwd_launch<T, INTERP, CT_STD_CONSTANT>(a, level, s);
        case CT_STD_MULTIPLIER: return fwd_launch<T, INTERP, CT_STD_MULTIPLIER>(a, level, s);
        case CT_STD_EXPLICIT: return fwd_launch<T, INTERP, CT_STD_EXPLICIT>(a, level, s);
    }
    return CT_ERR_INVALID_ARGUMENT;
}

template <typename T>
static int fwd_dispatch(const PairArgs &a, int interp, int std_mode, int level, hipStream_t s)
{
#ifdef CT_PAIRS_MINIMAL
    return interp == CT_INTERP_LINEAR ? fwd_dispatch_std<T, CT_INTERP_LINEAR>(a, std_mode, level, s) : CT_ERR_UNSUPPORTED;
#endif
    switch (interp) {
        case CT_INTERP_LOOKUP: return fwd_dispatch_std<T, CT_INTERP_LOOKUP>(a, std_mode, level, s);
        case CT_INTERP_LINEAR: return fwd_dispatch_std<T, CT_INTERP_LINEAR>(a, std_mode, level, s);
        case CT_INTERP_CATMULL: return fwd_dispatch_std<T, CT_INTERP_CATMULL>(a, std_mode, level, s);
        case CT_INTERP_NONE: return fwd_dispatch_std<T, CT_INTERP_NONE>(a, std_mode, level, s);
    }
    return CT_ERR_INVALID_ARGUMENT;
}

// Workspace of the backward: [first: N + 1 offsets | C "channel has gradient" flags | C "lane kernel has the channel"
// flags] [(C, P) OnceEntry] [(C, 64, 64) float4 lane table, only for N <= 64]
static size_t once_header_bytes(int n_images, int channels)
{
    return ((size_t)(n_images + 1 + 2 * channels) * 4 + 31) & ~(size_t)31;
}
static size_t once_table_bytes(int n_pairs, int channels) { return (size_t)channels * n_pairs * sizeof(OnceEntry); }
static size_t once_workspace_bytes(int n_images, int n_pairs, int channels)
{
    return once_header_bytes(n_images, channels) + once_table_bytes(n_pairs, channels) +
           (n_images <= 64 ? (size_t)channels * 64 * 64 * sizeof(float4) : 0);
}

template <typename T, int INTERP, int STD>
static int bwd_launch_once(PairArgs a, void *workspace, size_t workspace_bytes, hipStream_t s)
{
    if (a.n_images > 1024) return CT_ERR_TOO_LARGE;
    if (!workspace || workspace_bytes < once_workspace_bytes(a.n_images, a.n_pairs, a.channels) ||
        reinterpret_cast<uintptr_t>(workspace) % 32 != 0)
        return CT_ERR_INVALID_ARGUMENT;
    const size_t lut_bytes = ((size_t)a.channels * a.n_points * lut_entry_bytes(INTERP) + 15) & ~(size_t)15;
    const size_t cl = (size_t)a.channels * a.n_points;
    const uint32_t plane = a.plane_local;
    int32_t *first = static_cast<int32_t *>(workspace);
    OnceEntry *table = reinterpret_cast<OnceEntry *>(static_cast<char *>(workspace) + once_header_bytes(a.n_images, a.channels));
    float4 *lane_table = reinterpret_cast<float4 *>(reinterpret_cast<char *>(table) + once_table_bytes(a.n_pairs, a.channels));
    a.first_g = first;
    a.table_g = table;
    a.lane_table_g = nullptr;

    // lane <-> sample variant (no uncertainty weighting): the caller promises a band, the list fills at least 65 % of the
    // band x 64 lane-steps the kernel walks (measured on C3: 5.3 us per 1000 lane-steps against 8.7 us per 1000 pairs of the
    // generic kernel; below that the exact pair walk does less work), and the staged tile leaves room for two workgroups
    // per CU.  The entries kernel verifies the promise; a broken one makes the lane kernel return at once and
    // the generic kernel (always launched) do the work.
    PairArgs la = a;
    bool lane = false;
    size_t lane_lds = 0;
    if constexpr (STD == CT_STD_NONE) {
        const int band = (a.lane_band + 1) & ~1;  // the kernel walks the band two steps at a time
        la.lane_band = band;
        if (a.lane_band >= 1 && band <= 64 && a.n_images <= 64 && (double)a.n_pairs >= 0.65 * band * 64) {
            const size_t fixed = (lut_bytes + cl * 8 + 15) & ~(size_t)15;  // LUT | float64 histogram
            for (int tp = 64; tp >= 32 && !lane; tp /= 2) {
                // + (band, 64) constants | (N, pitch) (value, weight) | (N, pitch) LUT coordinate | colrow[tp]
                // (the pair phase reads val rows up to 63 + band: the allocation covers them whatever N is)
                const size_t tile_bytes = std::max((size_t)a.n_images * (tp + 1) * 12 + (size_t)tp * 4, (size_t)(64 + band) * (tp + 1) * 8);
                lane_lds = fixed + (size_t)band * 64 * sizeof(float4) + tile_bytes;
                la.lane_lds_bytes = (int32_t)lane_lds;
                if (lane_lds > 80 * 1024) continue;
                la.tp = tp;
                la.tp_shift = tp == 64 ? 6 : 5;
                la.val_offset = (int32_t)fixed;
                la.row_pitch = tp + 1;
                la.vec = vec_ok<T>(la, kLaneBlock, 2);
                la.lane_table_g = lane_table;
                lane = true;
            }
        }
    }

    // generic kernel geometry: LUT | float64 histogram | (N, pitch) float64 accumulators | staged tile
    const size_t fixed = ((lut_bytes + cl * 8 + 15) & ~(size_t)15) + 256;  // + colrow[tp] at the very end
    // accumulator + (value, weight) + LUT coordinate [+ linearized std], per tile column (+ 256 B of rows)
    const int per_sample = 8 + 12 + (STD == CT_STD_NONE ? 0 : 4);
    const int tp = pick_tile(a.n_images, fixed, per_sample, 64);
    if (tp == 0) return CT_ERR_TOO_LARGE;
    a.tp = tp;
    a.tp_shift = tp == 64 ? 6 : 5;
    a.val_offset = (int32_t)(fixed - 256);
    a.row_pitch = tp + 1;
    a.vec = vec_ok<T>(a, kBwdBlock, 4);
    a.lane_band = lane ? la.lane_band : 0;
    hipLaunchKernelGGL(pair_entries_kernel, dim3(a.channels), dim3(256), 0, s, a, first, table, lane ? lane_table : nullptr);
    if constexpr (STD == CT_STD_NONE) {
        if (lane) {
            const uint32_t tiles = (plane + la.tp - 1) / la.tp;
            const int grid = workgroups_per_channel(lane_lds, kLaneBlock, tiles) * la.channels;
            if (la.use_relative)
                hipLaunchKernelGGL((pair_bwd_lane_kernel<T, INTERP, true>), dim3(grid), dim3(kLaneBlock), lane_lds, s, la);
            else
                hipLaunchKernelGGL((pair_bwd_lane_kernel<T, INTERP, false>), dim3(grid), dim3(kLaneBlock), lane_lds, s, la);
        }
    }
    const size_t lds = fixed + (size_t)a.n_images * a.row_pitch * per_sample;
    const uint32_t tiles = (a.plane_local + tp - 1) / tp;
    const int per_chan = workgroups_per_channel(lds, kBwdBlock, tiles);
    const int grid = per_chan * a.channels;
    if (a.use_relative)
        hipLaunchKernelGGL((pair_bwd_once_kernel<T, INTERP, true, STD>), dim3(grid), dim3(kBwdBlock), lds, s, a);
    else
        hipLaunchKernelGGL((pair_bwd_once_kernel<T, INTERP, false, STD>), dim3(grid), dim3(kBwdBlock), lds, s, a);
    return hipGetLastError() == hipSuccess ? CT_OK : CT_ERR_LAUNCH;
}

template <typename T, int INTERP>
static int bwd_dispatch_std(const PairArgs &a, int std_mode, void *ws, size_t ws_bytes, hipStream_t s)
{
#ifdef CT_PAIRS_MINIMAL
    return std_mode == CT_STD_NONE ? bwd_launch_once<T, INTERP, CT_STD_NONE>(a, ws, ws_bytes, s) : CT_ERR_UNSUPPORTED;
#endif
    if constexpr (INTERP == CT_INTERP_LOOKUP) {  // with uncertainties LOOKUP has no gradient path (rejected by the caller)
        return std_mode == CT_STD_NONE ? bwd_launch_once<T, INTERP, CT_STD_NONE>(a, ws, ws_bytes, s) : CT_ERR_NO_GRADIENT_PATH;
    }
    switch (std_mode) {
        case CT_STD_NONE: return bwd_launch_once<T, INTERP, CT_STD_NONE>(a, ws, ws_bytes, s);
        case CT_STD_CONSTANT: return bwd_launch_once<T, INTERP, CT_STD_CONSTANT>(a, ws, ws_bytes, s);
        case CT_STD_MULTIPLIER: return bwd_launch_once<T, INTERP, CT_STD_MULTIPLIER>(a, ws, ws_bytes, s);
        case CT_STD_EXPLICIT: return bwd_launch_once<T, INTERP, CT_STD_EXPLICIT>(a, ws, ws_bytes, s);
    }
    return CT_ERR_INVALID_ARGUMENT;
}

template <typename T>
static int bwd_dispatch(const PairArgs &a, int interp, int std_mode, void *ws, size_t ws_bytes, hipStream_t s)
{
#ifdef CT_PAIRS_MINIMAL
    return interp == CT_INTERP_LINEAR ? bwd_dispatch_std<T, CT_INTERP_LINEAR>(a, std_mode, ws, ws_bytes, s) : CT_ERR_UNSUPPORTED;
#endif
    switch (interp) {
        case CT_INTERP_LOOKUP: return bwd_dispatch_std<T, CT_INTERP_LOOKUP>(a, std_mode, ws, ws_bytes, s);
        case CT_INTERP_LINEAR: return bwd_dispatch_std<T, CT_INTERP_LINEAR>(a, std_mode, ws, ws_bytes, s);
        case CT_INTERP_CATMULL: return bwd_dispatch_std<T, CT_INTERP_CATMULL>(a, std_mode, ws, ws_bytes, s);
    }
    return CT_ERR_INVALID_ARGUMENT;
}

static int fill_common(PairArgs &a, const void *stack_dev, int32_t n_images, const ct_geometry *g, const float *std_dev,
                       const ct_icrf *icrf, const ct_pair_params *prm, int32_t n_pairs)
{
    if (!stack_dev || !g || !icrf || !prm || n_images < 2 || n_pairs < 0) return CT_ERR_INVALID_ARGUMENT;
    if (g->channels <= 0 || g->h_tile <= 0 || g->width <= 0 || g->h_global < g->h_tile || g->row_offset < 0 ||
        g->row_offset + g->h_tile > g->h_global)
        return CT_ERR_INVALID_ARGUMENT;
    if (g->h_global * g->width * g->channels >= (int64_t)1 << 31) return CT_ERR_TOO_LARGE;
    if (g->image_stride < g->h_tile * g->width * g->channels) return CT_ERR_INVALID_ARGUMENT;
    if (g->layout < CT_LAYOUT_NCHW || g->layout > CT_LAYOUT_NHWC_BGR) return CT_ERR_INVALID_ARGUMENT;
    if (icrf->interp < CT_INTERP_LOOKUP || icrf->interp > CT_INTERP_NONE) return CT_ERR_INVALID_ARGUMENT;
    if (icrf->interp != CT_INTERP_NONE && (!icrf->lut_dev || icrf->n_points < 2)) return CT_ERR_INVALID_ARGUMENT;
    if (prm->std_mode < CT_STD_NONE || prm->std_mode > CT_STD_EXPLICIT) return CT_ERR_INVALID_ARGUMENT;
    if (prm->std_mode == CT_STD_EXPLICIT && !std_dev) return CT_ERR_INVALID_ARGUMENT;
    if (prm->lower > prm->upper) return CT_ERR_INVALID_ARGUMENT;
    a.stack = stack_dev;
    a.std_stack = prm->std_mode == CT_STD_EXPLICIT ? std_dev : nullptr;
    a.lut = icrf->lut_dev;
    a.image_stride = g->image_stride;
    a.tile.plane_local = (uint32_t)(g->h_tile * g->width);
    a.tile.chan_skip = (uint32_t)((g->h_global - g->h_tile) * g->width);
    a.tile.base = (uint32_t)(g->row_offset * g->width);
    a.tile.layout = (uint32_t)g->layout;  // planar, or interleaved RGB / BGR (the stack AND an explicit std stack)
    a.tile.channels = (uint32_t)g->channels;
    a.plane_local = a.tile.plane_local;
    a.n_images = n_images;
    a.n_pairs = n_pairs;
    a.channels = g->channels;
    a.n_points = icrf->interp == CT_INTERP_NONE ? 2 : icrf->n_points;
    a.lower = prm->lower;
    a.upper = prm->upper;
    a.neg_scale_log2e = -prm->weight_scale * 1.4426950408889634f;
    a.std_value = prm->std_value;
    a.use_relative = prm->use_relative;
    a.use_unc_weight = prm->use_uncertainty_weighting;
    a.lane_band = prm->pair_band;
    return CT_OK;
}

}  // namespace ct

extern "C" int ct_norm_constants(float max_code, float *hi, float *lo);
extern "C" int ct_pivot_floor_constants(float max_code, int n_points, float *rcp_step);

namespace ct {
// Code-domain staging is taken for integer stacks at the type's full range with a LINEAR curve whose step is a whole
// number of codes (host-verified for every code: ct_pivot_floor_constants) and no uncertainties.  The reference's validity
// mask lower <= fl(u / max) <= upper (general_functions.py:302) becomes a code interval by bisection with the same float32
// division (monotone in u); an empty interval keeps the generic staging.
static void fill_code_domain(PairArgs &a, int32_t dtype, float max_code, int interp, int std_mode, float weight_scale)
{
    a.code_domain = 0;
    if (interp != CT_INTERP_LINEAR || std_mode != CT_STD_NONE) return;
    // (Extending this staging to max_code below the container's range and to LUT steps that are not whole numbers of codes
    // was built and measured in round 3 -- value and coordinate in the reference's order on the proven interval, bit-identical
    // to the generic staging -- and bought nothing: C3 with 12-bit codes 8.94 ms against 8.90 ms generic; at full range the
    // two stagings now differ by 0.5 %, the loads of the next tile being in flight behind the pair phase either way.)
    static const bool disabled = getenv("CT_PAIRS_NO_CODE_DOMAIN") != nullptr;  // diagnostics: time the generic staging
    if (disabled) return;
    if (max_code != (dtype == CT_DTYPE_U8 ? 255.0f : 65535.0f)) return;
    float rcp = 0.0f;
    if (ct_pivot_floor_constants(max_code, a.n_points, &rcp) != CT_OK) return;  // whole steps, verified for every code
    const float step = (float)((int)max_code / (a.n_points - 1));
    const int maxc = (int)max_code;
    auto norm = [&](int u) { const float uf = (float)u; return fmaf(uf, a.norm.hi, uf * a.norm.lo); };  // == fl(u / max), ct_norm_constants
    int lo = 0, hi = maxc + 1;  // first code with norm >= lower
    while (lo < hi) { const int mid = (lo + hi) / 2; if (norm(mid) >= a.lower) hi = mid; else lo = mid + 1; }
    const int code_lo = lo;
    lo = 0; hi = maxc + 1;      // first code with norm > upper
    while (lo < hi) { const int mid = (lo + hi) / 2; if (norm(mid) > a.upper) hi = mid; else lo = mid + 1; }
    const int code_hi = lo - 1;
    if (code_lo > code_hi) return;
    const double kk = sqrt((double)weight_scale * 1.4426950408889634);
    a.code_rcp = rcp;
    a.code_step = step;
    a.code_inv_step = (float)(1.0 / (double)step);
    a.code_lo = (float)code_lo;
    a.code_hi = (float)code_hi;
    a.code_dk_mul = (float)(kk / (double)max_code);
    a.code_dk_add = (float)(-0.5 * kk);
    a.code_domain = 1;
}
}  // namespace ct

extern "C" int ct_pair_residual_fwd(const void *stack_dev, int32_t dtype, float max_code, int32_t n_images,
                                    const ct_geometry *geom, const float *std_dev, const ct_icrf *icrf,
                                    const int32_t *i_idx_dev, const int32_t *j_idx_dev, const double *ratio_dev,
                                    int32_t n_pairs, const ct_pair_params *params, int32_t level,
                                    const double *center_dev, double *sums_dev, void *stream)
{
    using namespace ct;
    PairArgs a{};
    int rc = fill_common(a, stack_dev, n_images, geom, std_dev, icrf, params, n_pairs);
    if (rc != CT_OK) return rc;
    if (n_pairs == 0) return CT_OK;
    if (!i_idx_dev || !j_idx_dev || !ratio_dev || !sums_dev || level < 0 || level > 1) return CT_ERR_INVALID_ARGUMENT;
    // icrf_training.py:117-124 / measure_linearity.py:57-62: autograd.grad raises for LOOKUP when stds are given
    if (params->std_mode != CT_STD_NONE && icrf->interp == CT_INTERP_LOOKUP) return CT_ERR_NO_GRADIENT_PATH;
    a.i_idx = i_idx_dev;
    a.j_idx = j_idx_dev;
    a.ratio = ratio_dev;
    a.sums = sums_dev;
    a.center = center_dev;
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (dtype) {
#ifndef CT_PAIRS_MINIMAL
        case CT_DTYPE_U8:
            if (ct_norm_constants(max_code, &a.norm.hi, &a.norm.lo) != CT_OK) return CT_ERR_UNSUPPORTED;
            fill_code_domain(a, dtype, max_code, icrf->interp, params->std_mode, params->weight_scale);
            return fwd_dispatch<uint8_t>(a, icrf->interp, params->std_mode, level, s);
        case CT_DTYPE_F32: return fwd_dispatch<float>(a, icrf->interp, params->std_mode, level, s);
#endif
        case CT_DTYPE_U16:
            if (ct_norm_constants(max_code, &a.norm.hi, &a.norm.lo) != CT_OK) return CT_ERR_UNSUPPORTED;
            fill_code_domain(a, dtype, max_code, icrf->interp, params->std_mode, params->weight_scale);
            return fwd_dispatch<uint16_t>(a, icrf->interp, params->std_mode, level, s);
    }
    return CT_ERR_UNSUPPORTED;
}

extern "C" int ct_pair_residual_bwd(const void *stack_dev, int32_t dtype, float max_code, int32_t n_images,
                                    const ct_geometry *geom, const float *std_dev, const ct_icrf *icrf,
                                    const double *ratio_dev, int32_t n_pairs, const int32_t *partner_offsets_dev,
                                    const int32_t *partner_sample_dev, const int32_t *partner_pair_dev,
                                    const ct_pair_params *params, const double *coef_dev, const double *smean_dev,
                                    double *lut_grad_dev, void *workspace_dev, int64_t workspace_bytes, void *stream)
{
    using namespace ct;
    if (!params) return CT_ERR_INVALID_ARGUMENT;
    // uncertainties only matter to the backward through the weights: without uncertainty weighting they are ignored
    ct_pair_params prm = *params;
    if (!prm.use_uncertainty_weighting) prm.std_mode = CT_STD_NONE;
    PairArgs a{};
    int rc = fill_common(a, stack_dev, n_images, geom, std_dev, icrf, &prm, n_pairs);
    if (rc != CT_OK) return rc;
    if (n_pairs == 0) return CT_OK;
    if (!ratio_dev || !partner_offsets_dev || !partner_sample_dev || !partner_pair_dev || !coef_dev || !lut_grad_dev ||
        workspace_bytes < 0)
        return CT_ERR_INVALID_ARGUMENT;
    if (icrf->interp == CT_INTERP_NONE) return CT_ERR_INVALID_ARGUMENT;
    if (prm.std_mode != CT_STD_NONE) {
        if (!smean_dev) return CT_ERR_INVALID_ARGUMENT;
        if (icrf->interp == CT_INTERP_LOOKUP) return CT_ERR_NO_GRADIENT_PATH;
    }
    a.ratio = ratio_dev;
    a.part_off = partner_offsets_dev;
    a.part_sample = partner_sample_dev;
    a.part_pair = partner_pair_dev;
    a.coef = coef_dev;
    a.smean = smean_dev;
    a.lut_grad = lut_grad_dev;
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (dtype) {
#ifndef CT_PAIRS_MINIMAL
        case CT_DTYPE_U8:
            if (ct_norm_constants(max_code, &a.norm.hi, &a.norm.lo) != CT_OK) return CT_ERR_UNSUPPORTED;
            fill_code_domain(a, dtype, max_code, icrf->interp, prm.std_mode, prm.weight_scale);
            return bwd_dispatch<uint8_t>(a, icrf->interp, prm.std_mode, workspace_dev, (size_t)workspace_bytes, s);
        case CT_DTYPE_F32: return bwd_dispatch<float>(a, icrf->interp, prm.std_mode, workspace_dev, (size_t)workspace_bytes, s);
#endif
        case CT_DTYPE_U16:
            if (ct_norm_constants(max_code, &a.norm.hi, &a.norm.lo) != CT_OK) return CT_ERR_UNSUPPORTED;
            fill_code_domain(a, dtype, max_code, icrf->interp, prm.std_mode, prm.weight_scale);
            return bwd_dispatch<uint16_t>(a, icrf->interp, prm.std_mode, workspace_dev, (size_t)workspace_bytes, s);
    }
    return CT_ERR_UNSUPPORTED;
}

extern "C" int64_t ct_pair_residual_bwd_workspace(int32_t n_images, int32_t n_pairs, int32_t channels)
{
    if (n_images < 0 || n_pairs < 0 || channels < 0) return 0;
    return (int64_t)ct::once_workspace_bytes(n_images, n_pairs, channels);
}
