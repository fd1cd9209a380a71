// ct_merge_exact.hip -- HDR merge of one batch with the uncertainty evaluated in the REFERENCE'S OWN float32 order (gfx950).
//
// compute_hdr_image obtains the variance from torch.autograd.grad (clair_torch/inference/hdr_merge.py:107-115).  That
// backward is a fixed sequence of float32 / float64 operations, and two of its steps cancel by two orders of magnitude in
// float32: the Catmull-Rom basis backward (the upstream gradient multiplies the four LUT taps first, then the signed sums
// d/dt3, d/dt2, d/dt) and, for every mode, the weight path fl32(y_n G) - fl32(S G / D) on a consistent stack.  The closed
// forms the fast kernels evaluate (ct_merge.hip) are better conditioned than that -- and therefore differ from the
// reference by its own rounding noise: up to 2e-5 element-wise for CATMULL on uint16 data, 5e-6 for the other modes
// (tests/test_oracle_golden.py measures it).  This kernel follows the reference instead, operation by operation:
//
//   pass 1 over the batch:  w_n = exp(-30 (x - 1/2)^2) (float32 ops, the exp rounded once from float64), lin_n = model forward in
//                           the reference's un-fused order, y_n = lin_n / t_n (float64), W_b = sum w (float32, in the
//                           summation order of torch.sum), S = sum w y (float64)
//   per element:            D, m_b, W_t, frac, mean and the gradients of the scalar chain exactly as autograd's nodes run
//                           (statistics.py:64-109 backward; DivBackward0 = -grad * ((self / other) / other), gradients
//                           meeting at W_b added in arrival order)
//   pass 2 over the batch:  G_n = fl32((g_S w_n) / t_n) into the model backward (icrf_grad_reference_order: G first), the
//                           weight path fl32(g_S y_n) + g_Wb through exp / (-30) / pow 2, their sum times sigma, squared,
//                           summed in torch.sum's order.
//
// oracle/eager_torch.merge_stack_reference_order is the same sequence on the CPU; with torch's own exp it reproduces every
// recorded uncertainty of the reference bit for bit, with a correctly rounded exp -- what this kernel uses -- it is within
// 1.1e-5 (CATMULL, uint16) / 3e-6 (everything else) of them, and this kernel equals THAT emulation bit for bit
// (tests/test_gpu_merge.py).  Used for CATMULL with uncertainties by default and for any mode with
// CT_MERGE_REFERENCE_ORDER.
//
// Cost: two passes over the batch (the second mostly from L2 / Infinity Cache), a float64 exp per sample and pass and the
// float64 scalar chain: 3-5x the time of the fast kernels.  Roofline: not bandwidth -- float64 VALU.  It is a parity instrument for the
// modes whose reference result is dominated by float32 cancellation, not the headline path.
#include <algorithm>
#include <cstdlib>
#include "ct_merge.hpp"

namespace ct {

// torch.sum over the leading dimension of a contiguous float32 tensor on the CPU (ATen/native/cpu/SumKernel.cpp,
// multi_row_sum): rows are added one after the other into level 0; after every 16 rows level 0 is folded into level 1,
// after every 256 into level 2, after every 4096 into level 3; the levels are added up at the end.  (Level width
// max(4, ceil(log2 n) / 4) bits: 16 rows up to n = 65536; the dispatcher refuses larger batches.)  For n <= 16 this is a
// plain left-to-right sum.  tests/test_oracle_golden.py pins the emulation of this order against torch itself.
struct TorchRowSum {
    float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
    __device__ __forceinline__ void add(float v, int row)
    {
        a0 = a0 + v;
        const int n = row + 1;
        if ((n & 15) == 0) {
            a1 = a1 + a0;
            a0 = 0.0f;
            if ((n & 0xf0) == 0) {
                a2 = a2 + a1;
                a1 = 0.0f;
                if ((n & 0xf00) == 0) {
                    a3 = a3 + a2;
                    a2 = 0.0f;
                }
            }
        }
    }
    __device__ __forceinline__ float total() const { return ((a0 + a1) + a2) + a3; }
};

// exp of a float32 argument, rounded once from a float64 value with relative error < 3e-13 (range reduction by ln 2 in two
// parts, Taylor polynomial of degree 10 on |r| <= 0.347, one ldexp): the float32 result is the correctly rounded one for
// all but ~2e-7 of the arguments (checked against numpy's float64 exp on 5 M arguments).  A third of the cost of the
// library's float64 exp, which mattered: this kernel evaluates it twice per sample.  torch's CPU exp (Sleef expf, 1 ULP)
// returns the correctly rounded value for 98.9 % of arguments; the other 1.1 % are the irreducible difference to the
// recorded vectors.
__device__ __forceinline__ float exp_correctly_rounded(float v)
{
    const double x = (double)v;
    const double k = __builtin_rint(x * 1.4426950408889634);
    double r = __builtin_fma(k, -0x1.62e42fefa38p-1, x);     // ln 2, high part (low bits cleared: k * hi is exact)
    r = __builtin_fma(k, -0x1.ef35793c7673p-45, r);           // ln 2, low part
    double p = 2.755731922398589e-07;                         // 1 / 10!
    p = __builtin_fma(p, r, 2.7557319223985893e-06);
    p = __builtin_fma(p, r, 2.48015873015873e-05);
    p = __builtin_fma(p, r, 1.984126984126984e-04);
    p = __builtin_fma(p, r, 1.3888888888888889e-03);
    p = __builtin_fma(p, r, 8.333333333333333e-03);
    p = __builtin_fma(p, r, 4.1666666666666664e-02);
    p = __builtin_fma(p, r, 1.6666666666666666e-01);
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    p = __builtin_fma(p, r, 1.0);
    const double kk = k < -1100.0 ? -1100.0 : k;              // (very negative arguments underflow to zero like exp)
    return (float)__builtin_ldexp(p, (int)kk);
}

// Correctly rounded float32 square root.  hipcc's own expansion (__fsqrt_rn / sqrtf: v_sqrt_f32 plus a one-step fix-up)
// was measured 1 ULP low on 17 % of the variances of the recorded fixtures on gfx950 (true root up to 0.86 ULP above the
// returned value; tools/debug/exact_dump2.py), so the candidate from the float64 root is checked here against the exact
// rounding boundaries: a midpoint of two neighbouring floats has 25 significant bits, its square 50 -- exact in float64.
__device__ __forceinline__ float sqrt_correctly_rounded(float v)
{
    if (!(v > 0.0f) || v > 3.0e38f) return __builtin_sqrtf(v);  // zeros, negatives (NaN), infinities, NaN: the plain result
    float s = (float)__builtin_sqrt((double)v);
    const double vd = (double)v;
    const float up = __uint_as_float(__float_as_uint(s) + 1u), dn = __uint_as_float(__float_as_uint(s) - 1u);
    const double m_up = 0.5 * ((double)s + (double)up), m_dn = 0.5 * ((double)s + (double)dn);
    if (m_up * m_up < vd)
        s = up;
    else if (m_dn * m_dn > vd)
        s = dn;
    return s;
}

// CACHE (small batches -- the reference's scripts stream batches of 4): pass 1 leaves every sample's weight (CACHE >= 1) and
// linearized value (CACHE == 2) in LDS, wc[n][e][thread] / lc[n][e][thread], so that pass 2 neither exponentiates nor
// interpolates a second time -- identical values by construction (the same float32 results, read back).  4 B (8 B) of LDS
// per sample: batches of up to 8 (4) exposures at four workgroups per CU.  (One element per thread would fit batches of 32,
// and was measured slower than not caching at all: 5.4 against 3.9 ms on C2 LOOKUP -- the per-exposure overhead of the
// loop is then paid per sample.)
template <typename T, int V, int INTERP, int WEIGHT, int STD, int CACHE = 0>
__global__ __launch_bounds__(kBlock) void merge_reference_order_kernel(const MergeArgs a, const MergeBatches mb)
{
    extern __shared__ __align__(16) char lds[];
    constexpr bool kGauss = WEIGHT == CT_WEIGHT_GAUSS;
    constexpr bool kHasStd = STD != CT_STD_NONE;
    constexpr int kEntry = lut_entry_bytes(INTERP);
    const int C = a.channels, L = a.n_points;
    const int n_exposures = a.batch;  // of all batches of the launch
    stage_lut<INTERP, false>(lds, a.lut, C, L);
    // 1 / t_n in float64, once per workgroup: y_n = lin_n / t_n and G_n = (g_S w_n) / t_n are formed as products with it
    // (at most one float64 ulp from the divisions the reference performs, i.e. invisible after the float32 casts except on
    // ~1e-9 of the samples) -- three float64 divisions per sample were a quarter of this kernel's time
    double *inv_t = reinterpret_cast<double *>(lds + ((INTERP == CT_INTERP_NONE ? 0 : C * L * kEntry) + 7 & ~7));
    for (int n = threadIdx.x; n < n_exposures; n += kBlock) inv_t[n] = 1.0 / a.exposure[n];
    int b_max = n_exposures;  // largest batch: the cache arrays are sized for it (host: exact_launch)
    if (mb.n_batches > 0) {
        b_max = 0;
        for (int k = 0; k < mb.n_batches; ++k) b_max = mb.batch_size[k] > b_max ? mb.batch_size[k] : b_max;
    }
    [[maybe_unused]] float *wc = reinterpret_cast<float *>(inv_t + n_exposures) + threadIdx.x;   // + (n * V + e) * kBlock: conflict-free
    [[maybe_unused]] float *lc = wc + (size_t)b_max * V * kBlock;
    __syncthreads();
    const float top = INTERP == CT_INTERP_NONE ? 1.0f : (float)(L - 1);
    const bool fresh = a.flags & CT_MERGE_FIRST_BATCH, finalize = a.flags & CT_MERGE_FINALIZE;
    const bool keep_state = a.mean_state != nullptr;

    const uint32_t vec = blockIdx.x * (uint32_t)kBlock + threadIdx.x;
    if (vec * (uint32_t)V >= a.q_count) return;
    const uint32_t q0 = a.q_begin + vec * (uint32_t)V;

    uint32_t qp[V];       // index of each element in the state / output arrays (planar unless CT_MERGE_OUT_AS_INPUT)
    const char *row[V];   // its LUT row in LDS
#pragma unroll
    for (int e = 0; e < V; ++e) {
        qp[e] = a.out_index(q0 + e);
        int ch;
        uint32_t qg;
        a.tile.locate(a.tile.planar_index(q0 + e), ch, qg);
        row[e] = lds + (INTERP == CT_INTERP_NONE ? 0 : lut_row<INTERP>(qg, ch, C) * L * kEntry);
    }

    // the forward of one sample in the reference's float32 order: pixel, weight, linearized value
    auto forward = [&](T code, const char *r, float &x, float &xm, float &w, float &lin) {
        x = to_pixel<T>(code, a.norm);                                   // CastTo + Normalize (general_functions.py:377)
        xm = x - 0.5f;                                                   // losses.py:205
        w = kGauss ? exp_correctly_rounded(-a.weight_scale * (xm * xm)) : 1.0f;   // hdr_merge.py:95 (ones without a weight_fn)
        float unused;
        lin = icrf_sample<INTERP, true, false>(x, r, top, unused);       // base.py:135-226, un-fused
    };

    // The streaming state: read once, carried in registers over the batches of the launch (internal_detach, hdr_merge.py:128:
    // a batch's result is the next batch's state), written once.
    double state_mean[V];
    float state_w[V], state_var[V];
#pragma unroll
    for (int e = 0; e < V; ++e) {
        state_mean[e] = fresh ? 0.0 : a.mean_state[qp[e]];
        state_w[e] = fresh ? 0.0f : a.sumw_state[qp[e]];
        state_var[e] = (fresh || !kHasStd) ? 0.0f : a.var_state[qp[e]];
    }
    const int n_batches = mb.n_batches > 0 ? mb.n_batches : 1;
    int n_first = 0;  // first exposure of the current batch among the launch's exposures
    for (int bi = 0; bi < n_batches; ++bi) {
    const int B = mb.n_batches > 0 ? mb.batch_size[bi] : n_exposures;
    const T *src = static_cast<const T *>(mb.n_batches > 0 ? mb.batch_ptr[bi] : a.stack) + q0;
    const float *ssrc = STD == CT_STD_EXPLICIT ? (mb.n_batches > 0 ? mb.std_ptr[bi] : a.std_stack) + q0 : nullptr;
    const bool first = fresh && bi == 0;
    const double *inv_tb = inv_t + n_first;

    // ---- pass 1: W_b (float32, torch.sum order), S = sum w y (float64) ----
    TorchRowSum Wsum[V];
    double S[V];
#pragma unroll
    for (int e = 0; e < V; ++e) S[e] = 0.0;
    for (int n = 0; n < B; ++n) {
        const Packet<T, V> pk = *reinterpret_cast<const Packet<T, V> *>(src + (int64_t)n * a.image_stride);
        const double it = inv_tb[n];
#pragma unroll
        for (int e = 0; e < V; ++e) {
            float x, xm, w, lin;
            forward(pk.v[e], row[e], x, xm, w, lin);
            if constexpr (CACHE >= 1) wc[(n * V + e) * kBlock] = w;
            if constexpr (CACHE == 2) lc[(n * V + e) * kBlock] = lin;
            const double y = (double)lin * it;                           // hdr_merge.py:103 (float32 / float64)
            Wsum[e].add(w, n);
            S[e] = S[e] + (double)w * y;                                 // statistics.py:79
        }
    }

    // ---- per element: the scalar chain and its backward (statistics.py:78-80, 99-109) ----
    float Wb[V], D[V], Wt[V], frac[V], g_wb[V], var_in[V];
    double mean[V], g_s[V];
#pragma unroll
    for (int e = 0; e < V; ++e) {
        Wb[e] = Wsum[e].total();
        D[e] = Wb[e] + 1e-6f;                                            // float32 tensor + Python float
        const double mb = S[e] / (double)D[e];
        const float WA = first ? 0.0f : state_w[e];
        const double meanA = first ? 0.0 : state_mean[e];
        Wt[e] = WA + Wb[e];
        frac[e] = __fdiv_rn(Wb[e], Wt[e]);
        const double diff = mb - meanA;
        mean[e] = meanA + (double)frac[e] * diff;
        var_in[e] = (first || !kHasStd) ? 0.0f : state_var[e];
        if constexpr (kHasStd) {
            const float g_frac = (float)diff;                             // d mean / d frac, cast at the float32 tensor
            g_s[e] = (double)frac[e] / (double)D[e];                      // d mean / d S
            const float g_d = (float)(-(double)frac[e] * ((S[e] / (double)D[e]) / (double)D[e]));   // DivBackward0, other
            float g = __fdiv_rn(g_frac, Wt[e]);                           // arrival 1 at W_b: W_B / W, self
            g = g + (-g_frac * __fdiv_rn(__fdiv_rn(Wb[e], Wt[e]), Wt[e]));   // arrival 2: through W = W_A + W_B
            g_wb[e] = g + g_d;                                            // arrival 3: through W_B + 1e-6
        }
    }

    // ---- pass 2: per-sample gradient in autograd's order, squared and summed like torch.sum ----
    float var_o[V];
    if constexpr (kHasStd) {
        TorchRowSum U[V];
        for (int n = 0; n < B; ++n) {
            const Packet<T, V> pk = *reinterpret_cast<const Packet<T, V> *>(src + (int64_t)n * a.image_stride);
            Packet<float, V> sp;
            if constexpr (STD == CT_STD_EXPLICIT) sp = *reinterpret_cast<const Packet<float, V> *>(ssrc + (int64_t)n * a.image_stride);
            const double it = inv_tb[n];
#pragma unroll
            for (int e = 0; e < V; ++e) {
                float x, xm, w, lin;
                if constexpr (CACHE == 0) {
                    forward(pk.v[e], row[e], x, xm, w, lin);
                } else {
                    x = to_pixel<T>(pk.v[e], a.norm);
                    xm = x - 0.5f;
                    w = wc[(n * V + e) * kBlock];
                    if constexpr (CACHE == 2) {
                        lin = lc[(n * V + e) * kBlock];
                    } else {
                        float unused;
                        lin = icrf_sample<INTERP, true, false>(x, row[e], top, unused);
                    }
                }
                const double y = (double)lin * it;
                const float g_lin = (float)((g_s[e] * (double)w) * it);   // mul backward (float64), / exposure, cast at the model output
                float g_x = icrf_grad_reference_order<INTERP>(x, row[e], top, g_lin);   // model nodes run before the weight nodes
                if constexpr (kGauss) {
                    const float g_w = (float)(g_s[e] * y) + g_wb[e];      // the product's gradient arrives first, then the expanded sum's
                    const float g_a = ((g_w * w) * -a.weight_scale) * (2.0f * xm);   // exp, * (-scale), pow(2) backward
                    g_x = INTERP == CT_INTERP_LOOKUP ? g_a : g_x + g_a;
                }
                float sigma = a.std_value;                                // CONSTANT
                if constexpr (STD == CT_STD_MULTIPLIER) sigma = x * a.std_value;   // datasets/base.py:133
                if constexpr (STD == CT_STD_EXPLICIT) sigma = sp.v[e];
                const float gs = g_x * sigma;
                U[e].add(gs * gs, n);                                     // hdr_merge.py:114
            }
        }
#pragma unroll
        for (int e = 0; e < V; ++e) var_o[e] = first ? U[e].total() : var_in[e] + U[e].total();   // :115
    }

#pragma unroll
    for (int e = 0; e < V; ++e) {
        state_mean[e] = mean[e];
        state_w[e] = Wt[e];
        if constexpr (kHasStd) state_var[e] = var_o[e];
    }
    n_first += B;
    }  // batches

#pragma unroll
    for (int e = 0; e < V; ++e) {
        if (keep_state) {
            a.mean_state[qp[e]] = state_mean[e];
            a.sumw_state[qp[e]] = state_w[e];
            if constexpr (kHasStd) a.var_state[qp[e]] = state_var[e];
        }
        if (finalize) {
            if (a.flags & CT_MERGE_MEAN_OUT_F32)
                static_cast<float *>(a.mean_out)[qp[e]] = (float)state_mean[e];
            else
                static_cast<double *>(a.mean_out)[qp[e]] = state_mean[e];
            if constexpr (kHasStd) a.std_out[qp[e]] = sqrt_correctly_rounded(state_var[e]);   // hdr_merge.py:155
        }
    }
}

static size_t exact_fixed_lds(const MergeArgs &a, int interp)
{
    return ((interp == CT_INTERP_NONE ? 0 : (size_t)a.channels * a.n_points * lut_entry_bytes(interp)) + 7 & ~(size_t)7) +
           sizeof(double) * (size_t)a.batch;
}

// Values kept per sample between the passes: 2 (weight and linearized value) while a workgroup's LDS stays within 37 KB --
// four workgroups per CU, the occupancy the kernel's registers allow anyway --, else 1 (the weight: the float64 exp is the
// expensive half) up to 45 KB (three workgroups per CU), else 0.  Measured on C2 through the API (CATMULL, batches of 4 / 8):
// both values at 44 KB -5 % against no cache, the weight alone at 28 KB -11 %; batches of 8 with the weight at 44 KB -6 %.
constexpr size_t kCacheBudgetBoth = 37 * 1024, kCacheBudgetWeight = 45 * 1024;
static int exact_cache_level(const MergeArgs &a, int b_max, int interp, int weight_mode, int std_mode, int v)
{
    if (std_mode == CT_STD_NONE) return 0;  // no second pass
    static const bool disabled = getenv("CT_EXACT_NO_CACHE") != nullptr;  // diagnostics: time the two-pass form alone
    if (disabled) return 0;
    const size_t per_level = (size_t)b_max * v * kBlock * sizeof(float);  // (b_max: the largest batch of the launch)
    if (exact_fixed_lds(a, interp) + 2 * per_level <= kCacheBudgetBoth) return 2;
    // the weight alone is only worth keeping when it is an exp
    if (weight_mode == CT_WEIGHT_GAUSS && exact_fixed_lds(a, interp) + per_level <= kCacheBudgetWeight) return 1;
    return 0;
}

struct ExactRoute {
    MergeBatches mb;  // n_batches == 0: the single batch of MergeArgs
    int b_max;        // exposures of the largest batch
};

template <typename T, int V, int INTERP, int WEIGHT, int STD, int CACHE = 0>
static int exact_launch(const MergeArgs &a, const ExactRoute &r, hipStream_t s)
{
    if (a.q_count == 0) return CT_OK;
    const uint32_t vecs = (a.q_count + V - 1) / V, grid = (vecs + kBlock - 1) / kBlock;
    const size_t lds = exact_fixed_lds(a, INTERP) + (size_t)CACHE * r.b_max * V * kBlock * sizeof(float);
    if (lds > 160 * 1024) return CT_ERR_TOO_LARGE;
    hipLaunchKernelGGL((merge_reference_order_kernel<T, V, INTERP, WEIGHT, STD, CACHE>), dim3(grid), dim3(kBlock), lds, s, a, r.mb);
    return hipGetLastError() == hipSuccess ? CT_OK : CT_ERR_LAUNCH;
}

template <typename T, int V, int INTERP, int WEIGHT>
static int exact_std(const MergeArgs &a, const ExactRoute &r, int std_mode, hipStream_t s)
{
    if constexpr (V == 4) {  // the cached variants (packets only; ragged tails are a handful of elements)
        const int level = exact_cache_level(a, r.b_max, INTERP, WEIGHT, std_mode, V);
        if (level == 2) {
            switch (std_mode) {
                case CT_STD_CONSTANT: return exact_launch<T, V, INTERP, WEIGHT, CT_STD_CONSTANT, 2>(a, r, s);
                case CT_STD_MULTIPLIER: return exact_launch<T, V, INTERP, WEIGHT, CT_STD_MULTIPLIER, 2>(a, r, s);
                case CT_STD_EXPLICIT: return exact_launch<T, V, INTERP, WEIGHT, CT_STD_EXPLICIT, 2>(a, r, s);
            }
        } else if (level == 1) {
            switch (std_mode) {
                case CT_STD_CONSTANT: return exact_launch<T, V, INTERP, WEIGHT, CT_STD_CONSTANT, 1>(a, r, s);
                case CT_STD_MULTIPLIER: return exact_launch<T, V, INTERP, WEIGHT, CT_STD_MULTIPLIER, 1>(a, r, s);
                case CT_STD_EXPLICIT: return exact_launch<T, V, INTERP, WEIGHT, CT_STD_EXPLICIT, 1>(a, r, s);
            }
        }
    }
    switch (std_mode) {
        case CT_STD_NONE: return exact_launch<T, V, INTERP, WEIGHT, CT_STD_NONE>(a, r, s);
        case CT_STD_CONSTANT: return exact_launch<T, V, INTERP, WEIGHT, CT_STD_CONSTANT>(a, r, s);
        case CT_STD_MULTIPLIER: return exact_launch<T, V, INTERP, WEIGHT, CT_STD_MULTIPLIER>(a, r, s);
        case CT_STD_EXPLICIT: return exact_launch<T, V, INTERP, WEIGHT, CT_STD_EXPLICIT>(a, r, s);
    }
    return CT_ERR_INVALID_ARGUMENT;
}

template <typename T, int V>
static int exact_interp(const MergeArgs &a, const ExactRoute &r, int interp, int weight_mode, int std_mode, hipStream_t s)
{
    const bool gauss = weight_mode == CT_WEIGHT_GAUSS;
    switch (interp) {
        case CT_INTERP_LOOKUP:
            return gauss ? exact_std<T, V, CT_INTERP_LOOKUP, CT_WEIGHT_GAUSS>(a, r, std_mode, s)
                         : exact_std<T, V, CT_INTERP_LOOKUP, CT_WEIGHT_NONE>(a, r, std_mode, s);
        case CT_INTERP_LINEAR:
            return gauss ? exact_std<T, V, CT_INTERP_LINEAR, CT_WEIGHT_GAUSS>(a, r, std_mode, s)
                         : exact_std<T, V, CT_INTERP_LINEAR, CT_WEIGHT_NONE>(a, r, std_mode, s);
        case CT_INTERP_CATMULL:
            return gauss ? exact_std<T, V, CT_INTERP_CATMULL, CT_WEIGHT_GAUSS>(a, r, std_mode, s)
                         : exact_std<T, V, CT_INTERP_CATMULL, CT_WEIGHT_NONE>(a, r, std_mode, s);
        case CT_INTERP_NONE:
            return gauss ? exact_std<T, V, CT_INTERP_NONE, CT_WEIGHT_GAUSS>(a, r, std_mode, s)
                         : exact_std<T, V, CT_INTERP_NONE, CT_WEIGHT_NONE>(a, r, std_mode, s);
    }
    return CT_ERR_INVALID_ARGUMENT;
}

template <typename T>
static int exact_typed(MergeArgs a, const ExactRoute &r, uint32_t Q, int interp, int weight_mode, int std_mode, hipStream_t s)
{
    // packets of 4 where every exposure's packet is naturally aligned; the rest (and everything, on odd strides) one by one
    constexpr int V = 4;
    auto aligned = [](const void *p, size_t bytes) { return p == nullptr || (reinterpret_cast<uintptr_t>(p) % bytes) == 0; };
    bool vec_ok = aligned(a.stack, sizeof(T) * V) && (a.image_stride % V) == 0 && aligned(a.std_stack, 4 * V);
    for (int b = 0; b < r.mb.n_batches; ++b) vec_ok = vec_ok && aligned(r.mb.batch_ptr[b], sizeof(T) * V) && aligned(r.mb.std_ptr[b], 4 * V);
    const uint32_t q_vec = vec_ok ? (Q / V) * V : 0;
    int rc = CT_OK;
    if (q_vec) {
        a.q_begin = 0;
        a.q_count = q_vec;
        rc = exact_interp<T, V>(a, r, interp, weight_mode, std_mode, s);
        if (rc != CT_OK) return rc;
    }
    if (q_vec < Q) {
        a.q_begin = q_vec;
        a.q_count = Q - q_vec;
        rc = exact_interp<T, 1>(a, r, interp, weight_mode, std_mode, s);
    }
    return rc;
}

int merge_reference_order(const MergeArgs &a, int dtype, uint32_t q_total, int interp, int weight_mode, int std_mode,
                          hipStream_t stream, const MergeBatches *batches)
{
    if (a.batch > 65536) return CT_ERR_TOO_LARGE;  // TorchRowSum's level width
    ExactRoute r{};
    r.b_max = a.batch;
    if (batches && batches->n_batches > 0) {
        if (batches->n_batches > kMaxMergeBatches) return CT_ERR_INVALID_ARGUMENT;
        r.mb = *batches;
        r.b_max = 0;
        for (int b = 0; b < batches->n_batches; ++b) r.b_max = std::max(r.b_max, (int)batches->batch_size[b]);
    }
    switch (dtype) {
        case CT_DTYPE_U8: return exact_typed<uint8_t>(a, r, q_total, interp, weight_mode, std_mode, stream);
        case CT_DTYPE_U16: return exact_typed<uint16_t>(a, r, q_total, interp, weight_mode, std_mode, stream);
        case CT_DTYPE_F32: return exact_typed<float>(a, r, q_total, interp, weight_mode, std_mode, stream);
    }
    return CT_ERR_UNSUPPORTED;
}

}  // namespace ct
