// ct_merge.hip -- fused HDR merge + propagated uncertainty for one batch of exposures (gfx950).
//
// Replaces the interior of compute_hdr_image's loop body (clair_torch/inference/hdr_merge.py:61-128) and,
// with CT_MERGE_FINALIZE, its return statement (hdr_merge.py:155).  The reference runs ~50 full-tensor eager
// kernels plus an autograd backward per batch; here each thread owns V consecutive output elements, streams
// the B samples of those elements once from HBM (16-byte coalesced loads), looks the ICRF up in an LDS copy
// of the LUT, and keeps five running sums per element in registers:
//
//     W   = sum w_n                      (float32, as torch.sum over the batch dim of the float32 weights)
//     Swy = sum w_n y_n                  y_n = f(x_n) / t_n
//     Saa = sum a_n^2, Sab = sum a_n b_n, Sbb = sum b_n^2          (float64)
//         a_n = w'_n sigma_n,  b_n = (w'_n y_n + w_n y'_n) sigma_n
//
// from which the closed form of the reference's autograd variance follows (oracle/ct_oracle.c):
//     m_b  = Swy / (W + 1e-6)                     mean = mean_A + (W/Wt)(m_b - mean_A),  Wt = W_A + W
//     dmean/dx_n = alpha w'_n + beta (w'_n y_n + w_n y'_n)
//         beta  = (W/Wt) / (W + 1e-6),   alpha = (W_A/Wt^2)(m_b - mean_A) - beta m_b
//     var += alpha^2 Saa + 2 alpha beta Sab + beta^2 Sbb
// The three second moments are accumulated in float64: the quadratic form cancels by up to ~100x where
// w'(y - m) and w y' nearly cancel, which float32 sums cannot carry at the 1e-5 parity bar.
//
// Roofline: HBM.  Algorithmic bytes per output element = B * sizeof(T) (+ 4 B with an explicit std stack)
// read + 12 written (float64 mean + float32 std).  No MFMA: this is a gather/reduce, not a contraction.
#include "ct_device.hpp"

namespace ct {

struct MergeArgs {
    const void *stack;
    const float *std_stack;
    const double *exposure;
    const float *lut;
    double *mean_state;
    float *sumw_state;
    float *var_state;
    void *mean_out;
    float *std_out;
    int64_t image_stride;  // elements
    uint32_t q_begin;      // first local element handled by this launch
    uint32_t q_count;      // number of local elements handled by this launch (multiple of V)
    TileMap tile;
    int32_t batch, channels, n_points;
    NormConst norm;       // code -> pixel (un-folded path)
    NormConst index;      // code -> LUT coordinate s = u * (L-1) / max_code (folded path)
    float inv_max_code;   // 1 / max_code (1 for float input)
    float std_value;
    float weight_scale;   // Gaussian scale (30)
    uint32_t flags;
};

template <typename T, int V>
struct alignas(sizeof(T) * V) Packet {
    T v[V];
};

// Arithmetic of one sample, written so that every constant factor is folded out of the loop:
//   dk  = kk (x - 1/2),  kk = sqrt(scale log2 e)          w = exp2(-dk^2)            (= exp(-scale (x-1/2)^2))
//   av  = dk w s'                                           true a = w' sigma           = av * (K / kk) * sig_scale
//   bv  = av y + (w s' f'_u) cq_n,  cq_n = kk top / (K t_n) true b = (w' y + w y') sigma = bv * (K / kk) * sig_scale
// with K = -2 scale, f'_u = df/ds (per unit of LUT coordinate), s' = sigma / sig_scale (the code u, the pixel x,
// 1, or the explicit std), so the loop body has no multiply by scale, top, 1/max_code or std_value.
// FOLD (integer codes only): the pixel value x is never formed; s and dk come straight from the code.
template <typename T, int V, int INTERP, int WEIGHT, int STD, bool FOLD, int PF = 2>
__global__ __launch_bounds__(kBlock) void merge_kernel(const MergeArgs a)
{
    extern __shared__ __align__(16) char lds[];
    constexpr bool kInt = sizeof(T) != 4;
    constexpr bool kRanged = kInt;
    constexpr bool kHasStd = STD != CT_STD_NONE;
    constexpr bool kGauss = WEIGHT == CT_WEIGHT_GAUSS;
    constexpr int kEntry = lut_entry_bytes(INTERP);
    static_assert(!FOLD || kInt, "FOLD is for integer codes");
    const int C = a.channels, L = a.n_points, B = a.batch;
    const int lut_bytes = INTERP == CT_INTERP_NONE ? 0 : C * L * kEntry;
    float *inv_t = reinterpret_cast<float *>(lds + lut_bytes);  // 1 / t_n
    float *cq = inv_t + B;                                      // derivative scale per exposure
    const float top = INTERP == CT_INTERP_NONE ? 1.0f : (float)(L - 1);
    const float kk = sqrtf(a.weight_scale * 1.4426950408889634f);
    const float K = -2.0f * a.weight_scale;

    stage_lut<INTERP, true>(lds, a.lut, C, L);
    for (int n = threadIdx.x; n < B; n += blockDim.x) {
        const float it = (float)(1.0 / a.exposure[n]);
        inv_t[n] = it;
        cq[n] = kGauss ? kk * top * it / K : top * it;
    }
    __syncthreads();

    const uint32_t vec = blockIdx.x * (uint32_t)kBlock + threadIdx.x;
    if (vec * (uint32_t)V >= a.q_count) return;
    const uint32_t q0 = a.q_begin + vec * (uint32_t)V;

    int row_off[V];  // byte offset of each element's LUT row inside the LDS table
    if (a.tile.layout == CT_LAYOUT_NCHW) {
        // planar input: one division and one modulo per packet; the next element's global index is one further (plus
        // the rows of the other bands when the packet runs into the next channel plane), so its row follows by an add
        // and a conditional subtract
        int ch;
        uint32_t qg;
        a.tile.locate(q0, ch, qg);
        uint32_t off = q0 - (uint32_t)ch * a.tile.plane_local;
        int r = (int)(qg % (uint32_t)C);
        const int skip_mod = (int)(a.tile.chan_skip % (uint32_t)C);
#pragma unroll
        for (int e = 0; e < V; ++e) {
            row_off[e] = (INTERP == CT_INTERP_LOOKUP ? ch : r) * L * kEntry;
            int inc = 1;
            if (++off == a.tile.plane_local) {
                off = 0;
                ++ch;
                inc += skip_mod;
            }
            r += inc;
            r = r >= C ? r - C : r;
        }
    } else {
#pragma unroll
        for (int e = 0; e < V; ++e) {
            int ch;
            uint32_t qg;
            a.tile.locate(a.tile.planar_index(q0 + e), ch, qg);
            row_off[e] = lut_row<INTERP>(qg, ch, C) * L * kEntry;
        }
    }

    float W[V], Swy[V];
    double Saa[V], Sab[V], Sbb[V];
#pragma unroll
    for (int e = 0; e < V; ++e) {
        W[e] = 0.0f;
        Swy[e] = 0.0f;
        Saa[e] = 0.0;
        Sab[e] = 0.0;
        Sbb[e] = 0.0;
    }

    const T *src = static_cast<const T *>(a.stack) + q0;
    const float *ssrc = STD == CT_STD_EXPLICIT ? a.std_stack + q0 : nullptr;
    const float dk_mul = FOLD ? kk * a.inv_max_code : kk, dk_add = -0.5f * kk;

    // Software pipeline: PF packets (16-byte loads) are in flight per thread ahead of the one being reduced, and
    // the V LDS gathers of a packet are issued together before any of them is consumed.
    Packet<T, V> ring[PF];  // ring[0] is the packet being reduced; rotation is by register renaming after unroll
    Packet<float, V> sring[STD == CT_STD_EXPLICIT ? PF : 1];
#pragma unroll
    for (int k = 0; k < PF; ++k) {
        const int nn = k < B ? k : B - 1;
        ring[k] = *reinterpret_cast<const Packet<T, V> *>(src + (int64_t)nn * a.image_stride);
        if constexpr (STD == CT_STD_EXPLICIT)
            sring[k] = *reinterpret_cast<const Packet<float, V> *>(ssrc + (int64_t)nn * a.image_stride);
    }
#pragma unroll PF
    for (int n = 0; n < B; ++n) {
        const int nn = n + PF < B ? n + PF : B - 1;  // tail re-loads the last exposure (cache hit, unused)
        // The exposure offset is laundered through an empty asm each iteration: otherwise LLVM proves that the packet
        // consumed in iteration n equals a fresh load of exposure n and re-loads it at the point of use, which
        // deletes the prefetch (seen in the ISA: load, s_waitcnt vmcnt(0), use).
        int64_t opaque_zero = 0;
        asm volatile("" : "+s"(opaque_zero));
        const int64_t eoff = (int64_t)nn * a.image_stride + opaque_zero;
        const Packet<T, V> incoming = *reinterpret_cast<const Packet<T, V> *>(src + eoff);
        Packet<float, V> sincoming;
        if constexpr (STD == CT_STD_EXPLICIT) sincoming = *reinterpret_cast<const Packet<float, V> *>(ssrc + eoff);
        const Packet<T, V> pk = ring[0];
        const Packet<float, V> sp = sring[0];
#pragma unroll
        for (int k = 0; k + 1 < PF; ++k) {
            ring[k] = ring[k + 1];
            if constexpr (STD == CT_STD_EXPLICIT) sring[k] = sring[k + 1];
        }
        ring[PF - 1] = incoming;
        if constexpr (STD == CT_STD_EXPLICIT) sring[PF - 1] = sincoming;
        const float it = inv_t[n];
        const float cqn = cq[n];
        // ---- stage A: pixel / LUT coordinate, issue the LDS gathers ----
        float pxv[V], frv[V], passv[V];
        float ga[V], gb[V], gc[V], gd[V];  // LUT taps (LINEAR: a,b; CATMULL: a..d; LOOKUP: a)
#pragma unroll
        for (int e = 0; e < V; ++e) {
            float px, s;
            if constexpr (FOLD) {
                px = (float)pk.v[e];
                s = __builtin_fmaf(px, a.index.hi, px * a.index.lo);
            } else {
                px = to_pixel<T>(pk.v[e], a.norm);
                s = px * top;
            }
            pxv[e] = px;
            passv[e] = 1.0f;
            frv[e] = 0.0f;
            ga[e] = gb[e] = gc[e] = gd[e] = 0.0f;
            if constexpr (INTERP == CT_INTERP_LOOKUP) {
                float r = rintf(s);
                if constexpr (!kRanged) r = fminf(fmaxf(r, 0.0f), top);
                ga[e] = reinterpret_cast<const float *>(lds + row_off[e])[(int)r];
            } else if constexpr (INTERP != CT_INTERP_NONE) {
                if constexpr (!kRanged) {
                    passv[e] = (s >= 0.0f && s <= top) ? 1.0f : 0.0f;
                    s = fminf(fmaxf(s, 0.0f), top);
                }
                const int i0 = (int)s;  // s >= 0: truncation is floor
                frv[e] = __builtin_amdgcn_fractf(s);
                if constexpr (INTERP == CT_INTERP_LINEAR) {
                    const float2 g = reinterpret_cast<const float2 *>(lds + row_off[e])[i0];
                    ga[e] = g.x;
                    gb[e] = g.y;
                } else {
                    const float4 g = reinterpret_cast<const float4 *>(lds + row_off[e])[i0];
                    ga[e] = g.x;
                    gb[e] = g.y;
                    gc[e] = g.z;
                    gd[e] = g.w;
                }
            }
        }
        // ---- stage B: f(x), weight, running sums ----
#pragma unroll
        for (int e = 0; e < V; ++e) {
            const float px = pxv[e];
            float lin, dfds;
            if constexpr (INTERP == CT_INTERP_NONE) {
                lin = FOLD ? px * a.inv_max_code : px;
                dfds = 1.0f;
            } else if constexpr (INTERP == CT_INTERP_LOOKUP) {
                lin = ga[e];
                dfds = 0.0f;
            } else if constexpr (INTERP == CT_INTERP_LINEAR) {
                dfds = gb[e];  // the staged table holds {g[i], g[i+1] - g[i]}
                lin = __builtin_fmaf(dfds, frv[e], ga[e]);
                if constexpr (!kRanged) dfds *= passv[e];
            } else {
                const float t = frv[e], t2 = t * t, t3 = t2 * t;
                const float w0 = -0.5f * t3 + t2 - 0.5f * t, w1 = 1.5f * t3 - 2.5f * t2 + 1.0f;
                const float w2 = -1.5f * t3 + 2.0f * t2 + 0.5f * t, w3 = 0.5f * t3 - 0.5f * t2;
                lin = ((w0 * ga[e] + w1 * gb[e]) + w2 * gc[e]) + w3 * gd[e];
                const float d0 = __builtin_fmaf(__builtin_fmaf(-1.5f, t, 2.0f), t, -0.5f);
                const float d2 = __builtin_fmaf(__builtin_fmaf(-4.5f, t, 4.0f), t, 0.5f);
                const float d3 = __builtin_fmaf(1.5f, t, -1.0f) * t;
                dfds = __builtin_fmaf(d0, ga[e] - gb[e], __builtin_fmaf(d2, gc[e] - gb[e], d3 * (gd[e] - gb[e])));
                if constexpr (!kRanged) dfds *= passv[e];
            }
            const float y = lin * it;
            float sg = 1.0f;
            if constexpr (STD == CT_STD_EXPLICIT) sg = sp.v[e];
            if constexpr (STD == CT_STD_MULTIPLIER) sg = px;
            if constexpr (kGauss) {
                const float dk = __builtin_fmaf(px, dk_mul, dk_add);
                const float w = __builtin_amdgcn_exp2f(-dk * dk);
                W[e] += w;
                Swy[e] = __builtin_fmaf(w, y, Swy[e]);
                if constexpr (kHasStd) {
                    const float wu = (STD == CT_STD_CONSTANT) ? w : w * sg;
                    const float av = dk * wu;
                    float bv;
                    if constexpr (INTERP == CT_INTERP_LOOKUP)
                        bv = av * y;
                    else if constexpr (INTERP == CT_INTERP_NONE)
                        bv = __builtin_fmaf(av, y, wu * cqn);
                    else
                        bv = __builtin_fmaf(av, y, (wu * dfds) * cqn);
                    // float64 FMAs: the quadratic form below cancels by 1e2..1e4 (LOOKUP: b = a y exactly);
                    // float32 block sums were measured 7 % faster and 1.3e-4 off on such cases -- not worth it.
                    const double ad = (double)av, bd = (double)bv;
                    Saa[e] = __builtin_fma(ad, ad, Saa[e]);
                    Sab[e] = __builtin_fma(ad, bd, Sab[e]);
                    Sbb[e] = __builtin_fma(bd, bd, Sbb[e]);
                }
            } else {
                Swy[e] += y;
                if constexpr (kHasStd) {
                    const float bv = (INTERP == CT_INTERP_NONE ? sg : dfds * sg) * cqn;
                    const double bd = (double)bv;
                    Sbb[e] = __builtin_fma(bd, bd, Sbb[e]);
                }
            }
        }
    }

    const bool first = a.flags & CT_MERGE_FIRST_BATCH;
    const bool finalize = a.flags & CT_MERGE_FINALIZE;
    const bool keep_state = a.mean_state != nullptr;
    // scale of the folded second moments back to true units
    double fs = 1.0;
    if constexpr (kGauss) fs = (double)K / (double)kk;
    if constexpr (STD == CT_STD_CONSTANT) fs *= (double)a.std_value;
    if constexpr (STD == CT_STD_MULTIPLIER) fs *= (double)a.std_value * (FOLD ? (double)a.inv_max_code : 1.0);
    const double sv2 = fs * fs;
    double mean_o[V];
    float std_o[V];
#pragma unroll
    for (int e = 0; e < V; ++e) {
        const uint32_t q = a.tile.planar_index(q0 + e);  // state and outputs are planar (C, H, W)
        float Wb = W[e];
        if constexpr (!kGauss) Wb = (float)B;
        const float Df = Wb + 1e-6f;  // float32 tensor + python float stays float32 (statistics.py:79-80)
        const double D = (double)Df;
        const double mb = (double)Swy[e] / D;
        const float WA = first ? 0.0f : a.sumw_state[q];
        const double meanA = first ? 0.0 : a.mean_state[q];
        const float Wt = WA + Wb;
        const float frac = Wb / Wt;  // float32 division (statistics.py:105)
        const double mean = meanA + (double)frac * (mb - meanA);
        float var = 0.0f;
        if constexpr (kHasStd) {
            const double beta = (double)frac / D;
            const double alpha = ((double)WA / ((double)Wt * (double)Wt)) * (mb - meanA) - beta * mb;
            const double upd = (alpha * alpha * Saa[e] + 2.0 * alpha * beta * Sab[e] + beta * beta * Sbb[e]) * sv2;
            var = (first ? 0.0f : a.var_state[q]) + (float)upd;
        }
        if (keep_state) {
            a.mean_state[q] = mean;
            a.sumw_state[q] = Wt;
            if constexpr (kHasStd) a.var_state[q] = var;
        }
        mean_o[e] = mean;
        std_o[e] = __builtin_amdgcn_sqrtf(var);
    }
    if (finalize && a.tile.layout != CT_LAYOUT_NCHW) {
        // interleaved input: the V elements of this thread belong to different planes -> element-wise stores
#pragma unroll
        for (int e = 0; e < V; ++e) {
            const uint32_t q = a.tile.planar_index(q0 + e);
            if (a.flags & CT_MERGE_MEAN_OUT_F32)
                static_cast<float *>(a.mean_out)[q] = (float)mean_o[e];
            else
                static_cast<double *>(a.mean_out)[q] = mean_o[e];
            if constexpr (kHasStd) a.std_out[q] = std_o[e];
        }
    } else if (finalize) {
        if (a.flags & CT_MERGE_MEAN_OUT_F32) {
            Packet<float, V> o;
#pragma unroll
            for (int e = 0; e < V; ++e) o.v[e] = (float)mean_o[e];
            store_stream(reinterpret_cast<Packet<float, V> *>(static_cast<float *>(a.mean_out) + q0), o);
        } else {
            Packet<double, V> o;
#pragma unroll
            for (int e = 0; e < V; ++e) o.v[e] = mean_o[e];
            store_stream(reinterpret_cast<Packet<double, V> *>(static_cast<double *>(a.mean_out) + q0), o);
        }
        if constexpr (kHasStd) {
            Packet<float, V> o;
#pragma unroll
            for (int e = 0; e < V; ++e) o.v[e] = std_o[e];
            store_stream(reinterpret_cast<Packet<float, V> *>(a.std_out + q0), o);
        }
    }
}

template <typename T, int V, int INTERP, int WEIGHT, int STD>
static int launch_one(const MergeArgs &a, hipStream_t stream, bool fold)
{
    if (a.q_count == 0) return CT_OK;
    const uint32_t vecs = a.q_count / V;
    const uint32_t grid = (vecs + kBlock - 1) / kBlock;
    const size_t lds = (INTERP == CT_INTERP_NONE ? 0 : (size_t)a.channels * a.n_points * lut_entry_bytes(INTERP)) +
                       2 * sizeof(float) * (size_t)a.batch;
    if (lds > 160 * 1024) return CT_ERR_TOO_LARGE;
    if constexpr (sizeof(T) != 4) {
        if (fold)
            hipLaunchKernelGGL((merge_kernel<T, V, INTERP, WEIGHT, STD, true>), dim3(grid), dim3(kBlock), lds, stream, a);
        else
            hipLaunchKernelGGL((merge_kernel<T, V, INTERP, WEIGHT, STD, false>), dim3(grid), dim3(kBlock), lds, stream, a);
    } else {
        hipLaunchKernelGGL((merge_kernel<T, V, INTERP, WEIGHT, STD, false>), dim3(grid), dim3(kBlock), lds, stream, a);
    }
    return hipGetLastError() == hipSuccess ? CT_OK : CT_ERR_LAUNCH;
}

template <typename T, int V, int INTERP, int WEIGHT>
static int dispatch_std(const MergeArgs &a, int std_mode, hipStream_t s, bool fold)
{
    switch (std_mode) {
        case CT_STD_NONE: return launch_one<T, V, INTERP, WEIGHT, CT_STD_NONE>(a, s, fold);
        case CT_STD_CONSTANT: return launch_one<T, V, INTERP, WEIGHT, CT_STD_CONSTANT>(a, s, fold);
        case CT_STD_MULTIPLIER: return launch_one<T, V, INTERP, WEIGHT, CT_STD_MULTIPLIER>(a, s, fold);
        case CT_STD_EXPLICIT: return launch_one<T, V, INTERP, WEIGHT, CT_STD_EXPLICIT>(a, s, fold);
    }
    return CT_ERR_INVALID_ARGUMENT;
}

template <typename T, int V, int INTERP>
static int dispatch_weight(const MergeArgs &a, int weight_mode, int std_mode, hipStream_t s, bool fold)
{
    return weight_mode == CT_WEIGHT_GAUSS ? dispatch_std<T, V, INTERP, CT_WEIGHT_GAUSS>(a, std_mode, s, fold)
                                          : dispatch_std<T, V, INTERP, CT_WEIGHT_NONE>(a, std_mode, s, fold);
}

template <typename T, int V>
static int dispatch_interp(const MergeArgs &a, int interp, int weight_mode, int std_mode, hipStream_t s, bool fold)
{
    switch (interp) {
        case CT_INTERP_LOOKUP: return dispatch_weight<T, V, CT_INTERP_LOOKUP>(a, weight_mode, std_mode, s, fold);
        case CT_INTERP_LINEAR: return dispatch_weight<T, V, CT_INTERP_LINEAR>(a, weight_mode, std_mode, s, fold);
        case CT_INTERP_CATMULL: return dispatch_weight<T, V, CT_INTERP_CATMULL>(a, weight_mode, std_mode, s, fold);
        case CT_INTERP_NONE: return dispatch_weight<T, V, CT_INTERP_NONE>(a, weight_mode, std_mode, s, fold);
    }
    return CT_ERR_INVALID_ARGUMENT;
}

// Elements per thread.  Measured on MI355X (tools/merge_bench.hip, C2 shape, Gaussian + MULTIPLIER std, PF = 2):
// uint16 V=8 (16-byte packets) and V=4 (8-byte) tie within device-to-device noise with the uncertainty on
// (1.20-1.26 ms vs 1.18-1.30 ms: VALU-bound either way) and V=8 is 7 % faster without it (0.72-0.75 vs 0.78-0.81 ms,
// HBM-bound), so 16-byte packets are used.  Rejected on measurement (profiles/r01_harness_*.log): float32 block
// moments (7 % faster, 1.3e-4 parity error), a two-phase variant caching every sample's (a_n, b_n) in registers to
// drop the float64 FMAs (exact, but 256 VGPRs and 4-byte loads: 2.4 ms), auto-SLP packed float32 (10 % slower).
template <typename T>
struct VecWidth {
    // uint8: 8 codes (8-byte loads); uint16: 4 codes (8-byte loads) -- measured 6 % faster than 8 codes per thread
    // in sustained runs (1.21 vs 1.28 ms on C2: 71 instead of 160 VGPRs); float32: 4 pixels (16-byte loads)
    static constexpr int value = sizeof(T) == 1 ? 8 : 4;
};

template <typename T>
static int merge_typed(MergeArgs a, uint32_t Q, int interp, int weight_mode, int std_mode, hipStream_t s, bool fold)
{
    constexpr int V = VecWidth<T>::value;
    // The packet path needs every packet naturally aligned in every exposure: base pointers and the image
    // stride multiples of the packet.  Anything else (odd widths, ragged tiles) goes through the V = 1 kernel;
    // a ragged tail of an otherwise aligned stack is a second, tiny V = 1 launch.
    auto aligned = [](const void *p, size_t bytes) { return p == nullptr || (reinterpret_cast<uintptr_t>(p) % bytes) == 0; };
    const bool vec_ok = aligned(a.stack, sizeof(T) * V) && (a.image_stride % V) == 0 && aligned(a.std_stack, 4 * V) &&
                        aligned(a.mean_state, 8 * V) && aligned(a.sumw_state, 4 * V) && aligned(a.var_state, 4 * V) &&
                        aligned(a.mean_out, 8 * V) && aligned(a.std_out, 4 * V);
    uint32_t q_vec = vec_ok ? (Q / V) * V : 0;
    int rc = CT_OK;
    if (q_vec) {
        a.q_begin = 0;
        a.q_count = q_vec;
        rc = dispatch_interp<T, V>(a, interp, weight_mode, std_mode, s, fold);
        if (rc != CT_OK) return rc;
    }
    if (q_vec < Q) {
        a.q_begin = q_vec;
        a.q_count = Q - q_vec;
        rc = dispatch_interp<T, 1>(a, interp, weight_mode, std_mode, s, fold);
    }
    return rc;
}

}  // namespace ct

// Host check that fma(u, hi, u*lo) == u / max_code for every code (see NormConst in ct_device.hpp).
extern "C" int ct_norm_constants(float max_code, float *hi, float *lo);
extern "C" int ct_index_constants(float max_code, int n_points, float *hi, float *lo);

extern "C" int ct_hdr_merge_batch(const void *stack_dev, int32_t dtype, float max_code, int32_t batch,
                                  const ct_geometry *geom, const float *std_dev, int32_t std_mode, float std_value,
                                  const double *exposure_dev, const ct_icrf *icrf, int32_t weight_mode,
                                  double *mean_state_dev, float *sumw_state_dev, float *var_state_dev,
                                  void *mean_out_dev, float *std_out_dev, uint32_t flags, void *stream)
{
    using namespace ct;
    if (!stack_dev || !geom || !icrf || !exposure_dev || batch <= 0) return CT_ERR_INVALID_ARGUMENT;
    if (geom->channels <= 0 || geom->h_tile <= 0 || geom->width <= 0 || geom->h_global < geom->h_tile ||
        geom->row_offset < 0 || geom->row_offset + geom->h_tile > geom->h_global)
        return CT_ERR_INVALID_ARGUMENT;
    if (geom->layout < CT_LAYOUT_NCHW || geom->layout > CT_LAYOUT_NHWC_BGR) return CT_ERR_INVALID_ARGUMENT;
    const int interp = icrf->interp;
    if (interp < CT_INTERP_LOOKUP || interp > CT_INTERP_NONE) return CT_ERR_INVALID_ARGUMENT;
    if (interp != CT_INTERP_NONE && (!icrf->lut_dev || icrf->n_points < 2)) return CT_ERR_INVALID_ARGUMENT;
    if (std_mode < CT_STD_NONE || std_mode > CT_STD_EXPLICIT) return CT_ERR_INVALID_ARGUMENT;
    if (std_mode == CT_STD_EXPLICIT && !std_dev) return CT_ERR_INVALID_ARGUMENT;
    if (weight_mode != CT_WEIGHT_NONE && weight_mode != CT_WEIGHT_GAUSS) return CT_ERR_INVALID_ARGUMENT;
    // hdr_merge.py:107-113: autograd.grad raises when nothing connects the mean to the image
    if (std_mode != CT_STD_NONE && interp == CT_INTERP_LOOKUP && weight_mode == CT_WEIGHT_NONE)
        return CT_ERR_NO_GRADIENT_PATH;
    const bool first = flags & CT_MERGE_FIRST_BATCH, finalize = flags & CT_MERGE_FINALIZE;
    const bool has_state = mean_state_dev && sumw_state_dev && (std_mode == CT_STD_NONE || var_state_dev);
    if (!has_state && !(first && finalize)) return CT_ERR_INVALID_ARGUMENT;
    if (finalize && (!mean_out_dev || (std_mode != CT_STD_NONE && !std_out_dev))) return CT_ERR_INVALID_ARGUMENT;

    const int64_t plane_g = geom->h_global * geom->width, plane_l = geom->h_tile * geom->width;
    const int64_t Qg = plane_g * geom->channels, Ql = plane_l * geom->channels;
    if (Qg >= (int64_t)1 << 31) return CT_ERR_TOO_LARGE;
    if (geom->image_stride < Ql) return CT_ERR_INVALID_ARGUMENT;

    MergeArgs a{};
    a.stack = stack_dev;
    a.std_stack = std_mode == CT_STD_EXPLICIT ? std_dev : nullptr;
    a.exposure = exposure_dev;
    a.lut = icrf->lut_dev;
    a.mean_state = has_state ? mean_state_dev : nullptr;
    a.sumw_state = has_state ? sumw_state_dev : nullptr;
    a.var_state = has_state ? var_state_dev : nullptr;
    a.mean_out = mean_out_dev;
    a.std_out = std_out_dev;
    a.image_stride = geom->image_stride;
    a.tile.plane_local = (uint32_t)plane_l;
    a.tile.chan_skip = (uint32_t)(plane_g - plane_l);
    a.tile.base = (uint32_t)(geom->row_offset * geom->width);
    a.tile.layout = (uint32_t)geom->layout;
    a.tile.channels = (uint32_t)geom->channels;
    a.batch = batch;
    a.channels = geom->channels;
    a.n_points = interp == CT_INTERP_NONE ? 2 : icrf->n_points;
    a.std_value = std_value;
    a.weight_scale = 30.0f;  // gaussian_value_weights default scale, hdr_merge.py:95
    a.inv_max_code = 1.0f;
    a.flags = flags;
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (dtype) {
        case CT_DTYPE_U8:
        case CT_DTYPE_U16: {
            if (ct_norm_constants(max_code, &a.norm.hi, &a.norm.lo) != CT_OK) return CT_ERR_UNSUPPORTED;
            // FOLD needs the LUT index formed from the code to equal the reference's float32 index for every code
            const bool fold = ct_index_constants(max_code, a.n_points, &a.index.hi, &a.index.lo) == CT_OK;
            a.inv_max_code = (float)(1.0 / (double)max_code);
            return dtype == CT_DTYPE_U8 ? merge_typed<uint8_t>(a, (uint32_t)Ql, interp, weight_mode, std_mode, s, fold)
                                        : merge_typed<uint16_t>(a, (uint32_t)Ql, interp, weight_mode, std_mode, s, fold);
        }
        case CT_DTYPE_F32: return merge_typed<float>(a, (uint32_t)Ql, interp, weight_mode, std_mode, s, false);
    }
    return CT_ERR_UNSUPPORTED;
}
