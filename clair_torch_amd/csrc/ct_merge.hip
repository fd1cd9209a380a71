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
#include "ct_merge.hpp"
#include <type_traits>

// The file compiles as two translation units so that the two halves of its (many) kernel instantiations build in parallel:
// CT_MERGE_PART 0 (this file as given to the compiler): everything but the several-batches-per-launch instantiations of
// merge_pivot_kernel; CT_MERGE_PART 1 (ct_merge_multi.hip = this file with the macro set): those, behind
// ct::merge_pivot_multi.  CT_MERGE_PART 2 (tools/merge_bench.hip, which includes this file verbatim): both.
#ifndef CT_MERGE_PART
#define CT_MERGE_PART 0
#endif

namespace ct {

// Arithmetic of one sample, written so that every constant factor is folded out of the loop:
//   dk  = kk (x - 1/2),  kk = sqrt(scale log2 e)          w = exp2(-dk^2)            (= exp(-scale (x-1/2)^2))
//   av  = dk w s'                                           true a = w' sigma           = av * (K / kk) * sig_scale
//   bv  = av y + (w s' f'_u) cq_n,  cq_n = kk top / (K t_n) true b = (w' y + w y') sigma = bv * (K / kk) * sig_scale
// with K = -2 scale, f'_u = df/ds (per unit of LUT coordinate), s' = sigma / sig_scale (the code u, the pixel x,
// 1, or the explicit std), so the loop body has no multiply by scale, top, 1/max_code or std_value.
// FOLD (integer codes only): the pixel value x is never formed; s and dk come straight from the code.
// PIVOT (the default since round 3): the second moments are float32 sums about a per-pixel pivot p ~ m_b, exactly as in
// merge_pivot_kernel below (c_n = b_n - p a_n; Saa, Sac, Scc; conditioning check and one repeat about the known mean) --
// the float64 moments Saa, Sab, Sbb of round 1 remain behind CT_MERGE_F64_MOMENTS as the independent comparand of the
// tests.  Besides being cheaper, the pivoted form is the more accurate one where it matters: b_n = a_n y_n is rounded to
// float32 before the float64 sums ever see it, and for LOOKUP (b = a y exactly) the whole variance is the cancelling
// part; y_n - p by one FMA does not lose those bits (LOOKUP against the recorded vectors: 1.25e-5 -> ~5e-6).
template <typename T, int V, int INTERP, int WEIGHT, int STD, bool FOLD, int PF = 2, bool PIVOT = true>
__global__ __launch_bounds__(kBlock) void merge_kernel(const MergeArgs a)
{
    extern __shared__ __align__(16) char lds[];
    constexpr bool kInt = sizeof(T) != 4;
    constexpr bool kRanged = kInt;
    constexpr bool kHasStd = STD != CT_STD_NONE;
    constexpr bool kGauss = WEIGHT == CT_WEIGHT_GAUSS;
    constexpr int kEntry = lut_entry_bytes(INTERP);
    static_assert(!FOLD || kInt, "FOLD is for integer codes");
    using Moment = std::conditional_t<PIVOT, float, double>;
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

    const T *src = static_cast<const T *>(a.stack) + q0;
    const float *ssrc = STD == CT_STD_EXPLICIT ? a.std_stack + q0 : nullptr;
    const float dk_mul = FOLD ? kk * a.inv_max_code : kk, dk_add = -0.5f * kk;
    const bool first = a.flags & CT_MERGE_FIRST_BATCH;
    const bool finalize = a.flags & CT_MERGE_FINALIZE;
    const bool keep_state = a.mean_state != nullptr;

    // one sample: pixel (or raw code when FOLD), f(x), df/ds per unit of LUT coordinate
    auto sample = [&](T code, int roff, float &px, float &lin, float &dfds) {
        float s;
        if constexpr (FOLD) {
            px = (float)code;
            s = __builtin_fmaf(px, a.index.hi, px * a.index.lo);
        } else {
            px = to_pixel<T>(code, a.norm);
            s = px * top;
        }
        if constexpr (INTERP == CT_INTERP_NONE) {
            lin = FOLD ? px * a.inv_max_code : px;
            dfds = 1.0f;
        } else if constexpr (INTERP == CT_INTERP_LOOKUP) {
            float r = rintf(s);
            r = kRanged ? fminf(r, top) : fminf(fmaxf(r, 0.0f), top);  // codes are >= 0: only the upper clamp can act
            lin = reinterpret_cast<const float *>(lds + roff)[(int)r];
            dfds = 0.0f;
        } else {
            float pass = 1.0f;
            if constexpr (!kRanged) {
                pass = (s >= 0.0f && s <= top) ? 1.0f : 0.0f;
                s = fminf(fmaxf(s, 0.0f), top);
            } else {
                // a code above max_code: clamp to the top of the LUT like the reference (base.py:166,190); LINEAR's
                // last staged interval has zero slope, CATMULL needs the explicit gradient mask
                if constexpr (INTERP == CT_INTERP_CATMULL) pass = s <= top ? 1.0f : 0.0f;
                s = fminf(s, top);
            }
            const int i0 = (int)s;  // s >= 0: truncation is floor
            const float fr = __builtin_amdgcn_fractf(s);
            if constexpr (INTERP == CT_INTERP_LINEAR) {
                const float2 g = reinterpret_cast<const float2 *>(lds + roff)[i0];  // {g[i], g[i+1] - g[i]}
                dfds = g.y;
                lin = __builtin_fmaf(dfds, fr, g.x);
                if constexpr (!kRanged) dfds *= pass;
            } else {
                const float4 g = reinterpret_cast<const float4 *>(lds + roff)[i0];
                const float t = fr, t2 = t * t, t3 = t2 * t;
                const float w0 = -0.5f * t3 + t2 - 0.5f * t, w1 = 1.5f * t3 - 2.5f * t2 + 1.0f;
                const float w2 = -1.5f * t3 + 2.0f * t2 + 0.5f * t, w3 = 0.5f * t3 - 0.5f * t2;
                lin = ((w0 * g.x + w1 * g.y) + w2 * g.z) + w3 * g.w;
                const float d0 = __builtin_fmaf(__builtin_fmaf(-1.5f, t, 2.0f), t, -0.5f);
                const float d2 = __builtin_fmaf(__builtin_fmaf(-4.5f, t, 4.0f), t, 0.5f);
                const float d3 = __builtin_fmaf(1.5f, t, -1.0f) * t;
                dfds = __builtin_fmaf(d0, g.x - g.y, __builtin_fmaf(d2, g.z - g.y, d3 * (g.w - g.y)));
                dfds *= pass;
            }
        }
    };

    // ---- pivot (PIVOT): the running mean of the earlier batches, else the middle exposure's sample ----
    [[maybe_unused]] float p[V];
    if constexpr (PIVOT) {
        if (first) {
            const int probe = B / 2;
            const Packet<T, V> pk = *reinterpret_cast<const Packet<T, V> *>(src + (int64_t)probe * a.image_stride);
            const float itp = inv_t[probe];
#pragma unroll
            for (int e = 0; e < V; ++e) {
                float px, lin, dfds;
                sample(pk.v[e], row_off[e], px, lin, dfds);
                p[e] = lin * itp;
            }
        } else {
#pragma unroll
            for (int e = 0; e < V; ++e) p[e] = (float)a.mean_state[a.out_index(q0 + e)];
        }
    }

    double mean_o[V];
    float std_o[V];
    for (int pass_no = 0;; ++pass_no) {
    float W[V], Swy[V];
    Moment Saa[V], Sab[V], Sbb[V];  // PIVOT: Saa, Sac, Scc about the pivot (float32); else the raw float64 moments
#pragma unroll
    for (int e = 0; e < V; ++e) {
        W[e] = 0.0f;
        Swy[e] = 0.0f;
        Saa[e] = 0;
        Sab[e] = 0;
        Sbb[e] = 0;
    }

    // Software pipeline: PF packets (16-byte loads) are in flight per thread ahead of the one being reduced.
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
        float pxv[V], linv[V], dfv[V];
#pragma unroll
        for (int e = 0; e < V; ++e) sample(pk.v[e], row_off[e], pxv[e], linv[e], dfv[e]);  // the V LDS gathers issue together
#pragma unroll
        for (int e = 0; e < V; ++e) {
            const float px = pxv[e], lin = linv[e], dfds = dfv[e];
            const float y = PIVOT ? __builtin_fmaf(lin, it, -p[e]) : lin * it;  // PIVOT: y_n - p
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
                    float bv;  // PIVOT: c_n = b_n - p a_n
                    if constexpr (INTERP == CT_INTERP_LOOKUP)
                        bv = av * y;
                    else if constexpr (INTERP == CT_INTERP_NONE)
                        bv = __builtin_fmaf(av, y, wu * cqn);
                    else
                        bv = __builtin_fmaf(av, y, (wu * dfds) * cqn);
                    if constexpr (PIVOT) {
                        Saa[e] = __builtin_fmaf(av, av, Saa[e]);
                        Sab[e] = __builtin_fmaf(av, bv, Sab[e]);
                        Sbb[e] = __builtin_fmaf(bv, bv, Sbb[e]);
                    } else {
                        // float64 FMAs: the quadratic form below cancels by 1e2..1e4 (LOOKUP: b = a y exactly)
                        const double ad = (double)av, bd = (double)bv;
                        Saa[e] = __builtin_fma(ad, ad, Saa[e]);
                        Sab[e] = __builtin_fma(ad, bd, Sab[e]);
                        Sbb[e] = __builtin_fma(bd, bd, Sbb[e]);
                    }
                }
            } else {
                Swy[e] += y;
                if constexpr (kHasStd) {
                    const float bv = (INTERP == CT_INTERP_NONE ? sg : dfds * sg) * cqn;
                    if constexpr (PIVOT) {
                        Sbb[e] = __builtin_fmaf(bv, bv, Sbb[e]);
                    } else {
                        const double bd = (double)bv;
                        Sbb[e] = __builtin_fma(bd, bd, Sbb[e]);
                    }
                }
            }
        }
    }

    // scale of the folded second moments back to true units
    double fs = 1.0;
    if constexpr (kGauss) fs = (double)K / (double)kk;
    if constexpr (STD == CT_STD_CONSTANT) fs *= (double)a.std_value;
    if constexpr (STD == CT_STD_MULTIPLIER) fs *= (double)a.std_value * (FOLD ? (double)a.inv_max_code : 1.0);
    const double sv2 = fs * fs;
    bool any_bad = false;
    [[maybe_unused]] float mb_f[V];
    [[maybe_unused]] bool bad[V];
#pragma unroll
    for (int e = 0; e < V; ++e) {
        const uint32_t q = a.out_index(q0 + e);  // state and outputs are planar (C, H, W) unless CT_MERGE_OUT_AS_INPUT
        float Wb = W[e];
        if constexpr (!kGauss) Wb = (float)B;
        const float Df = Wb + 1e-6f;  // float32 tensor + python float stays float32 (statistics.py:79-80)
        const float WA = first ? 0.0f : a.sumw_state[q];
        const double meanA = first ? 0.0 : a.mean_state[q];
        const float Wt = WA + Wb;
        const float frac = Wb / Wt;  // float32 division (statistics.py:105)
        double mean;
        float var = 0.0f;
        if constexpr (PIVOT) {
            // as merge_pivot_kernel's epilogue: m_b - p from the sums about the pivot, variance from the three float32 moments
            float r = __builtin_amdgcn_rcpf(Df);
            r = r * __builtin_fmaf(-Df, r, 2.0f);
            const float num = __builtin_fmaf(-p[e], 1e-6f, Swy[e]);  // sum w y - p (W + 1e-6)
            float qd = num * r;
            qd = __builtin_fmaf(__builtin_fmaf(-qd, Df, num), r, qd);  // m_b - p
            const double diff = ((double)p[e] - meanA) + (double)qd;   // m_b - mean_A
            mean = __builtin_fma((double)frac, diff, meanA);
            mb_f[e] = p[e] + qd;
            bad[e] = false;
            if constexpr (kHasStd) {
                const float gam = first ? 0.0f : (WA / (Wt * Wt)) * (float)diff;
                const float beta = frac * r;
                const float kap = __builtin_fmaf(-beta, qd, gam);
                const float t1 = beta * beta * Sbb[e];
                const float t2 = 2.0f * beta * kap * Sab[e];
                const float t3 = kap * kap * Saa[e];
                const float upd = (t1 + t2) + t3;
                if constexpr (kGauss) bad[e] = (t1 + fabsf(t2)) + t3 > kPivotCondLimit * upd;
                var = (first ? 0.0f : a.var_state[q]) + fmaxf(upd, 0.0f) * (float)sv2;
            }
            any_bad |= bad[e];
        } else {
            const double D = (double)Df;
            const double mb = (double)Swy[e] / D;
            mean = meanA + (double)frac * (mb - meanA);
            if constexpr (kHasStd) {
                const double beta = (double)frac / D;
                const double alpha = ((double)WA / ((double)Wt * (double)Wt)) * (mb - meanA) - beta * mb;
                const double upd = (alpha * alpha * (double)Saa[e] + 2.0 * alpha * beta * (double)Sab[e] + beta * beta * (double)Sbb[e]) * sv2;
                var = (first ? 0.0f : a.var_state[q]) + (float)upd;
            }
        }
        mean_o[e] = mean;
        std_o[e] = var;  // the variance until the stores below
        W[e] = Wt;       // (re-used as the output total weight)
    }
    if constexpr (PIVOT) {
        // an ill-conditioned pivot anywhere in the wavefront: repeat the batch once with those elements' pivot at the now
        // known mean; the others recompute bit-identically, so an element's result does not depend on its neighbours
        if (pass_no == 0 && __any(any_bad)) {
#pragma unroll
            for (int e = 0; e < V; ++e) p[e] = bad[e] ? mb_f[e] : p[e];
            continue;
        }
    }
    if (keep_state) {
#pragma unroll
        for (int e = 0; e < V; ++e) {
            const uint32_t q = a.out_index(q0 + e);
            a.mean_state[q] = mean_o[e];
            a.sumw_state[q] = W[e];
            if constexpr (kHasStd) a.var_state[q] = std_o[e];
        }
    }
    break;
    }
#pragma unroll
    for (int e = 0; e < V; ++e) std_o[e] = __builtin_amdgcn_sqrtf(std_o[e]);
    if (finalize && a.tile.layout != CT_LAYOUT_NCHW && !(a.flags & CT_MERGE_OUT_AS_INPUT)) {
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

// =====================================================================================================================
// Round 2 kernel for raw integer codes (LINEAR / no model): single-precision moments about a per-pixel pivot.
//
// Why: the kernel above is VALU-bound on the 2 B/sample headline form (profiles/r01_merge_c2_sq_counters.md), and 45 %
// of its loop is the float64 moments plus the float -> index conversions; another 20 % of all VALU work sat outside
// the loop (per-workgroup LUT staging with integer divides, three IEEE float64 divisions per output element).  Here:
//   * moments about a pivot p ~ m_b:   c_n = b_n - p a_n = w'_n sigma_n (y_n - p) + w_n y'_n sigma_n   (float32)
//       sum (alpha a_n + beta b_n)^2 = beta^2 Scc + 2 beta kappa Sac + kappa^2 Saa,   kappa = gamma - beta (m_b - p),
//       gamma = (W_A / Wt^2)(m_b - mean_A).  With |m_b - p| << m_b the expansion no longer cancels (for LINEAR
//       |a_n (m_b - p)| <= 13.6 |m_b - p| / m_b times the y' term), so float32 sums carry it.  The pivot is the running
//       mean of the earlier batches, or for a first batch the sample of the middle exposure.  Every output element
//       checks its own conditioning (sum of |terms| against the result); a wavefront with an ill-conditioned element
//       runs the batch a second time about the now known mean (explicit fallback, exact to float32 rounding).
//   * the batch mean is accumulated about the same pivot: sum w (y - p), mean = p + ..., in float64 only at the end;
//   * the codes reach the registers as floats through typed buffer loads (conversion in the texture-data path, not on
//     the VALU; ct_device.hpp), the LUT interval floor(code / step) is the mantissa of ONE FMA that rounds toward minus
//     infinity (host-verified for every code against the reference's float32 index), and the 8-byte LUT entry {A, S}
//     gives f = A + S * code in one more FMA: no float coordinate, no fract, no float -> int, no integer -> float;
//   * persistent workgroups: the table and 1/t are staged once per workgroup, not once per 1024 elements;
//   * the epilogue has no float64 division (one v_rcp_f32 + Newton step, shared by the mean and the variance).
// LOOKUP (b = a y exactly: the whole variance is the cancelling part) and CATMULL stay on the float64 kernel above.
// Raw-code build (CT_PIVOT_TYPED_LOAD=0, kept for A/B: profiles/r02_typed_load_ab.log): element e of a packet of raw
// codes as float, and its LUT interval, straight from the packed dwords -- on uint16 both are one SDWA instruction
// (v_cvt_f32_u32 / v_mul_hi_u32_u24 with a word select), so the codes are never unpacked.  (Left to itself LLVM unpacks
// with v_and / v_lshrrev first because the code has two users, and turns ((code * M) >> 32) << 4 into a 64-bit alignbit
// + and + add.)
template <int WORD>
__device__ __forceinline__ float word_to_float(uint32_t dw)
{
    float r;
    if constexpr (WORD == 0)
        asm("v_cvt_f32_u32_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0" : "=v"(r) : "v"(dw));
    else
        asm("v_cvt_f32_u32_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(r) : "v"(dw));
    return r;
}
template <int WORD>
__device__ __forceinline__ uint32_t word_mul_hi_u24(uint32_t dw, uint32_t mul)
{
    uint32_t r;
    if constexpr (WORD == 0)
        asm("v_mul_hi_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD"
            : "=v"(r) : "v"(dw), "s"(mul));
    else
        asm("v_mul_hi_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD"
            : "=v"(r) : "v"(dw), "s"(mul));
    return r;
}
template <typename T, int V, int E>
__device__ __forceinline__ float code_to_float(const Packet<T, V> &pk)
{
    if constexpr (sizeof(T) == 2 && V % 2 == 0)
        return word_to_float<E & 1>(reinterpret_cast<const uint32_t *>(&pk)[E >> 1]);
    else
        return (float)pk.v[E];
}
// floor(code * (L-1) / max_code) (uint16, by the host-verified multiplier) or the code itself (uint8)
template <typename T, int V, int E>
__device__ __forceinline__ uint32_t code_to_interval(const Packet<T, V> &pk, uint32_t mul)
{
    if constexpr (sizeof(T) == 2 && V % 2 == 0)
        return word_mul_hi_u24<E & 1>(reinterpret_cast<const uint32_t *>(&pk)[E >> 1], mul);
    else if constexpr (sizeof(T) == 2)
        return (uint32_t)(((uint64_t)pk.v[E] * (uint64_t)(mul & 0xffffffu)) >> 32);
    else
        return pk.v[E];
}

// One packet through a buffer descriptor based at `base` (wave-uniform) + a 32-bit per-thread byte offset:
// buffer_load_* v, v_offset, s[descriptor], 0 offen.  The descriptor spans 4 GiB from the base, so the offset (an
// element index inside ONE image times the element size) must stay below that -- the callers check.
#ifndef CT_STACK_LOAD_AUX
#define CT_STACK_LOAD_AUX 0  // MUBUF cache-policy bits of the stack loads (bit 0 sc0, bit 1 nt, bit 4 sc1)
#endif
template <typename P>
__device__ __forceinline__ P load_buffer(uint64_t base, uint32_t offset)
{
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(base), 0, 0xffffffff, 0x00020000 /* gfx9 raw dword format */);
    P out;
    if constexpr (sizeof(P) == 1) {
        const uint8_t v = __builtin_amdgcn_raw_buffer_load_b8(rsrc, offset, 0, CT_STACK_LOAD_AUX);
        __builtin_memcpy(&out, &v, sizeof(P));
    } else if constexpr (sizeof(P) == 2) {
        const uint16_t v = __builtin_amdgcn_raw_buffer_load_b16(rsrc, offset, 0, CT_STACK_LOAD_AUX);
        __builtin_memcpy(&out, &v, sizeof(P));
    } else if constexpr (sizeof(P) == 4) {
        const uint32_t v = __builtin_amdgcn_raw_buffer_load_b32(rsrc, offset, 0, CT_STACK_LOAD_AUX);
        __builtin_memcpy(&out, &v, sizeof(P));
    } else if constexpr (sizeof(P) == 8) {
        typedef uint32_t vec_t __attribute__((ext_vector_type(2)));
        const vec_t v = __builtin_amdgcn_raw_buffer_load_b64(rsrc, offset, 0, CT_STACK_LOAD_AUX);
        __builtin_memcpy(&out, &v, sizeof(P));
    } else {
        static_assert(sizeof(P) == 16, "packets are at most 16 bytes");
        typedef uint32_t vec_t __attribute__((ext_vector_type(4)));
        const vec_t v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, offset, 0, CT_STACK_LOAD_AUX);
        __builtin_memcpy(&out, &v, sizeof(P));
    }
    return out;
}

// Typed buffer loads (load_codes_as_float, ct_device.hpp): the codes reach the registers as floats, converted in the
// texture-data path instead of by one half-rate v_cvt_f32_u32_sdwa per sample on the VALU this kernel is bound by.
#ifndef CT_PIVOT_TYPED_LOAD
#define CT_PIVOT_TYPED_LOAD 1
#endif
template <typename T, int V>
__device__ __forceinline__ Packet<float, V> load_codes_as_float(uint64_t base, uint32_t offset)
{
    Packet<float, V> out;
    load_codes_as_float<T, V, CT_STACK_LOAD_AUX>(base, offset, out.v);
    return out;
}

// (floor_index_bits, lds_row_constant, lds_entry_address: ct_device.hpp -- shared with the training kernels)

// compile-time loop: f(std::integral_constant<int, 0>{}) ... f(<N-1>)
template <int N, int I = 0, typename F>
__device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<N, I + 1>(f);
    }
}

struct PivotArgs {
    uint32_t index_mul;  // floor(code * (L-1) / max_code) == (code * index_mul) >> 32 for every code (raw-load build, uint16 only)
    float step;          // max_code / (L-1): codes per LUT interval (any positive value; a whole number on the headline shapes)
    uint32_t n_tiles;    // tiles of kBlock * V elements
    int32_t probe;       // exposure whose sample seeds the pivot of a first batch
    float index_rcp;     // table entry of a code = floor(code * index_rcp) by one round-down FMA (typed-load path): (L-1) / max_code
                         // for LINEAR, 2 (L-1) / max_code for LOOKUP (half intervals), rounded so that EVERY code the container
                         // can hold lands in the reference's entry (ct_pivot_interval_constants)
    float max_code;      // what Normalize divides by
    float tf_max;        // CLAMP: 1.5 * 2^23 + last table entry (codes above max_code clamp to the top of the LUT, base.py:166)
    uint32_t n_entries;  // table entries per LUT row: L (LINEAR), 2 L (LOOKUP)
    unsigned long long *retry_count;  // diagnostics: wavefronts that ran the fallback pass (may be NULL)
    // MULTI (ct_hdr_merge_batches): several consecutive batches per launch, the streaming state in registers in between
    int32_t n_batches;                          // 1 .. kMaxMultiBatches
    int32_t fresh;                              // the first batch starts a merge (no state to read)
    int32_t batch_size[16];                     // exposures per batch; a.batch = their sum, a.exposure in the same order
    const void *batch_ptr[16];                  // each batch's (B_b, C, H_tile, W) stack
    const float *std_ptr[16];                   // CT_STD_EXPLICIT: each batch's std stack
    // interleaved RGB / BGR input with planar packet stores (see the kernel's epilogue): the results of a wavefront are
    // regrouped by channel plane through its 3 KB staging area at `stage_off`
    int32_t rgb252;
    uint32_t stage_off;                         // LDS byte offset of the 4 x 3072-byte staging areas
};
constexpr int kMaxMultiBatches = 16;
constexpr int kPivotV = 4;  // elements per thread of merge_pivot_kernel (uint16: 8-byte loads, uint8: 4-byte)

#ifndef CT_PIVOT_DEPTH
#define CT_PIVOT_DEPTH 2
#endif
constexpr int kPivotDepth = CT_PIVOT_DEPTH;  // exposures in flight per thread
constexpr float kRoughLimit = 64.0f;  // |A| / max(|g[i]|, |g[i+1]|) above which the table keeps {g[i], S}: error bound 2^-25 * 64 = 2e-6  // sum |terms| / result above which a wavefront repeats the batch about the mean

// Experiments with the weight evaluation (measured in profiles/r03_merge_weight_variants.md):
//   0  shipped: one v_exp_f32 per sample
//   1  fine linear weight table in LDS ((max_code + 1) >> CT_PIVOT_WT_SHIFT entries {Wa, Ws}, w = Wa + Ws * code): a second
//      8-byte gather per sample instead of the squaring multiply and the transcendental
//   2  16-byte LUT entries {A, S, W_i, D_i}: w = W_i * exp2(delta (D_i - m^2 delta)), delta = code - i * step, |exponent| <= 0.17,
//      degree-4 polynomial: one ds_read_b128 instead of ds_read_b64, no transcendental, seven more full-rate instructions
#ifndef CT_PIVOT_WEIGHT
#define CT_PIVOT_WEIGHT 0
#endif
#ifndef CT_PIVOT_WT_SHIFT
#define CT_PIVOT_WT_SHIFT 4
#endif

// The first-batch kernels (no state carried through the loop): 7 wavefronts per SIMD (72 VGPRs).  With the codes held as
// four floats per packet instead of two packed dwords the 64-VGPR build spills 19 registers (1.24 ms); 7 and 6
// wavefronts measure the same within 1 % (0.856 / 0.865 ms sustained, profiles/r02_typed_load_ab.log).  Raw-code
// builds (CT_PIVOT_TYPED_LOAD=0) fit 64.  The state-carrying kernels keep the default allocation.
#ifndef CT_RGB252_WAVES
#define CT_RGB252_WAVES 6
#endif
#ifndef CT_PIVOT_KERNEL_ATTR
#define CT_PIVOT_KERNEL_ATTR __attribute__((amdgpu_waves_per_eu(FIRST && V <= 4 && STD != CT_STD_EXPLICIT ? (CT_PIVOT_TYPED_LOAD ? (RGB252 ? CT_RGB252_WAVES : 7) : 8) : 4, 8)))
#endif
// MULTI: the launch walks x.n_batches consecutive batches per element with (mean, sum of weights, variance) in registers and
// the per-batch recurrence of WBOMean (statistics.py:64-109, state detached after every batch, hdr_merge.py:128) applied
// between them -- bit for bit what one launch per batch gives (the first-batch arithmetic with zero state IS the
// state-carrying arithmetic: W_A = 0 makes frac = 1 and gamma = 0 exactly), without the 32 B per element and batch of
// state traffic the reference's default batch_size: 4 costs beside 8 B of samples.
template <typename T, int V, int INTERP, int WEIGHT, int STD, bool FIRST, bool CLAMP = false, bool MULTI = false, bool RGB252 = false>
__global__ __launch_bounds__(kBlock) CT_PIVOT_KERNEL_ATTR void merge_pivot_kernel(const MergeArgs a, const PivotArgs x)
{
    static_assert(!MULTI || (!FIRST && CT_PIVOT_TYPED_LOAD), "MULTI carries state and uses the typed loads");
    static_assert(!RGB252 || (V == 4 && FIRST && !MULTI && CT_PIVOT_TYPED_LOAD), "RGB252: single-batch packets of the typed-load kernel");
    extern __shared__ __align__(16) char lds[];
    static_assert(sizeof(T) != 4, "raw integer codes only");
    static_assert(CT_PIVOT_TYPED_LOAD || ((INTERP == CT_INTERP_LINEAR || INTERP == CT_INTERP_NONE) && !CLAMP), "raw-load build: whole-step LINEAR only");
    constexpr bool kLut = INTERP != CT_INTERP_NONE;  // a table in LDS
    constexpr bool kLookup = INTERP == CT_INTERP_LOOKUP;  // piecewise constant: entry j = half interval j, slope 0, row = channel
    // CATMULL (r03): entry i holds the interval's cubic in the code offset, f = d + o (c + o (b + o a)), o = code - i step --
    // the Catmull-Rom basis of base.py:199-224 on the taps g[i-1..i+2] (edges replicated) collected by powers of t = o / step
    // in float64 and rounded once; three FMAs for the value, four more instructions for df/dcode.  The closed-form kernel
    // for CATMULL stacks WITHOUT uncertainties (and with CT_MERGE_CLOSED_FORM); the default with uncertainties stays
    // the reference-order kernel.
    constexpr bool kCat = INTERP == CT_INTERP_CATMULL;
    constexpr bool kHasStd = STD != CT_STD_NONE;
    constexpr bool kGauss = WEIGHT == CT_WEIGHT_GAUSS;
    constexpr bool kTyped = CT_PIVOT_TYPED_LOAD;  // codes arrive as floats from typed buffer loads
    using CodePk = std::conditional_t<kTyped, Packet<float, V>, Packet<T, V>>;
    const int C = a.channels, L = a.n_points, B = a.batch;
    constexpr int kWV = (INTERP == CT_INTERP_LINEAR && kGauss && kTyped && sizeof(T) == 2 && V == 4) ? CT_PIVOT_WEIGHT : 0;  // weight evaluation variant
    constexpr int kEntryShift = (kWV == 2 || kCat) ? 4 : 3;
    const int E = kLut ? (int)x.n_entries : 0;  // table entries per row
    const int lut_bytes = C * E * (1 << kEntryShift);
    float2 *expo = reinterpret_cast<float2 *>(lds + lut_bytes);  // per exposure {1 / t_n, chain factor of the y' term}
    [[maybe_unused]] const uint32_t wt_base = (uint32_t)lut_bytes + 8u * (uint32_t)B;  // kWV == 1: the weight table
    const float kk = sqrtf(a.weight_scale * 1.4426950408889634f);
    const float dk_mul = kk * a.inv_max_code, dk_add = -0.5f * kk;
    const float K = -2.0f * a.weight_scale;
    // y' = (df/dcode) max_code / t_n;  the loop forms (w s' df/dcode) * cq_n with cq_n = max_code (kk / K) / t_n
    const float max_code = kLut ? x.max_code : 1.0f;  // (no model: df/dcode * max_code = 1, folded)
    const float ce = kGauss ? max_code * kk / K : max_code;

    bool rough = false;
    if constexpr (kLookup) {
        // entry j of row c covers LUT coordinates [j / 2, (j + 1) / 2): the reference's round-half-even index is (j + 1) / 2
        // for every code (host-verified), so f = g[c][(j + 1) >> 1] and the slope is zero
        const int total = C * E;
        for (int k = threadIdx.x; k < total; k += kBlock) {
            const int r = k / E, j = k - r * E;
            const int idx = (j + 1) >> 1;
            reinterpret_cast<float2 *>(lds)[k] = make_float2(a.lut[(size_t)r * L + (idx < L ? idx : L - 1)], 0.0f);
        }
    } else if constexpr (kCat) {
        const int total = C * L;
        const double st = (double)x.step;
        for (int k = threadIdx.x; k < total; k += kBlock) {
            const int r = k / L, i = k - r * L;
            const float *row = a.lut + (size_t)r * L;
            const double p0 = row[i > 0 ? i - 1 : 0], p1 = row[i], p2 = row[i + 1 < L ? i + 1 : L - 1], p3 = row[i + 2 < L ? i + 2 : L - 1];
            // w0 p0 + w1 p1 + w2 p2 + w3 p3 with the basis of base.py:199-224 = p1 + t c + t^2 b + t^3 a
            const double c1 = 0.5 * (p2 - p0), b1 = 0.5 * (2.0 * p0 - 5.0 * p1 + 4.0 * p2 - p3), a1 = 0.5 * (-p0 + 3.0 * p1 - 3.0 * p2 + p3);
            // (entry L - 1 is met at offset 0 only -- code == max_code, where the reference's clamp still passes the gradient
            // -- or, with CLAMP, by codes above max_code, whose offset and slope are zeroed in the loop)
            reinterpret_cast<float4 *>(lds)[k] = make_float4((float)p1, (float)(c1 / st), (float)(b1 / (st * st)), (float)(a1 / (st * st * st)));
        }
    } else if constexpr (kLut) {
        // entry i of row r: f(code) = A + S * code on [i * step, (i + 1) * step):  S = (g[i+1] - g[i]) / step (the
        // reference backward's g1 - g0), A = g[i] - S * i * step formed in float64 and rounded once.  One FMA per
        // sample, but A carries an absolute rounding error of 2^-25 |A|, and |A| <= |g[i]| + i |g[i+1] - g[i]| exceeds
        // the LUT values themselves when the curve is steep: a factor 1 + p for g = x^p, unbounded for a LUT with a
        // jump.  A workgroup that meets |A| > kRoughLimit max(|g[i]|, |g[i+1]|) anywhere therefore stages {g[i], S}
        // instead and evaluates f = g[i] + S (code - i * step) with the offset formed exactly when the step is a whole
        // number of codes (two more instructions per sample); every workgroup sees the same LUT, so all take the same branch.
        const int total = C * L;
        const double stepd = (double)x.max_code / (double)(L - 1);
        bool viol = false;
        for (int k = threadIdx.x; k < total; k += kBlock) {
            const int r = k / L, i = k - r * L;
            const float *row = a.lut + (size_t)r * L;
            const float g0 = row[i], g1 = row[i + 1 < L ? i + 1 : L - 1];
            const float slope = (g1 - g0) / x.step;
            const float A = (float)((double)g0 - (double)slope * ((double)i * stepd));
            viol |= !(fabsf(A) <= kRoughLimit * fmaxf(fmaxf(fabsf(g0), fabsf(g1)), 1e-30f));
        }
        rough = __syncthreads_or(viol);
        for (int k = threadIdx.x; k < total; k += kBlock) {
            const int r = k / L, i = k - r * L;
            const float *row = a.lut + (size_t)r * L;
            const float g0 = row[i], g1 = row[i + 1 < L ? i + 1 : L - 1];
            const float slope = (g1 - g0) / x.step;
            const float A = (float)((double)g0 - (double)slope * ((double)i * stepd));
            if constexpr (kWV == 2) {
                const double dki = (double)i * (double)x.step * (double)dk_mul + (double)dk_add;
                reinterpret_cast<float4 *>(lds)[k] =
                    make_float4(rough ? g0 : A, slope, (float)exp2(-dki * dki), (float)(-2.0 * dki * (double)dk_mul));
            } else {
                reinterpret_cast<float2 *>(lds)[k] = make_float2(rough ? g0 : A, slope);
            }
        }
    }
    if constexpr (kWV == 1) {
        // entry j covers codes [j << s, (j + 1) << s): w = Wa + Ws * code, chord of exp2(-dk^2) over the interval
        constexpr int s = CT_PIVOT_WT_SHIFT, nW = 65536 >> s;
        for (int j = threadIdx.x; j < nW; j += kBlock) {
            const double c0 = (double)(j << s), c1 = (double)((j + 1) << s);
            const double d0 = c0 * (double)dk_mul + (double)dk_add, d1 = c1 * (double)dk_mul + (double)dk_add;
            const double w0 = exp2(-d0 * d0), w1 = exp2(-d1 * d1), ws = (w1 - w0) / (c1 - c0);
            *reinterpret_cast<float2 *>(lds + wt_base + 8u * (uint32_t)j) = make_float2((float)(w0 - ws * c0), (float)ws);
        }
    }
    for (int n = threadIdx.x; n < B; n += kBlock) {
        const float it = (float)(1.0 / a.exposure[n]);
        expo[n] = make_float2(it, ce * it);
    }
    __syncthreads();  // the only barrier: everything below is per wavefront

    const bool finalize = a.flags & CT_MERGE_FINALIZE;
    const bool keep_state = a.mean_state != nullptr;
    const bool planar = a.tile.layout == CT_LAYOUT_NCHW;
    const bool planar_out = planar || (a.flags & CT_MERGE_OUT_AS_INPUT);  // state / outputs at the memory index itself
    float fsf = 1.0f;  // scale of the folded moments back to true units
    if constexpr (kGauss) fsf = K / kk;
    if constexpr (STD == CT_STD_CONSTANT) fsf *= a.std_value;
    if constexpr (STD == CT_STD_MULTIPLIER) fsf *= a.std_value * a.inv_max_code;
    const float sv2 = fsf * fsf;
    [[maybe_unused]] const uint32_t index_mul = x.index_mul & 0xffffffu;
    [[maybe_unused]] const float index_rcp = x.index_rcp;
    [[maybe_unused]] float floor_magic = kFloorMagic;
    asm volatile("" : "+v"(floor_magic));  // one VGPR for the whole kernel (a VOP3 FMA cannot carry a literal)

    constexpr bool rgb252 = RGB252;  // interleaved RGB / BGR with packet stores: its own instantiation (the regrouping costs
                                     // registers the planar headline kernel, capped at 72, does not have)
    for (uint32_t tile = blockIdx.x; tile < x.n_tiles; tile += gridDim.x) {
        const uint32_t vec = tile * (uint32_t)kBlock + threadIdx.x;
        if (vec * (uint32_t)V >= a.q_count) continue;  // ragged last tile (no barrier below: lanes may leave)
        const uint32_t q0 = a.q_begin + vec * (uint32_t)V;

        int row_off[V];  // byte offset of each element's LUT row inside the LDS table
        if constexpr (kLut) {
            if (planar) {
                // channel by comparisons, row by a constant-divisor modulo when C == 3: a runtime 32-bit division costs
                // ~30 instructions, and this runs once per tile per thread
                int ch = 0;
                for (int c = 1; c < C; ++c) ch += q0 >= (uint32_t)c * a.tile.plane_local ? 1 : 0;
                const uint32_t qg = q0 + (uint32_t)ch * a.tile.chan_skip + a.tile.base;
                uint32_t off = q0 - (uint32_t)ch * a.tile.plane_local;
                int r = C == 3 ? (int)(qg % 3u) : (int)(qg % (uint32_t)C);
                const int skip_mod = (int)(a.tile.chan_skip % (uint32_t)C);
#pragma unroll
                for (int e = 0; e < V; ++e) {
                    row_off[e] = (kLookup ? ch : r) * E * 8;  // LOOKUP: the true channel (base.py:149-155)
                    int inc = 1;
                    if (++off == a.tile.plane_local) {
                        off = 0;
                        ++ch;
                        inc += skip_mod;
                    }
                    r += inc;
                    r = r >= C ? r - C : r;
                }
            } else if (C == 3) {
                // interleaved RGB / BGR: constant divisors (channel = m % 3, pixel = m / 3, row = global index % 3)
                const uint32_t plane_g = a.tile.plane_local + a.tile.chan_skip;
#pragma unroll
                for (int e = 0; e < V; ++e) {
                    uint32_t c, pixel;
                    a.tile.interleaved3(q0 + e, c, pixel);
                    const uint32_t qg = c * plane_g + a.tile.base + pixel;
                    row_off[e] = (int)(kLookup ? c : qg % 3u) * E * 8;
                }
            } else {
#pragma unroll
                for (int e = 0; e < V; ++e) {
                    int ch;
                    uint32_t qg;
                    a.tile.locate(a.tile.planar_index(q0 + e), ch, qg);
                    row_off[e] = (kLookup ? ch : (int)(qg % (uint32_t)C)) * E * 8;
                }
            }
        } else {
#pragma unroll
            for (int e = 0; e < V; ++e) row_off[e] = 0;
        }
        [[maybe_unused]] uint32_t rowc[V];  // typed path: the row offset as the addend of lds_entry_address
        if constexpr (kTyped && kLut) {
#pragma unroll
            for (int e = 0; e < V; ++e) rowc[e] = lds_row_constant<kEntryShift>(row_off[e] << (kEntryShift - 3));
        }

        // loads are addressed as (wave-uniform exposure base) + (32-bit per-thread byte offset): no 64-bit VALU address math
        const uint32_t voff = q0 * (uint32_t)sizeof(T), svoff = q0 * 4u;
        // planar (state / output) index of memory element m: the identity for planar stacks, constant divisors for RGB / BGR
        auto planar_of = [&](uint32_t m) -> uint32_t {
            if (planar_out) return m;
            if (C == 3) {
                uint32_t c, pixel;
                a.tile.interleaved3(m, c, pixel);
                return c * a.tile.plane_local + pixel;
            }
            return a.tile.planar_index(m);
        };

        // ---- pivot: the running mean of the earlier batches, else the middle exposure's sample ----
        constexpr int VS = FIRST ? 1 : V;  // state registers exist only when there is state
        float p[V], WA[VS], varA[VS];
        double meanA[VS];
        [[maybe_unused]] auto probe_pivot = [&](uint64_t stack_base) {  // typed loads: p = the probe exposure's y
            const Packet<float, V> pk = load_codes_as_float<T, V>(
                stack_base + (uint64_t)((int64_t)x.probe * a.image_stride * (int64_t)sizeof(T)), voff);
            const float itp = expo[x.probe].x;
            float tf[V];
            if constexpr (kLut) floor_index_bits<V>(pk.v, index_rcp, floor_magic, tf);
            if constexpr (kLut && CLAMP) {
#pragma unroll
                for (int e = 0; e < V; ++e) tf[e] = fminf(tf[e], x.tf_max);
            }
#pragma unroll
            for (int e = 0; e < V; ++e) {
                const float px = pk.v[e];
                float lin = px * a.inv_max_code;
                if constexpr (kCat) {
                    const float4 g = *reinterpret_cast<const float4 *>(lds + lds_entry_address<kEntryShift>(tf[e], rowc[e]));
                    float o = __builtin_fmaf(tf[e] - floor_magic, -x.step, px);
                    if constexpr (CLAMP) o = px > x.max_code ? 0.0f : o;
                    lin = __builtin_fmaf(__builtin_fmaf(__builtin_fmaf(g.w, o, g.z), o, g.y), o, g.x);
                } else if constexpr (kLut) {
                    const float2 g = *reinterpret_cast<const float2 *>(lds + lds_entry_address<kEntryShift>(tf[e], rowc[e]));
                    lin = __builtin_fmaf(g.y, rough ? __builtin_fmaf(tf[e] - floor_magic, -x.step, px) : px, g.x);
                }
                p[e] = lin * itp;
            }
        };
        if constexpr (FIRST && kTyped) {
            probe_pivot(reinterpret_cast<uint64_t>(a.stack));
        } else if constexpr (FIRST) {
            const Packet<T, V> pk = load_buffer<Packet<T, V>>(
                reinterpret_cast<uint64_t>(a.stack) + (uint64_t)((int64_t)x.probe * a.image_stride * (int64_t)sizeof(T)), voff);
            const float itp = expo[x.probe].x;
            static_for<V>([&](auto ec) {
                constexpr int e = decltype(ec)::value;
                float lin = code_to_float<T, V, e>(pk) * a.inv_max_code;
                if constexpr (kLut) {
                    const uint32_t i0 = code_to_interval<T, V, e>(pk, index_mul);
                    const float2 g = *reinterpret_cast<const float2 *>(lds + (row_off[e] + (int)(i0 << 3)));
                    const float px = code_to_float<T, V, e>(pk);
                    lin = __builtin_fmaf(g.y, rough ? __builtin_fmaf((float)i0, -x.step, px) : px, g.x);
                }
                p[e] = lin * itp;
            });
        } else {
            bool fresh = false;
            if constexpr (MULTI) fresh = x.fresh != 0;
            if (fresh) {  // a new merge: WBOMean starts at mean 0, weight 0 (statistics.py:30-31)
                if constexpr (MULTI) {
#pragma unroll
                    for (int e = 0; e < V; ++e) {
                        meanA[e] = 0.0;
                        WA[e] = 0.0f;
                        varA[e] = 0.0f;
                    }
                    probe_pivot(reinterpret_cast<uint64_t>(x.batch_ptr[0]));
                }
            } else {
#pragma unroll
                for (int e = 0; e < V; ++e) {
                    const uint32_t q = planar_of(q0 + e);
                    meanA[e] = a.mean_state[q];
                    WA[e] = a.sumw_state[q];
                    if constexpr (kHasStd) varA[e] = a.var_state[q];
                    p[e] = (float)meanA[e];
                }
            }
        }

        double mean_o[V];
        float var_o[V], Wt_o[V];
        const int n_batches = MULTI ? x.n_batches : 1;
        int n0 = 0;  // first exposure of the current batch within the launch (a.exposure / the LDS constants)
        for (int bi = 0; bi < n_batches; ++bi) {
        const int Bb = MULTI ? x.batch_size[bi] : B;  // exposures of this batch
        const uint64_t batch_base = MULTI ? reinterpret_cast<uint64_t>(x.batch_ptr[bi]) : reinterpret_cast<uint64_t>(a.stack);
        [[maybe_unused]] const uint64_t batch_std_base = MULTI ? reinterpret_cast<uint64_t>(x.std_ptr[bi]) : reinterpret_cast<uint64_t>(a.std_stack);
        for (int pass = 0;; ++pass) {
            float W[V], Swy[V], Saa[V], Sac[V], Scc[V];
#pragma unroll
            for (int e = 0; e < V; ++e) W[e] = Swy[e] = Saa[e] = Sac[e] = Scc[e] = 0.0f;

            auto run_batch = [&](auto rough_c, auto moments_c) {
            constexpr bool kRough = decltype(rough_c)::value;  // see the staging: exact but slower interval arithmetic
            constexpr bool kMoments = decltype(moments_c)::value;  // false: sum of weights and weighted sum only (kMeanFirst)
            // one exposure of this thread's V elements
            auto reduce = [&](const CodePk &pk, const Packet<float, V> &sp, uint32_t expo_adr) {
                const float2 ex = *reinterpret_cast<const float2 *>(lds + expo_adr);  // {1 / t_n, chain factor} of this exposure
                const float it = ex.x, cqn = ex.y;
                float pxv[V], ga[V], gs[V];
                [[maybe_unused]] float pxl[V];
                [[maybe_unused]] float dkv[V], wv[V];
                [[maybe_unused]] float tf[V], tw[V], gw[V], gd[V];
                if constexpr (kTyped) {
#pragma unroll
                    for (int e = 0; e < V; ++e) pxv[e] = pk.v[e];
                    if constexpr (kLut && kWV == 1)
                        floor_index_bits2<V>(pxv, index_rcp, 1.0f / (float)(1 << CT_PIVOT_WT_SHIFT), floor_magic, tf, tw);
                    else if constexpr (kLut)
                        floor_index_bits<V>(pxv, index_rcp, floor_magic, tf);
                    if constexpr (kLut && CLAMP) {  // a code above max_code: the last entry (top of the LUT, zero slope)
#pragma unroll
                        for (int e = 0; e < V; ++e) tf[e] = fminf(tf[e], x.tf_max);
                    }
                }
                static_for<V>([&](auto ec) {  // stage A: the V table gathers and the V transcendentals, each issued together
                    constexpr int e = decltype(ec)::value;
                    if constexpr (kTyped) {
                        if constexpr (kLut && kWV == 2) {
                            float4 g = *reinterpret_cast<const float4 *>(lds + lds_entry_address<4>(tf[e], rowc[e]));
                            asm volatile("" : "+v"(g.x), "+v"(g.y), "+v"(g.z), "+v"(g.w));  // keep the ds_read_b128 whole
                            ga[e] = g.x;
                            gs[e] = g.y;
                            gw[e] = g.z;
                            gd[e] = g.w;
                            pxl[e] = __builtin_fmaf(tf[e] - floor_magic, -x.step, pxv[e]);  // delta = code - i * step, exact
                        } else if constexpr (kCat) {
                            const float4 g = *reinterpret_cast<const float4 *>(lds + lds_entry_address<4>(tf[e], rowc[e]));
                            float o = __builtin_fmaf(tf[e] - floor_magic, -x.step, pxv[e]);  // code - i * step
                            [[maybe_unused]] bool above = false;  // a code above max_code: the model clamps it to the top, gradient 0
                            if constexpr (CLAMP) {
                                above = pxv[e] > x.max_code;
                                o = above ? 0.0f : o;
                            }
                            ga[e] = __builtin_fmaf(__builtin_fmaf(__builtin_fmaf(g.w, o, g.z), o, g.y), o, g.x);   // f
                            gs[e] = 0.0f;
                            if constexpr (kHasStd) {
                                gs[e] = __builtin_fmaf(__builtin_fmaf(3.0f * g.w, o, g.z + g.z), o, g.y);   // df / dcode
                                if constexpr (CLAMP) gs[e] = above ? 0.0f : gs[e];
                            }
                        } else if constexpr (kLut) {
                            const float2 g = *reinterpret_cast<const float2 *>(lds + lds_entry_address(tf[e], rowc[e]));
                            ga[e] = g.x;
                            gs[e] = g.y;
                            if constexpr (kRough) pxl[e] = __builtin_fmaf(tf[e] - floor_magic, -x.step, pxv[e]);  // code - i * step, exact
                            if constexpr (kWV == 1) {
                                const float2 wt = *reinterpret_cast<const float2 *>(
                                    lds + lds_entry_address(tw[e], wt_base - (kFloorMagicBits << 3)));
                                gw[e] = wt.x;
                                gd[e] = wt.y;
                            }
                        }
                    } else {
                    pxv[e] = code_to_float<T, V, e>(pk);
                    if constexpr (kLut) {
                        const uint32_t i0 = code_to_interval<T, V, e>(pk, index_mul);
                        const float2 g = *reinterpret_cast<const float2 *>(lds + (row_off[e] + (int)(i0 << 3)));
                        ga[e] = g.x;
                        gs[e] = g.y;
                        if constexpr (kRough) pxl[e] = __builtin_fmaf((float)i0, -x.step, pxv[e]);  // code - i * step, exact
                    }
                    }
                    if constexpr (kGauss) {
                        dkv[e] = __builtin_fmaf(pxv[e], dk_mul, dk_add);
                        if constexpr (kWV == 0) wv[e] = __builtin_amdgcn_exp2f(-dkv[e] * dkv[e]);
                    }
                });
                if constexpr (kWV == 1) {
#pragma unroll
                    for (int e = 0; e < V; ++e) wv[e] = __builtin_fmaf(gd[e], pxv[e], gw[e]);
                }
                if constexpr (kWV == 2) {
                    // exp2(v), v = delta (D_i - m^2 delta), |v| <= 0.17: 1 + v (c1 + v (c2 + v (c3 + v c4))), error < 2e-7
                    const float m2 = dk_mul * dk_mul;
#pragma unroll
                    for (int e = 0; e < V; ++e) {
                        const float v = pxl[e] * __builtin_fmaf(pxl[e], -m2, gd[e]);
                        float q = __builtin_fmaf(v, 0.009618129107628477f, 0.05550410866482158f);
                        q = __builtin_fmaf(q, v, 0.2402265069591007f);
                        q = __builtin_fmaf(q, v, 0.6931471805599453f);
                        q = __builtin_fmaf(q, v, 1.0f);
                        wv[e] = gw[e] * q;
                    }
                }
                if constexpr (kGauss && kWV == 0) {
                    // pins the four v_exp_f32 ahead of the dependent arithmetic: measured 4 % faster than letting the
                    // scheduler sink each one next to its first use (profiles/r02_merge_ablation.md)
#pragma unroll
                    for (int e = 0; e < V; ++e) asm volatile("" : "+v"(wv[e]));
                }
#pragma unroll
                for (int e = 0; e < V; ++e) {  // stage B: f, weight, running sums
                    const float px = pxv[e];
                    const float lin = (kLookup || kCat) ? ga[e] : kLut ? __builtin_fmaf(gs[e], kRough ? pxl[e] : px, ga[e]) : px * a.inv_max_code;
                    const float yd = __builtin_fmaf(lin, it, -p[e]);  // y_n - p
                    if constexpr (kGauss) {
                        const float dk = dkv[e], w = wv[e];
                        W[e] += w;
                        Swy[e] = __builtin_fmaf(w, yd, Swy[e]);
                        if constexpr (kHasStd && kMoments) {
                            float wu = w;
                            if constexpr (STD == CT_STD_MULTIPLIER) wu = w * px;
                            if constexpr (STD == CT_STD_EXPLICIT) wu = w * sp.v[e];
                            const float av = dk * wu;
                            float cv;
                            if constexpr (kLookup) {
                                cv = av * yd;  // no gradient through the index: the whole variance is the weight path
                            } else {
                                const float ev = kLut ? (wu * gs[e]) * cqn : wu * cqn;
                                cv = __builtin_fmaf(av, yd, ev);
                            }
                            Saa[e] = __builtin_fmaf(av, av, Saa[e]);
                            Sac[e] = __builtin_fmaf(av, cv, Sac[e]);
                            Scc[e] = __builtin_fmaf(cv, cv, Scc[e]);
                        }
                    } else {
                        Swy[e] += yd;
                        if constexpr (kHasStd) {
                            float ev = kLut ? gs[e] * cqn : cqn;
                            if constexpr (STD == CT_STD_MULTIPLIER) ev *= px;
                            if constexpr (STD == CT_STD_EXPLICIT) ev *= sp.v[e];
                            Scc[e] = __builtin_fmaf(ev, ev, Scc[e]);
                        }
                    }
                }
            };

            // Software pipeline: kDepth exposures in flight per thread, kDepth + 1 per trip through rotating registers
            // (the slot freed by one step is re-filled by the next), so that no packet is ever copied -- a copy would
            // make the wavefront wait for the load it has just issued.
            auto fetch = [&](int n, CodePk &pk, Packet<float, V> &sp) {
                const int nn = n < Bb ? n : Bb - 1;  // past the end: re-load the last exposure (cache hit, unused)
                // Buffer loads: (scalar descriptor rebased to the exposure) + (32-bit per-thread byte offset) -- no vector
                // address arithmetic.  The base is laundered through an empty asm so LLVM cannot prove the prefetched
                // packet equal to a fresh load at its use (it would re-load there and drop the prefetch).
                uint64_t base = batch_base + (uint64_t)((int64_t)nn * a.image_stride * (int64_t)sizeof(T));
                asm volatile("" : "+s"(base));
                if constexpr (kTyped)
                    pk = load_codes_as_float<T, V>(base, voff);
                else
                    pk = load_buffer<Packet<T, V>>(base, voff);
                if constexpr (STD == CT_STD_EXPLICIT) {
                    uint64_t sbase = batch_std_base + (uint64_t)((int64_t)nn * a.image_stride * 4);
                    asm volatile("" : "+s"(sbase));
                    sp = load_buffer<Packet<float, V>>(sbase, svoff);
                }
            };
            constexpr int kRing = kPivotDepth + 1;
            CodePk ring[kRing];
            Packet<float, V> sring[STD == CT_STD_EXPLICIT ? kRing : 1];
            static_for<kPivotDepth>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                fetch(j, ring[j], sring[STD == CT_STD_EXPLICIT ? j : 0]);
            });
            uint32_t expo_adr = (uint32_t)lut_bytes + 8u * (uint32_t)n0;  // LDS byte address of this trip's per-exposure constants
            for (int n = 0; n < Bb; n += kRing) {
                static_for<kRing>([&](auto jc) {
                    constexpr int j = decltype(jc)::value, slot = (j + kPivotDepth) % kRing;
                    fetch(n + j + kPivotDepth, ring[slot], sring[STD == CT_STD_EXPLICIT ? slot : 0]);
                    if (j == 0 || n + j < Bb) reduce(ring[j], sring[STD == CT_STD_EXPLICIT ? j : 0], expo_adr + 8u * j);
                });
                expo_adr += 8u * kRing;
            }
            };
            // LOOKUP's closed-form variance is the weight path alone, sum a_n^2 (y_n - m)^2: about any pivot that is not the
            // mean it cancels, and every wavefront of C2 used to repeat its batch (tools/debug/retry_rate.py: 196 608 of
            // 196 608).  So its first pass computes the mean only (9 instead of 17 instructions per sample) and the second,
            // about that mean, the moments.
            constexpr bool kMeanFirst = kLookup && kHasStd;
            if constexpr (kMeanFirst) {
                if (pass == 0)
                    run_batch(std::false_type{}, std::false_type{});
                else
                    run_batch(std::false_type{}, std::true_type{});
            } else {
                if (rough)
                    run_batch(std::true_type{}, std::true_type{});
                else
                    run_batch(std::false_type{}, std::true_type{});
            }

            // ---- epilogue: WBOMean update (statistics.py:64-109) and the closed-form variance, division-free ----
            bool bad[V];
            float mb_f[V];
#pragma unroll
            for (int e = 0; e < V; ++e) {
                const float Wb = kGauss ? W[e] : (float)Bb;
                const float Df = Wb + 1e-6f;  // float32 tensor + python float stays float32 (statistics.py:79-80)
                // v_rcp_f32 (1 ulp) as it comes: the quotient q is corrected against D below, and a last-bit error of
                // beta = frac / D or of frac moves the variance / the mean update by 1e-7 of themselves.  (Newton steps on
                // both reciprocals and the term-by-term quadratic form cost 8 of this epilogue's ~45 instructions, and
                // with several batches per launch the epilogue runs once per batch and element.)
                const float r = __builtin_amdgcn_rcpf(Df);
                const float num = __builtin_fmaf(-p[e], 1e-6f, Swy[e]);  // sum w y - p (W + 1e-6)
                float q = num * r;
                q = __builtin_fmaf(__builtin_fmaf(-q, Df, num), r, q);  // m_b - p
                float Wt = Wb, frac = 1.0f, var = 0.0f, gam = 0.0f;
                bad[e] = false;
                if constexpr (FIRST) {
                    mean_o[e] = (double)p[e] + (double)q;
                } else {
                    Wt = WA[e] + Wb;
                    const float rw = __builtin_amdgcn_rcpf(Wt);  // statistics.py:105, division-free
                    frac = WA[e] == 0.0f ? 1.0f : Wb * rw;  // (a fresh merge inside a MULTI launch: W_A = 0, W_B / W_B = 1 exactly)
                    const double diff = ((double)p[e] - meanA[e]) + (double)q;  // m_b - mean_A
                    mean_o[e] = __builtin_fma((double)frac, diff, meanA[e]);
                    gam = ((WA[e] * rw) * rw) * (float)diff;
                    var = varA[e];
                }
                Wt_o[e] = Wt;
                mb_f[e] = p[e] + q;
                if constexpr (kHasStd) {
                    const float beta = frac * r;
                    const float kap = __builtin_fmaf(-beta, q, gam);
                    // beta^2 Scc + 2 beta kappa Sac + kappa^2 Saa: the two squares first (S >= 0), then the cross term.
                    // Cancellation test: t1 + |t2| + t3 > kPivotCondLimit * upd  <=>  t2 < 0 and upd < S * 2 / (limit + 1)
                    const float bk = beta * kap;
                    const float S = __builtin_fmaf(kap * kap, Saa[e], (beta * beta) * Scc[e]);
                    const float upd = __builtin_fmaf(bk + bk, Sac[e], S);
                    if constexpr (kGauss) bad[e] = upd * (0.5f * (kPivotCondLimit + 1.0f)) < S;
                    if constexpr (kMeanFirst) bad[e] = bad[e] || pass == 0;  // (the first pass had no moments: go on about the mean)
                    var += fmaxf(upd, 0.0f) * sv2;
                }
                var_o[e] = var;
            }
            bool any_bad = false;
#pragma unroll
            for (int e = 0; e < V; ++e) any_bad |= bad[e];
            if (pass == 1 || !__any(any_bad)) break;
            if (x.retry_count && (threadIdx.x & 63) == 0) atomicAdd(x.retry_count, 1ull);
            // only the ill-conditioned elements move their pivot: the others recompute exactly what they had, so an
            // element's result does not depend on which other elements share its wavefront (tiles == whole, bit for bit)
#pragma unroll
            for (int e = 0; e < V; ++e) p[e] = bad[e] ? mb_f[e] : p[e];
        }
        if constexpr (MULTI) {  // internal_detach (hdr_merge.py:128): the batch's result is the next batch's state and pivot
#pragma unroll
            for (int e = 0; e < V; ++e) {
                meanA[e] = mean_o[e];
                WA[e] = Wt_o[e];
                varA[e] = var_o[e];
                p[e] = (float)mean_o[e];
            }
            n0 += Bb;
        }
        }  // batches

        if (keep_state) {
#pragma unroll
            for (int e = 0; e < V; ++e) {
                const uint32_t q = planar_of(q0 + e);
                a.mean_state[q] = mean_o[e];
                a.sumw_state[q] = Wt_o[e];
                if constexpr (kHasStd) a.var_state[q] = var_o[e];
            }
        }
        if (finalize && rgb252) {
            if constexpr (RGB252) {
                // Regroup the WORKGROUP's results by channel plane through LDS: a full tile's 1024 consecutive memory
                // elements contain 84-85 whole groups of 12 elements = 4 pixels x 3 channels; thread 3 i + c takes plane c of
                // group i and writes its four consecutive pixels as 16-byte packets.  Only the <= 11 elements before the first
                // and after the last whole group of the TILE are stored one by one (1 % of the elements; regrouping per
                // wavefront left 4 % of them to such partial-line stores: FETCH_SIZE +11 %, WRITE_SIZE +8 %,
                // profiles/r03_layout_ingest.md).  Two workgroup barriers per tile; the ragged last tile of the image, where
                // threads have left the loop, stores element by element.  (A mapping that gives every wavefront 252 elements
                // = 84 whole pixels was measured first: its 504-byte wave loads cost 11 % more HBM fetch.)
                const uint32_t tile_first = tile * (uint32_t)(kBlock * V);     // relative to q_begin (0 in this mode)
                const bool tile_full = tile_first + (uint32_t)(kBlock * V) <= a.q_count;   // workgroup-uniform
                float sdv[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) sdv[e] = kHasStd ? __builtin_amdgcn_sqrtf(var_o[e]) : 0.0f;
                if (tile_full) {
                    char *stage = lds + x.stage_off;
                    double *sm = reinterpret_cast<double *>(stage);            // 1024 means in memory order
                    float *ss = reinterpret_cast<float *>(stage + 8192);       // 1024 standard uncertainties
                    typedef double d2 __attribute__((ext_vector_type(2)));
                    typedef float f4 __attribute__((ext_vector_type(4)));
                    __syncthreads();  // the previous tile's readers are done with the stage
                    d2 m01 = {mean_o[0], mean_o[1]}, m23 = {mean_o[2], mean_o[3]};
                    *reinterpret_cast<d2 *>(sm + 4u * threadIdx.x) = m01;
                    *reinterpret_cast<d2 *>(sm + 4u * threadIdx.x + 2) = m23;
                    if constexpr (kHasStd) {
                        f4 sv = {sdv[0], sdv[1], sdv[2], sdv[3]};
                        *reinterpret_cast<f4 *>(ss + 4u * threadIdx.x) = sv;
                    }
                    __syncthreads();
                    const uint32_t g_first = (tile_first + 11u) / 12u, g_end = (tile_first + (uint32_t)(kBlock * V)) / 12u;   // whole groups
#pragma unroll
                    for (int e = 0; e < 4; ++e) {   // the ragged ends of the tile
                        const uint32_t m = q0 + (uint32_t)e - a.q_begin;
                        if (m < 12u * g_first || m >= 12u * g_end) {
                            const uint32_t q = planar_of(q0 + e);
                            static_cast<double *>(a.mean_out)[q] = mean_o[e];
                            if constexpr (kHasStd) a.std_out[q] = sdv[e];
                        }
                    }
                    // plane-major: threads 0 .. n-1 take plane 0 of the tile's n groups, the next n plane 1, ... -- consecutive
                    // lanes then store consecutive 32-byte packets of ONE plane (whole lines per wavefront; group-major
                    // threads 3 i + c alternated between the planes: WRITE_SIZE +20 %)
                    const uint32_t n_groups = g_end - g_first;               // 84 or 85: 3 n <= 256 threads
                    const uint32_t c = (threadIdx.x >= n_groups ? 1u : 0u) + (threadIdx.x >= 2u * n_groups ? 1u : 0u);
                    const uint32_t tri = threadIdx.x - c * n_groups;
                    if (threadIdx.x < 3u * n_groups) {
                        const uint32_t g = g_first + tri;                      // global group: pixels 4 g .. 4 g + 3
                        const uint32_t cm = a.tile.layout == CT_LAYOUT_NHWC_BGR ? 2u - c : c;
                        const uint32_t local = 12u * g - tile_first + cm;     // index of (pixel 4 g, memory channel cm) in the stage
                        Packet<double, 4> mo;
                        Packet<float, 4> so;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            mo.v[j] = sm[local + 3u * (uint32_t)j];
                            if constexpr (kHasStd) so.v[j] = ss[local + 3u * (uint32_t)j];
                        }
                        const size_t dst = (size_t)c * a.tile.plane_local + 4u * (size_t)g;
                        store_stream(reinterpret_cast<Packet<double, 4> *>(static_cast<double *>(a.mean_out) + dst), mo);
                        if constexpr (kHasStd) store_stream(reinterpret_cast<Packet<float, 4> *>(a.std_out + dst), so);
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const uint32_t q = planar_of(q0 + e);
                        static_cast<double *>(a.mean_out)[q] = mean_o[e];
                        if constexpr (kHasStd) a.std_out[q] = sdv[e];
                    }
                }
            }
        } else if (finalize && !planar_out) {
#pragma unroll
            for (int e = 0; e < V; ++e) {
                const uint32_t q = planar_of(q0 + e);
                if (a.flags & CT_MERGE_MEAN_OUT_F32)
                    static_cast<float *>(a.mean_out)[q] = (float)mean_o[e];
                else
                    static_cast<double *>(a.mean_out)[q] = mean_o[e];
                if constexpr (kHasStd) a.std_out[q] = __builtin_amdgcn_sqrtf(var_o[e]);
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
                for (int e = 0; e < V; ++e) o.v[e] = __builtin_amdgcn_sqrtf(var_o[e]);
                store_stream(reinterpret_cast<Packet<float, V> *>(a.std_out + q0), o);
            }
        }
    }
}

// Workgroups of `kernel` that fit one compute unit (registers, LDS, waves), cached per instantiation.
template <typename KernelT>
static int pivot_blocks_per_cu(KernelT kernel, size_t lds)
{
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel, kBlock, lds) != hipSuccess || n < 1) n = 1;
    return n < 8 ? n : 8;
}

// Persistent grid: as many workgroups as are resident at once (so every workgroup walks the same number of tiles, +-1).
template <auto kernel>
static int launch_pivot_grid(const MergeArgs &a, const PivotArgs &x, size_t lds, hipStream_t stream)
{
    // residency per kernel (the kernel is a template argument, so these statics are per kernel); it is re-derived when
    // a later call needs more LDS (a larger LUT)
    static int per_cu = 0;
    static size_t per_cu_lds = 0;
    if (per_cu == 0 || lds > per_cu_lds) {
        per_cu = pivot_blocks_per_cu(kernel, lds);
        per_cu_lds = lds;
    }
    uint32_t grid = (uint32_t)(compute_units() * per_cu);
    if (grid > x.n_tiles) grid = x.n_tiles;
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(kBlock), lds, stream, a, x);
    return hipGetLastError() == hipSuccess ? CT_OK : CT_ERR_LAUNCH;
}

#if CT_MERGE_PART != 1
template <typename T, int V, int INTERP, int WEIGHT, int STD, bool CLAMP>
static int launch_pivot(const MergeArgs &a, PivotArgs x, hipStream_t stream)
{
    if (a.q_count == 0) return CT_OK;
    if constexpr (INTERP == CT_INTERP_LOOKUP && WEIGHT == CT_WEIGHT_NONE && STD != CT_STD_NONE) {
        return CT_ERR_NO_GRADIENT_PATH;  // (refused by ct_hdr_merge_batch before it gets here)
    } else {
        x.n_tiles = (a.q_count + (uint32_t)(kBlock * V) - 1) / (uint32_t)(kBlock * V);  // a.q_count is a multiple of V
        constexpr bool kTable = INTERP != CT_INTERP_NONE;
        constexpr int kWV = (INTERP == CT_INTERP_LINEAR && WEIGHT == CT_WEIGHT_GAUSS && CT_PIVOT_TYPED_LOAD && sizeof(T) == 2 && V == 4) ? CT_PIVOT_WEIGHT : 0;
        size_t lds = (kTable ? (size_t)a.channels * x.n_entries * ((kWV == 2 || INTERP == CT_INTERP_CATMULL) ? 16 : 8) : 0) +
                     2 * sizeof(float) * (size_t)a.batch + (kWV == 1 ? (size_t)(65536 >> CT_PIVOT_WT_SHIFT) * 8 : 0);
        x.rgb252 = 0;
        if constexpr (V == 4 && CT_PIVOT_TYPED_LOAD) {
            // interleaved RGB / BGR, the whole image in this launch, outputs only (no streaming state): packet stores
            auto aligned16 = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) % 16) == 0; };
            if (a.tile.layout != CT_LAYOUT_NCHW && !(a.flags & CT_MERGE_OUT_AS_INPUT) && a.channels == 3 && a.tile.plane_local % 4 == 0 && !a.mean_state &&
                (a.flags & CT_MERGE_FINALIZE) && !(a.flags & CT_MERGE_MEAN_OUT_F32) && a.q_begin == 0 &&
                a.q_count == 3u * a.tile.plane_local && aligned16(a.mean_out) && aligned16(a.std_out)) {
                x.rgb252 = 1;
                lds = (lds + 15) & ~(size_t)15;
                x.stage_off = (uint32_t)lds;
                lds += 4 * 3072;
            }
        }
        if (lds > 160 * 1024) return CT_ERR_TOO_LARGE;
        if constexpr (V == 4 && CT_PIVOT_TYPED_LOAD) {
            if (x.rgb252)
                return (a.flags & CT_MERGE_FIRST_BATCH)
                           ? launch_pivot_grid<merge_pivot_kernel<T, V, INTERP, WEIGHT, STD, true, CLAMP, false, true>>(a, x, lds, stream)
                           : CT_ERR_INVALID_ARGUMENT;  // (no state and not the first batch: refused earlier)
        }
        return (a.flags & CT_MERGE_FIRST_BATCH)
                   ? launch_pivot_grid<merge_pivot_kernel<T, V, INTERP, WEIGHT, STD, true, CLAMP>>(a, x, lds, stream)
                   : launch_pivot_grid<merge_pivot_kernel<T, V, INTERP, WEIGHT, STD, false, CLAMP>>(a, x, lds, stream);
    }
}

template <typename T, int V, int INTERP, int WEIGHT, bool CLAMP>
static int dispatch_pivot_std(const MergeArgs &a, const PivotArgs &x, int std_mode, hipStream_t s)
{
    switch (std_mode) {
        case CT_STD_NONE: return launch_pivot<T, V, INTERP, WEIGHT, CT_STD_NONE, CLAMP>(a, x, s);
        case CT_STD_CONSTANT: return launch_pivot<T, V, INTERP, WEIGHT, CT_STD_CONSTANT, CLAMP>(a, x, s);
        case CT_STD_MULTIPLIER: return launch_pivot<T, V, INTERP, WEIGHT, CT_STD_MULTIPLIER, CLAMP>(a, x, s);
        case CT_STD_EXPLICIT: return launch_pivot<T, V, INTERP, WEIGHT, CT_STD_EXPLICIT, CLAMP>(a, x, s);
    }
    return CT_ERR_INVALID_ARGUMENT;
}

template <typename T, int V, bool CLAMP>
static int dispatch_pivot_interp(const MergeArgs &a, const PivotArgs &x, int interp, int weight_mode, int std_mode, hipStream_t s)
{
    const bool gauss = weight_mode == CT_WEIGHT_GAUSS;
    if (interp == CT_INTERP_LINEAR)
        return gauss ? dispatch_pivot_std<T, V, CT_INTERP_LINEAR, CT_WEIGHT_GAUSS, CLAMP>(a, x, std_mode, s)
                     : dispatch_pivot_std<T, V, CT_INTERP_LINEAR, CT_WEIGHT_NONE, CLAMP>(a, x, std_mode, s);
    if constexpr (CT_PIVOT_TYPED_LOAD) {
        if (interp == CT_INTERP_LOOKUP)
            return gauss ? dispatch_pivot_std<T, V, CT_INTERP_LOOKUP, CT_WEIGHT_GAUSS, CLAMP>(a, x, std_mode, s)
                         : dispatch_pivot_std<T, V, CT_INTERP_LOOKUP, CT_WEIGHT_NONE, CLAMP>(a, x, std_mode, s);
        if (interp == CT_INTERP_CATMULL)
            return gauss ? dispatch_pivot_std<T, V, CT_INTERP_CATMULL, CT_WEIGHT_GAUSS, CLAMP>(a, x, std_mode, s)
                         : dispatch_pivot_std<T, V, CT_INTERP_CATMULL, CT_WEIGHT_NONE, CLAMP>(a, x, std_mode, s);
    }
    // no model: no table, nothing to clamp
    return gauss ? dispatch_pivot_std<T, V, CT_INTERP_NONE, CT_WEIGHT_GAUSS, false>(a, x, std_mode, s)
                 : dispatch_pivot_std<T, V, CT_INTERP_NONE, CT_WEIGHT_NONE, false>(a, x, std_mode, s);
}

#endif  // CT_MERGE_PART != 1

#if CT_MERGE_PART != 0
// ---- several batches per launch (MULTI): packets of kPivotV only, state-carrying instantiation ----
template <typename T, int INTERP, int WEIGHT, int STD, bool CLAMP>
static int launch_pivot_multi(const MergeArgs &a, PivotArgs x, hipStream_t stream)
{
    if (a.q_count == 0) return CT_OK;
    if constexpr (INTERP == CT_INTERP_LOOKUP && WEIGHT == CT_WEIGHT_NONE && STD != CT_STD_NONE) {
        return CT_ERR_NO_GRADIENT_PATH;
    } else if constexpr (!CT_PIVOT_TYPED_LOAD) {
        return CT_ERR_UNSUPPORTED;
    } else {
        constexpr int V = kPivotV;
        x.n_tiles = (a.q_count + (uint32_t)(kBlock * V) - 1) / (uint32_t)(kBlock * V);
        constexpr bool kTable = INTERP != CT_INTERP_NONE;
        constexpr int kWV = (INTERP == CT_INTERP_LINEAR && WEIGHT == CT_WEIGHT_GAUSS && sizeof(T) == 2) ? CT_PIVOT_WEIGHT : 0;
        const size_t lds = (kTable ? (size_t)a.channels * x.n_entries * ((kWV == 2 || INTERP == CT_INTERP_CATMULL) ? 16 : 8) : 0) +
                           2 * sizeof(float) * (size_t)a.batch + (kWV == 1 ? (size_t)(65536 >> CT_PIVOT_WT_SHIFT) * 8 : 0);
        if (lds > 160 * 1024) return CT_ERR_TOO_LARGE;
        return launch_pivot_grid<merge_pivot_kernel<T, V, INTERP, WEIGHT, STD, false, CLAMP, true>>(a, x, lds, stream);
    }
}

template <typename T, int INTERP, int WEIGHT, bool CLAMP>
static int dispatch_multi_std(const MergeArgs &a, const PivotArgs &x, int std_mode, hipStream_t s)
{
    switch (std_mode) {
        case CT_STD_NONE: return launch_pivot_multi<T, INTERP, WEIGHT, CT_STD_NONE, CLAMP>(a, x, s);
        case CT_STD_CONSTANT: return launch_pivot_multi<T, INTERP, WEIGHT, CT_STD_CONSTANT, CLAMP>(a, x, s);
        case CT_STD_MULTIPLIER: return launch_pivot_multi<T, INTERP, WEIGHT, CT_STD_MULTIPLIER, CLAMP>(a, x, s);
        case CT_STD_EXPLICIT: return launch_pivot_multi<T, INTERP, WEIGHT, CT_STD_EXPLICIT, CLAMP>(a, x, s);
    }
    return CT_ERR_INVALID_ARGUMENT;
}

template <typename T, bool CLAMP>
static int dispatch_multi_interp(const MergeArgs &a, const PivotArgs &x, int interp, int weight_mode, int std_mode, hipStream_t s)
{
    const bool gauss = weight_mode == CT_WEIGHT_GAUSS;
    if (interp == CT_INTERP_LINEAR)
        return gauss ? dispatch_multi_std<T, CT_INTERP_LINEAR, CT_WEIGHT_GAUSS, CLAMP>(a, x, std_mode, s)
                     : dispatch_multi_std<T, CT_INTERP_LINEAR, CT_WEIGHT_NONE, CLAMP>(a, x, std_mode, s);
    if (interp == CT_INTERP_LOOKUP)
        return gauss ? dispatch_multi_std<T, CT_INTERP_LOOKUP, CT_WEIGHT_GAUSS, CLAMP>(a, x, std_mode, s)
                     : dispatch_multi_std<T, CT_INTERP_LOOKUP, CT_WEIGHT_NONE, CLAMP>(a, x, std_mode, s);
    if (interp == CT_INTERP_CATMULL)
        return gauss ? dispatch_multi_std<T, CT_INTERP_CATMULL, CT_WEIGHT_GAUSS, CLAMP>(a, x, std_mode, s)
                     : dispatch_multi_std<T, CT_INTERP_CATMULL, CT_WEIGHT_NONE, CLAMP>(a, x, std_mode, s);
    return gauss ? dispatch_multi_std<T, CT_INTERP_NONE, CT_WEIGHT_GAUSS, false>(a, x, std_mode, s)
                 : dispatch_multi_std<T, CT_INTERP_NONE, CT_WEIGHT_NONE, false>(a, x, std_mode, s);
}


// The several-batches launch of ct_hdr_merge_batches (defined in the CT_MERGE_PART 1 translation unit).
int merge_pivot_multi(const MergeArgs &a, const PivotArgs &px, int dtype, bool clamp, int interp, int weight_mode, int std_mode,
                      hipStream_t s)
{
    if (dtype == CT_DTYPE_U8) return dispatch_multi_interp<uint8_t, false>(a, px, interp, weight_mode, std_mode, s);
    return clamp ? dispatch_multi_interp<uint16_t, true>(a, px, interp, weight_mode, std_mode, s)
                 : dispatch_multi_interp<uint16_t, false>(a, px, interp, weight_mode, std_mode, s);
}
#else
int merge_pivot_multi(const MergeArgs &a, const PivotArgs &px, int dtype, bool clamp, int interp, int weight_mode, int std_mode,
                      hipStream_t s);
#endif  // CT_MERGE_PART != 0

#if CT_MERGE_PART != 1
// CLAMP (codes above max_code exist: max_code below the container's range) costs one v_min per sample, so it is its own
// instantiation for uint16 packets; the one-element launch of a ragged tail always carries it (its cost is irrelevant);
// uint8 packets with max_code < 255 are left to the generic kernel (pivot_eligible).
template <typename T, int V>
static int dispatch_pivot(const MergeArgs &a, const PivotArgs &x, int interp, int weight_mode, int std_mode, bool clamp, hipStream_t s)
{
    if constexpr (!CT_PIVOT_TYPED_LOAD) {
        return dispatch_pivot_interp<T, V, false>(a, x, interp, weight_mode, std_mode, s);
    } else if constexpr (V == 1) {
        return dispatch_pivot_interp<T, V, true>(a, x, interp, weight_mode, std_mode, s);
    } else if constexpr (sizeof(T) == 2) {
        return clamp ? dispatch_pivot_interp<T, V, true>(a, x, interp, weight_mode, std_mode, s)
                     : dispatch_pivot_interp<T, V, false>(a, x, interp, weight_mode, std_mode, s);
    } else {
        return dispatch_pivot_interp<T, V, false>(a, x, interp, weight_mode, std_mode, s);
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
    const bool f64 = a.flags & CT_MERGE_F64_MOMENTS;  // diagnostic: the round-1 float64 moments
    if constexpr (sizeof(T) != 4) {
        if (fold && f64)
            hipLaunchKernelGGL((merge_kernel<T, V, INTERP, WEIGHT, STD, true, 2, false>), dim3(grid), dim3(kBlock), lds, stream, a);
        else if (fold)
            hipLaunchKernelGGL((merge_kernel<T, V, INTERP, WEIGHT, STD, true, 2, true>), dim3(grid), dim3(kBlock), lds, stream, a);
        else if (f64)
            hipLaunchKernelGGL((merge_kernel<T, V, INTERP, WEIGHT, STD, false, 2, false>), dim3(grid), dim3(kBlock), lds, stream, a);
        else
            hipLaunchKernelGGL((merge_kernel<T, V, INTERP, WEIGHT, STD, false, 2, true>), dim3(grid), dim3(kBlock), lds, stream, a);
    } else if (f64) {
        hipLaunchKernelGGL((merge_kernel<T, V, INTERP, WEIGHT, STD, false, 2, false>), dim3(grid), dim3(kBlock), lds, stream, a);
    } else {
        hipLaunchKernelGGL((merge_kernel<T, V, INTERP, WEIGHT, STD, false, 2, true>), dim3(grid), dim3(kBlock), lds, stream, a);
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
static int merge_typed(MergeArgs a, uint32_t Q, int interp, int weight_mode, int std_mode, hipStream_t s, bool fold,
                       const PivotArgs *pivot = nullptr, bool pivot_clamp = false)
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
    if constexpr (sizeof(T) != 4) {
        // Eligible integer stacks go through the pivoted float32 kernel whole: packets of kPivotV where the alignment
        // allows, the ragged rest one element per thread.  Per-element arithmetic is identical in both, so how a stack
        // is cut into tiles does not change a single bit of the result.
        if (pivot) {
            const bool pv_ok = aligned(a.stack, sizeof(T) * kPivotV) && (a.image_stride % kPivotV) == 0 &&
                               aligned(a.std_stack, 4 * kPivotV) && aligned(a.mean_out, 8 * kPivotV) &&
                               aligned(a.std_out, 4 * kPivotV);
            const uint32_t q_pv = pv_ok ? (Q / kPivotV) * kPivotV : 0;
            if (q_pv) {
                a.q_begin = 0;
                a.q_count = q_pv;
                rc = dispatch_pivot<T, kPivotV>(a, *pivot, interp, weight_mode, std_mode, pivot_clamp, s);
                if (rc != CT_OK) return rc;
            }
            if (q_pv < Q) {
                a.q_begin = q_pv;
                a.q_count = Q - q_pv;
                rc = dispatch_pivot<T, 1>(a, *pivot, interp, weight_mode, std_mode, pivot_clamp, s);
            }
            return rc;
        }
    }
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

#endif  // CT_MERGE_PART != 1

}  // namespace ct

#if CT_MERGE_PART != 1

// Host check that fma(u, hi, u*lo) == u / max_code for every code (see NormConst in ct_device.hpp).
extern "C" int ct_norm_constants(float max_code, float *hi, float *lo);
extern "C" int ct_index_constants(float max_code, int n_points, float *hi, float *lo);
// Host check that (code * index_mul) >> 32 (uint16) / the code itself (uint8) is the reference's LUT interval for every code.
extern "C" int ct_pivot_index_constants(float max_code, int n_points, uint32_t *index_mul, float *step);
// Host check that the round-down FMA on the code held as a float gives the same interval (typed-load path).
extern "C" int ct_pivot_floor_constants(float max_code, int n_points, float *rcp_step);

// Diagnostics (not part of the data path): device counter that merge_pivot_kernel bumps once per wavefront that ran
// its fallback pass.  NULL (the default) disables counting.  Process-global; set it only around a measurement.
static unsigned long long *g_merge_retry_counter = nullptr;
extern "C" void ct_merge_set_retry_counter(unsigned long long *counter_dev) { g_merge_retry_counter = counter_dev; }

// Host proof behind the pivoted kernel's table addressing (ct_api.cpp).
extern "C" int ct_pivot_interval_constants(float max_code, int n_points, int lookup, int dtype_max, float *scale);

// Pivoted float32 kernel: LINEAR / LOOKUP / no model on raw integer codes whose table entry is an exact function of the
// code by one round-down FMA -- verified on the host for every code the container can hold, also above max_code (12- and
// 14-bit data in uint16) and for LUT steps that are not a whole number of codes; CT_MERGE_F64_MOMENTS opts out.
// *clamp: codes above max_code exist and must clamp to the last entry.
static bool pivot_eligible(int32_t dtype, float max_code, int interp, int n_points, uint32_t flags, ct::PivotArgs *px,
                           bool *clamp = nullptr)
{
    if (dtype != CT_DTYPE_U8 && dtype != CT_DTYPE_U16) return false;
    const int dtype_max = dtype == CT_DTYPE_U8 ? 255 : 65535;
    if (flags & CT_MERGE_F64_MOMENTS) return false;
    if (!(max_code >= 1.0f) || max_code > (float)dtype_max || floorf(max_code) != max_code) return false;
    if (interp < CT_INTERP_LOOKUP || interp > CT_INTERP_NONE) return false;
    if (interp == CT_INTERP_CATMULL && !CT_PIVOT_TYPED_LOAD) return false;
    const bool need_clamp = interp != CT_INTERP_NONE && max_code < (float)dtype_max;
    if (clamp) *clamp = need_clamp;
    px->step = 1.0f;
    px->index_mul = 0;
    px->index_rcp = 1.0f;
    px->max_code = max_code;
    px->n_entries = 0;
    px->tf_max = ct::kFloorMagic;
    if (interp == CT_INTERP_NONE) return true;
    if (CT_PIVOT_TYPED_LOAD) {
        if (need_clamp && dtype == CT_DTYPE_U8) return false;  // (no CLAMP instantiation for uint8 packets)
        const bool lookup = interp == CT_INTERP_LOOKUP;
        if (ct_pivot_interval_constants(max_code, n_points, lookup, dtype_max, &px->index_rcp) != CT_OK) return false;
        px->step = (float)((double)max_code / (double)(n_points - 1));
        px->n_entries = lookup ? 2u * (uint32_t)n_points : (uint32_t)n_points;
        px->tf_max = ct::kFloorMagic + (float)(lookup ? 2 * (n_points - 1) : n_points - 1);
        return true;
    }
    if (interp != CT_INTERP_LINEAR || need_clamp) return false;
    if (ct_pivot_index_constants(max_code, n_points, &px->index_mul, &px->step) != CT_OK) return false;
    if (dtype == CT_DTYPE_U8 && px->step != 1.0f) return false;   // raw uint8 codes: the code is the index
    if (dtype == CT_DTYPE_U16 && px->index_mul == 0) return false;
    px->n_entries = (uint32_t)n_points;
    return true;
}

// LOOKUP and CATMULL with uncertainties follow the reference's float32 autograd order by default (ct_merge_exact.hip):
// their reference results are dominated by float32 cancellation -- CATMULL in the cubic-basis backward, LOOKUP (whose
// variance is the weight path alone) in y_n - m_b with m_b formed from the float32-rounded sum of weights -- so a closed
// form, however accurate, differs from the reference by the reference's own noise (up to 2e-5 / 4e-5 on single elements).
// CT_MERGE_REFERENCE_ORDER asks for that path in any mode, CT_MERGE_CLOSED_FORM keeps the fast closed-form kernels.
static bool merge_uses_reference_order(int interp, int std_mode, uint32_t flags)
{
    if (flags & CT_MERGE_REFERENCE_ORDER) return true;
    return (interp == CT_INTERP_CATMULL || interp == CT_INTERP_LOOKUP) && std_mode != CT_STD_NONE &&
           !(flags & (CT_MERGE_CLOSED_FORM | CT_MERGE_F64_MOMENTS));
}

// Which kernel ct_hdr_merge_batch dispatches for these arguments (bench.py records it next to its numbers).
extern "C" const char *ct_hdr_merge_kernel_name(int32_t dtype, float max_code, int32_t interp, int32_t n_points,
                                                uint32_t flags)
{
    // (std mode unknown here: CT_MERGE_STD_HINT in flags says uncertainties are propagated)
    if (merge_uses_reference_order(interp, (flags & CT_MERGE_STD_HINT) ? CT_STD_CONSTANT : CT_STD_NONE, flags))
        return "ct::merge_reference_order_kernel (the reference's float32 autograd order, two passes, float64 exp and divisions)";
    ct::PivotArgs px{};
    if (pivot_eligible(dtype, max_code, interp, interp == CT_INTERP_NONE ? 2 : n_points, flags, &px))
        return (flags & CT_MERGE_FIRST_BATCH)
                   ? "ct::merge_pivot_kernel (float32 moments about a per-pixel pivot, persistent workgroups, 4 codes per "
                     "thread through typed buffer loads, 7 wavefronts per SIMD, first batch)"
                   : "ct::merge_pivot_kernel (float32 moments about the running mean, persistent workgroups, 4 codes per "
                     "thread, streaming state)";
    if (flags & CT_MERGE_F64_MOMENTS)
        return dtype == CT_DTYPE_F32 ? "ct::merge_kernel (float64 moments, float32 pixels, 4 per thread)"
                                     : "ct::merge_kernel (float64 moments, integer codes)";
    return dtype == CT_DTYPE_F32 ? "ct::merge_kernel (float32 moments about a per-pixel pivot, float32 pixels, 4 per thread)"
                                 : "ct::merge_kernel (float32 moments about a per-pixel pivot, integer codes through the float LUT coordinate)";
}

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
    if (merge_uses_reference_order(interp, std_mode, flags)) {
        if (dtype != CT_DTYPE_F32 && ct_norm_constants(max_code, &a.norm.hi, &a.norm.lo) != CT_OK) return CT_ERR_UNSUPPORTED;
        return merge_reference_order(a, dtype, (uint32_t)Ql, interp, weight_mode, std_mode, s);
    }
    switch (dtype) {
        case CT_DTYPE_U8:
        case CT_DTYPE_U16: {
            if (ct_norm_constants(max_code, &a.norm.hi, &a.norm.lo) != CT_OK) return CT_ERR_UNSUPPORTED;
            // FOLD needs the LUT index formed from the code to equal the reference's float32 index for every code
            const bool fold = ct_index_constants(max_code, a.n_points, &a.index.hi, &a.index.lo) == CT_OK;
            a.inv_max_code = (float)(1.0 / (double)max_code);
            PivotArgs px{};
            const PivotArgs *pivot = nullptr;
            bool clamp = false;
            if (pivot_eligible(dtype, max_code, interp, a.n_points, flags, &px, &clamp)) {
                px.probe = batch / 2;
                px.retry_count = g_merge_retry_counter;
                pivot = &px;
            }
            return dtype == CT_DTYPE_U8 ? merge_typed<uint8_t>(a, (uint32_t)Ql, interp, weight_mode, std_mode, s, fold, pivot, clamp)
                                        : merge_typed<uint16_t>(a, (uint32_t)Ql, interp, weight_mode, std_mode, s, fold, pivot, clamp);
        }
        case CT_DTYPE_F32: return merge_typed<float>(a, (uint32_t)Ql, interp, weight_mode, std_mode, s, false);
    }
    return CT_ERR_UNSUPPORTED;
}

// Several consecutive batches of one merge in ONE launch (hdr_merge.py:61-128 for k iterations of the loop).
extern "C" int ct_hdr_merge_batches(const void *const *stack_devs, const float *const *std_devs, const int32_t *batch_sizes,
                                    int32_t n_batches, int32_t dtype, float max_code, const ct_geometry *geom, int32_t std_mode,
                                    float std_value, const double *exposure_dev, const ct_icrf *icrf, int32_t weight_mode,
                                    double *mean_state_dev, float *sumw_state_dev, float *var_state_dev, void *mean_out_dev,
                                    float *std_out_dev, uint32_t flags, void *stream)
{
    using namespace ct;
    if (!stack_devs || !batch_sizes || n_batches <= 0 || !geom || !icrf || !exposure_dev) return CT_ERR_INVALID_ARGUMENT;
    if (std_mode == CT_STD_EXPLICIT && !std_devs) return CT_ERR_INVALID_ARGUMENT;
    int64_t total = 0;
    for (int b = 0; b < n_batches; ++b) {
        if (!stack_devs[b] || batch_sizes[b] <= 0 || (std_mode == CT_STD_EXPLICIT && !std_devs[b])) return CT_ERR_INVALID_ARGUMENT;
        total += batch_sizes[b];
    }
    const bool first = flags & CT_MERGE_FIRST_BATCH, finalize = flags & CT_MERGE_FINALIZE;
    const bool has_state = mean_state_dev && sumw_state_dev && (std_mode == CT_STD_NONE || var_state_dev);
    const int interp = icrf->interp;
    // the one-launch path: what ct::merge_pivot_kernel addresses in the code domain, whole packets everywhere
    bool fast = n_batches >= 2 && n_batches <= kMaxMultiBatches && total <= 0x7fffffff && (has_state || (first && finalize)) &&
                (dtype == CT_DTYPE_U8 || dtype == CT_DTYPE_U16) && geom->channels > 0 && geom->h_tile > 0 && geom->width > 0 &&
                interp >= CT_INTERP_LOOKUP && interp <= CT_INTERP_NONE && !(flags & CT_MERGE_F64_MOMENTS) &&
                !merge_uses_reference_order(interp, std_mode, flags) &&
                !(std_mode != CT_STD_NONE && interp == CT_INTERP_LOOKUP && weight_mode == CT_WEIGHT_NONE);
    PivotArgs px{};
    bool clamp = false;
    const int64_t Ql = geom->h_tile * geom->width * geom->channels;
    const size_t tsize = dtype == CT_DTYPE_U8 ? 1 : 2;
    if (fast) fast = pivot_eligible(dtype, max_code, interp, interp == CT_INTERP_NONE ? 2 : icrf->n_points, flags, &px, &clamp);
    if (fast) {
        auto aligned = [](const void *p, size_t bytes) { return p == nullptr || (reinterpret_cast<uintptr_t>(p) % bytes) == 0; };
        fast = Ql % kPivotV == 0 && geom->image_stride % kPivotV == 0 && aligned(mean_out_dev, 8 * kPivotV) &&
               aligned(std_out_dev, 4 * kPivotV);
        for (int b = 0; fast && b < n_batches; ++b)
            fast = aligned(stack_devs[b], tsize * kPivotV) && (std_mode != CT_STD_EXPLICIT || aligned(std_devs[b], 4 * kPivotV));
    }
    // the reference-order kernel walks several batches per launch too (any dtype; packets or single elements)
    if (!fast && n_batches >= 2 && n_batches <= kMaxMergeBatches && total <= 65536 && (has_state || (first && finalize)) &&
        interp >= CT_INTERP_LOOKUP && interp <= CT_INTERP_NONE && merge_uses_reference_order(interp, std_mode, flags) &&
        !(std_mode != CT_STD_NONE && interp == CT_INTERP_LOOKUP && weight_mode == CT_WEIGHT_NONE) &&
        (dtype == CT_DTYPE_U8 || dtype == CT_DTYPE_U16 || dtype == CT_DTYPE_F32) && geom->channels > 0 && geom->h_tile > 0 &&
        geom->width > 0 && geom->h_global >= geom->h_tile && geom->row_offset >= 0 &&
        geom->row_offset + geom->h_tile <= geom->h_global && geom->layout >= CT_LAYOUT_NCHW && geom->layout <= CT_LAYOUT_NHWC_BGR &&
        (interp == CT_INTERP_NONE || (icrf->lut_dev && icrf->n_points >= 2)) && std_mode >= CT_STD_NONE && std_mode <= CT_STD_EXPLICIT &&
        (weight_mode == CT_WEIGHT_NONE || weight_mode == CT_WEIGHT_GAUSS) &&
        (!finalize || (mean_out_dev && (std_mode == CT_STD_NONE || std_out_dev))) &&
        geom->h_global * geom->width * geom->channels < ((int64_t)1 << 31) && geom->image_stride >= Ql) {
        const int64_t plane_g = geom->h_global * geom->width, plane_l = geom->h_tile * geom->width;
        MergeArgs a{};
        a.stack = stack_devs[0];
        a.std_stack = std_mode == CT_STD_EXPLICIT ? std_devs[0] : nullptr;
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
        a.batch = (int32_t)total;
        a.channels = geom->channels;
        a.n_points = interp == CT_INTERP_NONE ? 2 : icrf->n_points;
        a.std_value = std_value;
        a.weight_scale = 30.0f;
        a.inv_max_code = 1.0f;
        a.flags = flags;
        if (dtype != CT_DTYPE_F32 && ct_norm_constants(max_code, &a.norm.hi, &a.norm.lo) != CT_OK) return CT_ERR_UNSUPPORTED;
        MergeBatches mb{};
        mb.n_batches = n_batches;
        for (int b = 0; b < n_batches; ++b) {
            mb.batch_size[b] = batch_sizes[b];
            mb.batch_ptr[b] = stack_devs[b];
            mb.std_ptr[b] = std_mode == CT_STD_EXPLICIT ? std_devs[b] : nullptr;
        }
        return merge_reference_order(a, dtype, (uint32_t)Ql, interp, weight_mode, std_mode, static_cast<hipStream_t>(stream), &mb);
    }
    if (!fast && (flags & CT_MERGE_REQUIRE_ONE_LAUNCH)) return CT_ERR_UNSUPPORTED;  // (tests: make the route explicit)
    if (!fast) {
        // one launch per batch with the state in memory (exactly what the caller would have done)
        if (n_batches > 1 && !has_state) return CT_ERR_INVALID_ARGUMENT;
        int64_t n0 = 0;
        for (int b = 0; b < n_batches; ++b) {
            const uint32_t f = (flags & ~(CT_MERGE_FIRST_BATCH | CT_MERGE_FINALIZE)) | ((first && b == 0) ? CT_MERGE_FIRST_BATCH : 0u) |
                               ((finalize && b == n_batches - 1) ? CT_MERGE_FINALIZE : 0u);
            const int rc = ct_hdr_merge_batch(stack_devs[b], dtype, max_code, batch_sizes[b], geom, std_devs ? std_devs[b] : nullptr,
                                              std_mode, std_value, exposure_dev + n0, icrf, weight_mode, mean_state_dev,
                                              sumw_state_dev, var_state_dev, mean_out_dev, std_out_dev, f, stream);
            if (rc != CT_OK) return rc;
            n0 += batch_sizes[b];
        }
        return CT_OK;
    }
    // argument checks of ct_hdr_merge_batch that the fast path still owes
    if (geom->h_global < geom->h_tile || geom->row_offset < 0 || geom->row_offset + geom->h_tile > geom->h_global) return CT_ERR_INVALID_ARGUMENT;
    if (geom->layout < CT_LAYOUT_NCHW || geom->layout > CT_LAYOUT_NHWC_BGR) return CT_ERR_INVALID_ARGUMENT;
    if (interp != CT_INTERP_NONE && (!icrf->lut_dev || icrf->n_points < 2)) return CT_ERR_INVALID_ARGUMENT;
    if (std_mode < CT_STD_NONE || std_mode > CT_STD_EXPLICIT) return CT_ERR_INVALID_ARGUMENT;
    if (weight_mode != CT_WEIGHT_NONE && weight_mode != CT_WEIGHT_GAUSS) return CT_ERR_INVALID_ARGUMENT;
    if (finalize && (!mean_out_dev || (std_mode != CT_STD_NONE && !std_out_dev))) return CT_ERR_INVALID_ARGUMENT;
    const int64_t plane_g = geom->h_global * geom->width, plane_l = geom->h_tile * geom->width;
    if (plane_g * geom->channels >= (int64_t)1 << 31) return CT_ERR_TOO_LARGE;
    if (geom->image_stride < Ql) return CT_ERR_INVALID_ARGUMENT;
    MergeArgs a{};
    a.stack = stack_devs[0];
    a.std_stack = std_mode == CT_STD_EXPLICIT ? std_devs[0] : nullptr;
    a.exposure = exposure_dev;
    a.lut = icrf->lut_dev;
    a.mean_state = has_state ? mean_state_dev : nullptr;
    a.sumw_state = has_state ? sumw_state_dev : nullptr;
    a.var_state = has_state ? var_state_dev : nullptr;
    a.mean_out = mean_out_dev;
    a.std_out = std_out_dev;
    a.image_stride = geom->image_stride;
    a.q_begin = 0;
    a.q_count = (uint32_t)Ql;
    a.tile.plane_local = (uint32_t)plane_l;
    a.tile.chan_skip = (uint32_t)(plane_g - plane_l);
    a.tile.base = (uint32_t)(geom->row_offset * geom->width);
    a.tile.layout = (uint32_t)geom->layout;
    a.tile.channels = (uint32_t)geom->channels;
    a.batch = (int32_t)total;
    a.channels = geom->channels;
    a.n_points = interp == CT_INTERP_NONE ? 2 : icrf->n_points;
    a.std_value = std_value;
    a.weight_scale = 30.0f;
    a.inv_max_code = (float)(1.0 / (double)max_code);
    a.flags = flags;
    if (ct_norm_constants(max_code, &a.norm.hi, &a.norm.lo) != CT_OK) return CT_ERR_UNSUPPORTED;
    px.n_batches = n_batches;
    px.fresh = first ? 1 : 0;
    px.probe = batch_sizes[0] / 2;
    px.retry_count = g_merge_retry_counter;
    for (int b = 0; b < n_batches; ++b) {
        px.batch_size[b] = batch_sizes[b];
        px.batch_ptr[b] = stack_devs[b];
        px.std_ptr[b] = std_mode == CT_STD_EXPLICIT ? std_devs[b] : nullptr;
    }
    return merge_pivot_multi(a, px, dtype, clamp, interp, weight_mode, std_mode, static_cast<hipStream_t>(stream));
}

#endif  // CT_MERGE_PART != 1
