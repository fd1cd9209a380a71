// ct_api.cpp -- host-only pieces of the C ABI: version, error strings, normalisation constants.
#include <math.h>
#include <mutex>

#include "../../include/clair_hip.h"

extern "C" int ct_abi_version(void) { return CT_ABI_VERSION; }

extern "C" const char *ct_error_string(int code)
{
    switch (code) {
        case CT_OK: return "ok";
        case CT_ERR_INVALID_ARGUMENT: return "invalid argument";
        case CT_ERR_UNSUPPORTED: return "unsupported dtype or mode";
        case CT_ERR_LAUNCH: return "HIP kernel launch failed";
        case CT_ERR_NO_GRADIENT_PATH: return "no gradient path from the output to the image (reference raises RuntimeError)";
        case CT_ERR_TOO_LARGE: return "dimension too large for the kernel's indexing";
    }
    return "unknown error";
}

// The reference normalises integer codes with one float32 division u / max_code
// (clair_torch/common/general_functions.py:377).  The kernels use fma(u, hi, u * lo) with hi + lo ~ 1/max_code;
// this routine derives the pair and proves on the host, over every code 0..max_code, that the two agree bit
// for bit.  Returns CT_ERR_UNSUPPORTED if they do not (then the caller must hand in float32 pixels instead).
extern "C" int ct_norm_constants(float max_code, float *hi, float *lo)
{
    static std::mutex mu;
    static float cached_max = 0.0f, cached_hi = 0.0f, cached_lo = 0.0f;
    static int cached_rc = CT_ERR_UNSUPPORTED;
    if (!(max_code >= 1.0f) || max_code > 65535.0f || floorf(max_code) != max_code) return CT_ERR_UNSUPPORTED;
    std::lock_guard<std::mutex> lock(mu);
    if (cached_max != max_code) {
        const double rd = 1.0 / (double)max_code;
        const float h = (float)rd;
        const float l = (float)(rd - (double)h);
        int rc = CT_OK;
        for (int u = 0; u <= (int)max_code; ++u) {
            const float uf = (float)u;
            volatile float ref = uf / max_code;
            volatile float got = fmaf(uf, h, uf * l);
            if (ref != got) {
                rc = CT_ERR_UNSUPPORTED;
                break;
            }
        }
        cached_max = max_code;
        cached_hi = h;
        cached_lo = l;
        cached_rc = rc;
    }
    *hi = cached_hi;
    *lo = cached_lo;
    return cached_rc;
}

// Folded LUT coordinate for integer codes: s = fma(u, hi, u * lo) with hi + lo ~ (L-1) / max_code, i.e. the correctly
// rounded u * (L-1) / max_code, instead of the reference's two roundings fl(fl(u / max_code) * (L-1))
// (clair_torch/models/base.py:166).  The interpolation interval floor(s) (and round-half-even(s) for LOOKUP) selects
// which LUT samples -- and which derivative -- a pixel uses, so the fold is only allowed when both agree with the
// reference's float32 arithmetic for EVERY code; this routine checks that exhaustively (<= 65536 codes) and caches
// the verdict per (max_code, L).
extern "C" int ct_index_constants(float max_code, int n_points, float *hi, float *lo)
{
    static std::mutex mu;
    static float cached_max = 0.0f, cached_hi = 0.0f, cached_lo = 0.0f;
    static int cached_L = 0, cached_rc = CT_ERR_UNSUPPORTED;
    if (!(max_code >= 1.0f) || max_code > 65535.0f || floorf(max_code) != max_code || n_points < 2)
        return CT_ERR_UNSUPPORTED;
    std::lock_guard<std::mutex> lock(mu);
    if (cached_max != max_code || cached_L != n_points) {
        const float top = (float)(n_points - 1);
        const double rd = (double)top / (double)max_code;
        const float h = (float)rd;
        const float l = (float)(rd - (double)h);
        int rc = CT_OK;
        for (int u = 0; u <= (int)max_code; ++u) {
            const float uf = (float)u;
            volatile float x = uf / max_code;
            volatile float s_ref = x * top;
            volatile float s_new = fmaf(uf, h, uf * l);
            if (floorf(s_ref) != floorf(s_new) || nearbyintf(s_ref) != nearbyintf(s_new) || s_new > top || s_new < 0.0f) {
                rc = CT_ERR_UNSUPPORTED;
                break;
            }
        }
        cached_max = max_code;
        cached_L = n_points;
        cached_hi = h;
        cached_lo = l;
        cached_rc = rc;
    }
    *hi = cached_hi;
    *lo = cached_lo;
    return cached_rc;
}

// merge_pivot_kernel addresses the LINEAR table by the interval index computed from the raw code in integer arithmetic:
// floor(u / step) with step = max_code / (L-1), as (u * M) >> 32 with M = ceil(2^32 / step) < 2^24 (one
// v_mul_hi_u32_u24), or the code itself when step == 1.  Allowed only if step is an integer and the result equals the
// reference's float32 interval floor(fl(fl(u / max_code) * (L-1))) (clair_torch/models/base.py:166-168) for EVERY code;
// checked exhaustively here and cached per (max_code, L).  *index_mul = 0 means "the code is the index".
extern "C" int ct_pivot_index_constants(float max_code, int n_points, uint32_t *index_mul, float *step)
{
    static std::mutex mu;
    static float cached_max = 0.0f, cached_step = 0.0f;
    static uint32_t cached_mul = 0;
    static int cached_L = 0, cached_rc = CT_ERR_UNSUPPORTED;
    if (!(max_code >= 1.0f) || max_code > 65535.0f || floorf(max_code) != max_code || n_points < 2)
        return CT_ERR_UNSUPPORTED;
    std::lock_guard<std::mutex> lock(mu);
    if (cached_max != max_code || cached_L != n_points) {
        const int maxc = (int)max_code, top = n_points - 1;
        int rc = CT_OK;
        uint32_t mul = 0;
        float st = 0.0f;
        if (top > maxc || maxc % top != 0) {
            rc = CT_ERR_UNSUPPORTED;
        } else {
            const uint32_t istep = (uint32_t)(maxc / top);
            st = (float)istep;
            if (istep > 1) {
                const uint64_t m = ((1ull << 32) + istep - 1) / istep;
                if (m >= (1ull << 24)) rc = CT_ERR_UNSUPPORTED;
                mul = (uint32_t)m;
            }
            for (int u = 0; rc == CT_OK && u <= maxc; ++u) {
                volatile float x = (float)u / max_code;
                volatile float s_ref = x * (float)top;
                const uint32_t ref = (uint32_t)floorf(s_ref);
                const uint32_t got = istep > 1 ? (uint32_t)(((uint64_t)(uint32_t)u * mul) >> 32) : (uint32_t)u;
                if (ref != got || got > (uint32_t)top) rc = CT_ERR_UNSUPPORTED;
            }
        }
        cached_max = max_code;
        cached_L = n_points;
        cached_mul = mul;
        cached_step = st;
        cached_rc = rc;
    }
    *index_mul = cached_mul;
    *step = cached_step;
    return cached_rc;
}

// The typed-load path of merge_pivot_kernel receives the codes as floats and forms the interval by ONE FMA that rounds
// toward minus infinity: as_uint(fma_rtn(u, r, 1.5 * 2^23)) - 0x4B400000 with r = 1 / step rounded up.  Emulated here in
// exact integer arithmetic (u * r is a multiple of 2^e with r = M 2^e, its floor is a 64-bit shift; adding it to the
// magic number and rounding down at ulp 1 is exact) and compared with the reference's float32 interval
// floor(fl(fl(u / max_code) * (L-1))) (clair_torch/models/base.py:166-168) for EVERY code; cached per (max_code, L).
extern "C" int ct_pivot_floor_constants(float max_code, int n_points, float *rcp_step)
{
    static std::mutex mu;
    static float cached_max = 0.0f, cached_rcp = 0.0f;
    static int cached_L = 0, cached_rc = CT_ERR_UNSUPPORTED;
    if (!(max_code >= 1.0f) || max_code > 65535.0f || floorf(max_code) != max_code || n_points < 2)
        return CT_ERR_UNSUPPORTED;
    if (n_points - 1 > (int)max_code || (int)max_code % (n_points - 1) != 0) return CT_ERR_UNSUPPORTED;  // whole steps only
    const float step = (float)((int)max_code / (n_points - 1));
    std::lock_guard<std::mutex> lock(mu);
    if (cached_max != max_code || cached_L != n_points) {
        const int maxc = (int)max_code, top = n_points - 1;
        float r = (float)(1.0 / (double)step);
        if ((double)r < 1.0 / (double)step) r = nextafterf(r, 2.0f);
        int e = 0;
        const double m = frexp((double)r, &e);         // r = m 2^e, m in [0.5, 1)
        const uint64_t M = (uint64_t)ldexp(m, 24);      // 24-bit integer significand (exact: r is a float)
        const int shift = 24 - e;                       // r = M 2^-shift (r <= 1, so shift >= 23)
        int rc = (shift >= 0 && shift < 63) ? CT_OK : CT_ERR_UNSUPPORTED;
        for (int u = 0; rc == CT_OK && u <= maxc; ++u) {
            volatile float x = (float)u / max_code;
            volatile float s_ref = x * (float)top;
            const uint64_t got = ((uint64_t)u * M) >> shift;
            if (got != (uint64_t)floorf(s_ref) || got > (uint64_t)top) rc = CT_ERR_UNSUPPORTED;
        }
        cached_max = max_code;
        cached_L = n_points;
        cached_rcp = r;
        cached_rc = rc;
    }
    *rcp_step = cached_rcp;
    return cached_rc;
}

// Table addressing of ct::merge_pivot_kernel for ANY (max_code, L): entry(u) = min(floor(u * scale), last) by one FMA that
// rounds toward minus infinity on the code held as a float, with scale ~ (L-1) / max_code (LINEAR: entry = interpolation
// interval) or 2 (L-1) / max_code (LOOKUP: entry j = half interval, LUT sample (j + 1) >> 1).  Proved here for EVERY code
// the container can hold (0..dtype_max, i.e. also codes above max_code, which the reference clamps to the top of the LUT,
// clair_torch/models/base.py:146,166) against the reference's float32 arithmetic fl(fl(u / max_code) * (L-1)): floor for
// LINEAR (base.py:166-168), round-half-even for LOOKUP (base.py:146).  u * scale is evaluated exactly in integers
// (scale = M 2^-shift), as the FMA does before its one rounding.  A few neighbouring floats are tried for the scale; the
// verdict is cached per argument set.  CT_ERR_UNSUPPORTED = no such scale (the generic kernel runs instead).
extern "C" int ct_pivot_interval_constants(float max_code, int n_points, int lookup, int dtype_max, float *scale)
{
    struct Entry { float max_code; int n_points, lookup, dtype_max, rc; float scale; };
    static std::mutex mu;
    static Entry cache[8];
    static int used = 0, next = 0;
    if (!(max_code >= 1.0f) || max_code > 65535.0f || floorf(max_code) != max_code || n_points < 2 || dtype_max < (int)max_code ||
        dtype_max > 65535)
        return CT_ERR_UNSUPPORTED;
    std::lock_guard<std::mutex> lock(mu);
    for (int k = 0; k < used; ++k)
        if (cache[k].max_code == max_code && cache[k].n_points == n_points && cache[k].lookup == lookup && cache[k].dtype_max == dtype_max) {
            *scale = cache[k].scale;
            return cache[k].rc;
        }
    const int top = n_points - 1, last = lookup ? 2 * top : top;
    const double want = (double)(lookup ? 2 * top : top) / (double)max_code;
    float cand[4];
    cand[0] = (float)want;
    if ((double)cand[0] < want) cand[0] = nextafterf(cand[0], INFINITY);   // rounded up first: exact multiples stay in their entry
    cand[1] = nextafterf(cand[0], INFINITY);
    cand[2] = nextafterf(cand[0], 0.0f);
    cand[3] = nextafterf(cand[2], 0.0f);
    int rc = CT_ERR_UNSUPPORTED;
    float found = 0.0f;
    for (int c = 0; c < 4 && rc != CT_OK; ++c) {
        const float r = cand[c];
        if (!((double)dtype_max * (double)r < 4194304.0)) continue;   // entry + 1.5 * 2^23 must stay exact at ulp 1
        int e = 0;
        const double m = frexp((double)r, &e);
        const uint64_t M = (uint64_t)ldexp(m, 24);
        const int shift = 24 - e;
        if (shift < 0 || shift >= 63) continue;
        bool ok = true;
        for (int u = 0; ok && u <= dtype_max; ++u) {
            volatile float x = (float)u / max_code;
            volatile float s_ref = x * (float)top;
            float sc = s_ref < 0.0f ? 0.0f : (s_ref > (float)top ? (float)top : s_ref);
            uint64_t got = ((uint64_t)u * M) >> shift;
            if (got > (uint64_t)last) got = (uint64_t)last;
            if (lookup) {
                float rr = nearbyintf(s_ref);
                rr = rr < 0.0f ? 0.0f : (rr > (float)top ? (float)top : rr);
                ok = ((got + 1) >> 1) == (uint64_t)rr;
            } else {
                ok = got == (uint64_t)floorf(sc);
            }
        }
        if (ok) {
            rc = CT_OK;
            found = r;
        }
    }
    Entry &slot = cache[used < 8 ? used++ : (next++ % 8)];
    slot = Entry{max_code, n_points, lookup, dtype_max, rc, found};
    *scale = found;
    return rc;
}
