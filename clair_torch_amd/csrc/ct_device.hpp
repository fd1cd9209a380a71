// ct_device.hpp -- device-side building blocks shared by the gfx950 kernels.
//
// Everything here is written for CDNA4 only (64-wide wavefronts, 160 KiB LDS per CU); there is no
// other backend.  Float32 expressions whose rounding must equal the reference's eager float32 ops are
// written with explicit __fmul_rn/__fadd_rn-free plain operators and the library is compiled with
// -ffp-contract=off, so the only fused multiply-adds are the ones spelled __builtin_fmaf below.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/clair_hip.h"

namespace ct {

constexpr int kBlock = 256;  // 4 wavefronts per workgroup

// ---- pixel value from a stored element ---------------------------------------------------------
// Integer codes are normalised exactly as the reference's Normalize(0, max_code) on a float32 tensor
// (clair_torch/common/general_functions.py:359-388): one correctly rounded float32 division u / max_code.
// fma(u, r_hi, u * r_lo) with r_hi + r_lo = 1/max_code reproduces that division bit for bit for every
// code (verified exhaustively on the host in ct_norm_constants before a launch uses it).
struct NormConst {
    float hi, lo;
};

template <typename T>
__device__ __forceinline__ float to_pixel(T v, NormConst nc)
{
    if constexpr (sizeof(T) == 4) {
        return v;
    } else {
        const float u = (float)v;
        return __builtin_fmaf(u, nc.hi, u * nc.lo);
    }
}

// ---- LUT staging in LDS ------------------------------------------------------------------------
// LINEAR : float2 {g[i], g[min(i+1, L-1)]}            -> one ds_read_b64 per sample
// CATMULL: float4 {g[i-1], g[i], g[i+1], g[i+2]} clamped -> one ds_read_b128 per sample
// LOOKUP : float  g[i]
// Rows are `row_pitch_bytes(interp, L)` apart.
__device__ __host__ __forceinline__ constexpr int lut_entry_bytes(int interp)
{
    return interp == CT_INTERP_LINEAR ? 8 : (interp == CT_INTERP_CATMULL ? 16 : 4);
}

// DELTA (LINEAR only): store {g[i], g[i+1] - g[i]} instead of {g[i], g[i+1]} -- the merge kernel's FMA lerp and its
// derivative both want the difference (the float32 subtraction is the same one the reference's backward performs).
template <int INTERP, bool DELTA = false>
__device__ __forceinline__ void stage_lut(char *lds, const float *__restrict__ lut, int C, int L)
{
    if constexpr (INTERP == CT_INTERP_NONE) {
        return;
    } else {
        const int total = C * L;
        for (int k = threadIdx.x; k < total; k += blockDim.x) {
            const int r = k / L, i = k - r * L;
            const float *row = lut + (size_t)r * L;
            const int im = i > 0 ? i - 1 : 0, i1 = i + 1 < L ? i + 1 : L - 1, i2 = i + 2 < L ? i + 2 : L - 1;
            if constexpr (INTERP == CT_INTERP_LINEAR) {
                reinterpret_cast<float2 *>(lds)[k] = make_float2(row[i], DELTA ? row[i1] - row[i] : row[i1]);
            } else if constexpr (INTERP == CT_INTERP_CATMULL) {
                reinterpret_cast<float4 *>(lds)[k] = make_float4(row[im], row[i], row[i1], row[i2]);
            } else {
                reinterpret_cast<float *>(lds)[k] = row[i];
            }
        }
    }
}

// d(G * result)/dt of the Catmull-Rom sum exactly as PyTorch's autograd evaluates the backward of
// clair_torch/models/base.py:199-224: the four products w_k * g_k run first (G_k = G * g_k), then the nodes of w3, w2, w1,
// w0; gradients meeting at t3, t2 and t are added in arrival order; t3 = t2 * t and t2 = t * t run last (t * t feeds t
// twice).  Un-fused float32 (-ffp-contract=off).  oracle/eager_torch.icrf_forward_backward_reference_order is the same
// sequence on the CPU and reproduces the reference's recorded uncertainties bit for bit (tests/test_oracle_golden.py).
// The float32 sum cancels ~100x, so any other order (e.g. the better conditioned closed form below) differs from the
// reference by up to 1e-4 on single samples.
__device__ __forceinline__ float catmull_backward_ref(float t, float t2, const float4 g, float G)
{
    const float g0 = G * g.x, g1 = G * g.y, g2 = G * g.z, g3 = G * g.w;
    const float a3 = ((0.5f * g3 - 1.5f * g2) + 1.5f * g1) - 0.5f * g0;
    const float a2 = (((2.0f * g2 - 0.5f * g3) - 2.5f * g1) + g0) + a3 * t;
    return (((0.5f * g2 - 0.5f * g0) + a3 * t2) + a2 * t) + a2 * t;
}

// One sample through the staged LUT.
//   EXACT   : reproduce the reference's float32 operation order bit for bit (linearize path);
//             otherwise the lerp is a single FMA (merge path, 1e-5 tolerance).
//   RANGED  : the value comes from an unsigned integer code, so it is >= 0 and only the upper clamp can act: a
//             code above max_code (12-bit data in a uint16 container normalised by 4095, say) gives x > 1, which the
//             reference clamps to the top of the LUT with zero gradient (base.py:146,166,190).  One v_min; LINEAR
//             needs no gradient mask because the last staged interval has zero slope (g[L-1] twice), CATMULL does.
// Returns f(x); dfdx receives df/dx (0 for LOOKUP).  `row_lds` points at this element's LUT row.
template <int INTERP, bool EXACT, bool RANGED>
__device__ __forceinline__ float icrf_sample(float x, const char *row_lds, float top, float &dfdx)
{
    if constexpr (INTERP == CT_INTERP_NONE) {
        dfdx = 1.0f;
        return x;
    } else if constexpr (INTERP == CT_INTERP_LOOKUP) {
        // clair_torch/models/base.py:146: (image * (L-1)).round().clamp(0, L-1); rintf = half to even
        float r = rintf(x * top);
        r = fminf(fmaxf(r, 0.0f), top);
        dfdx = 0.0f;
        return reinterpret_cast<const float *>(row_lds)[(int)r];
    } else {
        const float sraw = x * top;  // base.py:166 / 190
        float s, pass = 1.0f;
        if constexpr (!RANGED) {
            s = fminf(fmaxf(sraw, 0.0f), top);
            pass = (sraw >= 0.0f && sraw <= top) ? 1.0f : 0.0f;  // clamp backward mask
        } else {
            s = fminf(sraw, top);
            if constexpr (INTERP == CT_INTERP_CATMULL) pass = sraw <= top ? 1.0f : 0.0f;
        }
        const float fl = floorf(s);
        const int i0 = (int)fl;
        const float fr = s - fl;  // base.py:170 (exact: Sterbenz)
        if constexpr (INTERP == CT_INTERP_LINEAR) {
            const float2 g = reinterpret_cast<const float2 *>(row_lds)[i0];
            const float dg = g.y - g.x;
            dfdx = dg * top;
            if constexpr (!RANGED) dfdx *= pass;  // (RANGED: dg = 0 on the clamped interval)
            if constexpr (EXACT) {
                const float a = g.x * (1.0f - fr);  // base.py:182: g0 * (1 - w) + g1 * w, un-fused
                const float b = g.y * fr;
                return a + b;
            } else {
                return __builtin_fmaf(dg, fr, g.x);
            }
        } else {  // CATMULL, base.py:184-226
            const float4 g = reinterpret_cast<const float4 *>(row_lds)[i0];
            const float t = fr;  // already in [0,1)
            const float t2 = t * t, t3 = t2 * t;
            const float w0 = -0.5f * t3 + t2 - 0.5f * t;
            const float w1 = 1.5f * t3 - 2.5f * t2 + 1.0f;
            const float w2 = -1.5f * t3 + 2.0f * t2 + 0.5f * t;
            const float w3 = 0.5f * t3 - 0.5f * t2;
            float r = w0 * g.x;
            r = r + w1 * g.y;
            r = r + w2 * g.z;
            r = r + w3 * g.w;
            if constexpr (EXACT) {
                // the reference's own float32 backward with unit upstream gradient (linearization.py:100-105)
                dfdx = (catmull_backward_ref(t, t2, g, 1.0f) * pass) * top;
            } else {
                // derivative of the basis in t, chained through s = x * (L-1)
                const float d0 = __builtin_fmaf(__builtin_fmaf(-1.5f, t, 2.0f), t, -0.5f);
                const float d2 = __builtin_fmaf(__builtin_fmaf(-4.5f, t, 4.0f), t, 0.5f);
                const float d3 = __builtin_fmaf(1.5f, t, -1.0f) * t;
                // sum (g_k - g_1) d_k: the d_k sum to zero, subtracting g_1 removes the ~100x cancellation
                const float acc = __builtin_fmaf(d0, g.x - g.y, __builtin_fmaf(d2, g.z - g.y, d3 * (g.w - g.y)));
                dfdx = acc * top * pass;
            }
            return r;
        }
    }
}

// d(G * f(x))/dx for float32 x in autograd's operation order (the upstream gradient multiplies the LUT taps FIRST):
// what ICRFModelBase.forward's backward hands to the image (clair_torch/models/base.py:160-226), bit for bit.
template <int INTERP>
__device__ __forceinline__ float icrf_grad_reference_order(float x, const char *row_lds, float top, float G)
{
    if constexpr (INTERP == CT_INTERP_LOOKUP || INTERP == CT_INTERP_NONE) {
        return INTERP == CT_INTERP_NONE ? G : 0.0f;
    } else {
        const float sraw = x * top;
        const float s = fminf(fmaxf(sraw, 0.0f), top);
        const float pass = (sraw >= 0.0f && sraw <= top) ? 1.0f : 0.0f;
        const float fl = floorf(s);
        const int i0 = (int)fl;
        const float t = s - fl;
        if constexpr (INTERP == CT_INTERP_LINEAR) {
            const float2 g = reinterpret_cast<const float2 *>(row_lds)[i0];
            return ((G * g.y - G * g.x) * pass) * top;  // fr receives G g1 first, then -(G g0)
        } else {
            const float4 g = reinterpret_cast<const float4 *>(row_lds)[i0];
            return (catmull_backward_ref(t, t * t, g, G) * pass) * top;
        }
    }
}

// Which LUT row the reference uses for the element with GLOBAL in-image flat index q (= (c*H + h)*W + w):
// LOOKUP uses the channel (base.py:149-155); LINEAR / CATMULL use flat_index % C (base.py:173-176, 216-219),
// and n*C*H*W is a multiple of C so the exposure index drops out.
template <int INTERP>
__device__ __forceinline__ int lut_row(uint32_t q_global, int channel, int C)
{
    if constexpr (INTERP == CT_INTERP_LOOKUP)
        return channel;
    else
        return (int)(q_global % (uint32_t)C);
}

// Typed buffer loads: with DATA_FORMAT 16_16_16_16 / 8_8_8_8 (16 / 8 for single codes) and NUM_FORMAT USCALED in the
// descriptor, buffer_load_format_xyzw delivers (float)code for four packed integer codes -- the conversion happens in the
// texture-data path between L1 and the registers, which is idle in these kernels, instead of one half-rate v_cvt per
// sample on the VALU.  Exact for every code and at the streaming rate of a plain load (tools/typed_load_probe.hip checks
// all 65 536 codes in every component on the device; profiles/r02_typed_load_probe.log).
typedef float f32x4_t __attribute__((ext_vector_type(4)));
__device__ f32x4_t raw_buffer_load_format_v4f32(__amdgpu_buffer_rsrc_t rsrc, int voffset, int soffset, int aux)
    __asm("llvm.amdgcn.raw.ptr.buffer.load.format.v4f32");
__device__ float raw_buffer_load_format_f32(__amdgpu_buffer_rsrc_t rsrc, int voffset, int soffset, int aux)
    __asm("llvm.amdgcn.raw.ptr.buffer.load.format.f32");
// word 3 of the gfx9 buffer descriptor: DST_SEL_{X,Y,Z,W} = R,G,B,A (4,5,6,7) | NUM_FORMAT << 12 (2 = USCALED) |
// DATA_FORMAT << 15 (1 = 8, 2 = 16, 10 = 8_8_8_8, 12 = 16_16_16_16)
template <typename T, int V>
constexpr uint32_t uscaled_format_word()
{
    static_assert((V == 1 || V == 4) && (sizeof(T) == 1 || sizeof(T) == 2), "typed loads: 1 or 4 codes of 8 or 16 bits");
    constexpr uint32_t sel = V == 4 ? (4u | (5u << 3) | (6u << 6) | (7u << 9)) : 4u;
    constexpr uint32_t dfmt = V == 4 ? (sizeof(T) == 2 ? 12u : 10u) : (sizeof(T) == 2 ? 2u : 1u);
    return sel | (2u << 12) | (dfmt << 15);
}
// V = 1 or 4 codes at `base` (wave-uniform) + `offset` bytes (per thread, 32 bits: the descriptor spans 4 GiB from the base)
template <typename T, int V, int AUX = 0>
__device__ __forceinline__ void load_codes_as_float(uint64_t base, uint32_t offset, float (&out)[V])
{
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(base), 0, 0xffffffff, (int)uscaled_format_word<T, V>());
    if constexpr (V == 4) {
        const f32x4_t v = raw_buffer_load_format_v4f32(rsrc, (int)offset, 0, AUX);
        out[0] = v.x;
        out[1] = v.y;
        out[2] = v.z;
        out[3] = v.w;
    } else {
        out[0] = raw_buffer_load_format_f32(rsrc, (int)offset, 0, AUX);
    }
}
// the reference's CastTo(float32) + Normalize(0, max) on a code that is already a float
__device__ __forceinline__ float code_to_pixel(float u, NormConst nc) { return __builtin_fmaf(u, nc.hi, u * nc.lo); }

// LUT interval of V codes held as floats: floor(px / step) = mantissa of fma(px, r, 1.5 * 2^23) when that one FMA rounds
// toward minus infinity (r = 1 / step rounded up, so exact multiples of the step do not fall below their interval; the
// error of px * r stays below 1.6e-5 < 1 / step).  FP_ROUND of the MODE register is switched for exactly these V
// instructions -- one asm block, so the compiler cannot move any other arithmetic into it; the scalar unit is idle.
// The result keeps the magic number's bits: as_uint(t) = 0x4B400000 + interval; the callers fold that constant into
// the row offset they add anyway.  Host-verified for every code against the reference's float32 index
// (ct_pivot_index_constants) and on the device by tools/typed_load_probe.hip.
constexpr float kFloorMagic = 12582912.0f;        // 1.5 * 2^23: ulp 1
constexpr uint32_t kFloorMagicBits = 0x4B400000u;
template <int V>
__device__ __forceinline__ void floor_index_bits(const float (&px)[V], float r, float magic, float (&t)[V])
{
    if constexpr (V == 4) {
        asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 2\n\t"
                     "v_fma_f32 %0, %4, %8, %9\n\t"
                     "v_fma_f32 %1, %5, %8, %9\n\t"
                     "v_fma_f32 %2, %6, %8, %9\n\t"
                     "v_fma_f32 %3, %7, %8, %9\n\t"
                     "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0"
                     : "=&v"(t[0]), "=&v"(t[1]), "=&v"(t[2]), "=&v"(t[3])
                     : "v"(px[0]), "v"(px[1]), "v"(px[2]), "v"(px[3]), "v"(r), "v"(magic));
    } else {
        static_assert(V == 1, "1 or 4 codes");
        // (s_nop: gfx9 wants 2 wait states between two s_setreg writes of one hwreg; the assembler cannot see this block)
        asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 2\n\t"
                     "v_fma_f32 %0, %1, %2, %3\n\t"
                     "s_nop 1\n\t"
                     "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0"
                     : "=&v"(t[0])
                     : "v"(px[0]), "v"(r), "v"(magic));
    }
}

// Two intervals per code (experiment: a second table with its own step): eight FMAs in one round-down block.
template <int V>
__device__ __forceinline__ void floor_index_bits2(const float (&px)[V], float r1, float r2, float magic, float (&t1)[V], float (&t2)[V])
{
    static_assert(V == 4, "packets of 4 codes");
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 2\n\t"
                 "v_fma_f32 %0, %8, %12, %14\n\t"
                 "v_fma_f32 %1, %9, %12, %14\n\t"
                 "v_fma_f32 %2, %10, %12, %14\n\t"
                 "v_fma_f32 %3, %11, %12, %14\n\t"
                 "v_fma_f32 %4, %8, %13, %14\n\t"
                 "v_fma_f32 %5, %9, %13, %14\n\t"
                 "v_fma_f32 %6, %10, %13, %14\n\t"
                 "v_fma_f32 %7, %11, %13, %14\n\t"
                 "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0"
                 : "=&v"(t1[0]), "=&v"(t1[1]), "=&v"(t1[2]), "=&v"(t1[3]), "=&v"(t2[0]), "=&v"(t2[1]), "=&v"(t2[2]), "=&v"(t2[3])
                 : "v"(px[0]), "v"(px[1]), "v"(px[2]), "v"(px[3]), "v"(r1), "v"(r2), "v"(magic));
}

// LDS byte address of table entry `interval` of the row at byte offset `row`: (bits(t) << 3) + (row - (0x4B400000 << 3))
// mod 2^32, one v_lshl_add_u32; the row constant is formed once per tile.  (Tried and rejected: the same address by one
// full-rate FMA against the inline integer constant 8 read as the denormal 8 * 2^-149 -- exact, but denormal operands
// take a slow path: 0.879 against 0.856 ms on C2, profiles/r02_typed_load_ab.log.)
template <int SHIFT = 3>
__device__ __forceinline__ uint32_t lds_row_constant(int row_bytes) { return (uint32_t)row_bytes - (kFloorMagicBits << SHIFT); }
template <int SHIFT = 3>
__device__ __forceinline__ uint32_t lds_entry_address(float t, uint32_t row_constant) { return (__float_as_uint(t) << SHIFT) + row_constant; }

// Code-domain LINEAR table for integer stacks whose LUT step is a whole number of codes: entry i = {g[i], (g[i+1]-g[i]) / step}
// (the last interval has zero slope), f(code) = g[i] + slope * (code - i * step) with the offset formed exactly.
__device__ __forceinline__ void stage_lut_slope(char *lds, const float *__restrict__ lut, int C, int L, float step)
{
    const int total = C * L;
    for (int k = threadIdx.x; k < total; k += blockDim.x) {
        const int r = k / L, i = k - r * L;
        const float *row = lut + (size_t)r * L;
        const float g0 = row[i], g1 = row[i + 1 < L ? i + 1 : L - 1];
        reinterpret_cast<float2 *>(lds)[k] = make_float2(g0, (g1 - g0) / step);
    }
}

// Streaming store of a whole packet: outputs are written once and never re-read by these kernels, so they are
// stored non-temporally (no L2 allocation competing with the input stream).
#ifndef CT_STREAM_STORES_NT
#define CT_STREAM_STORES_NT 1
#endif
template <typename P>
__device__ __forceinline__ void store_stream(P *dst, const P &v)
{
    if constexpr (!CT_STREAM_STORES_NT) {
        *dst = v;
        return;
    }
    constexpr int kWords = sizeof(P) / 4;
    static_assert(sizeof(P) % 4 == 0, "packet must be dword sized");
    const uint32_t *src = reinterpret_cast<const uint32_t *>(&v);
    uint32_t *d = reinterpret_cast<uint32_t *>(dst);
    if constexpr (kWords % 4 == 0) {
        typedef uint32_t u4 __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int k = 0; k < kWords / 4; ++k)
            __builtin_nontemporal_store(*reinterpret_cast<const u4 *>(src + 4 * k), reinterpret_cast<u4 *>(d) + k);
    } else if constexpr (kWords % 2 == 0) {
        typedef uint32_t u2 __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int k = 0; k < kWords / 2; ++k)
            __builtin_nontemporal_store(*reinterpret_cast<const u2 *>(src + 2 * k), reinterpret_cast<u2 *>(d) + k);
    } else {
#pragma unroll
        for (int k = 0; k < kWords; ++k) __builtin_nontemporal_store(src[k], d + k);
    }
}

// Tile geometry -> global flat index of local element ql (see ct_geometry in clair_hip.h).
struct TileMap {
    uint32_t plane_local;  // H_tile * W
    uint32_t chan_skip;    // (H_global - H_tile) * W
    uint32_t base;         // row_offset * W
    uint32_t layout;       // CT_LAYOUT_* of the input stack
    uint32_t channels;     // C (used by the interleaved layouts)
    __device__ __forceinline__ void locate(uint32_t ql, int &channel, uint32_t &q_global) const
    {
        channel = (int)(ql / plane_local);
        q_global = ql + (uint32_t)channel * chan_skip + base;
    }
    // Interleaved RGB / BGR (C == 3, the layout OpenCV decodes to) with constant divisors: memory element m -> channel
    // plane c, pixel, and the planar index; a runtime 32-bit division costs ~30 instructions each, three of them per
    // element made the interleaved merge 9 % slower than the planar one (profiles/r03_layout_ingest.md).
    __device__ __forceinline__ void interleaved3(uint32_t m, uint32_t &c, uint32_t &pixel) const
    {
        pixel = m / 3u;
        const uint32_t cm = m - 3u * pixel;
        c = layout == CT_LAYOUT_NHWC_BGR ? 2u - cm : cm;
    }
    // Memory element m of one image -> planar (C, H_tile, W) index.  For NCHW this is the identity; for the
    // interleaved layouts element m is channel m % C of pixel m / C (channel order reversed for BGR).
    __device__ __forceinline__ uint32_t planar_index(uint32_t m) const
    {
        if (layout == CT_LAYOUT_NCHW) return m;
        const uint32_t pixel = m / channels, cm = m - pixel * channels;
        const uint32_t c = layout == CT_LAYOUT_NHWC_BGR ? channels - 1 - cm : cm;
        return c * plane_local + pixel;
    }
};

// ---- host: launch geometry -----------------------------------------------------------------------------------------
// Number of compute units of the current device (cached); 256 on MI355X.
static inline int compute_units()
{
    static int n_cu = 0;
    if (n_cu == 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess &&
            hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0)
            n_cu = v;
        else
            n_cu = 256;
    }
    return n_cu;
}

// Workgroups that can be resident on the whole device at once, for kernels that walk their work in a grid-stride
// loop: a grid that is a multiple of this runs as full rounds of equal work.  (1026 long-running workgroups on 256
// one-slot units run 4 full rounds plus 2 stragglers that cost a whole fifth round.)
static inline int resident_workgroups(size_t lds_bytes, int block_threads)
{
    const size_t by_lds = lds_bytes ? (size_t)(160 * 1024) / lds_bytes : 64;
    const size_t by_waves = (size_t)2048 / (size_t)block_threads;  // 32 wavefronts per unit
    size_t slots = by_lds < by_waves ? by_lds : by_waves;
    if (slots < 1) slots = 1;
    return compute_units() * (int)slots;
}

}  // namespace ct
