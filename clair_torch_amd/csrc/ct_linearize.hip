// ct_linearize.hip -- ICRF linearization of image frames (gfx950): value, propagated std, and backward.
//
//  * ct_linearize_std : body of linearize_dataset_generator (clair_torch/inference/linearization.py:95-106,132)
//                       for F frames per launch.  lin = f(x) with the reference's float32 operation order (bit-exact
//                       for LOOKUP/LINEAR), std = sqrt((f'(x) * sigma)^2) = |f'(x) * sigma|.
//  * ct_linearize_fwd : ICRFModelBase.forward (clair_torch/models/base.py:135-226) on a float32 (N,C,H,W) tensor.
//  * ct_linearize_bwd : its backward; the (C,L) LUT gradient is a scatter-add of every sample into <= 4 bins,
//                       privatised per workgroup in a float64 LDS histogram (ds_add_f64: 9-21 cycles per wavefront
//                       instruction on gfx950, against ~190 for ds_add_f32 on any address pattern,
//                       profiles/r01_lds_atomic_rates.log) and flushed with one global float atomic per bin per
//                       workgroup (global float atomics run at ~1.3 TB/s chip-wide; C*L*4 B per workgroup keeps the
//                       flush negligible).
//
// Roofline: HBM (sizeof(T) read + 4 or 8 B written per sample).  No reuse between workgroups, so no XCD remap.
#include <algorithm>
#include "ct_device.hpp"

namespace ct {

template <typename T, int V>
struct alignas(sizeof(T) * V) LPacket {
    T v[V];
};

struct LinArgs {
    const void *frames;
    const float *std_stack;
    const float *lut;
    float *lin_out;
    float *std_out;
    int64_t image_stride;
    int64_t out_stride;         // elements between consecutive output frames (C * H_tile * W, planar)
    uint32_t q_begin, q_count;  // local element range of this launch (per frame)
    uint32_t n_frames;
    TileMap tile;
    int32_t channels, n_points;
    NormConst norm;
    float std_value;
};

// grid.x covers the packets of one frame, grid.y walks frames
template <typename T, int V, int INTERP, int STD, bool WRITE_STD>
__global__ __launch_bounds__(kBlock) void linearize_kernel(const LinArgs a)
{
    extern __shared__ __align__(16) char lds[];
    constexpr bool kRanged = sizeof(T) != 4;
    constexpr int kEntry = lut_entry_bytes(INTERP);
    const int C = a.channels, L = a.n_points;
    stage_lut<INTERP>(lds, a.lut, C, L);
    __syncthreads();
    const uint32_t vec = blockIdx.x * (uint32_t)kBlock + threadIdx.x;
    if (vec * (uint32_t)V >= a.q_count) return;
    const uint32_t q0 = a.q_begin + vec * (uint32_t)V;
    const float top = (float)(L - 1);
    int row_off[V];
    if (a.tile.layout == CT_LAYOUT_NCHW) {
        // planar frames: the channel by comparisons, the row of the first element by ONE modulo (a constant divisor for
        // C == 3), the following elements by an add and a conditional subtract -- runtime 32-bit divisions cost ~30
        // instructions each, and two of them per element were a third of this kernel's arithmetic
        int ch = 0;
        for (int c = 1; c < C; ++c) ch += q0 >= (uint32_t)c * a.tile.plane_local ? 1 : 0;
        const uint32_t qg = q0 + (uint32_t)ch * a.tile.chan_skip + a.tile.base;
        uint32_t off = q0 - (uint32_t)ch * a.tile.plane_local;
        int r = C == 3 ? (int)(qg % 3u) : (int)(qg % (uint32_t)C);
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
    for (uint32_t f = blockIdx.y; f < a.n_frames; f += gridDim.y) {
        const int64_t off = (int64_t)f * a.image_stride + q0;
        // integer frames: typed buffer loads deliver the codes as floats (ct_device.hpp) -- frame base in SGPRs, 32-bit
        // per-thread byte offset, no v_cvt; float frames: plain 16-byte loads
        float xin[V];
        if constexpr (sizeof(T) != 4) {
            static_assert(V == 1 || V % 4 == 0, "packets of 1, 4 or 8 codes");
            const uint64_t base = reinterpret_cast<uint64_t>(a.frames) + (uint64_t)((int64_t)f * a.image_stride * (int64_t)sizeof(T));
            if constexpr (V == 1) {
                load_codes_as_float<T, 1>(base, q0 * (uint32_t)sizeof(T), xin);
            } else {
#pragma unroll
                for (int h = 0; h < V / 4; ++h) {
                    float part[4];
                    load_codes_as_float<T, 4>(base, (q0 + 4u * h) * (uint32_t)sizeof(T), part);
#pragma unroll
                    for (int k = 0; k < 4; ++k) xin[4 * h + k] = part[k];
                }
            }
        } else {
            const LPacket<T, V> pk = *reinterpret_cast<const LPacket<T, V> *>(static_cast<const T *>(a.frames) + off);
#pragma unroll
            for (int e = 0; e < V; ++e) xin[e] = pk.v[e];
        }
        LPacket<float, V> sp;
        if constexpr (STD == CT_STD_EXPLICIT) sp = *reinterpret_cast<const LPacket<float, V> *>(a.std_stack + off);
        LPacket<float, V> lo, so;
        [[maybe_unused]] bool tiny = false;  // some 0 < |grad * std| < 1e-18 in this packet
#pragma unroll
        for (int e = 0; e < V; ++e) {
            const float x = sizeof(T) != 4 ? code_to_pixel(xin[e], a.norm) : xin[e];
            float dfdx;
            lo.v[e] = icrf_sample<INTERP, true, kRanged>(x, lds + row_off[e], top, dfdx);
            if constexpr (WRITE_STD) {
                float sigma = 0.0f;
                if constexpr (STD == CT_STD_EXPLICIT) sigma = sp.v[e];
                if constexpr (STD == CT_STD_MULTIPLIER) sigma = x * a.std_value;  // datasets/base.py:133
                if constexpr (STD == CT_STD_CONSTANT) sigma = a.std_value;
                // linearization.py:106,132: sqrt((grad * std) ** 2).  In binary floating point the correctly rounded
                // square root of a correctly rounded square is |.| exactly unless the square underflows, so the sqrtf
                // expansion (~18 instructions) is only needed for 0 < |gs| < 1e-18
                const float ags = fabsf(dfdx * sigma);
                so.v[e] = STD == CT_STD_NONE ? 0.0f : ags;
                if constexpr (STD != CT_STD_NONE) tiny |= ags < 1e-18f && ags != 0.0f;
            }
        }
        if constexpr (WRITE_STD && STD != CT_STD_NONE) {
            // ... and runs behind a wave-uniform branch (practically never taken).  Written as a per-element `if` the
            // compiler if-converts it and every sample pays for the expansion: a third of this kernel's VALU work.
            if (__builtin_expect(__any(tiny), 0)) {
#pragma unroll
                for (int e = 0; e < V; ++e) {
                    const float g = so.v[e];  // |gs|; the square does not see the sign
                    if (g < 1e-18f && g != 0.0f) so.v[e] = sqrtf(g * g);
                }
            }
        }
        if (a.tile.layout != CT_LAYOUT_NCHW) {  // interleaved input -> planar outputs, element-wise stores
#pragma unroll
            for (int e = 0; e < V; ++e) {
                const int64_t oq = (int64_t)f * a.out_stride + a.tile.planar_index(q0 + e);
                a.lin_out[oq] = lo.v[e];
                if constexpr (WRITE_STD) a.std_out[oq] = so.v[e];
            }
            continue;
        }
        const int64_t ooff = (int64_t)f * a.out_stride + q0;
        store_stream(reinterpret_cast<LPacket<float, V> *>(a.lin_out + ooff), lo);
        if constexpr (WRITE_STD) store_stream(reinterpret_cast<LPacket<float, V> *>(a.std_out + ooff), so);
    }
}

// Planar frames, K packets of four elements per thread, the packets of one thread a whole workgroup apart: every load
// (8 bytes per lane for uint16 codes) and every store (16 bytes per lane) of a wavefront is dense, and K loads are in
// flight per thread.  The 8-elements-per-thread mapping of linearize_kernel writes 32 bytes per lane with two half-dense
// store instructions; the pixel-owning RGB kernel below, whose stores are dense, ran the same C4 workload in 0.70 ms
// against 0.84 ms (profiles/r03_layout_ingest.md) -- this kernel gives the planar layout the same access shape.
// Per-element arithmetic is the same function call as everywhere else (bit-identical results).  q_count: a multiple of 4.
template <typename T, int K, int INTERP, int STD, bool WRITE_STD>
__global__ __launch_bounds__(kBlock) void linearize_planar_kernel(const LinArgs a)
{
    extern __shared__ __align__(16) char lds[];
    constexpr bool kRanged = sizeof(T) != 4;
    constexpr int kEntry = lut_entry_bytes(INTERP);
    const int C = a.channels, L = a.n_points;
    stage_lut<INTERP>(lds, a.lut, C, L);
    __syncthreads();
    const float top = (float)(L - 1);
    const uint32_t n_packets = a.q_count / 4u;
    uint32_t q0[K];
    bool live[K];
    int row_off[K][4];
    const int skip_mod = (int)(a.tile.chan_skip % (uint32_t)C);
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const uint32_t pk = (blockIdx.x * (uint32_t)K + (uint32_t)k) * (uint32_t)kBlock + threadIdx.x;
        live[k] = pk < n_packets;
        q0[k] = a.q_begin + (live[k] ? pk : 0u) * 4u;
        int ch = 0;
        for (int c = 1; c < C; ++c) ch += q0[k] >= (uint32_t)c * a.tile.plane_local ? 1 : 0;
        const uint32_t qg = q0[k] + (uint32_t)ch * a.tile.chan_skip + a.tile.base;
        uint32_t off = q0[k] - (uint32_t)ch * a.tile.plane_local;
        int r = C == 3 ? (int)(qg % 3u) : (int)(qg % (uint32_t)C);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            row_off[k][e] = (INTERP == CT_INTERP_LOOKUP ? ch : r) * L * kEntry;
            int inc = 1;
            if (++off == a.tile.plane_local) {
                off = 0;
                ++ch;
                inc += skip_mod;
            }
            r += inc;
            r = r >= C ? r - C : r;
        }
    }
    if (!live[0]) return;
    for (uint32_t f = blockIdx.y; f < a.n_frames; f += gridDim.y) {
        float xin[K][4];
        LPacket<float, 4> sp[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {  // all loads first
            if constexpr (sizeof(T) != 4) {
                const uint64_t base = reinterpret_cast<uint64_t>(a.frames) + (uint64_t)((int64_t)f * a.image_stride * (int64_t)sizeof(T));
                load_codes_as_float<T, 4>(base, q0[k] * (uint32_t)sizeof(T), xin[k]);
            } else {
                const LPacket<T, 4> pk = *reinterpret_cast<const LPacket<T, 4> *>(static_cast<const T *>(a.frames) + (int64_t)f * a.image_stride + q0[k]);
#pragma unroll
                for (int e = 0; e < 4; ++e) xin[k][e] = pk.v[e];
            }
            if constexpr (STD == CT_STD_EXPLICIT) sp[k] = *reinterpret_cast<const LPacket<float, 4> *>(a.std_stack + (int64_t)f * a.image_stride + q0[k]);
        }
        LPacket<float, 4> lo[K], so[K];
        [[maybe_unused]] bool tiny = false;
#pragma unroll
        for (int k = 0; k < K; ++k) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float x = sizeof(T) != 4 ? code_to_pixel(xin[k][e], a.norm) : xin[k][e];
                float dfdx;
                lo[k].v[e] = icrf_sample<INTERP, true, kRanged>(x, lds + row_off[k][e], top, dfdx);
                so[k].v[e] = 0.0f;
                if constexpr (WRITE_STD && STD != CT_STD_NONE) {
                    float sigma = a.std_value;
                    if constexpr (STD == CT_STD_EXPLICIT) sigma = sp[k].v[e];
                    if constexpr (STD == CT_STD_MULTIPLIER) sigma = x * a.std_value;  // datasets/base.py:133
                    const float ags = fabsf(dfdx * sigma);  // linearization.py:106,132: sqrt((grad * std)^2) = |grad * std| ...
                    so[k].v[e] = ags;
                    tiny |= ags < 1e-18f && ags != 0.0f;    // ... unless the square underflows
                }
            }
        }
        if constexpr (WRITE_STD && STD != CT_STD_NONE) {
            if (__builtin_expect(__any(tiny), 0)) {
#pragma unroll
                for (int k = 0; k < K; ++k)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (so[k].v[e] < 1e-18f && so[k].v[e] != 0.0f) so[k].v[e] = sqrtf(so[k].v[e] * so[k].v[e]);
            }
        }
#pragma unroll
        for (int k = 0; k < K; ++k) {
            if (!live[k]) continue;
            const int64_t ooff = (int64_t)f * a.out_stride + q0[k];
            store_stream(reinterpret_cast<LPacket<float, 4> *>(a.lin_out + ooff), lo[k]);
            if constexpr (WRITE_STD) store_stream(reinterpret_cast<LPacket<float, 4> *>(a.std_out + ooff), so[k]);
        }
    }
}

constexpr int kLinPackets = 3;  // packets per thread of linearize_planar_kernel (12 elements, like the RGB kernel)

template <typename T, int INTERP, int STD, bool WRITE_STD>
static int lin_launch_planar(const LinArgs &a, hipStream_t s)
{
    if (a.q_count == 0 || a.n_frames == 0) return CT_OK;
    const uint32_t packets = a.q_count / 4, gx = (packets + kBlock * kLinPackets - 1) / (kBlock * kLinPackets);
    uint32_t gy = (a.n_frames + 1) / 2;
    if (gy < 1) gy = 1;
    if (gy > 65535) gy = 65535;
    const size_t lds = INTERP == CT_INTERP_NONE ? 0 : (size_t)a.channels * a.n_points * lut_entry_bytes(INTERP);
    if (lds > 160 * 1024) return CT_ERR_TOO_LARGE;
    hipLaunchKernelGGL((linearize_planar_kernel<T, kLinPackets, INTERP, STD, WRITE_STD>), dim3(gx, gy), dim3(kBlock), lds, s, a);
    return hipGetLastError() == hipSuccess ? CT_OK : CT_ERR_LAUNCH;
}

template <typename T, int INTERP>
static int lin_dispatch_planar_std(const LinArgs &a, int std_mode, bool write_std, hipStream_t s)
{
    if (!write_std) return lin_launch_planar<T, INTERP, CT_STD_NONE, false>(a, s);
    switch (std_mode) {
        case CT_STD_NONE: return lin_launch_planar<T, INTERP, CT_STD_NONE, true>(a, s);
        case CT_STD_CONSTANT: return lin_launch_planar<T, INTERP, CT_STD_CONSTANT, true>(a, s);
        case CT_STD_MULTIPLIER: return lin_launch_planar<T, INTERP, CT_STD_MULTIPLIER, true>(a, s);
        case CT_STD_EXPLICIT: return lin_launch_planar<T, INTERP, CT_STD_EXPLICIT, true>(a, s);
    }
    return CT_ERR_INVALID_ARGUMENT;
}

template <typename T>
static int lin_dispatch_planar(const LinArgs &a, int interp, int std_mode, bool write_std, hipStream_t s)
{
    switch (interp) {
        case CT_INTERP_LOOKUP: return lin_dispatch_planar_std<T, CT_INTERP_LOOKUP>(a, std_mode, write_std, s);
        case CT_INTERP_LINEAR: return lin_dispatch_planar_std<T, CT_INTERP_LINEAR>(a, std_mode, write_std, s);
        case CT_INTERP_CATMULL: return lin_dispatch_planar_std<T, CT_INTERP_CATMULL>(a, std_mode, write_std, s);
        case CT_INTERP_NONE: return lin_dispatch_planar_std<T, CT_INTERP_NONE>(a, std_mode, write_std, s);
    }
    return CT_ERR_INVALID_ARGUMENT;
}

// Interleaved RGB / BGR frames (C == 3: what OpenCV decodes to, clair_torch/common/data_io.py:125-154): a thread owns four
// consecutive PIXELS = 12 memory elements (three typed loads of four codes each, or three 16-byte loads of float pixels),
// so every channel plane of the planar outputs receives its four consecutive pixels as ONE 16-byte streaming store.  The
// element-per-thread mapping of linearize_kernel scatters 4-byte stores over three planes and pays three runtime divisions
// per element: 2.09 ms against 0.85 ms for the planar layout on C4 (profiles/r03_layout_ingest.md).  Per-element arithmetic
// is the same function call, so results are bit-identical to the planar path.  q_begin / q_count are multiples of 12.
template <typename T, int INTERP, int STD, bool WRITE_STD>
__global__ __launch_bounds__(kBlock) void linearize_rgb_kernel(const LinArgs a)
{
    extern __shared__ __align__(16) char lds[];
    constexpr bool kRanged = sizeof(T) != 4;
    constexpr int kEntry = lut_entry_bytes(INTERP);
    const int L = a.n_points;
    stage_lut<INTERP>(lds, a.lut, 3, L);
    __syncthreads();
    const uint32_t pv = blockIdx.x * (uint32_t)kBlock + threadIdx.x;
    // Loads: a thread's own 12 elements are 24 (48) bytes apart from lane to lane, so each of its three load instructions
    // would touch every cache line of the wavefront's span (measured 4-9 % behind the planar kernel).  Instead the wavefront
    // reads its 768 elements as three DENSE instructions (lane l takes packets l, 64 + l, 128 + l), parks them in 3 KB of
    // wave-private LDS and every lane reads its own 12 back (48-byte lane stride: conflict-free for ds_read_b128).  The DS
    // operations of one wavefront execute in order, so no barrier is needed.
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t wave_first = (pv - lane) * 12u;                  // first element of this wavefront, relative to q_begin
    if (wave_first >= a.q_count) return;                             // (wave-uniform)
    const bool active = pv * 12u < a.q_count;
    float *stage = reinterpret_cast<float *>(lds + ((INTERP == CT_INTERP_NONE ? 0 : 3 * L * kEntry) + 15 & ~15)) + wave * 768u;
    const uint32_t m0 = a.q_begin + pv * 12u;  // first memory element of this thread (within one frame)
    const uint32_t pix0 = m0 / 3u;             // local pixel index (m0 is a multiple of 12)
    const float top = (float)(L - 1);
    const bool bgr = a.tile.layout == CT_LAYOUT_NHWC_BGR;
    // LUT row of element (channel plane c, pixel pix0 + j): (c * plane_global + base + pix0 + j) % 3 (base.py:173-176)
    const uint32_t pg = (a.tile.plane_local + a.tile.chan_skip) % 3u, mrow = (a.tile.base + pix0) % 3u;
    int row_off[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) {
        const uint32_t j = k / 3, cm = k % 3, c = bgr ? 2u - cm : cm;
        row_off[k] = (int)(INTERP == CT_INTERP_LOOKUP ? c : (c * pg + mrow + j) % 3u) * L * kEntry;
    }
    for (uint32_t f = blockIdx.y; f < a.n_frames; f += gridDim.y) {
        float xin[12];
        {
            LPacket<float, 4> got[3];
#pragma unroll
            for (int h = 0; h < 3; ++h) {
                const uint32_t rel = wave_first + 4u * ((uint32_t)h * 64u + lane);  // this lane's packet of dense load h
                if (rel < a.q_count) {
                    if constexpr (sizeof(T) != 4) {
                        const uint64_t base = reinterpret_cast<uint64_t>(a.frames) + (uint64_t)((int64_t)f * a.image_stride * (int64_t)sizeof(T));
                        load_codes_as_float<T, 4>(base, (a.q_begin + rel) * (uint32_t)sizeof(T), got[h].v);
                    } else {
                        const T *src = static_cast<const T *>(a.frames) + (int64_t)f * a.image_stride + a.q_begin + rel;
                        const LPacket<T, 4> pk = *reinterpret_cast<const LPacket<T, 4> *>(src);
#pragma unroll
                        for (int k = 0; k < 4; ++k) got[h].v[k] = pk.v[k];
                    }
                }
            }
#pragma unroll
            for (int h = 0; h < 3; ++h) *reinterpret_cast<LPacket<float, 4> *>(stage + 4u * ((uint32_t)h * 64u + lane)) = got[h];
#pragma unroll
            for (int h = 0; h < 3; ++h) {
                const LPacket<float, 4> pk = *reinterpret_cast<const LPacket<float, 4> *>(stage + 12u * lane + 4u * h);
#pragma unroll
                for (int k = 0; k < 4; ++k) xin[4 * h + k] = pk.v[k];
            }
        }
        if (!active) continue;  // lanes past the end helped with the loads only
        float lin[12], sd[12];
        [[maybe_unused]] bool tiny = false;
#pragma unroll
        for (int k = 0; k < 12; ++k) {
            const float x = sizeof(T) != 4 ? code_to_pixel(xin[k], a.norm) : xin[k];
            float dfdx;
            lin[k] = icrf_sample<INTERP, true, kRanged>(x, lds + row_off[k], top, dfdx);
            sd[k] = 0.0f;
            if constexpr (WRITE_STD && STD != CT_STD_NONE) {
                float sigma = a.std_value;                                          // CONSTANT
                if constexpr (STD == CT_STD_MULTIPLIER) sigma = x * a.std_value;    // datasets/base.py:133
                const float ags = fabsf(dfdx * sigma);  // sqrt((grad * std)^2) = |grad * std| unless the square underflows
                sd[k] = ags;
                tiny |= ags < 1e-18f && ags != 0.0f;
            }
        }
        if constexpr (WRITE_STD && STD != CT_STD_NONE) {
            if (__builtin_expect(__any(tiny), 0)) {
#pragma unroll
                for (int k = 0; k < 12; ++k)
                    if (sd[k] < 1e-18f && sd[k] != 0.0f) sd[k] = sqrtf(sd[k] * sd[k]);
            }
        }
        const int64_t obase = (int64_t)f * a.out_stride + pix0;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int cm = bgr ? 2 - c : c;
            LPacket<float, 4> lo, so;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                lo.v[j] = lin[3 * j + cm];
                so.v[j] = sd[3 * j + cm];
            }
            store_stream(reinterpret_cast<LPacket<float, 4> *>(a.lin_out + obase + (int64_t)c * a.tile.plane_local), lo);
            if constexpr (WRITE_STD)
                store_stream(reinterpret_cast<LPacket<float, 4> *>(a.std_out + obase + (int64_t)c * a.tile.plane_local), so);
        }
    }
}

template <typename T, int INTERP, int STD, bool WRITE_STD>
static int lin_launch_rgb(const LinArgs &a, hipStream_t s)
{
    if (a.q_count == 0 || a.n_frames == 0) return CT_OK;
    const uint32_t vecs = a.q_count / 12, gx = (vecs + kBlock - 1) / kBlock;
    uint32_t gy = (a.n_frames + 1) / 2;
    if (gy < 1) gy = 1;
    if (gy > 65535) gy = 65535;
    const size_t lds = ((INTERP == CT_INTERP_NONE ? 0 : (size_t)3 * a.n_points * lut_entry_bytes(INTERP)) + 15 & ~(size_t)15) +
                       (size_t)(kBlock / 64) * 768 * sizeof(float);  // LUT | 3 KB of exchange space per wavefront
    if (lds > 160 * 1024) return CT_ERR_TOO_LARGE;
    hipLaunchKernelGGL((linearize_rgb_kernel<T, INTERP, STD, WRITE_STD>), dim3(gx, gy), dim3(kBlock), lds, s, a);
    return hipGetLastError() == hipSuccess ? CT_OK : CT_ERR_LAUNCH;
}

template <typename T, int INTERP>
static int lin_dispatch_rgb_std(const LinArgs &a, int std_mode, bool write_std, hipStream_t s)
{
    if (!write_std) return lin_launch_rgb<T, INTERP, CT_STD_NONE, false>(a, s);
    switch (std_mode) {
        case CT_STD_NONE: return lin_launch_rgb<T, INTERP, CT_STD_NONE, true>(a, s);
        case CT_STD_CONSTANT: return lin_launch_rgb<T, INTERP, CT_STD_CONSTANT, true>(a, s);
        case CT_STD_MULTIPLIER: return lin_launch_rgb<T, INTERP, CT_STD_MULTIPLIER, true>(a, s);
    }
    return CT_ERR_UNSUPPORTED;  // explicit std stacks with interleaved frames go through the element-wise kernel
}

template <typename T>
static int lin_dispatch_rgb(const LinArgs &a, int interp, int std_mode, bool write_std, hipStream_t s)
{
    switch (interp) {
        case CT_INTERP_LOOKUP: return lin_dispatch_rgb_std<T, CT_INTERP_LOOKUP>(a, std_mode, write_std, s);
        case CT_INTERP_LINEAR: return lin_dispatch_rgb_std<T, CT_INTERP_LINEAR>(a, std_mode, write_std, s);
        case CT_INTERP_CATMULL: return lin_dispatch_rgb_std<T, CT_INTERP_CATMULL>(a, std_mode, write_std, s);
        case CT_INTERP_NONE: return lin_dispatch_rgb_std<T, CT_INTERP_NONE>(a, std_mode, write_std, s);
    }
    return CT_ERR_INVALID_ARGUMENT;
}

template <typename T, int V, int INTERP, int STD, bool WRITE_STD>
static int lin_launch(const LinArgs &a, hipStream_t s)
{
    if (a.q_count == 0 || a.n_frames == 0) return CT_OK;
    const uint32_t vecs = a.q_count / V, gx = (vecs + kBlock - 1) / kBlock;
    // Two frames per workgroup (grid.y = F/2): measured best on C4 (64 frames: 57 % of the HBM peak vs 52 % with one
    // frame and 49 % with sixteen frames per workgroup); the LUT staging is amortised over two packets and the grid
    // stays far larger than the machine.  A write-heavy stream with the same ingredients and no frame loop reaches
    // 5.2-5.5 TB/s on the same device (tools/stream_write_heavy.hip), so this kernel is at ~85 % of what is achievable.
    uint32_t gy = (a.n_frames + 1) / 2;
    if (gy < 1) gy = 1;
    if (gy > 65535) gy = 65535;
    const size_t lds = INTERP == CT_INTERP_NONE ? 0 : (size_t)a.channels * a.n_points * lut_entry_bytes(INTERP);
    if (lds > 160 * 1024) return CT_ERR_TOO_LARGE;
    hipLaunchKernelGGL((linearize_kernel<T, V, INTERP, STD, WRITE_STD>), dim3(gx, gy), dim3(kBlock), lds, s, a);
    return hipGetLastError() == hipSuccess ? CT_OK : CT_ERR_LAUNCH;
}

template <typename T, int V, int INTERP>
static int lin_dispatch_std(const LinArgs &a, int std_mode, bool write_std, hipStream_t s)
{
    if (!write_std) return lin_launch<T, V, INTERP, CT_STD_NONE, false>(a, s);
    switch (std_mode) {
        case CT_STD_NONE: return lin_launch<T, V, INTERP, CT_STD_NONE, true>(a, s);
        case CT_STD_CONSTANT: return lin_launch<T, V, INTERP, CT_STD_CONSTANT, true>(a, s);
        case CT_STD_MULTIPLIER: return lin_launch<T, V, INTERP, CT_STD_MULTIPLIER, true>(a, s);
        case CT_STD_EXPLICIT: return lin_launch<T, V, INTERP, CT_STD_EXPLICIT, true>(a, s);
    }
    return CT_ERR_INVALID_ARGUMENT;
}

template <typename T, int V>
static int lin_dispatch(const LinArgs &a, int interp, int std_mode, bool write_std, hipStream_t s)
{
    switch (interp) {
        case CT_INTERP_LOOKUP: return lin_dispatch_std<T, V, CT_INTERP_LOOKUP>(a, std_mode, write_std, s);
        case CT_INTERP_LINEAR: return lin_dispatch_std<T, V, CT_INTERP_LINEAR>(a, std_mode, write_std, s);
        case CT_INTERP_CATMULL: return lin_dispatch_std<T, V, CT_INTERP_CATMULL>(a, std_mode, write_std, s);
        case CT_INTERP_NONE: return lin_dispatch_std<T, V, CT_INTERP_NONE>(a, std_mode, write_std, s);
    }
    return CT_ERR_INVALID_ARGUMENT;
}

template <typename T>
static int lin_typed(LinArgs a, uint32_t Q, int interp, int std_mode, bool write_std, hipStream_t s)
{
    // Elements per thread: 8 integer codes (two typed loads) or 8 float pixels (two 16-byte loads).  16 codes per thread was
    // measured 2.6x SLOWER on C4 (2.21 against 0.85 ms, profiles/r03_layout_ingest.md) although the pixel-owning RGB kernel
    // with 12 codes per thread is the fastest of all (0.68 ms).
#ifndef CT_LINEARIZE_V_INT
#define CT_LINEARIZE_V_INT 8
#endif
    constexpr int V = sizeof(T) == 4 ? 8 : CT_LINEARIZE_V_INT;
    auto aligned = [](const void *p, size_t b) { return p == nullptr || (reinterpret_cast<uintptr_t>(p) % b) == 0; };
    const bool vec_ok = aligned(a.frames, sizeof(T) * V) && (a.image_stride % V) == 0 && aligned(a.std_stack, 4 * V) &&
                        aligned(a.lin_out, 4 * V) && aligned(a.std_out, 4 * V);
    uint32_t q_vec = vec_ok ? (Q / V) * V : 0;
    int rc = CT_OK;
    // interleaved RGB / BGR with whole packets of four pixels per plane: the pixel-owning kernel (packet stores)
    if (a.tile.layout != CT_LAYOUT_NCHW && a.channels == 3 && std_mode != CT_STD_EXPLICIT && a.tile.plane_local % 4 == 0 &&
        aligned(a.frames, 16) && (a.image_stride % 4) == 0 && aligned(a.lin_out, 16) && aligned(a.std_out, 16) && a.out_stride % 4 == 0) {
        a.q_begin = 0;
        a.q_count = Q;  // 3 * plane_local: a multiple of 12
        return lin_dispatch_rgb<T>(a, interp, std_mode, write_std, s);
    }
    // planar frames with whole packets of four everywhere: the dense-access kernel (K packets per thread)
    if (a.tile.layout == CT_LAYOUT_NCHW && Q >= 4 && aligned(a.frames, sizeof(T) * 4) && (a.image_stride % 4) == 0 &&
        aligned(a.std_stack, 16) && aligned(a.lin_out, 16) && aligned(a.std_out, 16) && a.out_stride % 4 == 0) {
        a.q_begin = 0;
        a.q_count = (Q / 4) * 4;
        rc = lin_dispatch_planar<T>(a, interp, std_mode, write_std, s);
        if (rc != CT_OK || a.q_count == Q) return rc;
        a.q_begin = a.q_count;
        a.q_count = Q - a.q_begin;
        return lin_dispatch<T, 1>(a, interp, std_mode, write_std, s);
    }
    if (q_vec) {
        a.q_begin = 0;
        a.q_count = q_vec;
        rc = lin_dispatch<T, V>(a, interp, std_mode, write_std, s);
        if (rc != CT_OK) return rc;
    }
    if (q_vec < Q) {
        a.q_begin = q_vec;
        a.q_count = Q - q_vec;
        rc = lin_dispatch<T, 1>(a, interp, std_mode, write_std, s);
    }
    return rc;
}

// ---- backward ------------------------------------------------------------------------------------
struct BwdArgs {
    const float *x;
    const float *grad_out;
    const float *lut;
    float *grad_x;
    float *lut_grad;
    int64_t image_stride;
    uint32_t q_count;  // elements per image
    uint32_t n_images;
    TileMap tile;
    int32_t channels, n_points;
};

template <int INTERP>
__global__ __launch_bounds__(kBlock) void linearize_bwd_kernel(const BwdArgs a)
{
    extern __shared__ __align__(16) char lds[];
    constexpr int kEntry = lut_entry_bytes(INTERP);
    const int C = a.channels, L = a.n_points;
    const int lut_bytes = C * L * kEntry;
    // float64 histogram: ds_add_f64 is ~10x cheaper than ds_add_f32 on gfx950 (tools/lds_atomic_rates.hip)
    double *hist = reinterpret_cast<double *>(lds + ((lut_bytes + 15) & ~15));
    const bool want_lut = a.lut_grad != nullptr;
    stage_lut<INTERP>(lds, a.lut, C, L);
    if (want_lut)
        for (int k = threadIdx.x; k < C * L; k += blockDim.x) hist[k] = 0.0;
    __syncthreads();
    const float top = (float)(L - 1);
    const uint32_t stride = gridDim.x * (uint32_t)kBlock;
    for (uint32_t q = blockIdx.x * (uint32_t)kBlock + threadIdx.x; q < a.q_count; q += stride) {
        int ch;
        uint32_t qg;
        a.tile.locate(q, ch, qg);
        const int row = lut_row<INTERP>(qg, ch, C);
        const char *row_lds = lds + row * L * kEntry;
        double *hrow = hist + row * L;
        for (uint32_t n = blockIdx.y; n < a.n_images; n += gridDim.y) {
            const int64_t off = (int64_t)n * a.image_stride + q;
            const float x = a.x[off], go = a.grad_out[off];
            if (a.grad_x) a.grad_x[off] = icrf_grad_reference_order<INTERP>(x, row_lds, top, go);
            if (want_lut) {
                if constexpr (INTERP == CT_INTERP_LOOKUP) {
                    float r = rintf(x * top);
                    r = fminf(fmaxf(r, 0.0f), top);
                    atomicAdd(&hrow[(int)r], (double)go);
                } else {
                    const float s = fminf(fmaxf(x * top, 0.0f), top);
                    const float fl = floorf(s);
                    const int i0 = (int)fl;
                    const float t = s - fl;
                    if constexpr (INTERP == CT_INTERP_LINEAR) {
                        const int i1 = i0 + 1 < L ? i0 + 1 : L - 1;
                        atomicAdd(&hrow[i0], (double)(go * (1.0f - t)));
                        atomicAdd(&hrow[i1], (double)(go * t));
                    } else {
                        const float t2 = t * t, t3 = t2 * t;
                        const float w0 = -0.5f * t3 + t2 - 0.5f * t, w1 = 1.5f * t3 - 2.5f * t2 + 1.0f;
                        const float w2 = -1.5f * t3 + 2.0f * t2 + 0.5f * t, w3 = 0.5f * t3 - 0.5f * t2;
                        const int im = i0 > 0 ? i0 - 1 : 0, i1 = i0 + 1 < L ? i0 + 1 : L - 1,
                                  i2 = i0 + 2 < L ? i0 + 2 : L - 1;
                        atomicAdd(&hrow[im], (double)(go * w0));
                        atomicAdd(&hrow[i0], (double)(go * w1));
                        atomicAdd(&hrow[i1], (double)(go * w2));
                        atomicAdd(&hrow[i2], (double)(go * w3));
                    }
                }
            }
        }
    }
    if (want_lut) {
        __syncthreads();
        for (int k = threadIdx.x; k < C * L; k += blockDim.x) {
            const float v = (float)hist[k];
            if (v != 0.0f) atomicAdd(&a.lut_grad[k], v);
        }
    }
}

template <int INTERP>
static int bwd_launch(const BwdArgs &a, hipStream_t s)
{
    uint32_t gy = a.n_images;
    if (gy > 8) gy = 8;
    if (gy < 1) gy = 1;
    const size_t lut_bytes = (size_t)a.channels * a.n_points * lut_entry_bytes(INTERP);
    const size_t lds = ((lut_bytes + 15) & ~(size_t)15) + sizeof(double) * (size_t)a.channels * a.n_points;
    if (lds > 160 * 1024) return CT_ERR_TOO_LARGE;
    // the whole grid is what the device holds at once: one round of equal grid-stride work, and as few histogram
    // flushes (global atomics on C*L addresses) as possible
    uint32_t gx = (a.q_count + kBlock - 1) / kBlock;
    const uint32_t cap = std::max<uint32_t>(1u, (uint32_t)resident_workgroups(lds, kBlock) / gy);
    if (gx > cap) gx = cap;
    hipLaunchKernelGGL((linearize_bwd_kernel<INTERP>), dim3(gx, gy), dim3(kBlock), lds, s, a);
    return hipGetLastError() == hipSuccess ? CT_OK : CT_ERR_LAUNCH;
}

static int check_geom(const ct_geometry *g)
{
    if (!g || g->channels <= 0 || g->h_tile <= 0 || g->width <= 0 || g->h_global < g->h_tile || g->row_offset < 0 ||
        g->row_offset + g->h_tile > g->h_global)
        return CT_ERR_INVALID_ARGUMENT;
    if (g->h_global * g->width * g->channels >= (int64_t)1 << 31) return CT_ERR_TOO_LARGE;
    if (g->image_stride < g->h_tile * g->width * g->channels) return CT_ERR_INVALID_ARGUMENT;
    if (g->layout < CT_LAYOUT_NCHW || g->layout > CT_LAYOUT_NHWC_BGR) return CT_ERR_INVALID_ARGUMENT;
    return CT_OK;
}

static TileMap make_tile(const ct_geometry *g)
{
    TileMap t;
    t.plane_local = (uint32_t)(g->h_tile * g->width);
    t.chan_skip = (uint32_t)((g->h_global - g->h_tile) * g->width);
    t.base = (uint32_t)(g->row_offset * g->width);
    t.layout = (uint32_t)g->layout;
    t.channels = (uint32_t)g->channels;
    return t;
}

}  // namespace ct

extern "C" int ct_norm_constants(float max_code, float *hi, float *lo);

extern "C" int ct_linearize_std(const void *frames_dev, int32_t dtype, float max_code, int64_t n_frames,
                                const ct_geometry *geom, const float *std_dev, int32_t std_mode, float std_value,
                                const ct_icrf *icrf, float *lin_out_dev, float *std_out_dev, void *stream)
{
    using namespace ct;
    if (!frames_dev || !icrf || !lin_out_dev || n_frames <= 0 || n_frames > 0x7fffffff) return CT_ERR_INVALID_ARGUMENT;
    int rc = check_geom(geom);
    if (rc != CT_OK) return rc;
    const int interp = icrf->interp;
    if (interp < CT_INTERP_LOOKUP || interp > CT_INTERP_NONE) return CT_ERR_INVALID_ARGUMENT;
    if (interp != CT_INTERP_NONE && (!icrf->lut_dev || icrf->n_points < 2)) return CT_ERR_INVALID_ARGUMENT;
    if (std_mode < CT_STD_NONE || std_mode > CT_STD_EXPLICIT) return CT_ERR_INVALID_ARGUMENT;
    if (std_mode == CT_STD_EXPLICIT && !std_dev) return CT_ERR_INVALID_ARGUMENT;
    // linearization.py:100-105: autograd.grad raises for LOOKUP (no gradient path) when stds are present
    if (std_mode != CT_STD_NONE && interp == CT_INTERP_LOOKUP) return CT_ERR_NO_GRADIENT_PATH;
    LinArgs a{};
    a.frames = frames_dev;
    a.std_stack = std_mode == CT_STD_EXPLICIT ? std_dev : nullptr;
    a.lut = icrf->lut_dev;
    a.lin_out = lin_out_dev;
    a.std_out = std_out_dev;
    a.image_stride = geom->image_stride;
    a.out_stride = geom->h_tile * geom->width * geom->channels;
    a.n_frames = (uint32_t)n_frames;
    a.tile = make_tile(geom);
    a.channels = geom->channels;
    a.n_points = interp == CT_INTERP_NONE ? 2 : icrf->n_points;
    a.std_value = std_value;
    const uint32_t Q = (uint32_t)(geom->h_tile * geom->width * geom->channels);
    const bool write_std = std_out_dev != nullptr;
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (dtype) {
        case CT_DTYPE_U8:
            if (ct_norm_constants(max_code, &a.norm.hi, &a.norm.lo) != CT_OK) return CT_ERR_UNSUPPORTED;
            return lin_typed<uint8_t>(a, Q, interp, std_mode, write_std, s);
        case CT_DTYPE_U16:
            if (ct_norm_constants(max_code, &a.norm.hi, &a.norm.lo) != CT_OK) return CT_ERR_UNSUPPORTED;
            return lin_typed<uint16_t>(a, Q, interp, std_mode, write_std, s);
        case CT_DTYPE_F32: return lin_typed<float>(a, Q, interp, std_mode, write_std, s);
    }
    return CT_ERR_UNSUPPORTED;
}

extern "C" int ct_linearize_fwd(const float *x_dev, int64_t n_images, const ct_geometry *geom, const ct_icrf *icrf,
                                float *out_dev, void *stream)
{
    return ct_linearize_std(x_dev, CT_DTYPE_F32, 1.0f, n_images, geom, nullptr, CT_STD_NONE, 0.0f, icrf, out_dev,
                            nullptr, stream);
}

extern "C" int ct_linearize_bwd(const float *x_dev, const float *grad_out_dev, int64_t n_images,
                                const ct_geometry *geom, const ct_icrf *icrf, float *grad_x_dev, float *lut_grad_dev,
                                void *stream)
{
    using namespace ct;
    if (!x_dev || !grad_out_dev || !icrf || n_images <= 0 || n_images > 0x7fffffff) return CT_ERR_INVALID_ARGUMENT;
    if (!grad_x_dev && !lut_grad_dev) return CT_OK;
    int rc = check_geom(geom);
    if (rc != CT_OK) return rc;
    if (geom->layout != CT_LAYOUT_NCHW) return CT_ERR_UNSUPPORTED;
    const int interp = icrf->interp;
    if (interp < CT_INTERP_LOOKUP || interp > CT_INTERP_CATMULL || !icrf->lut_dev || icrf->n_points < 2)
        return CT_ERR_INVALID_ARGUMENT;
    BwdArgs a{};
    a.x = x_dev;
    a.grad_out = grad_out_dev;
    a.lut = icrf->lut_dev;
    a.grad_x = grad_x_dev;
    a.lut_grad = lut_grad_dev;
    a.image_stride = geom->image_stride;
    a.q_count = (uint32_t)(geom->h_tile * geom->width * geom->channels);
    a.n_images = (uint32_t)n_images;
    a.tile = make_tile(geom);
    a.channels = geom->channels;
    a.n_points = icrf->n_points;
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (interp) {
        case CT_INTERP_LOOKUP: return bwd_launch<CT_INTERP_LOOKUP>(a, s);
        case CT_INTERP_LINEAR: return bwd_launch<CT_INTERP_LINEAR>(a, s);
        case CT_INTERP_CATMULL: return bwd_launch<CT_INTERP_CATMULL>(a, s);
    }
    return CT_ERR_INVALID_ARGUMENT;
}
