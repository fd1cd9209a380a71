// ct_stats.hip -- streaming per-pixel mean and variance of video frames, gfx950.
//
// Replaces the loop body of compute_video_mean_and_std (clair_torch/inference/inferential_statistics.py:38-47):
// optional ICRF linearization of a batch of frames, then WBOMeanVar.update_values(frames, None)
// (clair_torch/common/statistics.py:213-259): batch mean, batch m2 = sum (x - mean_b)^2 and the pairwise merge
//   M = M_A + M_B + (W_A W_B / W)(mean_B - mean_A)^2,   mean = mean_A + (W_B / W)(mean_B - mean_A),   W = W_A + W_B
// with unit weights (W_B = batch size), all in float32 like the reference's float32 frames.
//
// One thread owns V consecutive elements.  The centred second moment needs the batch mean first, so the batch is
// walked twice -- out of registers when the batch has at most 32 frames (video_stats_cached_kernel: every frame is
// loaded and linearized exactly once, all loads in flight together), out of memory otherwise (video_stats_kernel).
// Both form the sums in the reference's order, so they agree bit for bit.  HBM-bound.
//
// Interleaved (F, H, W, C) frames as OpenCV decodes them (IL instantiations): a thread still owns V consecutive MEMORY
// elements -- the frame loads stay packets -- and only the LUT row and the position in the planar (C, H, W) state follow
// from TileMap::planar_index; the state is touched element by element, once per batch of frames.
#include "ct_device.hpp"

namespace ct {

struct StatsArgs {
    const void *frames;
    const float *lut;
    float *mean_state, *m2_state;
    int64_t image_stride;
    uint32_t q_begin, q_count;
    TileMap tile;
    int32_t batch, channels, n_points;
    NormConst norm;
    float count_before;  // W_A (number of frames merged so far)
};

template <typename T, int V>
struct alignas(sizeof(T) * V) SPacket {
    T v[V];
};

// Merge of the batch statistics into the running state (statistics.py:245-251), shared by both kernels.
template <int V, bool IL>
__device__ __forceinline__ void merge_state(const StatsArgs &a, uint32_t q0, const uint32_t (&pq)[V], const float (&mean_b)[V],
                                            const float (&m2)[V])
{
    const float WA = a.count_before, WB = (float)a.batch, W = WA + WB;
    if constexpr (IL) {
#pragma unroll
        for (int e = 0; e < V; ++e) {
            float mo = mean_b[e], vo = m2[e];
            if (WA != 0.0f) {
                const float ma = a.mean_state[pq[e]], va = a.m2_state[pq[e]];
                const float delta = mean_b[e] - ma;
                vo = va + m2[e] + (WA * WB / W) * (delta * delta);  // statistics.py:250
                mo = ma + (WB / W) * delta;                          // statistics.py:251
            }
            a.mean_state[pq[e]] = mo;
            a.m2_state[pq[e]] = vo;
        }
        return;
    }
    SPacket<float, V> mo, vo;
    if (WA == 0.0f) {
#pragma unroll
        for (int e = 0; e < V; ++e) {
            mo.v[e] = mean_b[e];
            vo.v[e] = m2[e];
        }
    } else {
        const SPacket<float, V> ma = *reinterpret_cast<const SPacket<float, V> *>(a.mean_state + q0);
        const SPacket<float, V> va = *reinterpret_cast<const SPacket<float, V> *>(a.m2_state + q0);
#pragma unroll
        for (int e = 0; e < V; ++e) {
            const float delta = mean_b[e] - ma.v[e];
            vo.v[e] = va.v[e] + m2[e] + (WA * WB / W) * (delta * delta);  // statistics.py:250
            mo.v[e] = ma.v[e] + (WB / W) * delta;                          // statistics.py:251
        }
    }
    *reinterpret_cast<SPacket<float, V> *>(a.mean_state + q0) = mo;
    *reinterpret_cast<SPacket<float, V> *>(a.m2_state + q0) = vo;
}

template <typename T, int V, int INTERP, bool IL>
__global__ __launch_bounds__(kBlock) void video_stats_kernel(const StatsArgs a)
{
    extern __shared__ __align__(16) char lds[];
    constexpr bool kRanged = sizeof(T) != 4;
    constexpr int kEntry = lut_entry_bytes(INTERP);
    const int C = a.channels, L = a.n_points, B = a.batch;
    stage_lut<INTERP>(lds, a.lut, C, L);
    __syncthreads();
    const uint32_t vec = blockIdx.x * (uint32_t)kBlock + threadIdx.x;
    if (vec * (uint32_t)V >= a.q_count) return;
    const uint32_t q0 = a.q_begin + vec * (uint32_t)V;
    const float top = INTERP == CT_INTERP_NONE ? 1.0f : (float)(L - 1);
    int row_off[V];
    uint32_t pq[V];  // position in the planar state (== q0 + e unless the frames are interleaved)
#pragma unroll
    for (int e = 0; e < V; ++e) {
        int ch;
        uint32_t qg;
        pq[e] = IL ? a.tile.planar_index(q0 + e) : q0 + e;
        a.tile.locate(pq[e], ch, qg);
        row_off[e] = lut_row<INTERP>(qg, ch, C) * L * kEntry;
    }
    const T *src = static_cast<const T *>(a.frames) + q0;
    float sum[V], m2[V];
#pragma unroll
    for (int e = 0; e < V; ++e) sum[e] = m2[e] = 0.0f;
    for (int n = 0; n < B; ++n) {
        const SPacket<T, V> pk = *reinterpret_cast<const SPacket<T, V> *>(src + (int64_t)n * a.image_stride);
#pragma unroll
        for (int e = 0; e < V; ++e) {
            float d;
            sum[e] += icrf_sample<INTERP, true, kRanged>(to_pixel<T>(pk.v[e], a.norm), lds + row_off[e], top, d);
        }
    }
    float mean_b[V];
#pragma unroll
    for (int e = 0; e < V; ++e) mean_b[e] = sum[e] / (float)B;  // torch.mean: float32 sum / count
    for (int n = 0; n < B; ++n) {
        const SPacket<T, V> pk = *reinterpret_cast<const SPacket<T, V> *>(src + (int64_t)n * a.image_stride);
#pragma unroll
        for (int e = 0; e < V; ++e) {
            float d;
            const float x = icrf_sample<INTERP, true, kRanged>(to_pixel<T>(pk.v[e], a.norm), lds + row_off[e], top, d);
            const float dv = x - mean_b[e];
            m2[e] += dv * dv;
        }
    }
    merge_state<V, IL>(a, q0, pq, mean_b, m2);
}

// Batch of at most BMAX frames: the linearized values stay in registers between the mean and the m2 pass.
template <typename T, int V, int INTERP, int BMAX, bool IL>
__global__ __launch_bounds__(kBlock) void video_stats_cached_kernel(const StatsArgs a)
{
    extern __shared__ __align__(16) char lds[];
    constexpr bool kRanged = sizeof(T) != 4;
    constexpr int kEntry = lut_entry_bytes(INTERP);
    const int C = a.channels, L = a.n_points, B = a.batch;
    stage_lut<INTERP>(lds, a.lut, C, L);
    __syncthreads();
    const uint32_t vec = blockIdx.x * (uint32_t)kBlock + threadIdx.x;
    if (vec * (uint32_t)V >= a.q_count) return;
    const uint32_t q0 = a.q_begin + vec * (uint32_t)V;
    const float top = INTERP == CT_INTERP_NONE ? 1.0f : (float)(L - 1);
    int row_off[V];
    uint32_t pq[V];  // position in the planar state (== q0 + e unless the frames are interleaved)
#pragma unroll
    for (int e = 0; e < V; ++e) {
        int ch;
        uint32_t qg;
        pq[e] = IL ? a.tile.planar_index(q0 + e) : q0 + e;
        a.tile.locate(pq[e], ch, qg);
        row_off[e] = lut_row<INTERP>(qg, ch, C) * L * kEntry;
    }
    const T *src = static_cast<const T *>(a.frames) + q0;
    SPacket<T, V> raw[BMAX];
#pragma unroll
    for (int n = 0; n < BMAX; ++n)
        if (n < B) raw[n] = *reinterpret_cast<const SPacket<T, V> *>(src + (int64_t)n * a.image_stride);
    float xs[BMAX][V], sum[V], m2[V], mean_b[V];
#pragma unroll
    for (int e = 0; e < V; ++e) sum[e] = m2[e] = 0.0f;
#pragma unroll
    for (int n = 0; n < BMAX; ++n) {
        if (n < B) {
#pragma unroll
            for (int e = 0; e < V; ++e) {
                float d;
                xs[n][e] = icrf_sample<INTERP, true, kRanged>(to_pixel<T>(raw[n].v[e], a.norm), lds + row_off[e], top, d);
                sum[e] += xs[n][e];
            }
        }
    }
#pragma unroll
    for (int e = 0; e < V; ++e) mean_b[e] = sum[e] / (float)B;  // torch.mean: float32 sum / count
#pragma unroll
    for (int n = 0; n < BMAX; ++n) {
        if (n < B) {
#pragma unroll
            for (int e = 0; e < V; ++e) {
                const float dv = xs[n][e] - mean_b[e];
                m2[e] += dv * dv;
            }
        }
    }
    merge_state<V, IL>(a, q0, pq, mean_b, m2);
}

template <typename T, int V, int INTERP, bool IL>
static int stats_launch_layout(const StatsArgs &a, hipStream_t s)
{
    if (a.q_count == 0) return CT_OK;
    const uint32_t vecs = a.q_count / V, grid = (vecs + kBlock - 1) / kBlock;
    const size_t lds = INTERP == CT_INTERP_NONE ? 0 : (size_t)a.channels * a.n_points * lut_entry_bytes(INTERP);
    if (lds > 160 * 1024) return CT_ERR_TOO_LARGE;
    if (a.batch <= 16)
        hipLaunchKernelGGL((video_stats_cached_kernel<T, V, INTERP, 16, IL>), dim3(grid), dim3(kBlock), lds, s, a);
    else if (a.batch <= 32)
        hipLaunchKernelGGL((video_stats_cached_kernel<T, V, INTERP, 32, IL>), dim3(grid), dim3(kBlock), lds, s, a);
    else
        hipLaunchKernelGGL((video_stats_kernel<T, V, INTERP, IL>), dim3(grid), dim3(kBlock), lds, s, a);
    return hipGetLastError() == hipSuccess ? CT_OK : CT_ERR_LAUNCH;
}

template <typename T, int V, int INTERP>
static int stats_launch(const StatsArgs &a, hipStream_t s)
{
    return a.tile.layout == CT_LAYOUT_NCHW ? stats_launch_layout<T, V, INTERP, false>(a, s) : stats_launch_layout<T, V, INTERP, true>(a, s);
}

template <typename T, int V>
static int stats_dispatch(const StatsArgs &a, int interp, hipStream_t s)
{
    switch (interp) {
        case CT_INTERP_LOOKUP: return stats_launch<T, V, CT_INTERP_LOOKUP>(a, s);
        case CT_INTERP_LINEAR: return stats_launch<T, V, CT_INTERP_LINEAR>(a, s);
        case CT_INTERP_CATMULL: return stats_launch<T, V, CT_INTERP_CATMULL>(a, s);
        case CT_INTERP_NONE: return stats_launch<T, V, CT_INTERP_NONE>(a, s);
    }
    return CT_ERR_INVALID_ARGUMENT;
}

template <typename T>
static int stats_typed(StatsArgs a, uint32_t Q, int interp, hipStream_t s)
{
    constexpr int V = 4;  // 4 elements per thread: 32 frames x 4 values fit the register file
    auto aligned = [](const void *p, size_t b) { return (reinterpret_cast<uintptr_t>(p) % b) == 0; };
    const bool vec_ok = aligned(a.frames, sizeof(T) * V) && (a.image_stride % V) == 0 && aligned(a.mean_state, 4 * V) &&
                        aligned(a.m2_state, 4 * V);
    const uint32_t q_vec = vec_ok ? (Q / V) * V : 0;
    int rc = CT_OK;
    if (q_vec) {
        a.q_begin = 0;
        a.q_count = q_vec;
        rc = stats_dispatch<T, V>(a, interp, s);
        if (rc != CT_OK) return rc;
    }
    if (q_vec < Q) {
        a.q_begin = q_vec;
        a.q_count = Q - q_vec;
        rc = stats_dispatch<T, 1>(a, interp, s);
    }
    return rc;
}

}  // namespace ct

extern "C" int ct_norm_constants(float max_code, float *hi, float *lo);

extern "C" int ct_video_stats_batch(const void *frames_dev, int32_t dtype, float max_code, int32_t batch,
                                    const ct_geometry *geom, const ct_icrf *icrf, float frames_before,
                                    float *mean_state_dev, float *m2_state_dev, void *stream)
{
    using namespace ct;
    if (!frames_dev || !geom || !icrf || !mean_state_dev || !m2_state_dev || batch <= 0 || frames_before < 0.0f)
        return CT_ERR_INVALID_ARGUMENT;
    if (geom->channels <= 0 || geom->h_tile <= 0 || geom->width <= 0 || geom->h_global < geom->h_tile ||
        geom->row_offset < 0 || geom->row_offset + geom->h_tile > geom->h_global)
        return CT_ERR_INVALID_ARGUMENT;
    const int interp = icrf->interp;
    if (interp < CT_INTERP_LOOKUP || interp > CT_INTERP_NONE) return CT_ERR_INVALID_ARGUMENT;
    if (interp != CT_INTERP_NONE && (!icrf->lut_dev || icrf->n_points < 2)) return CT_ERR_INVALID_ARGUMENT;
    const int64_t Qg = geom->h_global * geom->width * geom->channels, Ql = geom->h_tile * geom->width * geom->channels;
    if (Qg >= (int64_t)1 << 31) return CT_ERR_TOO_LARGE;
    if (geom->image_stride < Ql) return CT_ERR_INVALID_ARGUMENT;
    if (geom->layout < CT_LAYOUT_NCHW || geom->layout > CT_LAYOUT_NHWC_BGR) return CT_ERR_INVALID_ARGUMENT;
    StatsArgs a{};
    a.frames = frames_dev;
    a.lut = icrf->lut_dev;
    a.mean_state = mean_state_dev;
    a.m2_state = m2_state_dev;
    a.image_stride = geom->image_stride;
    a.tile.plane_local = (uint32_t)(geom->h_tile * geom->width);
    a.tile.chan_skip = (uint32_t)((geom->h_global - geom->h_tile) * geom->width);
    a.tile.base = (uint32_t)(geom->row_offset * geom->width);
    a.tile.layout = (uint32_t)geom->layout;  // frames planar or interleaved; the state is always planar (C, H, W)
    a.tile.channels = (uint32_t)geom->channels;
    a.batch = batch;
    a.channels = geom->channels;
    a.n_points = interp == CT_INTERP_NONE ? 2 : icrf->n_points;
    a.count_before = frames_before;
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (dtype) {
        case CT_DTYPE_U8:
            if (ct_norm_constants(max_code, &a.norm.hi, &a.norm.lo) != CT_OK) return CT_ERR_UNSUPPORTED;
            return stats_typed<uint8_t>(a, (uint32_t)Ql, interp, s);
        case CT_DTYPE_U16:
            if (ct_norm_constants(max_code, &a.norm.hi, &a.norm.lo) != CT_OK) return CT_ERR_UNSUPPORTED;
            return stats_typed<uint16_t>(a, (uint32_t)Ql, interp, s);
        case CT_DTYPE_F32: return stats_typed<float>(a, (uint32_t)Ql, interp, s);
    }
    return CT_ERR_UNSUPPORTED;
}
