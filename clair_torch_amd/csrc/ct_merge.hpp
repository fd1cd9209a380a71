// ct_merge.hpp -- argument block and packet type shared by the merge kernels (ct_merge.hip, ct_merge_exact.hip).
#pragma once
#include "ct_device.hpp"

namespace ct {

#define GLOBAL_AS __attribute__((address_space(1)))  // global memory (keeps laundered addresses off the flat path)

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
    // index of memory element m of one image in the state / output arrays: planar (C, H, W) unless CT_MERGE_OUT_AS_INPUT
    __device__ __forceinline__ uint32_t out_index(uint32_t m) const
    {
        return (flags & CT_MERGE_OUT_AS_INPUT) ? m : tile.planar_index(m);
    }
};

constexpr float kPivotCondLimit = 8.0f;  // sum |terms| / result above which a wavefront repeats the batch about the mean

template <typename T, int V>
struct alignas(sizeof(T) * V) Packet {
    T v[V];
};


// Several consecutive batches in one launch (ct_hdr_merge_batches): batch b has batch_size[b] exposures at batch_ptr[b]
// (explicit uncertainties at std_ptr[b]); the exposure times of all batches follow each other in MergeArgs::exposure and
// MergeArgs::batch is their total.  n_batches == 0: the one batch MergeArgs itself describes.
constexpr int kMaxMergeBatches = 16;
struct MergeBatches {
    int32_t n_batches;
    int32_t batch_size[kMaxMergeBatches];
    const void *batch_ptr[kMaxMergeBatches];
    const float *std_ptr[kMaxMergeBatches];
};

// ct_merge_exact.hip: the merge with the reference's float32 autograd order (two passes over each batch); `q_count` elements
// from `q_begin`, dispatched on dtype / interpolation / weight / std mode.  With `batches` the launch walks them with the
// streaming state in registers in between (bit-identical to one launch per batch).  Returns a CT_* status.
int merge_reference_order(const MergeArgs &a, int dtype, uint32_t q_total, int interp, int weight_mode, int std_mode,
                          hipStream_t stream, const MergeBatches *batches = nullptr);

}  // namespace ct
