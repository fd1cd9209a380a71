/*
 * clair_hip.h -- C ABI of the MI355X (gfx950) implementation of clair-torch's per-pixel hot path.
 *
 * The reference (samivout/clair-torch) is pure Python on eager PyTorch and has no FFI of its own; each entry
 * point below replaces the *interior* of one reference function (file:line cited per function, relative to
 * the reference repository).  INTEGRATION.md shows the ctypes binding a reference maintainer would add.
 *
 * Conventions (all entry points):
 *   - plain pointers and sizes, no torch / HIP types in the signatures; `stream` is a hipStream_t passed as void*
 *     (NULL = the default stream);
 *   - every pointer named *_dev is DEVICE memory owned by the caller; nothing is allocated, freed or retained;
 *   - launches are asynchronous on `stream`; no hidden device synchronisation; re-entrant across streams;
 *   - return value: CT_OK (0) or a negative CT_ERR_* code; never throws.  ct_error_string() names a code.
 *   - image stacks are NCHW, contiguous inside one exposure; consecutive exposures are `image_stride`
 *     elements apart (= C*H_tile*W for a dense stack).
 *
 * Tiling (multi-GPU row bands): a rank holds rows [row_offset, row_offset + H_tile) of every channel plane of a
 * global (C, H_global, W) image.  The reference picks the LUT row for LINEAR/CATMULL interpolation from the flat
 * NCHW index modulo C (clair_torch/models/base.py:173-176, 216-219), so kernels need the GLOBAL geometry to
 * reproduce it on a tile; pass H_global = H_tile and row_offset = 0 for an untiled image.
 */
#ifndef CLAIR_HIP_H
#define CLAIR_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CT_ABI_VERSION 3

/* status codes */
#define CT_OK 0
#define CT_ERR_INVALID_ARGUMENT (-1)
#define CT_ERR_UNSUPPORTED (-2)      /* dtype / mode combination not built */
#define CT_ERR_LAUNCH (-3)           /* HIP launch failure */
#define CT_ERR_NO_GRADIENT_PATH (-4) /* the reference raises RuntimeError here (no grad path to the image) */
#define CT_ERR_TOO_LARGE (-5)        /* a dimension exceeds what the kernels index */

/* element type of an image stack */
#define CT_DTYPE_U8 0  /* raw codes; x = code / max_code exactly as Normalize(0, max_code) on float32 */
#define CT_DTYPE_U16 1
#define CT_DTYPE_F32 2 /* already normalised float32 pixel values */

/* clair_torch/common/enums.py:10 InterpMode (+ "no model": icrf_model=None) */
#define CT_INTERP_LOOKUP 0
#define CT_INTERP_LINEAR 1
#define CT_INTERP_CATMULL 2
#define CT_INTERP_NONE 3

/* where the standard uncertainty of a sample comes from (clair_torch/datasets/base.py:128-135 MissingStdMode) */
#define CT_STD_NONE 0       /* no uncertainty propagated */
#define CT_STD_CONSTANT 1   /* sigma = std_value */
#define CT_STD_MULTIPLIER 2 /* sigma = std_value * x */
#define CT_STD_EXPLICIT 3   /* sigma read from std_dev (float32, same layout as the stack) */

/* weight_fn of compute_hdr_image: None -> ones, anything else -> Gaussian scale 30 (hdr_merge.py:95) */
#define CT_WEIGHT_NONE 0
#define CT_WEIGHT_GAUSS 1

/* flags of ct_hdr_merge_batch */
#define CT_MERGE_FIRST_BATCH 1u  /* state is not read (WBOMean starts at mean 0, weight 0) */
#define CT_MERGE_FINALIZE 2u     /* also write mean_out / std_out = sqrt(variance) after this batch */
#define CT_MERGE_MEAN_OUT_F32 4u /* mean_out is float32 instead of the reference's float64 */
#define CT_MERGE_F64_MOMENTS 8u  /* diagnostic: keep the float64-moment kernel where the pivoted float32 one would run */
#define CT_MERGE_REFERENCE_ORDER 16u /* evaluate the uncertainty in the reference's own float32 autograd order (two passes,
                                        float64 exp / divisions; ct_merge_exact.hip).  Default for LOOKUP and CATMULL with
                                        uncertainties, whose reference results are dominated by float32 cancellation */
#define CT_MERGE_CLOSED_FORM 32u     /* keep the fast closed-form kernels for LOOKUP / CATMULL with uncertainties as well */
#define CT_MERGE_REQUIRE_ONE_LAUNCH 128u /* ct_hdr_merge_batches: CT_ERR_UNSUPPORTED instead of one launch per batch */
#define CT_MERGE_STD_HINT 64u        /* ct_hdr_merge_kernel_name only: uncertainties are propagated */
#define CT_MERGE_OUT_AS_INPUT 256u   /* extension: state and outputs in the MEMORY ORDER OF THE INPUT stack instead of planar
                                        (C, H, W): an interleaved (H, W, C) RGB / BGR stack then gives (H, W, C) outputs in the
                                        same channel order -- what an OpenCV writer wants (the reference's save path permutes
                                        back, common/general_functions.py:338-358) -- and the merge stores dense packets
                                        without regrouping.  Use the same flag on every batch of a merge. */

/* Memory layout of one image of a stack.  Outputs and state are always planar (C, H, W) like the reference's tensors.
 * NHWC = the interleaved layout OpenCV decodes to (clair_torch/common/data_io.py:125-154); NHWC_BGR additionally
 * reverses the channel order on the fly, i.e. folds cv_to_torch (clair_torch/common/general_functions.py:315-335)
 * into the load.  Supported by ct_hdr_merge_batch(es), ct_linearize_std, ct_pair_residual_fwd / _bwd and
 * ct_video_stats_batch (an explicit std stack is then interleaved likewise); flat field, dark field and band
 * statistics work on planar data. */
#define CT_LAYOUT_NCHW 0
#define CT_LAYOUT_NHWC 1
#define CT_LAYOUT_NHWC_BGR 2

/* Geometry of a (tile of a) stack. */
typedef struct ct_geometry {
    int32_t channels;     /* C */
    int64_t h_tile;       /* rows held locally */
    int64_t width;        /* W */
    int64_t h_global;     /* rows of the full image (== h_tile when untiled) */
    int64_t row_offset;   /* first global row held locally */
    int64_t image_stride; /* elements between consecutive exposures / frames */
    int32_t layout;       /* CT_LAYOUT_* of the INPUT stack */
} ct_geometry;

/* ICRF model: LUT (C, L) float32 row-major as ICRFModelBase._icrf (clair_torch/models/base.py:69-71). */
typedef struct ct_icrf {
    const float *lut_dev; /* NULL with interp == CT_INTERP_NONE */
    int32_t n_points;     /* L */
    int32_t interp;       /* CT_INTERP_* */
} ct_icrf;

/* library / ABI */
int ct_abi_version(void);
const char *ct_error_string(int code);

/*
 * ct_hdr_merge_batch -- one batch of compute_hdr_image's loop body
 * (clair_torch/inference/hdr_merge.py:61-128: linearize, / exposure, weights, WBOMean.update_values
 *  [clair_torch/common/statistics.py:64-109], autograd variance [hdr_merge.py:107-115], internal_detach)
 * fused into one pass over the batch; with CT_MERGE_FINALIZE also hdr_merge.py:155 (squeeze + sqrt).
 *
 *   stack_dev      (B, C, H_tile, W) of `dtype`, exposures sorted as custom_collate does (datasets/collate.py:23)
 *   max_code       255 / 65535 / ... for integer dtypes (ignored for CT_DTYPE_F32)
 *   std_dev        explicit float32 std stack (CT_STD_EXPLICIT) else NULL; std_value for CONSTANT / MULTIPLIER
 *   exposure_dev   (B) float64 exposure times (collate.py:29 makes them float64)
 *   mean_state_dev (Q) float64, sumw_state_dev (Q) float32, var_state_dev (Q) float32, Q = C*H_tile*W:
 *                  WBOMean state + running variance, updated in place; may all be NULL when flags has both
 *                  FIRST_BATCH and FINALIZE (single-batch merge)
 *   mean_out_dev   (Q) float64 (or float32 with CT_MERGE_MEAN_OUT_F32), std_out_dev (Q) float32 or NULL when
 *                  std_mode == CT_STD_NONE; written only with CT_MERGE_FINALIZE
 * Returns CT_ERR_NO_GRADIENT_PATH for LOOKUP + CT_WEIGHT_NONE + std (the reference's autograd.grad raises).
 */
int ct_hdr_merge_batch(const void *stack_dev, int32_t dtype, float max_code, int32_t batch, const ct_geometry *geom,
                       const float *std_dev, int32_t std_mode, float std_value, const double *exposure_dev,
                       const ct_icrf *icrf, int32_t weight_mode, double *mean_state_dev, float *sumw_state_dev,
                       float *var_state_dev, void *mean_out_dev, float *std_out_dev, uint32_t flags, void *stream);

/*
 * ct_hdr_merge_batches -- n_batches consecutive iterations of compute_hdr_image's loop (hdr_merge.py:61-128) in one call:
 * exactly ct_hdr_merge_batch applied to batch 0 .. n_batches-1 in turn (CT_MERGE_FIRST_BATCH on batch 0 and
 * CT_MERGE_FINALIZE on the last one when set in `flags`), bit for bit.  Where the pivoted code-domain kernel applies and
 * every batch is packet-aligned (n_batches <= 16) it is ONE launch that keeps (mean, sum of weights, variance) in
 * registers between the batches -- the reference's default batch_size: 4 (scripts/config.yaml) otherwise moves 32 B of
 * state per element and batch beside 8 B of samples.  The reference-order kernel (CT_MERGE_REFERENCE_ORDER, the default
 * of LOOKUP / CATMULL with uncertainties) does the same for any dtype and alignment.  Anything else runs one launch per
 * batch.
 *   stack_devs / std_devs / batch_sizes  HOST arrays of n_batches device pointers / sizes (std_devs NULL unless EXPLICIT);
 *                                        every batch is (B_b, C, H_tile, W) with the common geometry, sorted as collate does
 *   exposure_dev                         the exposure times of all batches, concatenated in the same order (device)
 *   state / outputs / flags              as ct_hdr_merge_batch; the state may be NULL when flags has FIRST_BATCH and FINALIZE
 */
int ct_hdr_merge_batches(const void *const *stack_devs, const float *const *std_devs, const int32_t *batch_sizes,
                         int32_t n_batches, int32_t dtype, float max_code, const ct_geometry *geom, int32_t std_mode,
                         float std_value, const double *exposure_dev, const ct_icrf *icrf, int32_t weight_mode,
                         double *mean_state_dev, float *sumw_state_dev, float *var_state_dev, void *mean_out_dev,
                         float *std_out_dev, uint32_t flags, void *stream);

/*
 * Host-side proofs behind the folded integer paths (no device work; results cached per argument set):
 *   ct_norm_constants        fma(u, hi, u * lo) == the reference's float32 u / max_code for EVERY code
 *                            (clair_torch/common/general_functions.py:377) -- else CT_ERR_UNSUPPORTED
 *   ct_index_constants       the folded LUT coordinate has the reference's floor and round-half-even for every code
 *                            (clair_torch/models/base.py:146,166)
 *   ct_pivot_index_constants floor(u (L-1) / max_code) == (u * index_mul) >> 32 (or == u when *index_mul == 0) for every
 *                            code, with step = max_code / (L-1) an integer (what ct::merge_pivot_kernel addresses its
 *                            table with)
 *   ct_pivot_floor_constants the same interval from the code held as a float (typed buffer loads): mantissa of
 *                            fma(u, *rcp_step, 1.5 * 2^23) rounded toward minus infinity, for every code
 */
int ct_norm_constants(float max_code, float *hi, float *lo);
int ct_index_constants(float max_code, int n_points, float *hi, float *lo);
int ct_pivot_index_constants(float max_code, int n_points, uint32_t *index_mul, float *step);
int ct_pivot_floor_constants(float max_code, int n_points, float *rcp_step);
/*   ct_pivot_interval_constants  the general form of the above: table entry min(floor(u * *scale), last) equals the
 *                            reference's LINEAR interval (lookup = 0) / LOOKUP sample (entry j -> sample (j + 1) >> 1,
 *                            lookup = 1) for every code 0..dtype_max of the container, any max_code <= dtype_max, any L */
int ct_pivot_interval_constants(float max_code, int n_points, int lookup, int dtype_max, float *scale);

/* Diagnostics: a static string naming the kernel ct_hdr_merge_batch dispatches for these arguments. */
const char *ct_hdr_merge_kernel_name(int32_t dtype, float max_code, int32_t interp, int32_t n_points, uint32_t flags);

/* Diagnostics: device counter bumped once per wavefront of the pivoted merge kernel that had to repeat a batch about the
 * exact mean (ill-conditioned pivot); NULL disables.  Process-global, not part of the data path. */
void ct_merge_set_retry_counter(unsigned long long *counter_dev);

/*
 * ct_linearize_std -- body of linearize_dataset_generator for F independent frames
 * (clair_torch/inference/linearization.py:95-106,132): lin = f(x), std = sqrt((f'(x) * sigma)^2), float32,
 * bit-exact with the reference for LOOKUP / LINEAR.  std_out_dev may be NULL (value only).
 * Each frame is its own batch of one, as the reference requires (linearization.py:42).
 */
int ct_linearize_std(const void *frames_dev, int32_t dtype, float max_code, int64_t n_frames, const ct_geometry *geom,
                     const float *std_dev, int32_t std_mode, float std_value, const ct_icrf *icrf, float *lin_out_dev,
                     float *std_out_dev, void *stream);

/*
 * ct_linearize_fwd / ct_linearize_bwd -- ICRFModelBase.forward (clair_torch/models/base.py:135-226) on a
 * (N, C, H_tile, W) float32 tensor and its backward: grad wrt the image (analytic derivative of the LUT
 * interpolation incl. the clamp mask) and, when lut_grad_dev != NULL, the (C, L) float32 LUT gradient
 * accumulated (+=) with LDS-privatised scatter.  Used by the model's autograd wrapper.
 */
int ct_linearize_fwd(const float *x_dev, int64_t n_images, const ct_geometry *geom, const ct_icrf *icrf,
                     float *out_dev, void *stream);
int ct_linearize_bwd(const float *x_dev, const float *grad_out_dev, int64_t n_images, const ct_geometry *geom,
                     const ct_icrf *icrf, float *grad_x_dev, float *lut_grad_dev, void *stream);

/* Parameters of the exposure-pair linearity residual (train_icrf / measure_linearity). */
typedef struct ct_pair_params {
    float lower, upper;                 /* inclusive validity range of the raw pixel value (general_functions.py:302) */
    float weight_scale;                 /* Gaussian pair-weight scale, 10 in the reference (losses.py:212) */
    int32_t use_relative;               /* use_relative_linearity_loss */
    int32_t use_uncertainty_weighting;  /* add 1 / (err + 1e-6) to the weights (losses.py:96) */
    int32_t std_mode;                   /* CT_STD_*: source of sigma for the linearized std |f'(x) sigma| */
    float std_value;
    int32_t pair_band;                  /* backward only, a hint: 0 = nothing known; B > 0 = the caller promises that every
                                           pair (i, j) of the list has 0 < j - i <= B (get_valid_exposure_pairs on a sorted
                                           exposure series with a ratio limit gives such a band).  Enables the
                                           lane-per-sample backward for n_images <= 64; the promise is verified on the
                                           device and a broken one only costs the fast path. */
} ct_pair_params;

/*
 * ct_pair_residual_fwd -- spatial sums of the per-pixel linearity residual for P exposure pairs; replaces
 * get_pairwise_valid_pixel_mask + combined_gaussian_pair_weights + model forward + pixelwise_linearity_loss +
 * compute_spatial_linearity_loss of one train_icrf step (clair_torch/training/icrf_training.py:105-133) or of
 * measure_linearity (clair_torch/inference/measure_linearity.py:44-72).
 *   stack_dev (N, C, H_tile, W); i_idx_dev / j_idx_dev (P) int32, ratio_dev (P) float64 = t_i / t_j
 *   (get_valid_exposure_pairs, common/general_functions.py:242-272)
 *   level 0: sums 0..1 only (training), level 1: all five sums
 *   center_dev (P, C) float64 or NULL: value subtracted from v inside sum [2].  The weighted std needs
 *     sum (v - mean)^2 w m (general_functions.py:163-165); expanding it from raw moments cancels by (mean/std)^2, so
 *     callers run level 1 twice: once for the means, once with center = mean.
 *   sums_dev (P, C, 5) float64, ACCUMULATED (+=, caller zeroes; tiles / ranks add up):
 *     [0] sum w m   [1] sum v w m   [2] sum (v - center)^2 w m   [3] sum err m   [4] sum m
 *   with v the (relative) absolute residual, w the weight, m the validity mask, err the residual's uncertainty.
 *   spatial mean = [1] / max([0], 1e-8), etc. (general_functions.py:149-170).
 */
int ct_pair_residual_fwd(const void *stack_dev, int32_t dtype, float max_code, int32_t n_images,
                         const ct_geometry *geom, const float *std_dev, const ct_icrf *icrf, const int32_t *i_idx_dev,
                         const int32_t *j_idx_dev, const double *ratio_dev, int32_t n_pairs,
                         const ct_pair_params *params, int32_t level, const double *center_dev, double *sums_dev,
                         void *stream);

/*
 * ct_pair_residual_bwd -- gradient of the training loss's linearity term with respect to the (C, L) LUT
 * (what loss[c].backward() deposits in the model parameters, icrf_training.py:148-149, for that term).
 *   coef_dev (P, C) float64 = dL/d(spatial mean_pc) / max(sums[p][c][0], 1e-8)
 *   partner lists (CSR over samples): for sample n, entries partner_offsets[n] .. partner_offsets[n+1]-1 give the
 *   other sample of every pair containing n (partner_sample_dev) and the pair id (partner_pair_dev: p when n is
 *   the pair's first image i, ~p when it is the second image j)
 *   std_dev / params->std_mode: uncertainties, read only with params->use_uncertainty_weighting; the weights
 *   1/(err + 1e-6) then depend on the LUT (relative loss), and the backward needs smean_dev (P, C) float64 = the
 *   spatial means of the forward:  d mean = sum m [w dv + (v - mean) dw] / sum w m.  smean_dev may be NULL otherwise.
 *   lut_grad_dev (C, L) float64, ACCUMULATED (+=)
 *   workspace_dev: caller-owned scratch of at least ct_pair_residual_bwd_workspace(n_images, n_pairs, channels) bytes,
 *   32-byte aligned (the library allocates nothing): the per-channel partner tables the kernel reads with scalar loads
 *   are built there by a small preparatory launch on the same stream.  Contents are undefined afterwards.
 */
int64_t ct_pair_residual_bwd_workspace(int32_t n_images, int32_t n_pairs, int32_t channels);
int ct_pair_residual_bwd(const void *stack_dev, int32_t dtype, float max_code, int32_t n_images,
                         const ct_geometry *geom, const float *std_dev, const ct_icrf *icrf, const double *ratio_dev,
                         int32_t n_pairs, const int32_t *partner_offsets_dev, const int32_t *partner_sample_dev,
                         const int32_t *partner_pair_dev, const ct_pair_params *params, const double *coef_dev,
                         const double *smean_dev, double *lut_grad_dev, void *workspace_dev, int64_t workspace_bytes,
                         void *stream);

/*
 * Flat-field correction epilogues (clair_torch/inference/hdr_merge.py:131-153, linearization.py:48-57,118-130;
 * flat_field_mean / flatfield_correction, clair_torch/common/general_functions.py:182-238, whole-image ROI as both
 * call sites pass mid_area_side_fraction = 1.0).
 *
 * ct_flatfield_sums: sums_dev (C, 2) float64 += [sum flat, sum value / (flat + 1e-6)] over a (C, plane) band
 *   (value_dev may be NULL: only the first sum).  Additive over row bands: ranks all-reduce, then divide by the global
 *   pixel count to get M_c (float32) and the "through the mean" gradient term.
 * ct_flatfield_apply: value (F, C, plane) float64 or float32, in place: value / (flat + 1e-6) * M_c;
 *   var_or_std_dev (F, C, plane) float32 in place (or NULL): on entry the variance (input_is_variance != 0, merge)
 *   or the std (linearize) of the image term, on exit sqrt(var + (grad * flat_std)^2) with
 *   grad = -value * M_c / (flat + 1e-6)^2 + through_mean_dev[c]  (through_mean_dev NULL = 0, flat_std_dev NULL = no term).
 */
int ct_flatfield_sums(const void *value_dev, int32_t value_is_f64, const float *flat_dev, int32_t channels,
                      int64_t plane, double *sums_dev, void *stream);
int ct_flatfield_apply(void *value_dev, int32_t value_is_f64, int64_t n_frames, float *var_or_std_dev,
                       int32_t input_is_variance, const float *flat_dev, const float *flat_std_dev,
                       const float *flat_mean_dev, const double *through_mean_dev, int32_t channels, int64_t plane,
                       void *stream);

/*
 * ct_band_stats -- per-channel statistics of one merged row band, the quantity BASELINE configuration C5 gathers over
 * RCCL ("tile-sharded ... merge + uncertainty, RCCL gather of per-tile stats"; the reference has no such function: its
 * script saves the merged image, scripts/run_hdr_merging.py:60-86).  One pass over the band:
 *   out_dev (6, C) float64 rows = min mean, max mean, sum mean, min std, max std, sum std   (std rows 0 when std_dev NULL)
 * mean_dev (C, plane) float64 and std_dev (C, plane) float32 are ct_hdr_merge_batch's outputs.  Deterministic (no
 * atomics); min / max / sum combine over bands.  workspace_dev: ct_band_stats_workspace(channels) bytes, 8-byte aligned.
 */
int64_t ct_band_stats_workspace(int32_t channels);
int ct_band_stats(const double *mean_dev, const float *std_dev, int32_t channels, int64_t plane, void *workspace_dev,
                  int64_t workspace_bytes, double *out_dev, void *stream);

/*
 * ct_dark_field_blur -- conditional_gaussian_blur(images, dark, threshold, 3, differentiable=True)
 * (clair_torch/common/general_functions.py:440-486) as compute_hdr_image (inference/hdr_merge.py:76-92,117-126) and
 * linearize_dataset_generator (inference/linearization.py:73-92,108-116) apply it, together with the per-sample
 * uncertainty that carries BOTH variance terms of those call sites through the unchanged merge / linearize kernels:
 *   xb_out      (B, C, H_tile, W) float32 = m blur3x3(x) + (1 - m) x,  m = sigmoid(alpha (dark - threshold))
 *   std_out     (B, C, H_tile, W) float32 = sqrt(sigma^2 + ((blur(x) - x) alpha m (1 - m) sigma_dark)^2), or NULL
 *   stack_dev   (B, C, H_tile, W) codes / float32 pixels (NCHW only); std_dev / std_mode / std_value: sigma of the RAW image
 *   dark_dev, dark_std_dev (dark_batch, C, H_tile, W) float32, dark_batch = 1 (shared) or B (one matched dark field per frame)
 *   halo_dev    (B, C, 2, W) in the stack's element type: the global rows just above and below the band; required unless
 *               the band is the whole image (at the global top / bottom the blur reflects instead and ignores that row)
 * PARITY UNPINNED: the blur is torchvision's GaussianBlur(3, sigma=1), restated from its published algorithm.
 */
int ct_dark_field_blur(const void *stack_dev, int32_t dtype, float max_code, int32_t batch, const ct_geometry *geom,
                       const void *halo_dev, const float *std_dev, int32_t std_mode, float std_value,
                       const float *dark_dev, const float *dark_std_dev, int32_t dark_batch, float threshold, float alpha,
                       float *xb_out_dev, float *std_out_dev, void *stream);

/*
 * ct_video_stats_batch -- loop body of compute_video_mean_and_std
 * (clair_torch/inference/inferential_statistics.py:38-47): optional ICRF linearization of a batch of frames and the
 * unweighted WBOMeanVar update (clair_torch/common/statistics.py:213-259), float32 like the reference.
 *   frames_dev (B, C, H_tile, W); frames_before = number of frames already merged (0 for the first batch)
 *   mean_state_dev, m2_state_dev (C, H_tile, W) float32, updated in place.
 * After the last batch: mean = mean_state, std of the mean = sqrt(m2 / (n - 1)) / sqrt(n)  (SAMPLE_FREQUENCY).
 */
int ct_video_stats_batch(const void *frames_dev, int32_t dtype, float max_code, int32_t batch, const ct_geometry *geom,
                         const ct_icrf *icrf, float frames_before, float *mean_state_dev, float *m2_state_dev,
                         void *stream);

#ifdef __cplusplus
}
#endif
#endif /* CLAIR_HIP_H */
