"""Eager-PyTorch restatement of the clair-torch hot path -- TEST INFRASTRUCTURE ONLY.

This is the "reference-equivalent eager path": the same sequence of full-tensor PyTorch operations and
``torch.autograd.grad`` calls the reference performs (float32 images, float64 exposure times, detached
streaming state), written against plain tensors instead of DataLoaders.  It is used

* by tests as the oracle for quantities that need autograd (training-step LUT gradients), and
* by ``bench.py`` as the ``cpu_baseline`` ("port") timed on the host cores.

It is never imported by the product package.  Parity status: pinned against the golden vectors recorded
from the reference (tests/test_oracle_golden.py).  Citations are file:line in /root/reference.
"""
import torch

LOOKUP, LINEAR, CATMULL = "lookup", "linear", "catmull"


def icrf_forward(image, lut, mode=LINEAR):
    """ICRFModelBase.forward (clair_torch/models/base.py:135-226) for image (N,C,H,W), lut (C,L)."""
    n, c, h, w = image.shape
    top = lut.shape[1] - 1
    if mode == LOOKUP:  # base.py:138-158: nearest sample, true per-channel row, no gradient
        idx = (image * top).round().clamp(0, top).long()
        chan = torch.arange(c, device=image.device).view(1, c, 1, 1).expand(n, c, h, w)
        return lut[chan, idx]
    # base.py:173-176 / 216-219: the row is the flat NCHW position modulo C, not the channel
    rows = torch.arange(c, device=image.device).repeat(n * h * w)

    def take(ix):
        return lut[rows, ix.reshape(-1)].reshape(n, c, h, w)

    if mode == LINEAR:  # base.py:160-182
        s = (image * top).clamp_(0, top)
        i0 = s.floor().long()
        i1 = (i0 + 1).clamp_(0, top)
        fr = s - i0.float()
        return take(i0) * (1.0 - fr) + take(i1) * fr
    if mode == CATMULL:  # base.py:184-226
        s = (image * top).clamp(0, top)
        i0 = s.floor().long()
        t = (s - i0.float()).clamp(0, 1)
        t2 = t * t
        t3 = t2 * t
        basis = (-0.5 * t3 + t2 - 0.5 * t, 1.5 * t3 - 2.5 * t2 + 1.0, -1.5 * t3 + 2.0 * t2 + 0.5 * t,
                 0.5 * t3 - 0.5 * t2)
        taps = [take((i0 + k).clamp(0, top)) for k in (-1, 0, 1, 2)]
        return torch.stack([b * g for b, g in zip(basis, taps)], dim=0).sum(dim=0)
    raise ValueError(f"Unknown interpolation mode {mode}")


def gaussian_weight(image, scale=30.0):
    """gaussian_value_weights, clair_torch/training/losses.py:193-205."""
    return torch.exp(-scale * (image - 0.5) ** 2)


def merge_stack(vals, stds, exposures, lut, mode=LINEAR, use_gauss=True, partition=None, batches=None):
    """compute_hdr_image (clair_torch/inference/hdr_merge.py:61-155) without artefact corrections.

    vals (N,C,H,W) f32, stds same or None, exposures (N) f64.  ``partition`` lists the batch sizes.
    Streaming state follows WBOMean (clair_torch/common/statistics.py:64-109) incl. the per-batch detach.
    """
    n = vals.shape[0]
    if batches is None:   # ``partition`` = consecutive batch sizes; ``batches`` = explicit index lists (sorted per batch)
        partition = [n] if partition is None else list(partition)
        batches, k = [], 0
        for b in partition:
            batches.append(list(range(k, k + b)))
            k += b
    mean_a, w_a, variance = 0.0, 0.0, None
    for idx in batches:
        x = vals[idx].clone().requires_grad_(stds is not None)
        sd = None if stds is None else stds[idx]
        t = exposures[idx].to(torch.float64).view(-1, 1, 1, 1)
        wts = gaussian_weight(x) if use_gauss else torch.ones_like(x)
        with torch.set_grad_enabled(stds is not None):
            y = (icrf_forward(x, lut, mode) if lut is not None else x) / t
        w_b = wts.sum(dim=0, keepdim=True)
        m_b = (wts * y).sum(dim=0, keepdim=True) / (w_b + 1e-6)
        w_t = w_a + w_b
        mean = mean_a + (w_b / w_t) * (m_b - mean_a)
        if sd is not None:
            g = torch.autograd.grad(mean, x, torch.ones_like(mean), retain_graph=False)[0]
            upd = ((g * sd) ** 2).sum(dim=0, keepdim=True)
            variance = upd if variance is None else variance + upd
        mean_a, w_a = mean.detach(), w_t.detach()
    return mean_a.squeeze(0), (None if variance is None else torch.sqrt(variance.squeeze(0)))


# ---- float32-order emulation of the reference's backward (no autograd) ---------------------------------------------
# What torch.autograd.grad computes at hdr_merge.py:107-113, written out operation by operation in the order PyTorch's
# engine executes the nodes (highest sequence number first; gradients arriving at one tensor are added in arrival order)
# and in the dtype of each forward operation (float32 image-side, float64 once the exposure times enter).  With the same
# exp() it reproduces the reference's recorded uncertainties BIT FOR BIT (tests/test_oracle_golden.py), so it serves as
#   * the specification of the kernels' "reference order" paths (CATMULL derivative: ct_device.hpp catmull_backward_ref;
#     the exact-order merge pass: ct_merge.hip), and
#   * the instrument that tells rounding noise of the reference itself from error of the build: replacing ``exp`` by a
#     correctly rounded one (torch's CPU exp is Sleef's 1-ULP expf: 1.1 % of its results differ from correct rounding)
#     moves the reference's own CATMULL uncertainty by up to 1.1e-5 element-wise on the uint16 fixtures.
def icrf_forward_backward_reference_order(x, lut, mode):
    """ICRFModelBase.forward (clair_torch/models/base.py:135-226) and a closure G -> d(sum G * out)/dx evaluated in
    autograd's float32 operation order.  x (N,C,H,W) float32.  LOOKUP has no gradient (closure None)."""
    n, c, h, w = x.shape
    size = lut.shape[1]
    top = float(size - 1)
    if mode == LOOKUP:
        return icrf_forward(x, lut, mode), None
    rows = torch.arange(c).repeat(n * h * w)

    def take(ix):
        return lut[rows, ix.reshape(-1)].reshape(n, c, h, w)

    sraw = x * top
    s = sraw.clamp(0, top)
    smask = ((sraw >= 0) & (sraw <= top)).to(x.dtype)                # ClampBackward1
    i0 = s.floor().long()
    if mode == LINEAR:                                               # base.py:160-182
        fr = s - i0.float()
        g0, g1 = take(i0), take((i0 + 1).clamp(0, size - 1))
        out = g0 * (1.0 - fr) + g1 * fr

        def backward(grad):
            # nodes: rsub (1 - fr), mul g0 *, mul g1 *, add.  Engine: add, g1 * fr (fr receives G g1 first), g0 * (1 - fr),
            # rsub (fr receives -(G g0) second)
            gfr = (grad * g1) + (-(grad * g0))
            return (gfr * smask) * top

        return out, backward
    if mode == CATMULL:                                              # base.py:184-226
        traw = s - i0.float()
        t = traw.clamp(0, 1)
        tmask = ((traw >= 0) & (traw <= 1)).to(x.dtype)
        t2 = t * t
        t3 = t2 * t
        basis = (-0.5 * t3 + t2 - 0.5 * t, 1.5 * t3 - 2.5 * t2 + 1.0, -1.5 * t3 + 2.0 * t2 + 0.5 * t, 0.5 * t3 - 0.5 * t2)
        taps = [take((i0 + k).clamp(0, size - 1)) for k in (-1, 0, 1, 2)]
        out = torch.stack([b * g for b, g in zip(basis, taps)], dim=0).sum(dim=0)

        def backward(grad):
            # the four products w_k * g_k run first (latest nodes), then the nodes of w3, w2, w1, w0 in that order; each
            # feeds t3, t2 and t, whose buffers add in arrival order; t3 = t2 * t and t2 = t * t run last
            g0, g1, g2, g3 = [grad * g for g in taps]
            a3 = ((0.5 * g3 - 1.5 * g2) + 1.5 * g1) - 0.5 * g0                          # d / d t3
            a2 = ((((-(0.5 * g3)) + 2.0 * g2) - 2.5 * g1) + g0) + a3 * t                # d / d t2 (last arrival: via t3)
            at = (((0.5 * g2 - 0.5 * g0) + a3 * t2) + a2 * t) + a2 * t                  # d / d t  (t * t feeds t twice)
            return ((at * tmask) * smask) * top

        return out, backward
    raise ValueError(f"Unknown interpolation mode {mode}")


def merge_stack_reference_order(vals, stds, exposures, lut, mode=LINEAR, use_gauss=True, batches=None, exp=torch.exp):
    """compute_hdr_image (clair_torch/inference/hdr_merge.py:61-155) with the backward of :107-113 written out in
    autograd's order.  ``batches``: list of index lists (each already sorted by exposure as custom_collate does) or None
    for one batch.  ``exp``: the exponential used by gaussian_value_weights (default torch.exp = what the reference runs)."""
    n = vals.shape[0]
    batches = [list(range(n))] if batches is None else batches
    f32, f64 = torch.float32, torch.float64
    mean_a = w_a = variance = None
    for idx in batches:
        x = vals[idx]
        sd = None if stds is None else stds[idx]
        t = exposures[idx].to(f64).view(-1, 1, 1, 1)
        xm = x - 0.5
        wts = exp(-30.0 * xm ** 2) if use_gauss else torch.ones_like(x)             # losses.py:205 / hdr_merge.py:95
        if lut is None:
            lin, backward = x, (lambda grad: grad)
        else:
            lin, backward = icrf_forward_backward_reference_order(x, lut, mode)
        y = lin / t                                                                   # float64 from here on
        w_b = wts.sum(dim=0, keepdim=True)                                            # statistics.py:78-80 (float32)
        swy = (wts * y).sum(dim=0, keepdim=True)
        d = w_b + 1e-6
        m_b = swy / d
        first = mean_a is None
        w_t = w_b if first else w_a + w_b                                             # statistics.py:103 (0.0 + W_B first)
        frac = w_b / w_t                                                              # float32
        diff = m_b if first else m_b - mean_a
        mean = frac * diff if first else mean_a + frac * diff
        if sd is not None:
            g_frac = diff.to(f32)                                                     # d mean / d frac, cast at the float32 tensor
            g_swy = frac.to(f64) / d.to(f64)
            g_d = (-frac.to(f64) * ((swy / d) / d)).to(f32)                           # DivBackward0: -grad * ((self / other) / other)
            g_wb = g_frac / w_t                                                       # arrival 1: W_B / W, self
            g_wb = g_wb + (-g_frac * ((w_b / w_t) / w_t))                             # arrival 2: through W = W_A + W_B
            g_wb = g_wb + g_d                                                         # arrival 3: through W_B + 1e-6
            g_lin = ((g_swy * wts) / t).to(f32)                                       # into the model's output
            g_x = None if backward is None else backward(g_lin)                       # model nodes run before the weight nodes
            if use_gauss:
                g_w = (g_swy * y).to(f32) + g_wb                                      # mul first, then the expanded sum
                g_a = ((g_w * wts) * -30.0) * (2.0 * xm)                              # exp, * (-scale), pow 2 backward
                g_x = g_a if g_x is None else g_x + g_a
            if g_x is None:
                raise RuntimeError("element 0 of tensors does not require grad and does not have a grad_fn")
            upd = ((g_x * sd) ** 2).sum(dim=0, keepdim=True)
            variance = upd if variance is None else variance + upd
        mean_a, w_a = mean, w_t
    return mean_a.squeeze(0), (None if variance is None else torch.sqrt(variance.squeeze(0)))


def linearize_frame(val, std, lut, mode=LINEAR):
    """One iteration of linearize_dataset_generator (clair_torch/inference/linearization.py:95-106,132)."""
    x = val.unsqueeze(0).clone().requires_grad_(std is not None)
    with torch.set_grad_enabled(std is not None):
        lin = icrf_forward(x, lut, mode)
    var = torch.zeros_like(x)
    if std is not None:
        g = torch.autograd.grad(lin, x, torch.ones_like(lin))[0]
        var = var + (g * std.unsqueeze(0)) ** 2
    return lin.detach().squeeze(0), torch.sqrt(var).detach().squeeze(0)


def exposure_pairs(exposures, threshold):
    """get_valid_exposure_pairs, clair_torch/common/general_functions.py:242-272."""
    n = exposures.shape[0]
    ratios = exposures.view(n, 1) / exposures.view(1, n)
    i, j = torch.triu_indices(n, n, offset=1)
    r = ratios[i, j]
    if threshold is not None:
        keep = r >= threshold
        i, j, r = i[keep], j[keep], r[keep]
    return i, j, r


def masked_weighted_mean_std(values, weights, mask, eps=1e-8):
    """weighted_mean_and_std over (H,W) with mask, clair_torch/common/general_functions.py:118-178."""
    m = mask.to(values.dtype)
    v = values * m
    w = weights * m if weights is not None else m
    total = w.sum(dim=(2, 3), keepdim=True).clamp(min=eps)
    mean = (v * w).sum(dim=(2, 3), keepdim=True) / total
    std = torch.sqrt((((v - mean) ** 2) * w).sum(dim=(2, 3), keepdim=True) / total)
    return mean.squeeze((2, 3)), std.squeeze((2, 3))


def linearity_statistics(vals, stds, exposures, lut, mode, ratio_threshold, lo, hi, use_relative, use_unc_weight, forward=None):
    """Body shared by train_icrf (clair_torch/training/icrf_training.py:105-136) and measure_linearity
    (clair_torch/inference/measure_linearity.py:44-72).  ``lut`` may require grad (training) or be None.
    ``forward``: replaces icrf_forward(x, lut, mode) (linearity_lut_grad_f64 passes one built on explicit LUT taps).
    Returns (ratio, spatial mean (P,C), spatial std, spatial error|None)."""
    i, j, r = exposure_pairs(exposures.to(torch.float64), ratio_threshold)
    xi, xj = vals[i], vals[j]
    mask = (xi >= lo) & (xi <= hi) & (xj >= lo) & (xj <= hi)                  # general_functions.py:302
    gw = gaussian_weight(xi, 10.0) + gaussian_weight(xj, 10.0)               # losses.py:208-235
    x = vals.clone().requires_grad_(stds is not None or (lut is not None and lut.requires_grad))
    lin = (forward(x) if forward is not None else icrf_forward(x, lut, mode)) if lut is not None else x
    if stds is not None:
        g = torch.autograd.grad(lin, x, torch.ones_like(lin), retain_graph=True)[0]
        lsd = (g * stds).abs()
    else:
        lsd = None
    rr = r.view(-1, 1, 1, 1)
    li, lj = lin[i], lin[j]
    expected = lj * rr                                                         # losses.py:41
    diff = li - expected
    safe = expected + 1e-6
    if use_relative:
        diff = diff / safe
    loss = diff.abs()
    err = None
    if lsd is not None:                                                        # losses.py:50-63
        si, sj = lsd[i], lsd[j]
        if use_relative:
            err = torch.sqrt((si / safe) ** 2 + ((li * sj) / (safe * lj.clamp(min=1e-6))) ** 2 + 1e-6)
        else:
            err = torch.sqrt(si ** 2 + (rr * sj) ** 2)
    weights = torch.zeros_like(loss)                                           # losses.py:93-100
    if err is not None and use_unc_weight:
        weights = weights + 1 / (err + 1e-6)
    weights = weights + gw
    sp_mean, sp_std = masked_weighted_mean_std(loss, weights, mask)
    sp_err = None if err is None else masked_weighted_mean_std(err, None, mask)[0]
    return r, sp_mean, sp_std, sp_err


def curve_penalties(curve):
    """The four per-channel ICRF penalties, clair_torch/training/losses.py:111-190, curve (C,L)."""
    df = curve[:, 1:] - curve[:, :-1]
    mono = ((df <= 0).float() * df.pow(2)).sum(dim=1)
    rng = (torch.relu(-curve) + torch.relu(curve - 1)).sum(dim=1)
    endp = curve[:, 0] ** 2 + (curve[:, -1] - 1) ** 2
    smooth = (curve[:, :-2] - 2 * curve[:, 1:-1] + curve[:, 2:]).pow(2).sum(dim=1)
    return mono, rng, endp, smooth


def training_loss(vals, stds, exposures, lut, mode=LINEAR, ratio_threshold=0.25, lo=1 / 255, hi=254 / 255,
                  use_relative=True, use_unc_weight=False, alpha=1.0, beta=1.0, gamma=1.0, delta=1.0):
    """Per-channel loss of one train_icrf step (clair_torch/training/icrf_training.py:105-143).
    Returns (loss (C,), linearity term (C,), spatial (P,C))."""
    _, sp, _, _ = linearity_statistics(vals, stds, exposures, lut, mode, ratio_threshold, lo, hi, use_relative,
                                       use_unc_weight)
    lin_loss = torch.sqrt((sp ** 2).sum(dim=0))
    mono, rng, endp, smooth = curve_penalties(lut)
    return lin_loss + alpha * mono + beta * rng + gamma * endp + delta * smooth, lin_loss, sp


def linearity_lut_grad_f64(vals, stds, exposures, lut, mode=LINEAR, ratio_threshold=0.25, lo=1 / 255, hi=254 / 255,
                           use_relative=True, use_unc_weight=False):
    """The LUT gradient of the training loss's linearity term with a DETERMINISTIC accumulation: the comparand for the
    backward kernels.  The reference's gradient reaches the (C, L) LUT through the backward of ``icrf[rows, idx]``, a
    float32 index_put whose accumulation order depends on the CPU thread count (2.6e-6 norm-wise between 3 and 8
    threads on a uint8 CATMULL case).  Here the LUT taps are explicit leaves: out = sum_k basis_k * tap_k with the same
    float32 arithmetic as icrf_forward, autograd delivers d loss / d tap_k per sample exactly as it would hand them to
    index_put, and the scatter into the (C, L) bins is done in float64 (order-independent to 1e-15).
    Returns (linearity loss (C,), spatial means (P,C), LUT gradient (C,L) float64)."""
    n, c, h, w = vals.shape
    size = lut.shape[1]
    top = size - 1
    lut = lut.detach()
    rows = torch.arange(c).repeat(n * h * w)
    taps, tap_index = [], []

    def forward(x):
        def take(ix):
            tap_index.append(ix.reshape(-1))
            taps.append(lut[rows, ix.reshape(-1)].reshape(n, c, h, w).clone().requires_grad_(True))
            return taps[-1]

        if mode == LOOKUP:
            raise ValueError("LOOKUP carries no LUT gradient in the reference")
        if mode == LINEAR:
            s = (x * top).clamp(0, top)
            i0 = s.floor().long()
            fr = s - i0.float()
            return take(i0) * (1.0 - fr) + take((i0 + 1).clamp(0, top)) * fr
        s = (x * top).clamp(0, top)
        i0 = s.floor().long()
        t = (s - i0.float()).clamp(0, 1)
        t2 = t * t
        t3 = t2 * t
        basis = (-0.5 * t3 + t2 - 0.5 * t, 1.5 * t3 - 2.5 * t2 + 1.0, -1.5 * t3 + 2.0 * t2 + 0.5 * t, 0.5 * t3 - 0.5 * t2)
        return torch.stack([b * take((i0 + k).clamp(0, top)) for b, k in zip(basis, (-1, 0, 1, 2))], dim=0).sum(dim=0)

    _, sp, _, _ = linearity_statistics(vals, stds, exposures, lut, mode, ratio_threshold, lo, hi, use_relative, use_unc_weight,
                                       forward=forward)
    lin_loss = torch.sqrt((sp ** 2).sum(dim=0))
    grads = torch.autograd.grad(lin_loss.sum(), taps)
    out = torch.zeros((c, size), dtype=torch.float64)
    for g, ix in zip(grads, tap_index):
        out.index_put_((rows, ix), g.reshape(-1).double(), accumulate=True)
    return lin_loss.detach(), sp.detach(), out


def video_mean_std(frames, lut, mode, batch_sizes):
    """compute_video_mean_and_std (clair_torch/inference/inferential_statistics.py:19-49): unweighted WBOMeanVar over
    batches of frames (clair_torch/common/statistics.py:213-259), SAMPLE_FREQUENCY variance, std of the mean."""
    mean_a = w_a = m2_a = 0.0
    k = 0
    with torch.inference_mode():
        for b in batch_sizes:
            x = frames[k:k + b]
            k += b
            if lut is not None:
                x = icrf_forward(x, lut, mode)
            mean_b = x.mean(dim=0, keepdim=True)
            w_b = torch.full_like(mean_b, float(b))
            m2_b = ((x - mean_b) ** 2).sum(dim=0, keepdim=True)
            w = w_a + w_b
            m2_a = m2_a + m2_b + (w_a * w_b / w) * (mean_b - mean_a) ** 2
            mean_a = mean_a + (w_b / w) * (mean_b - mean_a)
            w_a = w
    var = m2_a * (1 / (w_a - 1))
    return mean_a.squeeze(0), torch.sqrt(var.squeeze(0)) / (k ** 0.5)


# ---- dark field -- PARITY UNPINNED --------------------------------------------------------------
# The blur is torchvision.transforms.GaussianBlur(kernel_size=3, sigma=1.0), a third-party dependency the reference does
# not vendor (version unpinned, pyproject.toml) and this image lacks.  Restated from its published algorithm
# (torchvision/transforms/_functional_tensor.py: _get_gaussian_kernel1d -> pdf = exp(-0.5 (x / sigma)^2) on
# linspace(-1, 1, 3), normalised; kernel2d = outer product; reflect padding; depthwise conv2d).  No vector recorded from
# the reference covers it, so everything below is pinned to nothing but this restatement.
def gaussian_blur3(image, sigma=1.0):
    c = image.shape[-3]
    x = torch.linspace(-1.0, 1.0, 3, dtype=image.dtype)
    pdf = torch.exp(-0.5 * (x / sigma) ** 2)
    k1 = pdf / pdf.sum()
    k2 = torch.mm(k1[:, None], k1[None, :]).expand(c, 1, 3, 3)
    padded = torch.nn.functional.pad(image, [1, 1, 1, 1], mode="reflect")
    return torch.nn.functional.conv2d(padded, k2, groups=c)


def conditional_gaussian_blur(image, mask_map, threshold=0.05, alpha=50.0):
    """clair_torch/common/general_functions.py:440-486 with differentiable=True, kernel_size=3."""
    mask = torch.sigmoid((mask_map - threshold) * alpha)
    if mask.shape[0] == 1 and image.shape[0] > 1:
        mask = mask.expand(image.shape[0], -1, -1, -1)
    return mask * gaussian_blur3(image) + (1 - mask) * image


def merge_stack_dark(vals, stds, exposures, lut, dark, dark_std, mode=LINEAR, use_gauss=True, partition=None):
    """compute_hdr_image with a dark-field dataset (clair_torch/inference/hdr_merge.py:61-128): ``dark`` / ``dark_std``
    (C,H,W) are matched to every frame of a batch (one copy per frame, datasets/base.py:225-255).  Note the reference
    rebinds ``images`` to the blurred batch (:90), so both autograd.grad calls differentiate w.r.t. tensors of the
    blurred graph."""
    n = vals.shape[0]
    partition = [n] if partition is None else list(partition)
    mean_a, w_a, variance, k = 0.0, 0.0, None, 0
    for b in partition:
        raw = vals[k:k + b].clone().requires_grad_(True)
        sd = stds[k:k + b]
        t = exposures[k:k + b].to(torch.float64).view(-1, 1, 1, 1)
        k += b
        d = dark.unsqueeze(0).expand(b, -1, -1, -1).clone().requires_grad_(True)
        x = conditional_gaussian_blur(raw, d)                                     # :90, images rebound
        wts = gaussian_weight(x) if use_gauss else torch.ones_like(x)
        y = (icrf_forward(x, lut, mode) if lut is not None else x) / t
        w_b = wts.sum(dim=0, keepdim=True)
        m_b = (wts * y).sum(dim=0, keepdim=True) / (w_b + 1e-6)
        w_t = w_a + w_b
        mean = mean_a + (w_b / w_t) * (m_b - mean_a)
        g = torch.autograd.grad(mean, x, torch.ones_like(mean), retain_graph=True)[0]          # :107-115
        upd = ((g * sd) ** 2).sum(dim=0, keepdim=True)
        variance = upd if variance is None else variance + upd
        gd = torch.autograd.grad(mean, d, torch.ones_like(mean), retain_graph=False)[0]        # :117-126
        variance = variance + ((gd * dark_std.unsqueeze(0)) ** 2).sum(dim=0, keepdim=True)
        mean_a, w_a = mean.detach(), w_t.detach()
    return mean_a.squeeze(0), torch.sqrt(variance.squeeze(0))


def linearize_frame_dark(val, std, lut, dark, dark_std, mode=LINEAR):
    """One iteration of linearize_dataset_generator with a dark field (clair_torch/inference/linearization.py:73-116)."""
    raw = val.unsqueeze(0).clone().requires_grad_(True)
    d = dark.unsqueeze(0).clone().requires_grad_(True)
    x = conditional_gaussian_blur(raw, d)
    lin = icrf_forward(x, lut, mode)
    g = torch.autograd.grad(lin, x, torch.ones_like(lin), retain_graph=True)[0]
    var = (g * std.unsqueeze(0)) ** 2
    gd = torch.autograd.grad(lin, d, torch.ones_like(lin))[0]
    var = var + (gd * dark_std.unsqueeze(0)) ** 2
    return lin.detach().squeeze(0), torch.sqrt(var).detach().squeeze(0)
