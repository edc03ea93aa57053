"""Hand-computed RoIAlign fixture (torchvision is not in the image, so the oracle's RoIAlign is a restatement and
the reference goldens go through that same restatement -- circular for this one op).  The expectations below are NOT produced
by any implementation: they follow from two facts about torchvision.ops.roi_align(output_size=7, aligned=True,
sampling_ratio=-1) (detr_roi_head.py:44-56) that can be checked on paper:

  * sample positions: start = size * (c - extent/2) - 0.5, bin = roi_extent / 7, grid = ceil(bin) points per bin at
    start + p*bin + (i + 0.5)*bin/grid; a sample outside [-1, size] contributes 0, a sample in [-1, 0] is read at 0, a sample in
    (size-1, size] at size-1; every bin is divided by grid_h*grid_w (not by the number of valid samples);
  * bilinear interpolation reproduces a linear function exactly and turns y^2 into y^2 + t(1-t), t = frac(y).

H = 15, W = 20 (the C5 map of a 480x640 image).  Each case lists its sample positions in the comments."""
import pytest
import torch

H, W = 15, 20


def nbox(y0, rh, x0=4.0, rw=6.3):
    """normalised cxcywh box whose ALIGNED sample window starts at pixel coordinate (y0, x0) with extent (rh, rw).
    Extents are chosen so that bin = extent / 7 is NOT an integer: ceil(7.0000005 / 7) = 2 -- the grid size flips on the
    fp32 rounding of the box arithmetic when a bin is exactly 1.0 (that sensitivity is torchvision's too)."""
    h, w = rh / H, rw / W
    return [(x0 + 0.5) / W + w / 2, (y0 + 0.5) / H + h / 2, w, h]


def feature_maps():
    """channels: 0 constant 1, 1 ramp y, 2 ramp x, 3 y^2, 4 3 + 2y - 0.5x"""
    y = torch.arange(H, dtype=torch.float32).view(H, 1).expand(H, W)
    x = torch.arange(W, dtype=torch.float32).view(1, W).expand(H, W)
    return torch.stack([torch.ones(H, W), y, x, y * y, 3 + 2 * y - 0.5 * x])      # (5, H, W)


# default x window: start 4.0, extent 6.3 -> bin 0.9, grid 1: x samples 4.45, 5.35, ..., 9.85, all interior, mean 7.15
XM = 7.15
CASES = [
    # interior, grid 1 (bin 0.9): y samples 2.45, 3.35, 4.25, 5.15, 6.05, 6.95, 7.85 -> mean 5.15;
    # bilinear y^2 = y^2 + t(1-t): mean 29.9071428...
    (nbox(2.0, 6.3), [1.0, 5.15, XM, 209.35 / 7, 3 + 2 * 5.15 - 0.5 * XM]),
    # interior, grid 2 (extent 7.7, bin 1.1): y samples 2 + 1.1p + {0.275, 0.825}: mean 5.85; y^2 channel 39.3107142...
    (nbox(2.0, 7.7), [1.0, 5.85, XM, 550.35 / 14, 3 + 2 * 5.85 - 0.5 * XM]),
    # grid 1 in y (extent 3.5, bin 0.5: samples 3.25 .. 6.25, mean 4.75, y^2 channel 23.75), grid 2 in x (extent 10.5, bin 1.5:
    # x samples 3 + 1.5p + {0.375, 1.125}, mean 8.25)
    (nbox(3.0, 3.5, 3.0, 10.5), [1.0, 4.75, 8.25, 23.75, 3 + 2 * 4.75 - 0.5 * 8.25]),
    # top border (start -2.0, bin 0.9): y samples -1.55 (outside [-1, H]: contributes 0), -0.65 (read at 0), 0.25, 1.15, 2.05, 2.95, 3.85
    # -> 6 of 7 valid; sum of read positions 10.25; sum of bilinear y^2 29.65
    (nbox(-2.0, 6.3), [6.0 / 7, 10.25 / 7, 6.0 / 7 * XM, 29.65 / 7, (6 * 3 + 2 * 10.25 - 0.5 * 6 * XM) / 7]),
    # bottom border (start 9.9, bin 0.9): y samples 10.35, 11.25, 12.15, 13.05, 13.95, 14.85 (floor = H-1: read at 14), 15.75 (> H: outside)
    # -> 6 of 7 valid; sum of read positions 74.75; sum of bilinear y^2 942.85
    (nbox(9.9, 6.3), [6.0 / 7, 74.75 / 7, 6.0 / 7 * XM, 942.85 / 7, (6 * 3 + 2 * 74.75 - 0.5 * 6 * XM) / 7]),
    # right border (x start 15.4, bin 0.9): x samples 15.85, 16.75, 17.65, 18.55, 19.45 (read at 19), 20.35 and 21.25 (> W: outside)
    # -> 5 of 7 valid, sum of read positions 87.8; y interior as in the first case
    (nbox(2.0, 6.3, 15.4, 6.3), [5.0 / 7, 5.0 / 7 * 5.15, 87.8 / 7, 5.0 / 7 * 209.35 / 7, (5 * 3 + 2 * 5 * 5.15 - 0.5 * 87.8) / 7]),
    # degenerate box (w = h = 0): grid = ceil(0 / 7) = 0 -> no samples -> 0
    ([0.5, 0.5, 0.0, 0.0], [0.0, 0.0, 0.0, 0.0, 0.0]),
]


def expected():
    return torch.tensor([e for _, e in CASES], dtype=torch.float32)


def boxes():
    return torch.tensor([b for b, _ in CASES], dtype=torch.float32)


def test_oracle_roi_align_matches_hand_computed_values():
    from oracle import gpv_oracle as O
    feat = feature_maps()
    got = O.extract_roi(feat[None], boxes()[None])[0]                     # (N, C) separable-weights form
    assert torch.allclose(got, expected(), rtol=1e-5, atol=1e-5), (got - expected()).abs().max()
    bx = boxes()
    xyxy = torch.stack([W * (bx[:, 0] - bx[:, 2] / 2), H * (bx[:, 1] - bx[:, 3] / 2), W * (bx[:, 0] + bx[:, 2] / 2), H * (bx[:, 1] + bx[:, 3] / 2)], 1)
    direct = O.roi_align_mean_direct(feat, xyxy)                          # literal per-sample form
    assert torch.allclose(direct, expected(), rtol=1e-5, atol=1e-5), (direct - expected()).abs().max()


@pytest.mark.gpu
@pytest.mark.parametrize('precise', [True, False])
def test_hip_roi_pool_matches_hand_computed_values(precise):
    import gpv1_amd.ops as ops
    ops.RT.set_precise(precise)
    try:
        feat = feature_maps().permute(1, 2, 0).reshape(1, H * W, 5)      # NHWC rows
        C = 8
        f = torch.zeros(2, H * W, C)
        f[:, :, :5] = feat
        got = ops.roi_pool(f.cuda().to(ops.RT.dtype), boxes()[None].expand(2, -1, -1).contiguous().cuda(), H, W).float().cpu()
        exp = expected()
        tol = 1e-5 if precise else 1.5e-2                                 # bf16: weights and features rounded to 8 bits
        for b in range(2):
            err = (got[b, :, :5] - exp).abs() / exp.abs().clamp(min=1.0)
            assert err.max() < tol, (precise, err.max())
            assert got[b, :, 5:].abs().max() == 0
    finally:
        ops.RT.set_precise(False)


# ---- bins that are exact integers, +- one ulp of the box size ---------------------------------------------
# grid = ceil(extent / 7) flips between g and g + 1 when extent / 7 is an integer and the fp32 box arithmetic lands one ulp above
# it (7/15 and 7/20 are not representable, so a "7-pixel" box is always such a case).  What must hold WHICHEVER way it falls:
#   * linear channels: a bin's mean of a linear function is its midpoint value for any grid -> one hand value;
#   * the y^2 channel: one of the two hand values (grid g or g + 1): sum over samples of y^2 + t(1 - t).
def _y2_mean(y0, extent, grid):
    """float64 mean over the 7 x grid interior samples of the bilinear interpolant of y^2"""
    b = extent / 7.0
    tot = 0.0
    for p in range(7):
        for i in range(grid):
            y = y0 + p * b + (i + 0.5) * b / grid
            t = y - int(y)
            tot += y * y + t * (1.0 - t)
    return tot / (7 * grid)


def _ulp_boxes():
    import numpy as np
    out = []
    for y0, ext in ((2.0, 7.0), (-0.25, 14.0), (3.25, 7.0)):       # (every sample stays inside [0, H - 1]: no border clamping)
        h = np.float32(ext / H)
        for k in (-1, 0, 1):
            hk = float(np.nextafter(h, np.float32(2.0 * k)) if k else h)              # one ulp below / the rounded value / one ulp above
            b = nbox(y0, ext)
            b[3] = hk
            g0 = int(round(ext / 7.0))
            ymid = y0 + ext / 2.0
            out.append((b, ymid, (_y2_mean(y0, ext, g0), _y2_mean(y0, ext, g0 + 1))))
    return out


def _check_ulp_rows(got, tol):
    for row, (b, ymid, (c0, c1)) in zip(got, _ulp_boxes()):
        lin = torch.tensor([1.0, ymid, XM, 3 + 2 * ymid - 0.5 * XM])
        assert (row[[0, 1, 2, 4]] - lin).abs().max() <= tol * lin.abs().max(), (b, row, lin)
        assert min(abs(float(row[3]) - c0), abs(float(row[3]) - c1)) <= tol * c0, (b, float(row[3]), c0, c1)


def test_oracle_roi_align_at_integer_bins_plus_minus_one_ulp():
    from oracle import gpv_oracle as O
    feat = feature_maps()
    bx = torch.tensor([b for b, _, _ in _ulp_boxes()], dtype=torch.float32)
    got = O.extract_roi(feat[None], bx[None])[0]
    _check_ulp_rows(got, 2e-5)
    xyxy = torch.stack([W * (bx[:, 0] - bx[:, 2] / 2), H * (bx[:, 1] - bx[:, 3] / 2), W * (bx[:, 0] + bx[:, 2] / 2), H * (bx[:, 1] + bx[:, 3] / 2)], 1)
    _check_ulp_rows(O.roi_align_mean_direct(feat, xyxy), 2e-5)


@pytest.mark.gpu
def test_hip_roi_pool_at_integer_bins_plus_minus_one_ulp():
    """the HIP kernel on the same boxes: one of the two hand values, and the SAME one the oracle's separable form picks (both
    derive the grid from the box in the same fp32 order: start = size * (c - e / 2) - 0.5, length = size * e)"""
    import gpv1_amd.ops as ops
    from oracle import gpv_oracle as O
    ops.RT.set_precise(True)
    try:
        feat = feature_maps()
        f = torch.zeros(1, H * W, 8)
        f[0, :, :5] = feat.permute(1, 2, 0).reshape(H * W, 5)
        bx = torch.tensor([b for b, _, _ in _ulp_boxes()], dtype=torch.float32)
        got = ops.roi_pool(f.cuda(), bx[None].contiguous().cuda(), H, W).float().cpu()[0, :, :5]
        _check_ulp_rows(got, 2e-5)
        ref = O.extract_roi(feat[None], bx[None])[0]
        assert (got - ref).abs().max() <= 2e-5 * ref.abs().max(), (got - ref).abs().max()
    finally:
        ops.RT.set_precise(False)
