"""CPU restatement of the reference's predict-time input transforms (TEST INFRASTRUCTURE ONLY: imported by tests/, never by the
product path).  numpy, fp32, one function per reference step:

  image_read                 utils/io/images.py:14-29          np.float32(img) / 255.0
  T.Resize(short, BILINEAR, antialias=True)   task/predict.py:87-93, dataloading/dataset/nvs_dataset.py:218-225
      -> torchvision computes (short, int(short * long / short_side)) and calls
         torch.nn.functional.interpolate(mode="bilinear", align_corners=False, antialias=True); the arithmetic restated here is
         ATen/native/cpu/UpSampleKernel.cpp (_compute_indices_min_size_weights_aa, separable: width pass, then height pass).
         torchvision is not vendored in the reference (environment.yaml pins 0.16.2) and absent here: PINNED against
         torch 2.10 F.interpolate outputs (tests/golden/make_golden_preprocess.py -> tests/golden/p*.npz).
  deterministic crop         dataloading/transformation/crop.py:8-25 (i = j = 0), nvs_dataset.py:227-241 (integer patches)
  T.Normalize(mean, std)     task/predict.py:68-74              (x - mean) / std
"""
from __future__ import annotations

import numpy as np

IMAGENET_MEAN = (0.485, 0.456, 0.406)  # utils/io/images.py:8-11
IMAGENET_STD = (0.229, 0.224, 0.225)


def resized_output_size(h: int, w: int, short: int):
    """torchvision.transforms.v2.functional._geometry._compute_resized_output_size for an int size (no max_size)."""
    if h <= w:
        return short, int(short * w / h)
    return int(short * h / w), short


def aa_axis_table(n_in: int, n_out: int):
    """(xmin[n_out], xsize[n_out], weights[n_out][taps]) of the antialiased triangle filter along one axis."""
    scale = np.float32(n_in) / np.float32(n_out)
    support = scale if scale >= 1.0 else np.float32(1.0)
    taps = int(np.ceil(support)) * 2 + 1
    invscale = np.float32(1.0) / scale if scale >= 1.0 else np.float32(1.0)
    xmin = np.zeros(n_out, np.int64)
    xsize = np.zeros(n_out, np.int64)
    w = np.zeros((n_out, taps), np.float32)
    for i in range(n_out):
        center = np.float32(np.float64(scale) * (i + 0.5))
        lo = max(int(np.float64(center) - np.float64(support) + 0.5), 0)
        hi = min(int(np.float64(center) + np.float64(support) + 0.5), n_in)
        n = hi - lo
        x = np.float32((np.arange(n, dtype=np.float64) + lo - np.float64(center) + 0.5) * np.float64(invscale))
        ww = np.maximum(np.float32(1.0) - np.abs(x), np.float32(0.0)).astype(np.float32)
        tot = np.float32(0.0)
        for v in ww:
            tot = np.float32(tot + v)
        if tot != 0:
            ww = (ww / tot).astype(np.float32)
        xmin[i], xsize[i] = lo, n
        w[i, :n] = ww
    return xmin, xsize, w


def _resize_axis(x: np.ndarray, n_out: int, axis: int) -> np.ndarray:
    n_in = x.shape[axis]
    xmin, xsize, w = aa_axis_table(n_in, n_out)
    x = np.moveaxis(x, axis, -1)
    out = np.zeros(x.shape[:-1] + (n_out,), np.float32)
    for i in range(n_out):
        acc = x[..., xmin[i]] * w[i, 0]
        for j in range(1, xsize[i]):
            acc = (acc + x[..., xmin[i] + j] * w[i, j]).astype(np.float32)
        out[..., i] = acc
    return np.moveaxis(out, -1, axis)


def resize_bilinear_aa(img_chw: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """F.interpolate(img[None], (out_h, out_w), mode="bilinear", align_corners=False, antialias=True)[0] on fp32 CHW."""
    x = np.asarray(img_chw, np.float32)
    if x.shape[-1] != out_w:
        x = _resize_axis(x, out_w, -1)
    if x.shape[-2] != out_h:
        x = _resize_axis(x, out_h, -2)
    return x


def preprocess_u8(img_hwc_u8: np.ndarray, rs_hw=None, crop=None, mean=IMAGENET_MEAN, std=IMAGENET_STD) -> np.ndarray:
    """uint8 HWC -> normalised fp32 CHW.  rs_hw: resized (h, w) or None; crop: (y, x, h, w) in the resized image or None."""
    x = (img_hwc_u8.astype(np.float32) / np.float32(255.0)).transpose(2, 0, 1)
    if rs_hw is not None and tuple(rs_hw) != x.shape[1:]:
        x = resize_bilinear_aa(x, int(rs_hw[0]), int(rs_hw[1]))
    if crop is not None:
        y0, x0, hh, ww = crop
        x = x[:, y0:y0 + hh, x0:x0 + ww]
    m = np.asarray(mean, np.float32)[:, None, None]
    s = np.asarray(std, np.float32)[:, None, None]
    return ((x - m) / s).astype(np.float32)
