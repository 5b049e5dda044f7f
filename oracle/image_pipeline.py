"""oracle/image_pipeline.py -- TEST INFRASTRUCTURE ONLY (CPU restatement; never imported by the product path).

The reference's image input pipeline (ref:vilmedic/datasets/base/ImageDataset.py:80-108):
    train:  Resize(resize) -> RandomCrop(crop) -> RandomHorizontalFlip() -> ToTensor() -> Normalize(mean, std)
    eval:   Resize((crop, crop)) -> ToTensor() -> Normalize(mean, std)
torchvision (not installed here) implements Resize on PIL images as ``img.resize(size, BILINEAR)``; the arithmetic
that matters is therefore Pillow's 8-bit resampler (Pillow 12.2.0, src/libImaging/Resample.c: precompute_coeffs,
normalize_coeffs_8bpc, ImagingResampleHorizontal_8bpc / Vertical_8bpc), restated below in numpy:
two passes (horizontal, then vertical), triangle filter widened by the scale (antialias), coefficients rounded to
22-bit fixed point, every pass rounded and clipped to uint8.
PARITY PINNED: tests/test_image_pipeline.py checks this restatement against Pillow itself (bit-exact) on random sizes.
"""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2
MEAN = (0.485, 0.456, 0.406)
STD = (0.229, 0.224, 0.225)


def _bilinear(x):
    x = -x if x < 0.0 else x
    return 1.0 - x if x < 1.0 else 0.0


def precompute_coeffs(in_size, out_size):
    """Resample.c precompute_coeffs + normalize_coeffs_8bpc for the full-image box -> (bounds [out,2], kk int [out,ksize])"""
    scale = float(in_size) / out_size
    filterscale = scale if scale >= 1.0 else 1.0
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        xmin = max(xmin, 0)
        xmax = int(center + support + 0.5)
        xmax = min(xmax, in_size)
        xmax -= xmin
        k = [_bilinear((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for w in k:
            ww += w
        for x in range(xmax):
            v = k[x] / ww if ww != 0.0 else k[x]
            kk[xx, x] = int(v * (1 << PRECISION_BITS) - 0.5) if v < 0 else int(v * (1 << PRECISION_BITS) + 0.5)
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _pass(img, bounds, kk, axis):
    """one 8-bit resampling pass along ``axis`` (0 = vertical, 1 = horizontal) of an HWC uint8 image"""
    src = img.astype(np.int64)
    out_size = bounds.shape[0]
    shape = list(img.shape)
    shape[axis] = out_size
    out = np.zeros(shape, dtype=np.uint8)
    for xx in range(out_size):
        xmin, n = bounds[xx]
        acc = np.full(np.take(src, 0, axis=axis).shape, 1 << (PRECISION_BITS - 1), dtype=np.int64)
        for x in range(n):
            acc = acc + np.take(src, xmin + x, axis=axis) * int(kk[xx, x])
        v = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
        if axis == 0:
            out[xx] = v
        else:
            out[:, xx] = v
    return out


def pil_bilinear_resize_u8(img, out_h, out_w):
    """== PIL.Image.fromarray(img).resize((out_w, out_h), Image.BILINEAR) for an HWC uint8 array"""
    h, w = img.shape[:2]
    if out_w != w:
        img = _pass(img, *precompute_coeffs(w, out_w), axis=1)
    if out_h != h:
        img = _pass(img, *precompute_coeffs(h, out_h), axis=0)
    return img


def resized_hw(h, w, resize):
    """torchvision.transforms.Resize(int): shorter side -> resize, longer side -> int(resize * long / short)"""
    if h <= w:
        return resize, int(resize * w / h)
    return int(resize * h / w), resize


def preprocess(img, *, resize, crop, top=0, left=0, flip=False, mean=MEAN, std=STD):
    """img uint8 [H,W,3] -> float32 [3,crop,crop].  resize > 0: the train chain (Resize(resize), crop window at (top, left)
    of the resized image, optional horizontal flip); resize == 0: the eval chain (Resize((crop, crop)))."""
    h, w = img.shape[:2]
    if resize > 0:
        nh, nw = resized_hw(h, w, resize)
        r = pil_bilinear_resize_u8(img, nh, nw)[top:top + crop, left:left + crop]
    else:
        r = pil_bilinear_resize_u8(img, crop, crop)
    if flip:
        r = r[:, ::-1]
    x = r.astype(np.float32) / np.float32(255.0)                      # ToTensor
    x = (x - np.asarray(mean, np.float32)) / np.asarray(std, np.float32)   # Normalize
    return np.ascontiguousarray(x.transpose(2, 0, 1))
