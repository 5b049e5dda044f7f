"""Image input pipeline on the device.

``DeviceImagePipeline(split, resize, crop)`` is the drop-in for the transform the reference builds in
``get_transforms`` (ref: vilmedic/datasets/base/ImageDataset.py:80-108) when the dataset hands over DECODED uint8 HWC
images (numpy arrays / torch uint8 tensors) instead of PIL images: the whole batch goes to HBM as bytes (3 B/pixel instead
of 12 B/pixel of normalised fp32) and one HIP kernel does Resize -> RandomCrop -> RandomHorizontalFlip -> ToTensor ->
Normalize, bit-exact with Pillow's resampler.  The random crop / flip draws follow torchvision's call order
(``RandomCrop.get_params``: two ``torch.randint`` unless the image already has the crop size; ``RandomHorizontalFlip``:
``torch.rand(1) < 0.5``), so a seeded run draws what the reference's transforms would draw."""
import ctypes as C
import math

import numpy as np
import torch

from .._lib import check, lib, ptr, stream

MEAN = (0.485, 0.456, 0.406)
STD = (0.229, 0.224, 0.225)


def resized_hw(h, w, resize):
    return (resize, int(resize * w / h)) if h <= w else (int(resize * h / w), resize)


def packed_layout(images):
    """byte offset of each image in the packed batch buffer (16-byte aligned), their (H, W), and the buffer's length"""
    sizes = [(int(im.shape[0]), int(im.shape[1])) for im in images]
    offs = np.zeros(len(images), dtype=np.int64)
    total = 0
    for b, (h, w) in enumerate(sizes):
        offs[b] = total
        total += (h * w * 3 + 15) // 16 * 16
    return offs, sizes, total


def pack_host(images, buffer):
    """the host half of ``DeviceImagePipeline.pack``: the images laid out in ONE byte buffer obtained from ``buffer(nbytes)`` (a
    pinned staging buffer, datasets/prefetch.py), by plain memcpys -- the batch then crosses PCIe as one asynchronous copy
    instead of one blocking pageable copy per image."""
    offs, sizes, total = packed_layout(images)
    buf = buffer(total)
    dst = buf.numpy()
    for b, im in enumerate(images):
        a = im if isinstance(im, np.ndarray) else im.numpy()
        if a.dtype != np.uint8 or a.ndim != 3 or a.shape[2] != 3:
            raise ValueError("DeviceImagePipeline takes decoded uint8 [H,W,3] images")
        np.copyto(dst[offs[b]:offs[b] + a.size].reshape(a.shape), a)
    return buf[:total]


class DeviceImagePipeline:
    def __init__(self, split, resize, crop, mean=MEAN, std=STD, flip_p=0.5, device="cuda", generator=None):
        self.train = split == "train"
        self.resize, self.crop = int(resize), int(crop)
        self.mean = (C.c_float * 3)(*mean)
        self.std = (C.c_float * 3)(*std)
        self.flip_p = flip_p
        self.device = torch.device(device)
        self.generator = generator
        self._ws = None

    def draw(self, sizes):
        """per image (top, left, flip) exactly as torchvision's RandomCrop + RandomHorizontalFlip would draw them"""
        tl = np.zeros((len(sizes), 2), dtype=np.int32)
        flip = np.zeros(len(sizes), dtype=np.uint8)
        if not self.train:
            return tl, flip
        g = self.generator
        for b, (h, w) in enumerate(sizes):
            nh, nw = resized_hw(h, w, self.resize)
            if nh < self.crop or nw < self.crop:
                raise ValueError(f"Required crop size {(self.crop, self.crop)} is larger than input image size {(nh, nw)}")
            if not (nw == self.crop and nh == self.crop):
                tl[b, 0] = int(torch.randint(0, nh - self.crop + 1, size=(1,), generator=g).item())
                tl[b, 1] = int(torch.randint(0, nw - self.crop + 1, size=(1,), generator=g).item())
            flip[b] = 1 if float(torch.rand(1, generator=g)) < self.flip_p else 0
        return tl, flip

    def pack(self, images):
        """list of uint8 [H,W,3] images (numpy / torch, host or device) -> (device byte buffer, offsets, sizes)"""
        offs, sizes, total = packed_layout(images)
        packed = torch.empty(total, dtype=torch.uint8, device=self.device)
        for b, im in enumerate(images):
            t = torch.as_tensor(np.ascontiguousarray(im) if isinstance(im, np.ndarray) else im.contiguous())
            if t.dtype != torch.uint8 or t.dim() != 3 or t.shape[2] != 3:
                raise ValueError("DeviceImagePipeline takes decoded uint8 [H,W,3] images")
            packed[offs[b]:offs[b] + t.numel()].copy_(t.reshape(-1), non_blocking=True)
        return packed, offs, sizes

    def run(self, packed, offs, sizes, top_left=None, flip=None):
        """the kernel on an already packed batch -> fp32 [B,3,crop,crop]"""
        B = len(sizes)
        if top_left is None:
            top_left, flip = self.draw(sizes)
        top_left = np.ascontiguousarray(top_left, dtype=np.int32)
        flip = np.ascontiguousarray(flip if flip is not None else np.zeros(B), dtype=np.uint8)
        hw = np.asarray(sizes, dtype=np.int32).reshape(B, 2)
        scale, max_out = 1.0, self.crop
        for h, w in sizes:
            nh, nw = resized_hw(h, w, self.resize) if self.train else (self.crop, self.crop)
            scale = max(scale, h / nh, w / nw)
            max_out = max(max_out, nh, nw)
        max_taps = int(math.ceil(scale)) * 2 + 1
        need = lib().vm_image_pipeline_ws(len(set(sizes)), max_out, max_taps) + B * 64
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        out = torch.empty(B, 3, self.crop, self.crop, dtype=torch.float32, device=self.device)
        check(lib().vm_image_pipeline_u8(ptr(packed), offs.ctypes.data_as(C.c_void_p), hw.ctypes.data_as(C.c_void_p), B,
                                         self.resize if self.train else 0, self.crop, top_left.ctypes.data_as(C.c_void_p),
                                         flip.ctypes.data_as(C.c_void_p), C.cast(self.mean, C.c_void_p), C.cast(self.std, C.c_void_p),
                                         ptr(out), max_taps, ptr(self._ws), self._ws.numel(), stream()), "vm_image_pipeline_u8")
        return out

    def __call__(self, images, top_left=None, flip=None):
        """images: list of uint8 [H,W,3] numpy arrays / torch tensors (host or device) -> fp32 [B,3,crop,crop] on the device"""
        packed, offs, sizes = self.pack(images)
        return self.run(packed, offs, sizes, top_left, flip)
