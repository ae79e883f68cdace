"""Device-side image preprocessing: the reference does this on CPU DataLoader workers
(exp/cxr_pt/inference/dataset.py:31-51: cv2 min-max to 8 bit; then the Blip image processor: RGB, PIL bicubic
resize of the uint8 image, rescale, normalise — processing.py:31-49, :90-91), which becomes the bottleneck once the
encoder runs at hundreds of images/s.

Host side here: Pillow's resampling coefficient tables (ImagingResample `precompute_coeffs` + `normalize_coeffs_8bpc`,
bicubic a = -0.5, 22-bit fixed point) restated in numpy, cached per (in, out) size; the byte work runs in
csrc/preprocess.hip through `rz_preprocess_image`.  The image-processor constants (mean/std/size) live in the
checkpoint's preprocessor_config.json, which is unreachable offline: they are parameters (defaults = BlipImageProcessor's:
OPENAI_CLIP mean/std, rescale 1/255).
"""
from __future__ import annotations

import ctypes
import math
from functools import lru_cache

import numpy as np
import torch

from . import _lib

OPENAI_CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
OPENAI_CLIP_STD = (0.26862954, 0.26130258, 0.27577711)
PRECISION_BITS = 32 - 8 - 2


def _bicubic(x: float) -> float:
    a = -0.5
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1.0
    if x < 2.0:
        return (((x - 5.0) * x + 8.0) * x - 4.0) * a
    return 0.0


@lru_cache(maxsize=64)
def resample_tables(in_size: int, out_size: int):
    """Pillow `precompute_coeffs` (src/libImaging/Resample.c) for the bicubic filter over the whole axis, then
    `normalize_coeffs_8bpc`.  Returns (bounds int32 (out, 2) = (xmin, count), kk int32 (out, ksize), ksize)."""
    support_base = 2.0
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = support_base * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        xmin = max(xmin, 0)
        xmax = int(center + support + 0.5)
        xmax = min(xmax, in_size)
        cnt = xmax - xmin
        w = [_bicubic((x + xmin - center + 0.5) * ss) for x in range(cnt)]
        ww = sum(w)
        if ww != 0.0:
            w = [v / ww for v in w]
        for x, v in enumerate(w):
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, cnt)
    return bounds, kk, ksize


class DevicePreprocessor:
    """`collate_fn` (dataset.py:31-51) + the image processor's `__call__` for a BATCH of raw images, on the GPU:
    cv2 min-max to 8 bit -> [keep_aspect_ratio: pad to a black square, AspectRatioBlipImageProcessor.pad_to_square processing.py:247-259]
    -> Pillow-exact bicubic resize of the 8-bit image -> rescale -> normalise -> fp32 (B, 3, S, S).  One set of launches per batch."""

    def __init__(self, size: int, image_mean=OPENAI_CLIP_MEAN, image_std=OPENAI_CLIP_STD, rescale_factor: float = 1.0 / 255.0,
                 minmax_normalize: bool = True, keep_aspect_ratio: bool = False, device="cuda:0"):
        self.size = int(size)
        self.mean = (ctypes.c_float * 3)(*image_mean)
        self.std = (ctypes.c_float * 3)(*image_std)
        self.rescale = float(rescale_factor)
        self.minmax = bool(minmax_normalize)
        self.keep_aspect_ratio = bool(keep_aspect_ratio)
        self.device = torch.device(device)
        self._lib = _lib.load()
        self._tables = {}
        self._ws = None

    def _dev_tables(self, in_size):
        key = (in_size, self.size)
        if key not in self._tables:
            b, k, ks = resample_tables(in_size, self.size)
            self._tables[key] = (torch.from_numpy(b).to(self.device), torch.from_numpy(k).to(self.device), ks)
        return self._tables[key]

    def _canonical(self, image: torch.Tensor):
        img = image.to(self.device, non_blocking=True)
        if img.dim() == 2:
            img = img.unsqueeze(-1)
        if img.dim() != 3 or img.shape[2] not in (1, 3):
            raise ValueError("image must be (H, W) or (H, W, 1|3)")
        if img.dtype == torch.uint8:
            code = 0
        elif img.dtype == torch.uint16:
            code = 1
        elif img.dtype in (torch.int16, torch.int32):
            img, code = img.to(torch.int32).clamp_(0, 65535).to(torch.uint16), 1
        else:
            img, code = img.float(), 2
        return img.contiguous(), code

    @torch.no_grad()
    def __call__(self, images, out: torch.Tensor | None = None) -> torch.Tensor:
        """images: ONE tensor (H, W) / (H, W, C) or a list of them (sizes and dtypes may differ): uint8 / uint16 (int16, int32
        accepted as the uint16 range) / float32 -> (B, 3, S, S) fp32 on the device (`out`, if given, receives it)."""
        single = torch.is_tensor(images)
        items = [self._canonical(im) for im in ([images] if single else list(images))]
        n = len(items)
        if n == 0:
            raise ValueError("empty image batch")
        s = self.size
        descs = (_lib.RzImageDesc * n)()
        for d, (img, code) in zip(descs, items):
            h, w, c = img.shape
            side = max(h, w) if self.keep_aspect_ratio else 0
            ph, pw = (side, side) if self.keep_aspect_ratio else (h, w)
            bh, kh, ksh = self._dev_tables(pw)
            bv, kv, ksv = self._dev_tables(ph)
            d.image_dev, d.src_dtype, d.height, d.width, d.channels = img.data_ptr(), code, h, w, c
            d.pad_left, d.pad_top = ((side - w) // 2, (side - h) // 2) if self.keep_aspect_ratio else (0, 0)
            d.padded_height, d.padded_width = ph, pw
            d.bounds_h_dev, d.coeffs_h_dev, d.ksize_h = bh.data_ptr(), kh.data_ptr(), ksh
            d.bounds_v_dev, d.coeffs_v_dev, d.ksize_v = bv.data_ptr(), kv.data_ptr(), ksv
        need = int(self._lib.rz_preprocess_batch_workspace(descs, n, s))
        if need == 0:
            raise ValueError("bad image batch")
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        if out is None:
            out = torch.empty((n, 3, s, s), dtype=torch.float32, device=self.device)
        elif tuple(out.shape) != (n, 3, s, s) or out.dtype != torch.float32 or not out.is_contiguous():
            raise ValueError("out must be a contiguous fp32 (B, 3, S, S) tensor")
        P = lambda t: ctypes.c_void_p(t.data_ptr())
        with torch.cuda.device(self.device):
            st = ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
            _lib.check(self._lib.rz_preprocess_batch(descs, n, s, self.mean, self.std, self.rescale, int(self.minmax), P(self._ws),
                                                     self._ws.numel(), P(out), st), "rz_preprocess_batch")
        # the raw tensors must outlive the launches on this stream: torch's caching allocator only recycles them for later work
        # of the same stream, so dropping `items` here is safe
        return out

    @torch.no_grad()
    def single_image_entry(self, image: torch.Tensor) -> torch.Tensor:
        """The one-image C entry point rz_preprocess_image (round 2's path; kept as the A/B of the batched kernels)."""
        if self.keep_aspect_ratio:
            raise NotImplementedError("rz_preprocess_image has no padding branch: use the batched call")
        img, code = self._canonical(image)
        h, w, c = img.shape
        bh, kh, ksh = self._dev_tables(w)
        bv, kv, ksv = self._dev_tables(h)
        s = self.size
        ws = torch.empty(16 + h * w * c + h * s * c + s * s * c, dtype=torch.uint8, device=self.device)
        out = torch.empty((1, 3, s, s), dtype=torch.float32, device=self.device)
        P = lambda t: ctypes.c_void_p(t.data_ptr())
        with torch.cuda.device(self.device):
            st = ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
            _lib.check(self._lib.rz_preprocess_image(P(img), code, h, w, c, s, P(bh), P(kh), ksh, P(bv), P(kv), ksv, self.mean, self.std,
                                                     self.rescale, int(self.minmax), P(ws), P(out), st), "rz_preprocess_image")
        return out
