"""Device-side input pipeline of the coco-attngan train step.

The reference augments on the host, per sample, inside DataLoader workers (code/coco/attngan/datasets.py:70-137): resize
to 268x268, ToTensor, random 256 crop + flip, ToPILImage + transforms.Resize for the 64/128 branches, Normalize -- and then
ships 1 MB of fp32 per sample to the GPU.  Here the workers stop after the JPEG decode + resize (`TextDataset(raw=True)`
yields the 268x268 u8 image and the unscaled boxes); `DeviceFeeder` draws the crop offsets / flip per sample with the
reference's rules (boxes rescaled and clamped on the host: a few floats), uploads 215 KB of u8 per sample from pinned
memory on a copy stream, and runs crop + flip + both PIL-compatible resamplings + normalisation as HIP kernels
(csrc/mogan_feed.hip).  The images are bit-identical to the host pipeline's (tests/test_feeder_cpu.py pins the coefficient
tables to Pillow, tests/test_feeder_gpu.py the kernels to datasets.crop_imgs/_multi_scale).
"""
import math

import numpy as np
import torch

PRECISION_BITS = 32 - 8 - 2          # Pillow, src/libImaging/Resample.c


def pil_bilinear_coeffs(in_size, out_size):
    """Pillow's precompute_coeffs() + normalize_coeffs_8bpc() for the BILINEAR filter (support 1.0) over the whole input
    range: bounds (out_size, 2) = (first input index, tap count), kk (out_size, ksize) int32 fixed point."""
    scale = filterscale = in_size / float(out_size)
    if filterscale < 1.0:
        filterscale = 1.0
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        k = []
        ww = 0.0
        for x in range(xmax):
            a = abs((x + xmin - center + 0.5) * ss)
            w = 1.0 - a if a < 1.0 else 0.0
            k.append(w)
            ww += w
        for x in range(xmax):
            w = k[x] / ww if ww != 0.0 else k[x]
            kk[xx, x] = int(-0.5 + w * (1 << PRECISION_BITS)) if w < 0 else int(0.5 + w * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def resample_u8_reference(img, out_size):
    """numpy restatement of Pillow's two-pass 8-bit resample with the tables above (img (S,S,3) u8) -- used by the CPU
    test that pins the tables to PIL itself."""
    S = img.shape[0]
    bounds, kk = pil_bilinear_coeffs(S, out_size)
    half = 1 << (PRECISION_BITS - 1)
    tmp = np.zeros((S, out_size, 3), dtype=np.uint8)
    a = img.astype(np.int64)
    for ox in range(out_size):
        x0, n = bounds[ox]
        acc = half + (a[:, x0:x0 + n, :] * kk[ox, :n].astype(np.int64)[None, :, None]).sum(1)
        tmp[:, ox, :] = np.clip(acc >> PRECISION_BITS, 0, 255)
    out = np.zeros((out_size, out_size, 3), dtype=np.uint8)
    t = tmp.astype(np.int64)
    for oy in range(out_size):
        y0, n = bounds[oy]
        acc = half + (t[y0:y0 + n, :, :] * kk[oy, :n].astype(np.int64)[:, None, None]).sum(0)
        out[oy] = np.clip(acc >> PRECISION_BITS, 0, 255)
    return out


def draw_crop(bbox, rng, ori_size=268, imsize=256, max_objects=3):
    """The random part of crop_imgs (datasets.py:95-137): flip, offsets, and the boxes rescaled to the crop with the
    reference's clamp -- same draws in the same order as datasets.crop_imgs."""
    flip = rng.random() < 0.5
    margin = ori_size - imsize
    h1 = int(np.floor(margin * rng.random()))
    w1 = int(np.floor(margin * rng.random()))
    out = np.full_like(np.asarray(bbox, dtype=np.float32), -1.0)
    for k in range(max_objects):
        b = bbox[k]
        if b[0] == -1:
            break
        x = max(b[0] * float(ori_size) - h1, 0) / float(imsize)
        y = max(b[1] * float(ori_size) - w1, 0) / float(imsize)
        w = min(float(ori_size) / imsize * b[2], 1.0)
        if x + w > 0.999:
            w = 1.0 - x - 0.001
        h = min(float(ori_size) / imsize * b[3], 1.0)
        if y + h > 0.999:
            h = 1.0 - y - 0.001
        if flip:
            x = 1.0 - x - w
        out[k] = [x, y, w, h]
    return (h1, w1, int(flip)), out


class DeviceFeeder:
    """u8 268x268 batches -> the train step's image list [64x64, 128x128, 256x256] (fp32, [-1,1], on the device).

    feeder = DeviceFeeder(device, batch, sizes=(64, 128, 256))
    imgs = feeder(u8_batch, params)        # u8_batch (B,268,268,3) uint8 CPU tensor, params (B,3) int32 (h1, w1, flip)
    Uploads go through pinned staging buffers on a copy stream; the kernels run on the caller's current stream behind an
    event, so batch n+1 can be uploaded while step n computes (double buffered)."""

    def __init__(self, device, batch, sizes=(64, 128, 256), ori_size=268, depth=2):
        from ..hip import lib
        self.lib = lib
        self.device = torch.device(device)
        self.B, self.sizes, self.ori = batch, tuple(sizes), ori_size
        self.S = self.sizes[-1]
        self.tables = {}
        for s in self.sizes[:-1]:
            bounds, kk = pil_bilinear_coeffs(self.S, s)
            self.tables[s] = (torch.from_numpy(bounds).to(self.device), torch.from_numpy(kk).to(self.device), kk.shape[1])
        self.copy_stream = torch.cuda.Stream(device=self.device)
        self.slots = []
        for _ in range(depth):
            self.slots.append(dict(
                pin=torch.empty((batch, ori_size, ori_size, 3), dtype=torch.uint8).pin_memory(),
                pin_par=torch.empty((batch, 3), dtype=torch.int32).pin_memory(),
                dev=torch.empty((batch, ori_size, ori_size, 3), dtype=torch.uint8, device=self.device),
                par=torch.empty((batch, 3), dtype=torch.int32, device=self.device),
                q=torch.empty((batch, self.S, self.S, 3), dtype=torch.uint8, device=self.device),
                tmp=torch.empty((batch, self.S, max(self.sizes[:-1] or (1,)), 3), dtype=torch.uint8, device=self.device),
                ready=torch.cuda.Event(), free=torch.cuda.Event()))
        self.turn = 0

    def upload(self, u8_batch, params):
        """Stage a batch (host tensors) and start its upload on the copy stream; returns the slot for `process`."""
        slot = self.slots[self.turn]
        self.turn = (self.turn + 1) % len(self.slots)
        B = u8_batch.shape[0]
        assert B <= self.B and tuple(u8_batch.shape[1:]) == (self.ori, self.ori, 3) and u8_batch.dtype == torch.uint8
        if slot.get("pending"):                         # uploaded but never processed: its copy may still be in flight
            slot["ready"].synchronize()
        slot["free"].synchronize()                      # the kernels that last read this slot's device buffers are done
        slot["pin"][:B].copy_(u8_batch)
        slot["pin_par"][:B].copy_(torch.as_tensor(params, dtype=torch.int32).reshape(B, 3))
        with torch.cuda.stream(self.copy_stream):
            slot["dev"][:B].copy_(slot["pin"][:B], non_blocking=True)
            slot["par"][:B].copy_(slot["pin_par"][:B], non_blocking=True)
            slot["ready"].record()
        slot["n"], slot["pending"] = B, True
        return slot

    def process(self, slot):
        """crop / flip / resample / normalise on the current stream -> [imgs64, imgs128, imgs256]"""
        call, sp = self.lib.call, self.lib.stream_ptr
        B = slot["n"]
        torch.cuda.current_stream().wait_event(slot["ready"])
        outs = [torch.empty((B, 3, s, s), dtype=torch.float32, device=self.device) for s in self.sizes]
        call("mogan_feed_crop_flip", slot["dev"].data_ptr(), slot["par"].data_ptr(), slot["q"].data_ptr(),
             outs[-1].data_ptr(), B, self.ori, self.S, sp())
        for i, s in enumerate(self.sizes[:-1]):
            bounds, kk, ksize = self.tables[s]
            call("mogan_feed_resample", slot["q"].data_ptr(), slot["tmp"].data_ptr(), outs[i].data_ptr(), bounds.data_ptr(),
                 kk.data_ptr(), ksize, B, self.S, s, sp())
        slot["free"].record()
        slot["pending"] = False
        return outs

    def __call__(self, u8_batch, params):
        return self.process(self.upload(u8_batch, params))
