"""Input side of the training step on the GPU, and the reference's checkpoint files.

    ImageBatcher        decoded 8-bit images -> the normalised float batch, one launch per batch
                        (replaces cv2.resize / crop / flip / float / Normalize of BasicGANs/utils.py:39-58,
                        Pix2Pix/dataset.py:225-318, StyleGAN/train.py:22-50 after the file has been decoded)
    save_checkpoint, save_weight, load_checkpoint      BasicGANs/utils.py:5-16: the same dictionary, the same keys

Decoding the image files (cv2.imread in the reference) stays with the caller: the batcher takes uint8 arrays or tensors of
shape [H, W, C] in the file's channel order."""
import ctypes

import numpy as np
import torch

from . import _lib

_JOB = np.dtype([("src", "<u8"), ("src_h", "<i4"), ("src_w", "<i4"), ("src_stride", "<i4"), ("resize_h", "<i4"), ("resize_w", "<i4"),
                 ("crop_y", "<i4"), ("crop_x", "<i4"), ("flip", "<i4")], align=True)
assert _JOB.itemsize == 40      # gc_image_job


class ImageBatcher:
    """size: (width, height) of the network input, as the reference's `image_size`.  jitter: None, or the factor by which
    the image is resized before a random `size` window is cut out of it (Pix2Pix/dataset.py:262, 286 / 256); flip: mirror a
    sample left-right with probability 1/2 (the reference's "flip" augmentation drawn against "noaug", :276-279).
    mean / std: per-channel constants of transforms.Normalize.  linear=False, divisor=1: label masks (INTER_NEAREST, raw
    values, :241, :313).  channels_last: memory format of the returned [N, C, H, W] tensor.

    With `jitter` the reference resizes twice, each time to 8 bits: Resize(SIZE) (Pix2Pix/train.py:49-52, dataset.py:232) and
    then the jitter's own resize (:285).  This class resizes once, from the source straight to int(size * jitter): the same
    pixels when the source already has `size` (then the reference's first resize is the identity), slightly different ones
    otherwise.  A source of another size is therefore refused unless single_resize=True says that one resize is wanted."""

    def __init__(self, size, mean=None, std=None, jitter=None, flip=False, linear=True, divisor=255.0, dtype=torch.float32,
                 channels_last=True, device="cuda", single_resize=False):
        self.single_resize = bool(single_resize)
        self.out_w, self.out_h = int(size[0]), int(size[1])
        self.mean = None if mean is None else (ctypes.c_float * len(mean))(*mean)
        self.std = None if std is None else (ctypes.c_float * len(std))(*std)
        self.jitter, self.flip, self.linear, self.divisor = jitter, bool(flip), bool(linear), float(divisor)
        self.dtype, self.channels_last, self.device = dtype, channels_last, torch.device(device)

    def plan(self, shapes, rng=None):
        """Per-sample (resize_h, resize_w, crop_y, crop_x, flip) drawn as the reference draws them: the resized size is
        int(size * jitter), the window start uniform in [0, max) (0 when there is no room), the flip a fair coin."""
        rng = np.random if rng is None else rng
        plans = []
        for _ in shapes:
            if self.jitter is None:
                rh, rw, y, x = self.out_h, self.out_w, 0, 0
            else:
                rh, rw = int(self.out_h * self.jitter), int(self.out_w * self.jitter)
                max_y, max_x = rh - self.out_h, rw - self.out_w
                x = int(rng.randint(0, max_x)) if (max_x > 0 and max_y > 0) else 0
                y = int(rng.randint(0, max_y)) if (max_x > 0 and max_y > 0) else 0
            plans.append((rh, rw, y, x, int(self.flip and rng.randint(0, 2) == 1)))
        return plans

    def __call__(self, images, plans=None, rng=None):
        """images: a list of uint8 [H, W, C] arrays / tensors (any sizes, one channel count).  Returns [N, C, out_h, out_w]."""
        tensors = []
        for im in images:
            t = torch.as_tensor(im)
            if t.dtype != torch.uint8 or t.dim() != 3 or t.shape[2] > 4:
                raise RuntimeError(f"ganconv: images must be uint8 [H, W, C <= 4], got {tuple(t.shape)} {t.dtype}")
            tensors.append(t.to(self.device, non_blocking=True).contiguous())
        n = len(tensors)
        channels = tensors[0].shape[2] if n else 3
        if any(t.shape[2] != channels for t in tensors):
            raise RuntimeError("ganconv: the images of a batch must have the same number of channels")
        if self.jitter is not None and not self.single_resize:
            for t in tensors:
                if (t.shape[0], t.shape[1]) != (self.out_h, self.out_w):
                    raise RuntimeError(f"ganconv: with jitter the reference resizes a {t.shape[1]}x{t.shape[0]} source to {self.out_w}x{self.out_h} first "
                                       "and then to the jitter size (two 8-bit passes); this class resizes once: pass sources of the network's "
                                       "input size, or single_resize=True")
        plans = self.plan([t.shape for t in tensors], rng) if plans is None else plans
        jobs = np.zeros(n, _JOB)
        for i, (t, (rh, rw, y, x, flip)) in enumerate(zip(tensors, plans)):
            if not (0 <= y and y + self.out_h <= rh and 0 <= x and x + self.out_w <= rw):
                raise RuntimeError(f"ganconv: window ({y}, {x}) + ({self.out_h}, {self.out_w}) leaves the resized image ({rh}, {rw})")
            jobs[i] = (t.data_ptr(), t.shape[0], t.shape[1], t.shape[1] * channels, rh, rw, y, x, flip)
        out = torch.empty((n, channels, self.out_h, self.out_w), dtype=self.dtype, device=self.device,
                          memory_format=torch.channels_last if self.channels_last else torch.contiguous_format)
        if n == 0:
            return out
        table = torch.from_numpy(jobs.view(np.uint8)).to(self.device)
        strides = (ctypes.c_int64 * 4)(*out.stride())
        code = _lib.GC_F32 if self.dtype == torch.float32 else _lib.GC_BF16
        _lib.check(_lib.lib().gc_image_batch_prepare(ctypes.c_void_p(table.data_ptr()), n, channels, self.out_h, self.out_w,
                                                     1 if self.linear else 0, self.divisor, self.mean, self.std,
                                                     ctypes.c_void_p(out.data_ptr()), code, strides,
                                                     ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)),
                   "gc_image_batch_prepare")
        self._keep = (tensors, table)        # until the next call: the launch reads them asynchronously
        return out


# ---- checkpoints (BasicGANs/utils.py:5-16) ------------------------------------------------------------------------------
def save_checkpoint(path, net_G, net_D, optimizer_G, optimizer_D):
    torch.save({"net_G": net_G.state_dict(), "net_D": net_D.state_dict(),
                "optimizer_G": optimizer_G.state_dict(), "optimizer_D": optimizer_D.state_dict()}, path)


def save_weight(path, net_G):
    torch.save(net_G.state_dict(), path)


def load_checkpoint(path, net_G, net_D, optimizer_G=None, optimizer_D=None, map_location=None):
    """Restores what save_checkpoint (this one or the reference's) wrote; tensors only (weights_only=True)."""
    ckpt = torch.load(path, map_location=map_location, weights_only=True)
    net_G.load_state_dict(ckpt["net_G"])
    net_D.load_state_dict(ckpt["net_D"])
    if optimizer_G is not None:
        optimizer_G.load_state_dict(ckpt["optimizer_G"])
    if optimizer_D is not None:
        optimizer_D.load_state_dict(ckpt["optimizer_D"])
    return ckpt
