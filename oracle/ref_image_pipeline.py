"""TEST INFRASTRUCTURE -- CPU restatement of the reference's per-sample image pipeline, for checking gc_image_batch_prepare.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this module; the product path never does.

Follows BasicGANs/utils.py:46-55 (cv2.imread -> cv2.resize(image, size) -> float / 255 -> Normalize), Pix2Pix/dataset.py:
273-318 (resize to the jitter size with INTER_LINEAR for the image and INTER_NEAREST for the label mask, random crop,
cv2.flip(., 1)) and StyleGAN/train.py:30-38.  cv2.resize itself lives in OpenCV (a dependency of the reference that is
absent from this image and from /root/reference; the reference pins no version).  Its published 8-bit INTER_LINEAR
algorithm is restated here (modules/imgproc/src/resize.cpp: coordinate (d + 0.5) * scale - 0.5, border clamps, weights
saturate_cast<short>(w * 2048), HResizeLinear in integers, VResizeLinear's ((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16)
+ 2) >> 2), and INTER_NEAREST's min(floor(d * scale), size - 1).  PARITY UNPINNED against OpenCV itself: the reference
holds no image fixtures and cv2 cannot be imported here; the restatement is pinned only by hand-computed cases in
tests/test_image_pipeline_oracle.py."""
import numpy as np


def _linear_taps(dst_size, src_size):
    d = np.arange(dst_size, dtype=np.float64)
    f = ((d + 0.5) * (src_size / dst_size) - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = f - s.astype(np.float32)
    low = s < 0
    f[low], s[low] = 0.0, 0
    high = s >= src_size - 1
    f[high], s[high] = 0.0, src_size - 1
    s1 = np.minimum(s + 1, src_size - 1)
    a0 = np.rint((np.float32(1.0) - f) * np.float32(2048.0)).astype(np.int64)
    a1 = np.rint(f * np.float32(2048.0)).astype(np.int64)
    return s, s1, a0, a1


def resize_linear_u8(image, width, height):
    """cv2.resize(image, (width, height), interpolation=cv2.INTER_LINEAR) for an [H, W, C] uint8 image."""
    src = image.astype(np.int64)
    y0, y1, b0, b1 = _linear_taps(height, image.shape[0])
    x0, x1, a0, a1 = _linear_taps(width, image.shape[1])
    rows = src[:, x0, :] * a0[None, :, None] + src[:, x1, :] * a1[None, :, None]          # horizontal pass: [H, width, C]
    out = (((b0[:, None, None] * (rows[y0] >> 4)) >> 16) + ((b1[:, None, None] * (rows[y1] >> 4)) >> 16) + 2) >> 2
    return out.astype(np.uint8)


def resize_nearest_u8(image, width, height):
    """cv2.resize(image, (width, height), interpolation=cv2.INTER_NEAREST)."""
    ys = np.minimum(np.floor(np.arange(height) * (image.shape[0] / height)).astype(np.int64), image.shape[0] - 1)
    xs = np.minimum(np.floor(np.arange(width) * (image.shape[1] / width)).astype(np.int64), image.shape[1] - 1)
    return image[ys][:, xs]


def prepare(image, resize_hw, crop_yx, out_hw, flip, linear=True, divisor=255.0, mean=None, std=None):
    """One sample of the pipeline: resize, crop window, optional left-right flip, / divisor, (x - mean) / std, [C, H, W] float32."""
    rh, rw = resize_hw
    r = resize_linear_u8(image, rw, rh) if linear else resize_nearest_u8(image, rw, rh)
    y, x = crop_yx
    r = r[y:y + out_hw[0], x:x + out_hw[1]]
    if flip:
        r = r[:, ::-1]
    t = np.ascontiguousarray(r.transpose(2, 0, 1)).astype(np.float32)
    if divisor != 1.0:
        t = t / np.float32(divisor)
    c = t.shape[0]
    m = np.zeros(c, np.float32) if mean is None else np.asarray(mean, np.float32)
    s = np.ones(c, np.float32) if std is None else np.asarray(std, np.float32)
    return (t - m[:, None, None]) / s[:, None, None]
