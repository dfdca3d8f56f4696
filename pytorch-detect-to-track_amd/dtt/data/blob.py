"""Image -> network-input helpers (reference: lib/model/utils/blob.py:19-52, roi_data_layer/minibatch.py:62-88).

The reference decodes with cv2.imread (BGR, uint8) and rescales with cv2.resize(..., fx, fy, INTER_LINEAR); OpenCV is not
part of this image, so decoding goes through PIL (channel order flipped to BGR) and `resize_linear` restates OpenCV's
bilinear resize for float images: destination size round-half-even(size * scale), source coordinate (d + 0.5) / scale -
0.5 with replicated borders, horizontal pass then vertical pass in float32.
"""
import numpy as np
from PIL import Image


def imread_bgr(path):
    """cv2.imread stand-in: H x W x 3 uint8 in BGR order (grey images are replicated, minibatch.py:73-75)."""
    with Image.open(path) as im:
        rgb = np.asarray(im.convert("RGB"))
    return np.ascontiguousarray(rgb[:, :, ::-1])


def _taps(n_dst, n_src, scale):
    """Left tap index and right-tap weight per destination coordinate (OpenCV resize, INTER_LINEAR, float path)."""
    pos = ((np.arange(n_dst, dtype=np.float64) + 0.5) / scale - 0.5).astype(np.float32)
    left = np.floor(pos).astype(np.int64)
    frac = (pos - left.astype(np.float32)).astype(np.float32)
    low = left < 0
    left[low] = 0; frac[low] = 0.0
    high = left >= n_src - 1
    left[high] = n_src - 1; frac[high] = 0.0
    return left, np.minimum(left + 1, n_src - 1), frac


def resize_linear(im, scale):
    """cv2.resize(im, None, None, fx=scale, fy=scale, interpolation=cv2.INTER_LINEAR) for a float32 H x W x C image."""
    im = np.asarray(im, dtype=np.float32)
    h, w = im.shape[:2]
    oh, ow = int(np.rint(h * scale)), int(np.rint(w * scale))
    x0, x1, fx = _taps(ow, w, scale)
    y0, y1, fy = _taps(oh, h, scale)
    fx = fx[None, :, None]
    rows = im[:, x0] * (np.float32(1) - fx) + im[:, x1] * fx
    fy = fy[:, None, None]
    return rows[y0] * (np.float32(1) - fy) + rows[y1] * fy


def prep_im_for_blob(im, pixel_means, target_size, max_size):
    """blob.py:34-52: mean-subtract, scale the short side to target_size (the max_size clamp is commented out in the
    reference's loader path, blob.py:45-47 -- reproduced: max_size is accepted and ignored)."""
    im = im.astype(np.float32) - np.asarray(pixel_means, dtype=np.float32)
    scale = float(target_size) / float(min(im.shape[0], im.shape[1]))
    return resize_linear(im, scale), scale


def im_list_to_blob(ims):
    """blob.py:19-32: zero-padded (N, Hmax, Wmax, 3) float32 stack."""
    hmax = max(im.shape[0] for im in ims); wmax = max(im.shape[1] for im in ims)
    blob = np.zeros((len(ims), hmax, wmax, 3), dtype=np.float32)
    for i, im in enumerate(ims):
        blob[i, :im.shape[0], :im.shape[1]] = im
    return blob
