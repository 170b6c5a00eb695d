"""numpy restatement of the OpenCV steps either side of the path in the reference's video loop -- TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED against the reference for this file: ``cv2`` is not installed here, so nothing below could be checked against
OpenCV itself; each function restates the algorithm OpenCV 4.x documents / implements (modules/imgproc: color_rgb, resize)
and serves as the checker of csrc/frameio.hip:
  * ``bgr2gray_u8``        cv2.cvtColor(img, cv2.COLOR_BGR2GRAY), 8-bit: (B*3735 + G*19235 + R*9798 + 2^14) >> 15
  * ``resize_area_u8``     cv2.resize(img, dsize, interpolation=cv2.INTER_AREA) for a non-integer ratio
                           (computeResizeAreaTab + ResizeArea_Invoker: float weights, horizontal pass first, rows in order,
                           saturate_cast<uchar> = round half to even)
  * ``resize_area_half_u8`` the 2 x 2 integer fast path of the same flag: (a + b + c + d + 2) >> 2
  * ``resize_area_u8_hwc``  the output frame's resize to (640, 360) from any source size (main_new.py:723): dispatch as cv::resize
Reference call sites: main_new.py:639 (window frames), :653-656 / :664-667 (new frame), :723-725 (output frame).
"""
import numpy as np


def bgr2gray_u8(img):
    b, g, r = (img[..., k].astype(np.int64) for k in range(3))
    return ((b * 3735 + g * 19235 + r * 9798 + (1 << 14)) >> 15).astype(np.uint8)


def _area_tab(ssize, dsize):
    scale = 1.0 / (float(dsize) / ssize)
    tab = []
    for d in range(dsize):
        fsx1 = d * scale
        fsx2 = fsx1 + scale
        cell = min(scale, ssize - fsx1)
        sx1, sx2 = int(np.ceil(fsx1)), int(np.floor(fsx2))
        sx2 = min(sx2, ssize - 1)
        sx1 = min(sx1, sx2)
        ent = []
        if sx1 - fsx1 > 1e-3:
            ent.append((sx1 - 1, np.float32((sx1 - fsx1) / cell)))
        for sx in range(sx1, sx2):
            ent.append((sx, np.float32(1.0 / cell)))
        if fsx2 - sx2 > 1e-3:
            ent.append((sx2, np.float32(min(min(fsx2 - sx2, 1.0), cell) / cell)))
        tab.append(ent)
    return tab


def resize_area_u8(gray, oh, ow):
    """gray: (H, W) uint8 -> (oh, ow) uint8."""
    h, w = gray.shape
    xt, yt = _area_tab(w, ow), _area_tab(h, oh)
    src = gray.astype(np.float32)
    # horizontal pass for every source row: buf[sy, dx] = sum_k S[sy, sx_k] * alpha_k  (accumulated in tab order, float32)
    buf = np.zeros((h, ow), np.float32)
    for dx, ent in enumerate(xt):
        acc = np.zeros(h, np.float32)
        for sx, a in ent:
            acc = (acc + src[:, sx] * a).astype(np.float32)
        buf[:, dx] = acc
    out = np.zeros((oh, ow), np.uint8)
    for dy, ent in enumerate(yt):
        acc = None
        for sy, beta in ent:
            term = (buf[sy] * beta).astype(np.float32)
            acc = term if acc is None else (acc + term).astype(np.float32)
        out[dy] = np.clip(np.rint(acc), 0, 255).astype(np.uint8)   # rint: half to even, as cvRound
    return out


def resize_area_half_u8(img):
    """img: (H, W, C) uint8 with even H, W -> (H/2, W/2, C)."""
    s = img.astype(np.int32)
    return ((s[0::2, 0::2] + s[0::2, 1::2] + s[1::2, 0::2] + s[1::2, 1::2] + 2) >> 2).astype(np.uint8)


def resize_area_u8_hwc(img, oh, ow):
    """cv2.resize(img (H, W, C) uint8, (ow, oh), interpolation=cv2.INTER_AREA), down-scaling, as cv::resize dispatches it
    (imgproc/resize.cpp): 2 x 2 -> the SIMD path (resize_area_half_u8); both ratios integer -> resizeAreaFast_<uchar, int>:
    int sum of the iscale_y x iscale_x block * float(1 / area), cvRound (half to even); else the area tables per channel."""
    h, w, c = img.shape
    if h == 2 * oh and w == 2 * ow:
        return resize_area_half_u8(img)
    if h % oh == 0 and w % ow == 0:
        ky, kx = h // oh, w // ow
        s = img.astype(np.int64).reshape(oh, ky, ow, kx, c).sum(axis=(1, 3))
        v = (s.astype(np.float32) * np.float32(np.float32(1.0) / np.float32(kx * ky))).astype(np.float32)
        return np.clip(np.rint(v), 0, 255).astype(np.uint8)
    return np.stack([resize_area_u8(img[..., k], oh, ow) for k in range(c)], axis=-1)


def window_plane(frame_bgr, size=256):
    """One plane of the generator's input window from a decoded frame (main_new.py:639-643)."""
    g = resize_area_u8(bgr2gray_u8(frame_bgr), size, size)
    return (g.astype(np.float32) / np.float32(255) * np.float32(2) - np.float32(1)).astype(np.float32)
