"""Minimal stand-in for OpenCV's Python module on hosts without it -- ONLY what generate_illusion.py itself touches:
`cv2.imread` for the Colab preview (:659, :673) and `cv2.cvtColor` with COLOR_RGB2BGR / COLOR_GRAY2BGR in pil_to_cv2
(:467-474).  The optical flow does not go through this module (it runs on the HIP engine)."""
import numpy as np

COLOR_BGR2RGB = 4
COLOR_RGB2BGR = 4
COLOR_GRAY2BGR = 8
COLOR_BGR2GRAY = 6
IMREAD_COLOR = 1


def imread(path, flags=IMREAD_COLOR):
    from PIL import Image
    try:
        return np.ascontiguousarray(np.asarray(Image.open(path).convert("RGB"))[:, :, ::-1])
    except (OSError, ValueError):
        return None  # OpenCV returns None for unreadable files


def imwrite(path, image):
    from PIL import Image
    a = np.asarray(image)
    Image.fromarray(a if a.ndim == 2 else np.ascontiguousarray(a[:, :, ::-1])).save(path)
    return True


def cvtColor(src, code):
    a = np.asarray(src)
    if code == COLOR_RGB2BGR:
        return np.ascontiguousarray(a[:, :, ::-1])
    if code == COLOR_GRAY2BGR:
        return np.ascontiguousarray(np.repeat(a[:, :, None], 3, axis=2))
    if code == COLOR_BGR2GRAY:  # OpenCV's 8-bit fixed-point weights (15 bits): B 3735, G 19235, R 9798
        b, g, r = (a[:, :, i].astype(np.int64) for i in range(3))
        return ((b * 3735 + g * 19235 + r * 9798 + (1 << 14)) >> 15).astype(np.uint8)
    raise NotImplementedError("cv2 stand-in: colour conversion code %r" % (code,))
