"""Output writer for visual regression.  LPRT_BUF_OUTPUT is already display-referred: the
finalize kernel applied pow(c, 1/2.2) like rt/rtrt/cuda.hpp:2451-2454, so an 8-bit PNG is a clamp + round."""
import numpy as np


def to_rgb8(rgba_float, flip_y=True):
    """[H, W, >=3] float -> uint8 RGB.  The frame's row 0 is the BOTTOM of the image (screen y grows upwards in
    get_primary_ray, rt/rtrt/cuda.hpp:799-832), PNG rows run top-down: flip."""
    a = np.nan_to_num(np.asarray(rgba_float, dtype=np.float32)[..., :3], nan=0.0, posinf=1.0, neginf=0.0)
    a = (np.clip(a, 0.0, 1.0) * 255.0 + 0.5).astype(np.uint8)
    return a[::-1] if flip_y else a


def save_png(path, rgba_float, flip_y=True):
    from PIL import Image
    Image.fromarray(to_rgb8(rgba_float, flip_y), "RGB").save(path)
    return path


def save_pfm(path, rgba_float, flip_y=False):
    """Lossless float32 dump (PFM, bottom-up rows by definition -> no flip needed by default)."""
    a = np.asarray(rgba_float, dtype=np.float32)[..., :3]
    if flip_y:
        a = a[::-1]
    with open(path, "wb") as f:
        f.write(f"PF\n{a.shape[1]} {a.shape[0]}\n-1.0\n".encode())
        f.write(np.ascontiguousarray(a).astype("<f4").tobytes())
    return path


def load_pfm(path):
    with open(path, "rb") as f:
        assert f.readline().strip() == b"PF"
        w, h = (int(x) for x in f.readline().split())
        scale = float(f.readline())
        a = np.frombuffer(f.read(), dtype="<f4" if scale < 0 else ">f4").reshape(h, w, 3)
    return a.astype(np.float32)
