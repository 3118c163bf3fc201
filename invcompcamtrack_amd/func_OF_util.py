"""Optical-flow file readers, EPE evaluation and patch extraction helpers.

Python-3 restatement of the reference's ``misc_src/func_OF_util.py`` call surface (same function and keyword
names). The reference module is Python-2 only (print statements, integer ``/``) and drags matplotlib / scipy /
func_util_geom in at import time; none of that is needed by the functions themselves, so this module depends on
NumPy alone. Integer divisions that Python 2 performed implicitly are written ``//`` here.

  func_eval_flowgt(imgflow, flowout)                       func_OF_util.py:18-36
  func_read_flo_file(name, ch=2)                           func_OF_util.py:40-57   (+ func_write_flo_file)
  func_read_pfm_file(name)                                 func_OF_util.py:60-84
  func_extract_bil_patch(ptin, img, pz, ...)               func_OF_util.py:87-129  (same bilinear convention as
                                                           util_getPatch, utilities.cpp:55-113, for even pz)
  func_extract_NN_patch(ptin, img, pz, ...)                func_OF_util.py:132-165
  func_get_pat_cosmask(psize)                              func_OF_util.py:169-175
  gauss2Dfilter(shape, sigma)                              func_OF_util.py:177-187

The reference holds no expected values for these: tests pin them against the formulas and
against util_getPatch's convention -- "parity unpinned by reference outputs".
"""
from __future__ import annotations

import numpy as np

__all__ = ["func_extract_bil_patches_hip", "func_eval_flowgt", "func_read_flo_file", "func_write_flo_file", "func_read_pfm_file",
           "func_extract_bil_patch", "func_extract_NN_patch", "func_get_pat_cosmask", "gauss2Dfilter"]

FLO_MAGIC = 202021.25  # Middlebury .flo tag ("PIEH" as float32)


def func_eval_flowgt(imgflow, flowout):
    """Mean end-point error overall and in the ground-truth magnitude bins <10, [10,40), >=40 px."""
    gtmag = np.sqrt(np.sum(imgflow ** 2, axis=2))
    errmag = np.sqrt(np.sum((imgflow - flowout) ** 2, axis=2))
    bins = [np.ones_like(gtmag).astype(bool), gtmag < 10, (gtmag >= 10) & (gtmag < 40), gtmag >= 40]

    def errcnt(mask):
        with np.errstate(invalid="ignore", divide="ignore"):
            return np.sum(errmag[mask]) / np.sum(mask)  # nan for an empty bin, like the reference

    return [errcnt(m) for m in bins]


def func_read_flo_file(name, ch=2):
    """Middlebury .flo: float32 magic 202021.25, int32 width, int32 height, then h*w*ch float32 (little endian).
    Returns an (h, w, ch) array, or [] (after printing a warning) when the header is wrong."""
    with open(name, "rb") as f:
        magic = np.fromfile(f, np.float32, count=1)
        data2d = []
        if magic.size != 1 or magic[0] != np.float32(FLO_MAGIC):
            print("Head incorrect. Invalid .flo file")
        else:
            w = int(np.fromfile(f, np.int32, count=1)[0])
            h = int(np.fromfile(f, np.int32, count=1)[0])
            data = np.fromfile(f, np.float32, count=ch * w * h)
            data2d = np.resize(data, (h, w, ch))
    return data2d


def func_write_flo_file(name, flow):
    """Inverse of func_read_flo_file (the reference only reads; its external DIS binaries write)."""
    flow = np.ascontiguousarray(flow, np.float32)
    h, w = flow.shape[:2]
    with open(name, "wb") as f:
        np.array([FLO_MAGIC], np.float32).tofile(f)
        np.array([w, h], np.int32).tofile(f)
        flow.tofile(f)


def func_read_pfm_file(name):
    """Single-channel .pfm ('Pf'): header lines 'Pf', 'w h', 'scale', then h*w float32 bottom-up rows."""
    with open(name, "rb") as f:
        magic = f.readline()
        data2d = []
        if magic == b"Pf\n":
            wh = f.readline()[:-1].split()
            w, h = int(wh[0]), int(wh[1])
            sc = float(f.readline()[:-1])
            print("Reading %d x %d pfm file with scaling %d" % (w, h, sc))
            data = np.fromfile(f, np.float32, count=w * h)
            data2d = np.resize(data, (h, w))[::-1, :]
        else:
            print("Head incorrect. Invalid .pfm file")
    return data2d


def _postprocess(pf, do_zeromean, use_mask, do_log, do_unitnorm, do_flatten):
    if do_log == 1:
        pf = np.log(np.minimum(255, np.maximum(0.1, pf)))
    if do_zeromean == 1:
        for i in range(pf.shape[2]):
            pf[:, :, i] -= np.mean(pf[:, :, i])
    if use_mask is not None:
        for i in range(pf.shape[2]):
            pf[:, :, i] *= use_mask
    if do_unitnorm == 1:
        for i in range(pf.shape[2]):
            pnorm = np.linalg.norm(pf[:, :, i])
            if pnorm < 1e-15:
                pnorm = 1e-15
            pf[:, :, i] /= pnorm
    if do_flatten == 1:
        pf = np.stack([np.ndarray.flatten(pf[:, :, i]) for i in range(pf.shape[2])], axis=1)
    return pf


def func_extract_bil_patch(ptin, img, pz, do_zeromean=0, use_mask=None, do_log=0, do_unitnorm=0, do_flatten=1):
    """pz x pz bilinear patch of a (H,W,C) image around the sub-pixel point ptin=(x,y).

    Patch-constant weights on four integer-shifted windows, exactly the tracker's convention
    (utilities.cpp:72-77): samples at x - pz//2 ... x + pz//2 - 1. Returns (pz*pz, C) when do_flatten else
    (pz, pz, C)."""
    ptin = np.asarray(ptin, dtype=float)
    cell = np.floor(ptin).astype(int)          # integer pixel below-left of the point
    fx, fy = ptin - cell                       # sub-pixel fractions: the four windows' weights are patch-constant
    half = pz // 2                             # Python-2 integer `/`: an odd pz gives a (pz-1) x (pz-1) patch
    x0, y0 = cell[0] - half, cell[1] - half

    def window(dy, dx):                        # the 2*half square whose first sample is (x0 + dx, y0 + dy)
        return img[(y0 + dy):(y0 + dy + 2 * half), (x0 + dx):(x0 + dx + 2 * half), :]

    # taps in the tracker's order (utilities.cpp:107): (x+1,y+1), (x,y+1), (x+1,y), (x,y) of the integer cell
    pf = (window(1, 1) * (fx * fy) + window(1, 0) * ((1 - fx) * fy)
          + window(0, 1) * (fx * (1 - fy)) + window(0, 0) * ((1 - fx) * (1 - fy)))
    return _postprocess(pf, do_zeromean, use_mask, do_log, do_unitnorm, do_flatten)


def func_extract_bil_patches_hip(pts, img, pz):
    """Batched raw ``func_extract_bil_patch`` on the device (``ictr_extract_bil_patches``): pts (K, 2) -> patches
    (K, side, side, C), side = 2 * (pz // 2), bit-identical to ``func_extract_bil_patch(pt, img, pz, do_flatten=0)``
    for every point (float64, same operation order). The post-processing options (log / zero-mean / mask / unit norm /
    flatten) stay per patch in ``_postprocess``. No CPU fallback: raises IctrError without a GPU."""
    from . import _lib
    pts = np.ascontiguousarray(pts, dtype=np.float64).reshape(-1, 2)
    img = np.ascontiguousarray(img, dtype=np.float64)
    if img.ndim != 3:
        raise ValueError("img must be (H, W, C)")
    side = 2 * (pz // 2)
    out = np.empty((pts.shape[0], side, side, img.shape[2]), np.float64)
    _lib.check(_lib.load().ictr_extract_bil_patches(_lib.dp(img), img.shape[0], img.shape[1], img.shape[2],
                                                    _lib.dp(pts), pts.shape[0], int(pz), _lib.dp(out)))
    return out


def func_extract_NN_patch(ptin, img, pz, do_zeromean=0, use_mask=None, do_log=0, do_unitnorm=0, do_flatten=1):
    """pz x pz nearest-neighbour patch around the integer point ptin=(x,y). Like the reference, the slice is a
    view into img until an option that subtracts/scales is applied to it (those modify a copy here)."""
    pz2 = pz // 2
    x, y = int(ptin[0]), int(ptin[1])
    pf = img[(y - pz2):(y + pz2), (x - pz2):(x + pz2), :]
    if do_zeromean == 1 or use_mask is not None or do_unitnorm == 1:
        pf = np.array(pf, dtype=float)  # the reference would write through into img here; we do not
    return _postprocess(pf, do_zeromean, use_mask, do_log, do_unitnorm, do_flatten)


def func_get_pat_cosmask(psize):
    """Radial cosine window: cos(pi/2 * min(1, r/(psize//2))), r measured from the patch centre."""
    cent = psize // 2
    idx = np.arange(psize) - cent + 0.5
    r2 = idx[:, None] ** 2 + idx[None, :] ** 2
    return np.cos(np.minimum(1, np.sqrt(r2 / float(cent ** 2))) * np.pi / 2)


def gauss2Dfilter(shape=(3, 3), sigma=0.5):
    """Normalised 2-D Gaussian window with MATLAB fspecial('gaussian') conventions: centre at ceil((n-1)/2) per
    axis, taps below machine epsilon relative to the peak dropped, unit sum."""
    rows, cols = int(shape[0]), int(shape[1])
    dy = np.arange(rows, dtype=float)[:, None] - float(np.ceil((rows - 1.0) / 2.0))
    dx = np.arange(cols, dtype=float)[None, :] - float(np.ceil((cols - 1.0) / 2.0))
    kernel = np.exp(-(dx * dx + dy * dy) / (2.0 * sigma * sigma))
    kernel[kernel < np.finfo(kernel.dtype).eps * kernel.max()] = 0.0
    total = kernel.sum()
    return kernel / total if total != 0 else kernel
