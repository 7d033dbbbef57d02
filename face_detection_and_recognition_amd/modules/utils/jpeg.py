"""JPEG decode in front of the path: host Huffman + device reconstruction (csrc/jpeg.hip), byte-identical to what the
reference's cv2.imread returns (libjpeg-turbo's default decompressor; fde/modules/utils/inference.py:68-76,
fde/face_extraction/extract_faces_from_dataset.py:393-420).

    decode_jpeg(data, device)                 one frame  -> (H, W, 3) u8 BGR tensor on `device`
    decode_jpeg_batch(datas, device)          many frames: Huffman decoding on a thread pool (the C function drops the GIL),
                                              coefficient upload and the device kernels on the caller's stream
    imread(path, device)                      cv2.imread for the device: JPEGs (sequential and progressive) through the above;
                                              what the decoder does not take (CMYK / arithmetic-coded JPEGs, PNG, ...) is
                                              decoded by Pillow on the host -- file I/O, not the hot path -- and uploaded

JpegUnsupported is raised by the first two for files outside csrc/jpeg.hip's scope (arithmetic-coded, lossless, 12-bit,
CMYK, sampling layouts other than 4:4:4 / 4:2:2 / 4:2:0).  All 1048 JPEG files of the reference's tree are inside it."""
import ctypes as C
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

from ... import _lib as L

FP_ERR_UNSUPPORTED = -3


class JpegUnsupported(L.FacepathError):
    pass


def parse(data):
    """JPEG bytes -> fp_jpeg_info (host only)."""
    info = L.FpJpegInfo()
    buf = (C.c_uint8 * len(data)).from_buffer_copy(data)
    rc = L.load().fp_jpeg_parse(buf, len(data), C.byref(info))
    if rc == FP_ERR_UNSUPPORTED:
        raise JpegUnsupported("a JPEG this decoder does not take (arithmetic-coded / lossless / 12-bit / CMYK / unusual sampling)")
    L.check(rc, "fp_jpeg_parse")
    return info, buf


def entropy_decode(data, pinned=False):
    """JPEG bytes -> (fp_jpeg_info, int16 coefficient tensor in host memory): the host half of the decode."""
    info, buf = parse(data)
    # (page-locking a buffer costs about a millisecond: only worth it for frames, not for face crops of a few KB)
    coefs = torch.empty((int(info.n_coefs),), dtype=torch.int16, pin_memory=pinned and info.n_coefs >= (1 << 19))
    L.check(L.load().fp_jpeg_entropy_decode(buf, len(data), C.byref(info), C.c_void_p(coefs.data_ptr())),
            "fp_jpeg_entropy_decode")
    return info, coefs


def reconstruct(info, coefs_dev, device, bgr=True, out=None):
    """The device half: coefficients (int16, on `device`) -> (H, W, 3) u8."""
    lib = L.load()
    ws_bytes = int(lib.fp_jpeg_workspace_bytes(C.byref(info)))
    ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=device)
    if out is None:
        out = torch.empty((info.height, info.width, 3), dtype=torch.uint8, device=device)
    assert out.is_contiguous() and tuple(out.shape) == (info.height, info.width, 3) and out.dtype == torch.uint8
    L.check(lib.fp_jpeg_reconstruct(L.ptr(coefs_dev), C.byref(info), L.ptr(ws), ws_bytes, L.ptr(out), 1 if bgr else 0,
                                    L.current_stream(device)), "fp_jpeg_reconstruct")
    return out


def decode_jpeg(data, device, bgr=True):
    device = torch.device(device)
    if device.type != "cuda":
        raise L.FacepathError("decode_jpeg reconstructs on a HIP device; there is no CPU path")
    info, coefs = entropy_decode(data, pinned=True)
    return reconstruct(info, coefs.to(device, non_blocking=True), device, bgr)


def decode_jpeg_batch(datas, device, bgr=True, threads=8):
    """List of JPEG byte strings -> list of (H, W, 3) u8 tensors on `device` (sizes may differ)."""
    device = torch.device(device)
    if device.type != "cuda":
        raise L.FacepathError("decode_jpeg_batch reconstructs on a HIP device; there is no CPU path")
    with ThreadPoolExecutor(max_workers=max(1, min(threads, len(datas)))) as pool:
        # in order, as each frame's Huffman decode finishes: its copy and reconstruction run under the decodes still going
        return [reconstruct(info, coefs.to(device, non_blocking=True), device, bgr)
                for info, coefs in pool.map(lambda d: entropy_decode(d, pinned=True), datas)]


def imread(path, device, bgr=True):
    """cv2.imread(path) as a device tensor: (H, W, 3) u8, BGR by default."""
    with open(path, "rb") as f:
        data = f.read()
    if data[:2] == b"\xff\xd8":
        try:
            return decode_jpeg(data, device, bgr)
        except JpegUnsupported:
            pass
    import io
    from PIL import Image
    rgb = np.asarray(Image.open(io.BytesIO(data)).convert("RGB"))
    arr = np.ascontiguousarray(rgb[..., ::-1] if bgr else rgb)
    return torch.from_numpy(arr).to(device)


def imread_batch(paths, device, bgr=True, threads=8):
    """cv2.imread over a list of files (the dataset driver reads its media this way,
    fde/face_extraction/extract_faces_from_dataset.py:393-420) -> (B, H, W, 3) u8 on `device` when every frame has the same size,
    else a list of (H, W, 3) tensors.  Baseline JPEGs: Huffman decoding on a thread pool, everything else of the decode on the
    device; other files through imread's host fallback."""
    device = torch.device(device)
    datas = []
    for p in paths:
        with open(p, "rb") as f:
            datas.append(f.read())

    def host(d):
        if d[:2] == b"\xff\xd8":
            try:
                return entropy_decode(d, pinned=True)
            except JpegUnsupported:
                pass
        return None
    with ThreadPoolExecutor(max_workers=max(1, min(threads, len(datas)))) as pool:
        frames = [reconstruct(h[0], h[1].to(device, non_blocking=True), device, bgr) if h is not None else imread(p, device, bgr)
                  for h, p in zip(pool.map(host, datas), paths)]
    if frames and all(f.shape == frames[0].shape for f in frames):
        return torch.stack(frames)
    return frames
