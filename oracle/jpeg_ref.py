"""Oracle for the JPEG decode in front of the path.  TEST INFRASTRUCTURE ONLY.

The reference reads frames with cv2.imread (fde/modules/utils/inference.py:68-76) and tf.io.decode_jpeg
(sff/filter_faces_using_reference.py:62): libjpeg(-turbo) in both cases, a third-party dependency that is not under
/root/reference (opencv-python 4.11.0.86 bundles libjpeg-turbo 3.0; cv2 and tensorflow are absent offline).  What IS here is
Pillow 12.2 with libjpeg-turbo (PIL.features: libjpeg_turbo, jpeglib 6.2 API): `decode_pil` is that library's own output on
the same bytes -- the pin.  `reconstruct` restates the library's published default algorithm after the entropy decoder
(jidctint.c jpeg_idct_islow, jdsample.c h2v2_fancy_upsample / h2v1_fancy_upsample, jdcolor.c ycc_rgb_convert) in numpy, so
that the CPU suite can check the product's host-side Huffman decoder (fp_jpeg_entropy_decode) against Pillow without a GPU,
and the GPU kernels against both."""
import io

import numpy as np


def decode_pil(data):
    """JPEG bytes -> (H, W, 3) u8 RGB by Pillow / libjpeg-turbo (defaults: JDCT_ISLOW, fancy upsampling)."""
    from PIL import Image
    im = Image.open(io.BytesIO(data))
    im.draft(None, None)
    return np.asarray(im.convert("RGB") if im.mode != "RGB" else im, dtype=np.uint8)


_C = dict(f0298=2446, f0390=3196, f0541=4433, f0765=6270, f0899=7373, f1175=9633, f1501=12299, f1847=15137, f1961=16069,
          f2053=16819, f2562=20995, f3072=25172)


def _descale(x, n):
    return (x + (1 << (n - 1))) >> n


def _idct8(v, shift):
    """One pass of jpeg_idct_islow along the LAST axis of v (int64 [..., 8])."""
    c = _C
    z2, z3 = v[..., 2], v[..., 6]
    z1 = (z2 + z3) * c["f0541"]
    tmp2 = z1 + z3 * (-c["f1847"])
    tmp3 = z1 + z2 * c["f0765"]
    z2, z3 = v[..., 0], v[..., 4]
    tmp0 = (z2 + z3) << 13
    tmp1 = (z2 - z3) << 13
    tmp10, tmp13, tmp11, tmp12 = tmp0 + tmp3, tmp0 - tmp3, tmp1 + tmp2, tmp1 - tmp2
    t0, t1, t2, t3 = v[..., 7], v[..., 5], v[..., 3], v[..., 1]
    z1 = t0 + t3
    z2 = t1 + t2
    z3 = t0 + t2
    z4 = t1 + t3
    z5 = (z3 + z4) * c["f1175"]
    t0 = t0 * c["f0298"]
    t1 = t1 * c["f2053"]
    t2 = t2 * c["f3072"]
    t3 = t3 * c["f1501"]
    z1 = z1 * -c["f0899"]
    z2 = z2 * -c["f2562"]
    z3 = z3 * -c["f1961"] + z5
    z4 = z4 * -c["f0390"] + z5
    t0 = t0 + z1 + z3
    t1 = t1 + z2 + z4
    t2 = t2 + z2 + z3
    t3 = t3 + z1 + z4
    out = np.stack([tmp10 + t3, tmp11 + t2, tmp12 + t1, tmp13 + t0, tmp13 - t0, tmp12 - t1, tmp11 - t2, tmp10 - t3], axis=-1)
    return _descale(out, shift)


def _planes(info, coefs):
    """Sample planes (u8, padded to whole blocks) of every component."""
    planes = []
    for c in range(info.ncomp):
        bw, bh = info.blocks_w[c], info.blocks_h[c]
        q = np.array(info.quant[c][:], np.int64).reshape(8, 8)
        blk = coefs[info.coef_off[c]: info.coef_off[c] + bw * bh * 64].astype(np.int64).reshape(bh, bw, 8, 8) * q
        ws = _idct8(blk.swapaxes(-1, -2), 11).swapaxes(-1, -2)       # pass 1: columns
        px = _idct8(ws, 18)                                          # pass 2: rows
        px = np.clip(px + 128, 0, 255).astype(np.uint8)
        planes.append(px.transpose(0, 2, 1, 3).reshape(bh * 8, bw * 8))
    return planes


def _h2v1_fancy(p):
    """(h, cw) -> (h, 2 cw): jdsample.c h2v1_fancy_upsample."""
    p = p.astype(np.int32)
    cw = p.shape[1]
    out = np.empty((p.shape[0], 2 * cw), np.int32)
    if cw == 1:
        out[:, 0] = out[:, 1] = p[:, 0]
        return out
    left = np.concatenate([p[:, :1], p[:, :-1]], 1)
    right = np.concatenate([p[:, 1:], p[:, -1:]], 1)
    out[:, 0::2] = (p * 3 + left + 1) >> 2
    out[:, 1::2] = (p * 3 + right + 2) >> 2
    out[:, 0] = p[:, 0]
    out[:, -1] = p[:, -1]
    return out


def _h2v2_fancy(p):
    """(ch, cw) -> (2 ch, 2 cw): jdsample.c h2v2_fancy_upsample (edges replicate inside the component)."""
    p = p.astype(np.int32)
    ch, cw = p.shape
    above = np.concatenate([p[:1], p[:-1]], 0)
    below = np.concatenate([p[1:], p[-1:]], 0)
    out = np.empty((2 * ch, 2 * cw), np.int32)
    for v, other in ((0, above), (1, below)):
        cs = p * 3 + other                                           # column sums
        left = np.concatenate([cs[:, :1], cs[:, :-1]], 1)
        right = np.concatenate([cs[:, 1:], cs[:, -1:]], 1)
        even = (cs * 3 + left + 8) >> 4
        odd = (cs * 3 + right + 7) >> 4
        even[:, 0] = (cs[:, 0] * 4 + 8) >> 4
        odd[:, -1] = (cs[:, -1] * 4 + 7) >> 4
        out[v::2, 0::2] = even
        out[v::2, 1::2] = odd
    return out


def reconstruct(info, coefs):
    """fp_jpeg_info (ctypes struct) + quantised coefficients (int16, host) -> (H, W, 3) u8 RGB."""
    H, W = info.height, info.width
    planes = _planes(info, np.asarray(coefs))
    y = planes[0][:H, :W].astype(np.int32)
    if info.ncomp == 1:
        return np.repeat(y[..., None], 3, axis=2).astype(np.uint8)
    ch = []
    for c in (1, 2):
        p = planes[c][:info.comp_h[c], :info.comp_w[c]]
        if info.hs[0] == 2 and info.vs[0] == 2:
            p = _h2v2_fancy(p)
        elif info.hs[0] == 2:
            p = _h2v1_fancy(p)
        ch.append(p[:H, :W].astype(np.int32) - 128)
    cb, cr = ch
    r = y + ((91881 * cr + 32768) >> 16)
    b = y + ((116130 * cb + 32768) >> 16)
    g = y + ((-22554 * cb + 32768 - 46802 * cr) >> 16)
    return np.clip(np.stack([r, g, b], axis=2), 0, 255).astype(np.uint8)
