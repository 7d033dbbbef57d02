"""Similarity-filter operators on device tensors (csrc/sim.hip).

  l2_mean_thres / l2_filter  — the reference's filter arithmetic
      (similar_face_filtering/filter_faces_using_reference.py:85-99, 186-189)
  cosine_filter              — batched cosine filter (cosine of
      face_detection_and_extraction/face_extraction/extract_and_label_faces_from_dataset.py:106)
"""
import torch

from . import _lib as L


def _f32c(t):
    assert t.is_cuda and t.dtype == torch.float32
    return t.contiguous()


def row_inv_norm(x):
    x = _f32c(x)
    out = torch.empty((x.shape[0],), dtype=torch.float32, device=x.device)
    L.check(L.load().fp_row_inv_norm(L.ptr(x), x.shape[0], x.shape[1], L.ptr(out), L.current_stream(x.device)),
            "fp_row_inv_norm")
    return out


def split3_rows(R):
    """The reference rows as three bf16 planes for cosine_filter's split-MFMA kernel (csrc/split.h: exact three-way split of
    every fp32 value; layout [D / 32][3][round_up(Nr, 128)][32]).  Done once per reference set."""
    R = _f32c(R)
    Nr, D = R.shape
    lib = L.load()
    out = torch.empty((lib.fp_split3_bytes(Nr, D),), dtype=torch.uint8, device=R.device)
    L.check(lib.fp_split3_rows(L.ptr(R), Nr, D, L.ptr(out), L.current_stream(R.device)), "fp_split3_rows")
    return out


def cosine_filter(G, R, tau, ginv=None, rinv=None, r3=None, x6=None):
    """G (M, D) gallery, R (Nr, D) reference -> best (M,), arg (M,) int32, keep (M,) bool.
    The M x Nr score matrix is never materialised.  x6 (default: PlanBuilder.X6 and D a multiple of 32): S = G R^T on the
    bf16 matrix cores with fp32-equivalent split arithmetic (r3 = split3_rows(R), computed here when not passed in)."""
    from .plan import PlanBuilder
    G, R = _f32c(G), _f32c(R)
    M, D = G.shape
    Nr = R.shape[0]
    assert R.shape[1] == D
    ginv = row_inv_norm(G) if ginv is None else ginv
    rinv = row_inv_norm(R) if rinv is None else rinv
    dev = G.device
    best = torch.empty((M,), dtype=torch.float32, device=dev)
    arg = torch.empty((M,), dtype=torch.int32, device=dev)
    keep = torch.empty((M,), dtype=torch.uint8, device=dev)
    packed = torch.empty((M,), dtype=torch.int64, device=dev)
    if x6 is None:
        x6 = PlanBuilder.X6 and D % 32 == 0
    if x6:
        r3 = split3_rows(R) if r3 is None else r3
        L.check(L.load().fp_cosine_filter_x6(L.ptr(G), L.ptr(ginv), M, L.ptr(r3), L.ptr(rinv), Nr, D, float(tau),
                                             L.ptr(best), L.ptr(arg), L.ptr(keep), L.ptr(packed), L.current_stream(dev)),
                "fp_cosine_filter_x6")
    else:
        L.check(L.load().fp_cosine_filter(L.ptr(G), L.ptr(ginv), M, L.ptr(R), L.ptr(rinv), Nr, D, float(tau),
                                          L.ptr(best), L.ptr(arg), L.ptr(keep), L.ptr(packed), L.current_stream(dev)),
                "fp_cosine_filter")
    return best, arg, keep.bool()


def l2_mean_thres(ref):
    """ref (R, D) -> mean (D,), thres (1,) on device (filter_faces_using_reference.py:85-99)."""
    ref = _f32c(ref)
    R, D = ref.shape
    mean = torch.empty((D,), dtype=torch.float32, device=ref.device)
    thres = torch.empty((1,), dtype=torch.float32, device=ref.device)
    L.check(L.load().fp_l2_mean_thres(L.ptr(ref), R, D, L.ptr(mean), L.ptr(thres), L.current_stream(ref.device)),
            "fp_l2_mean_thres")
    return mean, thres


def l2_filter(E, mean, thres):
    """E (M, D), mean (D,), thres (1,) -> dist (M,), keep (M,) bool (filter_faces_using_reference.py:186-189)."""
    E, mean = _f32c(E), _f32c(mean)
    M, D = E.shape
    dist = torch.empty((M,), dtype=torch.float32, device=E.device)
    keep = torch.empty((M,), dtype=torch.uint8, device=E.device)
    L.check(L.load().fp_l2_filter(L.ptr(E), M, D, L.ptr(mean), L.ptr(thres), L.ptr(dist), L.ptr(keep),
                                  L.current_stream(E.device)), "fp_l2_filter")
    return dist, keep.bool()
