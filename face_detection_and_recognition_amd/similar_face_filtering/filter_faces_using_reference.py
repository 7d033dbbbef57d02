"""filter_faces_using_reference on MI355X (similar_face_filtering/filter_faces_using_reference.py).

Same functions, flags (--ud --rd --td -m -b -r) and directory contract.  The feature extractor is a HIP
network (Mobile-FaceNet 112x112 by default; the reference's TF/Keras FaceNet SavedModel is an external download and
TensorFlow is not a dependency of this build); the filter arithmetic runs in csrc/sim.hip:
  reference-exact default   --metric l2_mean : mean of <= R reference embeddings, thres = max distance to the mean,
                                               keep iff ||e - mean|| <= thres            (:71-100, :183-197)
  batched cosine filter     --metric cosine  : keep iff max_j cos(e, ref_j) >= --tau     
"""
import argparse
import glob
import os
import shutil
from typing import List, Tuple

import numpy as np
import torch

from .. import similarity as S
from ..modules.mobile_facenet.utils import crops_to_input, mfn_lut


def _fix_path_for_globbing(dir: str) -> str:
    """:29-38: add * at the end of paths for globbing."""
    if dir[-1] == '/':
        dir += '*'
    elif dir[-1] != '*':
        dir += '/*'
    return dir


def get_class_name_list(base_dir: str) -> List[str]:
    """:41-57: sorted class sub-directory names."""
    return [d.split('/')[-1] for d in sorted(glob.glob(_fix_path_for_globbing(base_dir)))]


from ..modules.utils.jpeg import imread_batch  # noqa: E402


def read_image_bgr(img_path: str) -> np.ndarray:
    from PIL import Image
    return np.ascontiguousarray(np.asarray(Image.open(img_path).convert("RGB"))[..., ::-1])


def preprocess_tf_standardize(frames_u8_rgb, in_size=(160, 160)):
    """The TF FaceNet preprocess of the reference (filter_faces_using_reference.py:60-68 after decode_jpeg):
    frames (n, H, W, 3) uint8 RGB on the device -> (n, h, w, 3) fp32, resized and per-image standardised by
    fp_resize_standardize.  Selectable next to the Mobile-FaceNet preprocess ((x - 127.5) / 127.5 at 112x112)."""
    import torch
    from .. import _lib as L
    f = frames_u8_rgb.contiguous()
    if f.dtype != torch.uint8 or f.dim() != 4 or f.shape[-1] != 3 or f.device.type != "cuda":
        raise ValueError("expected a (n, H, W, 3) uint8 tensor on the HIP device")
    n, H, W, _ = f.shape
    oh, ow = int(in_size[0]), int(in_size[1])
    out = torch.empty((n, oh, ow, 3), dtype=torch.float32, device=f.device)
    stats = torch.empty((max(n, 1), 2), dtype=torch.float64, device=f.device)
    L.check(L.load().fp_resize_standardize(L.ptr(f), n, H, W, L.ptr(out), oh, ow, L.ptr(stats), L.current_stream(f.device)),
            "fp_resize_standardize")
    return out


def read_and_preprocess_img(img_path: str, in_size=(160, 160), dct_method: str = "INTEGER_FAST", device="cuda:0"):
    """filter_faces_using_reference.py:60-68.  The JPEG's Huffman stage runs on the host, the rest of the decode and everything
    after it on the device (modules/utils/jpeg.py; libjpeg's default "islow" inverse DCT -- dct_method is accepted for
    signature compatibility: TF's INTEGER_FAST transform has no fixture offline).  Returns a (h, w, 3) fp32 device tensor."""
    from ..modules.utils.jpeg import imread
    t = imread(img_path, str(device).replace("hip", "cuda"), bgr=False).unsqueeze(0)
    return preprocess_tf_standardize(t, in_size)[0]


def embed_images(model, paths, batch_size=32, preprocess="mobile_facenet"):
    """Decode (Huffman stage on a host thread pool, the rest on the device: modules/utils/jpeg.py), then resize + normalise and
    embed on device, batch by batch.  preprocess:
    "mobile_facenet" = cv2-style resize of the whole image to 112x112, (x - 127.5) / 127.5, BGR
    (fde/modules/mobile_facenet/utils.py:13-17); "tf_standardize" = the reference filter's own TF preprocess
    (filter_faces_using_reference.py:60-68: RGB, [0,1], bilinear resize, per-image standardisation) at the network's
    112x112 input size."""
    if preprocess not in ("mobile_facenet", "tf_standardize"):
        raise ValueError(f"unknown preprocess {preprocess!r}")
    dev = model._device()
    lut = mfn_lut(dev)
    feats = []
    for i in range(0, len(paths), batch_size):
        chunk = paths[i:i + batch_size]
        plan = model.plan_for(len(chunk))
        decoded = imread_batch(chunk, dev)       # (B, H, W, 3) when the sizes agree, else a list
        for j in range(len(chunk)):              # images differ in size: one resize launch per image
            img = decoded[j].unsqueeze(0)
            if preprocess == "tf_standardize":
                rgb = img.flip(-1).contiguous()
                plan.input[j, ..., :3].copy_(preprocess_tf_standardize(rgb, (112, 112))[0])
                plan.input[j, ..., 3:].zero_()
                continue
            h, w = img.shape[1:3]
            item = torch.tensor([[0, 0, 0, w, h, 0, 0, 112, 112]], dtype=torch.int32, device=dev)
            crops_to_input(img, item, 1, plan.input[j:j + 1], lut)
        plan.run()
        feats.append(plan.out.clone())
    return torch.cat(feats) if feats else torch.zeros((0, model.embedding_size), device=dev)


def get_ref_mean_vec_and_thres_from_imgs(model, ref_class_path: str, max_ref_img_count: int = 32,
                                         preprocess: str = "mobile_facenet") -> Tuple[np.ndarray, np.ndarray]:
    """:71-100: mean vector of the first <= max_ref_img_count reference embeddings and the max L2 distance to it."""
    X_imgs = glob.glob(ref_class_path + "/*.jpg")[:max_ref_img_count]
    feats = embed_images(model, X_imgs, batch_size=1 if len(X_imgs) < 2 else min(32, len(X_imgs)), preprocess=preprocess)
    mean, thres = S.l2_mean_thres(feats)
    print(f"number of samples considered for reference={len(X_imgs)}", f"ref mean shape={tuple(mean.shape)}")
    print("max dist from mean in the reference batch: ", float(thres))
    return mean, thres


def get_parsed_args(argv=None):
    """:103-124 plus --metric/--tau/--device."""
    parser = argparse.ArgumentParser()
    parser.add_argument('--ud', '--unfiltered_data_path', dest="unfiltered_data_path", type=str, required=True)
    parser.add_argument('--rd', '--reference_data_path', dest="reference_data_path", type=str, required=True)
    parser.add_argument('--td', '--target_data_path', dest="target_data_path", type=str, default="data/faces_filtered")
    parser.add_argument('-m', '--savedmodel_path', type=str, default="weights/mobile_facenet/mobile_facenet.pth",
                        help='Mobile-FaceNet state_dict (.pth). (default: %(default)s)')
    parser.add_argument('-b', '--batch_size', type=int, default=32)
    parser.add_argument('-r', '--ref_img_per_class', type=int, default=32)
    parser.add_argument('--metric', choices=["l2_mean", "cosine"], default="l2_mean")
    parser.add_argument('--tau', type=float, default=0.3)
    parser.add_argument('--preprocess', choices=["mobile_facenet", "tf_standardize"], default="mobile_facenet",
                        help='mobile_facenet: (x - 127.5)/127.5 BGR at 112x112 (mobile_facenet/utils.py:13-17); tf_standardize: '
                             'the reference filter\'s read_and_preprocess_img (:60-68) at 112x112. (default: %(default)s)')
    parser.add_argument('-d', '--device', default="cuda")
    return parser.parse_args(argv)


def filter_class(model, ref_class_path, unfiltered_class_path, clean_dir, unclean_dir, args):
    """One class of main()'s loop (:161-199).  Returns (similar_cnt, total)."""
    X_imgs = glob.glob(unfiltered_class_path + "/*.jpg")
    name = unfiltered_class_path.split('/')[-1]
    os.makedirs(os.path.join(clean_dir, name), exist_ok=True)
    os.makedirs(os.path.join(unclean_dir, name), exist_ok=True)
    pre = getattr(args, "preprocess", "mobile_facenet")
    feats = embed_images(model, X_imgs, args.batch_size, preprocess=pre)
    if args.metric == "l2_mean":
        mean, thres = get_ref_mean_vec_and_thres_from_imgs(model, ref_class_path, args.ref_img_per_class, preprocess=pre)
        _, keep = S.l2_filter(feats, mean, thres)
    else:
        refs = embed_images(model, glob.glob(ref_class_path + "/*.jpg")[:args.ref_img_per_class], args.batch_size,
                            preprocess=pre)
        _, _, keep = S.cosine_filter(feats, refs, args.tau)
    keep = keep.cpu().numpy()
    for pth, k in zip(X_imgs, keep):
        shutil.copy(pth, os.path.join(clean_dir if k else unclean_dir, name, pth.split('/')[-1]))
    return int(keep.sum()), len(X_imgs)


def main(argv=None):
    args = get_parsed_args(argv)
    print(args)
    from ..modules.mobile_facenet.mobile_facenet import MobileFaceNet
    model = MobileFaceNet(512)
    model.load_state_dict(torch.load(args.savedmodel_path, weights_only=True))
    model = model.to(args.device.replace("hip", "cuda"))
    ref_class_paths = glob.glob(_fix_path_for_globbing(args.reference_data_path))
    unfiltered_class_paths = glob.glob(_fix_path_for_globbing(args.unfiltered_data_path))
    if len(unfiltered_class_paths) != len(ref_class_paths):
        raise Exception("Class number Error. Unfiltered root and reference root must have the same number of classes")
    for r, u in zip(ref_class_paths, unfiltered_class_paths):
        if r.split('/')[-1] != u.split('/')[-1]:
            raise Exception(f"class {r} and {u} did not match")
    clean_dir = os.path.join(args.target_data_path, 'clean')
    unclean_dir = os.path.join(args.target_data_path, 'unclean')
    os.makedirs(clean_dir, exist_ok=True)
    os.makedirs(unclean_dir, exist_ok=True)
    for r, u in zip(ref_class_paths, unfiltered_class_paths):
        similar, total = filter_class(model, r, u, clean_dir, unclean_dir, args)
        print(f"Similar images ratio={similar / max(total, 1):2.2f}, positive={similar}, total={total}")


if __name__ == "__main__":
    main()
