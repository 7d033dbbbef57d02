"""Import harness for the upstream reference (this container only).

Used ONLY by tools/gen_golden.py to produce tests/golden/*.npz.  It injects
stub modules for packages that are absent here (cv2, torchvision, thop,
seaborn, onnxruntime) into sys.modules *before* importing the reference's own
model classes from /root/reference, so the reference's arithmetic runs
unmodified on CPU.  Nothing from the reference is copied into this repo and
this file never runs on the GPU box (the reference does not exist there).
"""
import os
import sys
import types

REF = "/root/reference"
FDE = os.path.join(REF, "face_detection_and_extraction")
Y5 = os.path.join(FDE, "modules", "yolov5_face", "pytorch")


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install_stubs():
    sys.dont_write_bytecode = True
    if "cv2" not in sys.modules:
        _stub("cv2", setNumThreads=lambda n: None, ocl=types.SimpleNamespace(setUseOpenCL=lambda b: None))
    if "torchvision" not in sys.modules:
        tv = _stub("torchvision")
        tv.transforms = _stub(
            "torchvision.transforms",
            Compose=lambda x: x, Resize=lambda *a, **k: None,
            ToTensor=lambda *a, **k: None, Normalize=lambda *a, **k: None)
        tv.ops = _stub("torchvision.ops", nms=None)
    if "thop" not in sys.modules:
        _stub("thop", profile=lambda *a, **k: (0, 0), clever_format=lambda *a, **k: ("", ""))
    if "seaborn" not in sys.modules:
        _stub("seaborn")
    if "onnxruntime" not in sys.modules:
        _stub("onnxruntime")
    if "onnx" not in sys.modules:
        _stub("onnx")
    if "requests" not in sys.modules:
        try:
            import requests  # noqa
        except Exception:
            _stub("requests")


def install_tensorflow_stub(feature_of_path, log):
    """A `tensorflow` stand-in with exactly the surface similar_face_filtering/filter_faces_using_reference.py
    touches (TF is absent offline).  Images are never decoded: tf.io.read_file returns the path itself and
    the tf.image.* calls pass it through, so read_and_preprocess_img(path) -> path; tf.data.Dataset keeps Python
    lists; the "model" maps a batch of paths to the feature rows feature_of_path(path) and appends the paths to `log`.
    Everything else -- glob order, the mean, the threshold, the <= comparison, the clean / unclean copies -- is the
    reference's own code."""
    import numpy as np

    class Dataset:
        def __init__(self, items, batch=None):
            self.items, self.bs = list(items), batch

        @staticmethod
        def from_tensor_slices(x):
            return Dataset(x)

        def map(self, fn):
            return Dataset([fn(i) for i in self.items])

        def batch(self, k):
            return Dataset(self.items, int(k))

        def __len__(self):
            return len(self.items) if self.bs is None else -(-len(self.items) // self.bs)

        def __iter__(self):
            if self.bs is None:
                return iter(self.items)
            return (self.items[i:i + self.bs] for i in range(0, len(self.items), self.bs))

    class Model:
        inputs, outputs = ["stub input"], ["stub output"]

        def predict(self, batch, verbose=0):
            log.extend(batch)
            return np.stack([np.asarray(feature_of_path(p), dtype=np.float32) for p in batch])

    ident = lambda x, *a, **k: x
    tf = _stub("tensorflow", Tensor=object, float32="float32")
    tf.random = types.SimpleNamespace(set_seed=lambda s: None)
    tf.io = types.SimpleNamespace(read_file=ident)
    tf.image = types.SimpleNamespace(decode_jpeg=ident, convert_image_dtype=ident, resize=ident,
                                     per_image_standardization=ident)
    tf.data = types.SimpleNamespace(Dataset=Dataset)
    tf.keras = types.SimpleNamespace(Model=Model, models=types.SimpleNamespace(load_model=lambda path, compile=False: Model()))
    return tf


def import_reference_filter(feature_of_path, log):
    """similar_face_filtering/filter_faces_using_reference.py imported unmodified on top of the tensorflow stub."""
    install_tensorflow_stub(feature_of_path, log)
    sff = os.path.join(REF, "similar_face_filtering")
    if sff not in sys.path:
        sys.path.insert(0, sff)
    env = dict(os.environ)                       # the module sets XLA / OMP / CUDA_VISIBLE_DEVICES at import: undo
    import filter_faces_using_reference as ffr
    os.environ.clear()
    os.environ.update(env)
    return ffr


def import_reference():
    """Returns a namespace with the reference's hot-path classes/functions."""
    install_stubs()
    if FDE not in sys.path:
        sys.path.insert(0, FDE)
    ns = types.SimpleNamespace()
    from modules.blazeface import blazeface as bf
    ns.blazeface = bf
    from modules.mobile_facenet import mobile_facenet as mfn
    ns.mobile_facenet = mfn
    from modules.utils import image as uimage
    ns.image = uimage
    from modules.utils import inference as uinf
    ns.inference = uinf
    return ns


def import_reference_yolo():
    """YOLOv5-face model classes; works around the reference's broken
    parse_model by replacing models.yolo.literal_eval with a
    resolver against the reference's own namespace (harness side only)."""
    install_stubs()
    if FDE not in sys.path:
        sys.path.insert(0, FDE)
    if Y5 not in sys.path:
        sys.path.append(Y5)
    import ast
    import torch.nn as nn
    import models.yolo as yolo
    import models.common as common

    def _resolve(s):
        if isinstance(s, str):
            if s.startswith("nn."):
                return getattr(nn, s[3:])
            if hasattr(yolo, s):
                return getattr(yolo, s)
            if hasattr(common, s):
                return getattr(common, s)
        return ast.literal_eval(s)

    yolo.literal_eval = _resolve

    def build_model(yaml_name):
        """Build the reference Model from its in-tree yaml ('yolov5n.yaml').
        'nc'/'anchors' argument strings are substituted harness-side (the
        reference would resolve them with eval(); its literal_eval cannot)."""
        import yaml
        with open(os.path.join(Y5, "models", yaml_name)) as f:
            d = yaml.safe_load(f)
        for sec in ("backbone", "head"):
            for layer in d[sec]:
                layer[3] = [d["nc"] if a == "nc" else d["anchors"] if a == "anchors" else a
                            for a in layer[3]]
        return yolo.Model(d)

    ns = types.SimpleNamespace(yolo=yolo, common=common, build_model=build_model)
    import utils.general as general
    ns.general = general
    import utils.torch_utils as torch_utils
    ns.torch_utils = torch_utils
    from modules.yolov5_face.onnx import onnx_utils
    ns.onnx_utils = onnx_utils
    return ns
