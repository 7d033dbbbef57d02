"""BlazeFaceModel plugin (face_detection_and_extraction/modules/blazeface/model.py:12-105) on the HIP path,
plus ``predict_batch`` — the batched fast path the reference does not have (it runs one frame per call)."""
import os

import numpy as np
import torch

from ... import _lib as L
from ..models.base import Model
from ..utils.image import bind_letterbox, letterbox_batch
from .blazeface import BlazeFace, generate_anchors

MODEL_IN_SIZES = {"back": (256, 256), "front": (128, 128)}

_REORDER = [1, 0, 3, 2] + list(range(4, 17))   # model.py:70  (ymin,xmin,ymax,xmax,..) -> (xmin,ymin,xmax,ymax,..)


def load_net(model_path: str, model_type: str, device: str):
    """model.py:16-38.  ``.pth`` state_dict -> HIP BlazeFace; anchors.npy next to the weights (the reference
    always loads anchors.npy) or, when absent, the generated MediaPipe anchors."""
    print(f"Using {model_type} type model")
    _, fext = os.path.splitext(model_path)
    anchors = os.path.join(os.path.dirname(model_path), "anchors.npy")
    is_back_model = model_type == "back"
    if fext == ".pth":
        net = BlazeFace(back_model=is_back_model).to(device)
        net.load_weights(model_path)
        if os.path.exists(anchors):
            net.load_anchors(anchors)
        else:
            net.set_anchors(generate_anchors(is_back_model))
        runtime = None
    elif fext == ".onnx":
        raise NotImplementedError("[ERROR] onnxruntime sessions are out of scope of the HIP build; pass the .pth")
    else:
        raise NotImplementedError(f"[ERROR] model with extension {fext} not implemented")
    return net, runtime


class BlazeFaceModel(Model):

    __slots__ = ["net", "runtime", "model_type"]
    accepts_device_frames = True     # __call__ / predict_batch take (H, W, 3) / (B, H, W, 3) u8 BGR CUDA tensors as well as numpy

    def __init__(self, model_path: str, det_thres: float, bbox_area_thres: float, model_type: str,
                 device: str = "cuda", net: BlazeFace = None):
        input_size = MODEL_IN_SIZES[model_type]
        Model.__init__(self, input_size, det_thres, bbox_area_thres)
        if net is not None:           # already-built network (tests / benchmark with synthetic weights)
            self.net, self.runtime = net, None
        else:
            self.net, self.runtime = load_net(model_path, model_type, device)
        self.model_type = model_type

    def __call__(self, cv2_img: np.ndarray) -> np.ndarray:
        """BGR HWC u8 -> (k, 17) [xmin, ymin, xmax, ymax, 6 landmarks, conf], normalised to [0,1] (model.py:59-71)."""
        return self.predict_batch(cv2_img[None])[0]

    def raw_batch(self, frames):
        """frames: (B, H, W, 3) u8 BGR (numpy or CUDA tensor) -> device dets (B, 896, 17) [ymin,xmin,...], counts (B,)."""
        net = self.net
        dev = net._device()
        if isinstance(frames, np.ndarray):
            frames = torch.from_numpy(np.ascontiguousarray(frames))
        frames = frames.to(dev)
        B, fh, fw, _ = frames.shape
        if BlazeFace.FUSE_LETTERBOX and fw >= 3 and fh <= 65535 and frames.dtype == torch.uint8:
            # pad_resize_image + BGR->RGB + x/127.5-1 (model.py:61,75; blazeface.py:248-250) happen inside the stem conv's
            # staging (FP_OP_STEM_U8): no fp32 canvas, no letterbox launch
            plan = net.plan_for(B, frame_hw=(fh, fw))
            bind_letterbox(plan, frames.contiguous(), net._preprocess_lut(), pad_value=125, swap_rb=True)
        else:
            plan = net.plan_for(B)
            letterbox_batch(frames, self.input_size, net._preprocess_lut(), plan.input, pad_value=125, swap_rb=True)
        plan.run()
        net.last_plan = plan          # measurement / tests: the plan (and its raw r, c views) of the last batch
        return net.postprocess(plan.r, plan.c)

    def predict_batch(self, frames):
        dets, counts = self.raw_batch(frames)
        dets = dets.cpu().numpy()
        counts = counts.cpu().numpy()
        return [dets[i, :counts[i]][:, _REORDER] if counts[i] > 0 else np.zeros((0, 17), dtype=np.float32)
                for i in range(dets.shape[0])]
