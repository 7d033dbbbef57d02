"""face_detection_and_extraction/modules/yolov5_face/pytorch/__init__.py:9-31 on the HIP path."""
import numpy as np
import torch

from ... import _lib as L
from ..utils.image import bind_letterbox, check_img_size, letterbox_batch
from .general import nms_face_device, non_max_suppression_face
from .yolo import Model, attempt_load


_LUTS = {}


def yolo_lut(device):
    """img.astype(float32) / 255.0 (__init__.py:18-19) for the 256 u8 values (one device copy per device)."""
    key = str(device)
    if key not in _LUTS:
        x = np.arange(256).astype(np.float32)
        x /= 255.0
        _LUTS[key] = torch.from_numpy(x).to(device)
    return _LUTS[key]


def preprocess_batch(net, frames_u8, input_size):
    """BGR->RGB, letterbox (grey 125), /255 (__init__.py:9-22) for a batch of same-sized frames.  Returns the plan to
    run: when the network's StemBlock can read u8 frames itself (FP_OP_YSTEM_U8) the letterbox happens inside its
    staging and no fp32 canvas exists; otherwise fp_resize_normalize fills the plan's NHWC input."""
    in_w, in_h = tuple(map(check_img_size, input_size))
    frames_u8 = frames_u8.contiguous()
    B, fh, fw, _ = frames_u8.shape
    if net.letterbox_fusable(in_h, in_w) and fw >= 3 and fh <= 65535 and in_h + in_w <= 2048:
        plan = net.plan_for(B, in_h, in_w, frame_hw=(fh, fw))
        bind_letterbox(plan, frames_u8, yolo_lut(frames_u8.device), pad_value=125, swap_rb=True)
        return plan
    plan = net.plan_for(B, in_h, in_w)
    letterbox_batch(frames_u8, (in_w, in_h), yolo_lut(frames_u8.device), plan.input, pad_value=125, swap_rb=True)
    return plan


def inference_pytorch_model_yolov5_face(net, cv2_img, input_size):
    """__init__.py:25-31: one BGR image -> (k, 16) detections in model-input pixels (conf 0.4, iou 0.5)."""
    dev = net._device()
    frames = torch.from_numpy(np.ascontiguousarray(cv2_img)).to(dev).unsqueeze(0)
    plan = preprocess_batch(net, frames, input_size)
    z = net.run_plan(plan)
    return non_max_suppression_face(z, conf_thres=0.4, iou_thres=0.5)[0]
