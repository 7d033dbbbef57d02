"""face_detection_and_extraction/modules/yolov5_face/pytorch/__init__.py:9-31 on the HIP path."""
import numpy as np
import torch

from ... import _lib as L
from ..utils.image import check_img_size, letterbox_batch
from .general import nms_face_device, non_max_suppression_face
from .yolo import Model, attempt_load


def yolo_lut(device):
    """img.astype(float32) / 255.0 (__init__.py:18-19) for the 256 u8 values."""
    x = np.arange(256).astype(np.float32)
    x /= 255.0
    return torch.from_numpy(x).to(device)


def preprocess_batch(net, frames_u8, input_size):
    """BGR->RGB, letterbox (grey 125), /255, NHWC float into the plan input (__init__.py:9-22).  Returns the plan."""
    in_w, in_h = tuple(map(check_img_size, input_size))
    B = frames_u8.shape[0]
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
