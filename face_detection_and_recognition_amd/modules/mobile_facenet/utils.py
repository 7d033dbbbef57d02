"""Mobile-FaceNet feature extraction front end
(face_detection_and_extraction/modules/mobile_facenet/utils.py:5-18): resize to 112x112, (x-127.5)/127.5,
BGR kept, NCHW — here the crop, resize and normalisation are one device kernel writing the network's NHWC input."""
import numpy as np
import torch

from ... import _lib as L


def mfn_lut(device):
    """(u8 - 127.5) / 127.5 evaluated in float64 then cast, as numpy does at utils.py:13-16."""
    return torch.from_numpy(((np.arange(256, dtype=np.uint8) - 127.5) / 127.5).astype(np.float32)).to(device)


def crops_to_input(frames_u8, items, n_items, canvas, lut):
    """frames_u8 (B,H,W,3) u8 CUDA; items int32 CUDA (n,9) fp_resize_item rows; canvas (n,112,112,C) fp32."""
    lib = L.load()
    B, H, W, _ = frames_u8.shape
    L.check(lib.fp_resize_normalize(L.ptr(frames_u8), B, H, W, L.ptr(items), int(n_items), L.ptr(canvas),
                                    canvas.shape[1], canvas.shape[2], canvas.shape[3], L.ptr(lut), 0, 0,
                                    L.current_stream(frames_u8.device)), "fp_resize_normalize")


def inference_onnx_model_mobile_facenet(feature_net, face, face_feat_in_size=(112, 112)):
    """utils.py:5-18 with ``feature_net`` = a HIP MobileFaceNet: one BGR face crop -> (E,) features."""
    dev = feature_net._device()
    face_t = torch.from_numpy(np.ascontiguousarray(face)).to(dev).unsqueeze(0)
    h, w = face.shape[:2]
    plan = feature_net.plan_for(1)
    items = torch.tensor([[0, 0, 0, w, h, 0, 0, face_feat_in_size[0], face_feat_in_size[1]]], dtype=torch.int32,
                         device=dev)
    crops_to_input(face_t, items, 1, plan.input, mfn_lut(dev))
    plan.run()
    return plan.out[0].cpu().numpy()
