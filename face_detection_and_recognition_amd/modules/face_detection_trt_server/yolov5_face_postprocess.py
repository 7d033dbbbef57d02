"""The Triton ensemble's python post-process step on the device:
face_detection_and_extraction/modules/face_detection_trt_server/models/yolov5_face_postprocess/1/model.py:32-113 with
the tensor names and dtypes of its config.pbtxt (inputs images / face_det_thres / face_bbox_area_thres /
stride_{8,16,32}_out; outputs face_detector_faces fp32 [-1,3,112,112], face_detector_bboxes int32 [-1,4],
face_detector_confs fp32).  Decode (`conv_strides_to_anchors`), the +1-pixel-IoU greedy NMS (`w_non_max_suppression`,
conf 0.4 / nms 0.3 as hard-coded at :51-52) and the crop + float bilinear resize + normalise run as HIP kernels; the
threshold / area filter on the handful of kept boxes is host arithmetic exactly as in the reference.
Not a Triton backend: a maintainer calls ``execute`` from one.
"""
import numpy as np
import torch

from ... import _lib as L
from ..yolov5_face.general import conv_strides_to_anchors, w_non_max_suppression


class YOLOv5FacePostprocess:
    model_face_out_size = (112, 112)   # mow, moh (model.py:20-22 reads it from the output config)

    def __init__(self, device="cuda:0"):
        self.dev = torch.device(str(device).replace("hip", "cuda"))

    def execute(self, requests):
        """requests: iterable of dicts name -> numpy array (the request's input tensors).  Returns one dict per
        request with the three output tensors as numpy arrays."""
        return [self._one(r) for r in requests]

    def _one(self, r):
        dev = self.dev
        face_det_thres = np.asarray(r["face_det_thres"]).reshape(-1)[0]
        face_bbox_area_thres = np.asarray(r["face_bbox_area_thres"]).reshape(-1)[0]
        image = torch.as_tensor(np.asarray(r["images"])[0]).to(dev, torch.float32).contiguous()   # (3, H, W) RGB [0,1]
        h, w = int(image.shape[1]), int(image.shape[2])
        outputx = conv_strides_to_anchors([r["stride_8_out"], r["stride_16_out"], r["stride_32_out"]], dev)
        detections = w_non_max_suppression(outputx, num_classes=1, conf_thres=0.4, nms_thres=0.3)[0]
        mow, moh = self.model_face_out_size
        if detections is None:                                                      # model.py:67-70
            return {"face_detector_faces": np.zeros((1, 3, moh, mow), np.float32),
                    "face_detector_bboxes": np.asarray([[0, 0, 0, 0]], dtype=np.int32),
                    "face_detector_confs": np.asarray([[0.]], dtype=np.float32)}
        det = detections.cpu().numpy()
        det = det[det[..., 4] > face_det_thres]                                     # :75
        bbox_area = (det[:, 2] - det[:, 0]) * (det[:, 3] - det[:, 1])
        det = det[100 * bbox_area / (w * h) > face_bbox_area_thres]                 # :78-82
        boxes = []
        for box in det[..., :4]:                                                    # :86-89 (offsets are 0)
            xmin, ymin, xmax, ymax = map(int, box)
            boxes.append([max(xmin, 0), max(ymin, 0), min(xmax, w), min(ymax, h)])
        n = len(boxes)
        if n == 0:   # np.asarray([]) in the reference: empty 1-D arrays
            return {"face_detector_faces": np.zeros((0,), np.float32), "face_detector_bboxes": np.zeros((0,), np.int32),
                    "face_detector_confs": np.zeros((0,), np.float32)}
        b = torch.tensor(boxes, dtype=torch.int32, device=dev)
        if bool(((b[:, 2] <= b[:, 0]) | (b[:, 3] <= b[:, 1])).any()):
            raise L.FacepathError("empty face crop (cv2.resize raises on it in the reference)")
        faces = torch.empty((n, 3, moh, mow), dtype=torch.float32, device=dev)
        L.check(L.load().fp_crop_resize_f32(L.ptr(image), h, w, L.ptr(b), n, L.ptr(faces), moh, mow,
                                            L.current_stream(dev)), "fp_crop_resize_f32")
        return {"face_detector_faces": faces.cpu().numpy(), "face_detector_bboxes": np.asarray(boxes, dtype=np.int32),
                "face_detector_confs": det[..., 4].astype(np.float32)}               # shape (n,), as the reference
