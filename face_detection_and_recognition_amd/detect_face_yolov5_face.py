"""``python -m face_detection_and_recognition_amd.detect_face_yolov5_face -i img.jpg --md yolov5n-face.pt``
Entry point with the reference's flags (face_detection_and_extraction/detect_face_yolov5_face.py:13-66)."""
import json
import os
from typing import Tuple

from .modules.utils.inference import inference_img
from .modules.utils.parser import get_argparse, torch_device
from .modules.yolov5_face import attempt_load, inference_pytorch_model_yolov5_face
from .modules.yolov5_face.model import YOLOV5FaceModel


def load_model(model_path: str, det_thres: float, bbox_area_thres: float, model_in_size: Tuple[int, int], device: str):
    """detect_face_yolov5_face.py:13-38: .pt/.pth (state_dict) -> HIP network; .onnx sessions are out of scope."""
    _, fext = os.path.splitext(model_path)
    if fext in {".pt", ".pth"}:
        net = attempt_load(model_path, torch_device(device))
    else:
        raise NotImplementedError(f"[ERROR] model with extension {fext} not implemented")
    return YOLOV5FaceModel(net, det_thres, bbox_area_thres, inference_pytorch_model_yolov5_face, model_in_size)


def main(argv=None):
    parser = get_argparse(description="YOLOv5-face face detection (MI355X HIP path)", conflict_handler='resolve')
    parser.remove_argument("model")
    parser.add_argument("--md", "--model", dest="model", default="weights/yolov5s/yolov5s-face.pt",
                        help='Path to weight file (.pt/.pth state_dict). (default: %(default)s).')
    parser.add_argument("--is", "--input_size", dest="input_size", nargs=2, default=(640, 640),
                        help='Input images are resized to this size (width, height). (default: %(default)s).')
    args = parser.parse_args(argv)
    print("Current Arguments: ", args)
    args.input_size = tuple(map(int, args.input_size))
    net = load_model(args.model, args.det_thres, args.bbox_area_thres, args.input_size, args.device)
    post = inference_img(net, args.input_src)
    print(json.dumps({"boxes": post.boxes.tolist(), "confs": post.bbox_confs.tolist(), "areas": post.bbox_areas.tolist()}))
    return post


if __name__ == "__main__":
    main()
