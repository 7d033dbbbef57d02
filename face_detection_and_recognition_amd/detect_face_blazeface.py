"""``python -m face_detection_and_recognition_amd.detect_face_blazeface -i img.jpg --md w.pth --mt back``
Entry point with the reference's flags (face_detection_and_extraction/detect_face_blazeface.py:7-34).  Images
only (the video / webcam GUI loops are out of scope); prints the post-processed detections as JSON."""
import json

from .modules.blazeface.model import BlazeFaceModel
from .modules.utils.inference import inference_img
from .modules.utils.parser import get_argparse, torch_device


def main(argv=None):
    parser = get_argparse(description="Blazeface face detection (MI355X HIP path)", conflict_handler='resolve')
    parser.add_argument("--md", "--model", dest="model", default="weights/blazeface/blazefaceback.pth",
                        help="Path to weight file (.pth). anchors.npy next to it is used when present, "
                             "otherwise the MediaPipe anchors are generated. (default: %(default)s)")
    parser.add_argument("--mt", "--model_type", dest="model_type", default="back", choices=["back", "front"],
                        help="Model type back or front; must match the weight file. (default: %(default)s)")
    args = parser.parse_args(argv)
    print("Current Arguments: ", args)
    net = BlazeFaceModel(args.model, args.det_thres, args.bbox_area_thres, args.model_type,
                         torch_device(args.device))
    post = inference_img(net, args.input_src)
    print(json.dumps({"boxes": post.boxes.tolist(), "confs": post.bbox_confs.tolist(),
                      "areas": post.bbox_areas.tolist(), "landmarks": post.bbox_lmarks.tolist()}))
    return post


if __name__ == "__main__":
    main()
