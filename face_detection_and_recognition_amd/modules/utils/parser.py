"""CLI surface of the reference (face_detection_and_extraction/modules/utils/parser.py:5-62), with the
device choices extended by the HIP spellings."""
import argparse


class ArgumentParserMod(argparse.ArgumentParser):
    """parser.py:5-34: an ArgumentParser whose arguments can be removed again by the entry points.
    Unlike the reference (which leaves the option strings registered), removal here is complete, so a
    removed flag is rejected and can be re-added without conflict_handler='resolve'."""

    def remove_argument(self, arg):
        names = arg if isinstance(arg, (list, tuple)) else [arg]
        for name in names:
            for action in list(self._actions):
                opts = vars(action)['option_strings']
                if (opts and opts[0] == name) or vars(action)['dest'] == name:
                    self._remove_action(action)
                    for o in opts:
                        self._option_string_actions.pop(o, None)
                    for group in self._action_groups:
                        if action in group._group_actions:
                            group._group_actions.remove(action)

    def remove_arguments(self, arg_list):
        for a in arg_list:
            self.remove_argument(a)


def get_argparse(*args, **kwargs):
    """parser.py:37-62: -i/--input_src, --md/--model, --dt/--det_thres 0.70, --at/--bbox_area_thres 0.12, -d/--device."""
    parser = ArgumentParserMod(*args, **kwargs)
    parser.add_argument("-i", "--input_src", default='0', dest="input_src",
                        help="Path to input image/video/cam_index")
    parser.add_argument("--md", "--model", dest="model", default=None, help="Path to model weights")
    parser.add_argument("--dt", "--det_thres", dest="det_thres", type=float, default=0.70,
                        help='score to filter weak detections. (default: %(default)s)')
    parser.add_argument("--at", "--bbox_area_thres", dest="bbox_area_thres", type=float, default=0.12,
                        help='bbox_area * 100 / image_area threshold. (default: %(default)s)')
    parser.add_argument('-d', "--device", default="hip",
                        choices=["cpu", "cuda", "cuda:0", "cuda:1", "cuda:2", "cuda:3", "hip"] +
                                [f"hip:{i}" for i in range(8)],
                        help="Device to inference on. (default: %(default)s); cpu is rejected at model build")
    return parser


def torch_device(name: str):
    """'hip' / 'hip:N' / 'cuda[:N]' -> torch device string (ROCm exposes HIP devices as 'cuda')."""
    if name == "cpu":
        raise NotImplementedError("this build has no CPU path: use -d hip")
    return name.replace("hip", "cuda")
