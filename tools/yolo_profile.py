import sys, os, ctypes, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tools"))
from plan_profile import profile
from face_detection_and_recognition_amd.modules.yolov5_face.yolo import Model
from face_detection_and_recognition_amd.synth import synth_state_dict
dev = torch.device("cuda:0")
for name in ("yolov5n", "yolov5s"):
    m = Model(name); m.load_state_dict(synth_state_dict(m.state_dict(), 11)); m = m.fuse().to(dev)
    p = m.plan_for(256, 640, 640); p.input.uniform_()
    profile(p, name + " B=256 640x640", reps=3)
