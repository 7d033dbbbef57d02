"""ORACLE (test infrastructure only): CPU restatement of the reference's face-tracker matching.

Follows fde/face_extraction/extract_and_label_faces_from_dataset.py:101-121 (Net.check_if_face_exists, Net.add_face)
and fde/modules/utils/image.py:124-143 (calculate_bbox_iou).  Pinned by tests/golden/tracker.npz, which
tools/gen_golden.py produces by calling those reference methods themselves (on a Net instance created without its
__init__, which only loads model files).
"""
import numpy as np


def calculate_bbox_iou(bbox1, bbox2):
    """image.py:124-143 — boxes (xmin, ymin, xmax, ymax); 0 when they do not intersect."""
    x1min, y1min, x1max, y1max = bbox1
    x2min, y2min, x2max, y2max = bbox2
    x_diff = min(x1max, x2max) - max(x1min, x2min)
    y_diff = min(y1max, y2max) - max(y1min, y2min)
    if x_diff < 0 or y_diff < 0:
        return 0
    intersect = x_diff * y_diff
    return intersect / (((x1max - x1min) * (y1max - y1min)) + ((x2max - x2min) * (y2max - y2min)) - intersect)


class FaceTrackerRef:
    """State and rules of extract_and_label_faces_from_dataset.py:82-84,101-123."""

    def __init__(self, feat_net_type="MOBILE_FACENET", normal_thres=1., harsh_thres=0.72):
        self.feat_net_type = feat_net_type
        self.normal_thres, self.harsh_thres = normal_thres, harsh_thres
        self.faces = []          # [faceid, feat, bbox]
        self.max_faceid = 0

    def check_if_face_exists(self, new_feat, new_bbox):   # :101-116
        for i, (faceid, feat, bbox) in enumerate(self.faces):
            if self.feat_net_type == "MOBILE_FACENET":
                dist = np.linalg.norm(feat - new_feat)
            else:
                dist = 1 - (np.inner(feat, new_feat) / (np.linalg.norm(feat) * np.linalg.norm(new_feat)))
            iou = calculate_bbox_iou(bbox, new_bbox)
            if (dist < self.normal_thres and iou > 0.1) or dist < self.harsh_thres:
                self.faces[i][1] = new_feat
                self.faces[i][2] = new_bbox
                return True, faceid
        return False, None

    def add_face(self, feat, bbox):                       # :118-121
        self.max_faceid += 1
        self.faces.append([self.max_faceid, feat, bbox])

    def track(self, feats, bboxes):
        """The per-face sequence of the reference's frame loop (:281-307): ids and exists flags."""
        ids, exists = [], []
        for f, b in zip(feats, bboxes):
            ok, fid = self.check_if_face_exists(f, tuple(int(v) for v in b))
            if not ok:
                fid = self.max_faceid + 1
                self.add_face(f, tuple(int(v) for v in b))
            ids.append(fid)
            exists.append(ok)
        return np.asarray(ids, np.int32), np.asarray(exists, bool)
