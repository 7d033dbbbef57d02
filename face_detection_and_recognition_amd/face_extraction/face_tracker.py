"""Device-side face tracker: the matching rules of the reference's ``Net`` in
face_detection_and_extraction/face_extraction/extract_and_label_faces_from_dataset.py:66-123
(``check_if_face_exists``, ``add_face``, ``clear_faces``, ``get_num_unique_faces``; thresholds :82-84), with the
gallery of a video resident in HBM and a whole frame's faces matched by one kernel launch (csrc/tracker.hip,
``fp_tracker_step``).  Age / gender are whatever the caller attaches to a face id (the reference asks for them
through a cv2 window, :301-307); they stay on the host.
"""
import torch

from .. import _lib as L

MODES = {"MOBILE_FACENET": 0, "FACE_REID_MNV3": 1}   # L2 distance / cosine distance (:104-106)


class FaceTracker:
    def __init__(self, feat_net_type="MOBILE_FACENET", feat_dim=512, device="cuda:0", max_faces=1024,
                 use_bbox_iou_to_track_face=True):
        if feat_net_type not in MODES:
            raise NotImplementedError(f"{feat_net_type} feature extraction net is not implemented."
                                      "Supported types are ['MOBILE_FACENET', 'FACE_REID_MNV3']")   # :98-100
        if not use_bbox_iou_to_track_face:
            # the reference reads `iou` unconditionally at :109-110, so use_bbox_iou=False raises NameError there
            raise ValueError("the reference's tracker only works with use_bbox_iou_to_track_face=True")
        self.feat_net_type = feat_net_type
        self.normal_thres = 1.
        self.harsh_thres = 0.72
        self.use_bbox_iou = True
        self.dev = torch.device(device)
        self.D, self.cap = int(feat_dim), int(max_faces)
        self.feats = torch.zeros((self.cap, self.D), dtype=torch.float32, device=self.dev)
        self.bboxes = torch.zeros((self.cap, 4), dtype=torch.int32, device=self.dev)
        self.count = torch.zeros((1,), dtype=torch.int32, device=self.dev)
        self.age_gender = {}     # faceid -> (age, gender)

    # -- batched entry point -------------------------------------------------------------------
    def track(self, feats, bboxes):
        """feats (F, D) fp32, bboxes (F, 4) int (x, y, xw, yh) in detection order -> (ids int32 (F,), exists uint8 (F,)).
        Face f sees the gallery as left by faces 0..f-1 (matches replace the stored feature / box, new faces append)."""
        feats = feats.to(self.dev, torch.float32).contiguous()
        bboxes = bboxes.to(self.dev, torch.int32).contiguous()
        F = feats.shape[0]
        if feats.dim() != 2 or feats.shape[1] != self.D or tuple(bboxes.shape) != (F, 4):
            raise ValueError(f"expected feats (F, {self.D}) and bboxes (F, 4), got {tuple(feats.shape)} {tuple(bboxes.shape)}")
        ids = torch.zeros((F,), dtype=torch.int32, device=self.dev)
        exists = torch.zeros((F,), dtype=torch.uint8, device=self.dev)
        lib = L.load()
        L.check(lib.fp_tracker_step(L.ptr(self.feats), L.ptr(self.bboxes), L.ptr(self.count), self.cap, self.D,
                                    L.ptr(feats), L.ptr(bboxes), F, MODES[self.feat_net_type], self.normal_thres,
                                    self.harsh_thres, L.ptr(ids), L.ptr(exists), L.current_stream(self.dev)),
                "fp_tracker_step")
        return ids, exists

    # -- the reference's per-face API ----------------------------------------------------------
    def check_if_face_exists(self, new_feat, new_bbox):
        """(:101-116) -> (exists, faceid, age, gender).  NOTE: like the reference, a match replaces the stored
        feature and box; unlike it, a miss already appends the face (the reference's caller does that next, :307),
        so call ``set_age_gender`` instead of ``add_face`` afterwards."""
        f = torch.as_tensor(new_feat, dtype=torch.float32).reshape(1, -1)
        b = torch.as_tensor([list(new_bbox)], dtype=torch.int32)
        ids, ex = self.track(f, b)
        fid, hit = int(ids[0]), bool(ex[0])
        if fid == 0:
            raise L.FacepathError(f"tracker gallery is full ({self.cap} faces)")
        if hit:
            age, gender = self.age_gender.get(fid, (None, None))
            return True, fid, age, gender
        return False, None, None, None

    def add_face(self, feat, bbox, age, gender):
        """(:118-121).  The face itself was appended by the miss in check_if_face_exists / track; this records the
        labels for the newest id."""
        self.age_gender[self.get_num_unique_faces()] = (age, gender)

    def set_age_gender(self, faceid, age, gender):
        self.age_gender[int(faceid)] = (age, gender)

    def clear_faces(self):   # :123-126
        self.count.zero_()
        self.age_gender = {}

    def get_num_unique_faces(self):   # :128-129
        return int(self.count.item())

    @property
    def max_faceid(self):
        return self.get_num_unique_faces()
