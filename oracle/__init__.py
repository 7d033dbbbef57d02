"""CPU oracle — TEST INFRASTRUCTURE ONLY.

A restatement of the reference's algorithms for the detect -> embed -> filter path in plain
torch-CPU fp32 functional ops and numpy, each function citing the reference file:line it follows
(paths relative to /root/reference; fde = face_detection_and_extraction).  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package; the product
(face_detection_and_recognition_amd/) never does.

Pinning: every function here is checked against outputs of the reference itself (imported in the
build container by tools/gen_golden.py, vectors committed under tests/golden/) by
tests/test_oracle_vs_golden.py.  Two boundaries have no reference output available offline and are
"parity unpinned": cv2.resize (oracle/image_ref.py: OpenCV's published INTER_LINEAR u8 scheme restated)
and torchvision.ops.nms (oracle/yolo_ref.py: torchvision's published greedy algorithm restated, cross-checked
against the reference's own pure-torch box_iou).
"""
