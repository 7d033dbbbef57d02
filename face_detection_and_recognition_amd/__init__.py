"""facepath-mi355x: MI355X-native detect -> embed -> similarity-filter hot path.

Drop-in for the hot path of SamSamhuns/face_detection_and_recognition: the
``modules/*`` Net API, ``detect_face_*`` and ``filter_faces_using_reference``
entry points keep the reference's names and semantics; the arithmetic runs in
hand-written gfx950 HIP kernels behind the C ABI in include/facepath.h.
"""
__version__ = "0.1.0"
