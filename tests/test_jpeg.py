"""JPEG decode in front of the path (csrc/jpeg.hip, modules/utils/jpeg.py).

Pin: libjpeg-turbo itself, through Pillow (oracle/jpeg_ref.decode_pil) -- the library family cv2.imread wraps in the reference
(fde/modules/utils/inference.py:68-76).  tests/golden/jpeg holds the reference's own test images (data fixtures) and the sha256
of Pillow's decode of each, recorded in the build container (tools/gen_golden.py jpeg).
  CPU:  the product's HOST half (fp_jpeg_parse / fp_jpeg_entropy_decode, plain C) + the oracle's numpy restatement of the
        device half against Pillow, byte for byte; refusals; damaged input.
  GPU:  the product's device half (fp_jpeg_reconstruct) against Pillow and against the oracle restatement, byte for byte.
cv2.resize after the decode stays parity-unpinned (cv2 is absent offline): that part of the row is unchanged."""
import ctypes
import glob
import hashlib
import io
import json
import os

import numpy as np
import pytest
import torch

from conftest import ROOT
from face_detection_and_recognition_amd import _lib as L
from oracle import jpeg_ref

JDIR = os.path.join(ROOT, "tests", "golden", "jpeg")
EXPECTED = json.load(open(os.path.join(JDIR, "expected.json")))
BASELINE = ["ref_test2_faces_3.jpg", "ref_test1_faces_0.jpg", "ref_selfie3.jpeg", "ref_selfie1_progressive.jpeg"]    # three sequential, one progressive


def _host_decode(lib, data):
    info = L.FpJpegInfo()
    buf = (ctypes.c_uint8 * len(data)).from_buffer_copy(data)
    rc = lib.fp_jpeg_parse(buf, len(data), ctypes.byref(info))
    if rc:
        return rc, info, None
    coefs = np.zeros(int(info.n_coefs), np.int16)
    rc = lib.fp_jpeg_entropy_decode(buf, len(data), ctypes.byref(info), coefs.ctypes.data_as(ctypes.c_void_p))
    return rc, info, coefs


def _synthetic(rng, w, h, **kw):
    """A smooth random image encoded by Pillow (libjpeg-turbo's compressor) with the given options."""
    from PIL import Image
    img = np.clip(np.cumsum(np.cumsum(rng.normal(0, 3, (h, w, 3)), 0), 1) + 128, 0, 255).astype(np.uint8)
    gray = kw.pop("gray", False)
    b = io.BytesIO()
    Image.fromarray(img[..., 0] if gray else img).save(b, "JPEG", **kw)
    return b.getvalue()


SYNTH = [(w, h, dict(quality=q, subsampling=sub, progressive=prog, **({"restart_marker_blocks": rst} if rst else {})))
         for (w, h) in ((64, 48), (67, 45), (1, 1), (17, 9), (250, 131))
         for sub in (0, 1, 2) for q, rst, prog in ((30, 0, False), (92, 3, False), (40, 0, True), (95, 3, True))] + \
        [(33, 70, dict(quality=75, gray=True)), (33, 70, dict(quality=75, gray=True, progressive=True)), (8, 8, dict(quality=100, subsampling=2))]


def test_pillow_still_decodes_the_fixtures_as_recorded():
    """The pin itself: libjpeg-turbo (through the Pillow of whatever box runs this) decodes the reference's test images to the
    bytes recorded in the build container.  If this fails the library under the oracle changed, not the product."""
    for name, exp in EXPECTED.items():
        a = jpeg_ref.decode_pil(open(os.path.join(JDIR, name), "rb").read())
        assert list(a.shape) == exp["shape"] and hashlib.sha256(a.tobytes()).hexdigest() == exp["sha256_rgb"], name


def test_host_huffman_and_oracle_restatement_vs_pillow(lib):
    """fp_jpeg_parse + fp_jpeg_entropy_decode (the product's host half) feed oracle/jpeg_ref.reconstruct (numpy restatement of
    jidctint.c islow IDCT, jdsample.c fancy upsampling, jdcolor.c): identical, byte for byte, to Pillow's decode -- the
    reference's 4:2:0 test images and synthetic files over 4:4:4 / 4:2:2 / 4:2:0, odd sizes down to 1 x 1, two qualities,
    restart intervals, grayscale, sequential and progressive (spectral selection + successive approximation) files.  (In the
    build container the same comparison passes on all 1048 JPEG files under /root/reference: 796 sequential, 252 progressive.)"""
    for name in BASELINE:
        data = open(os.path.join(JDIR, name), "rb").read()
        rc, info, coefs = _host_decode(lib, data)
        assert rc == 0, (name, rc)
        assert [info.height, info.width, 3] == EXPECTED[name]["shape"] and (info.hs[0], info.vs[0]) == (2, 2)
        assert bool(info.progressive) == ("progressive" in name)
        got = jpeg_ref.reconstruct(info, coefs)
        assert hashlib.sha256(got.tobytes()).hexdigest() == EXPECTED[name]["sha256_rgb"], name
    rng = np.random.default_rng(0)
    for w, h, kw in SYNTH:
        data = _synthetic(rng, w, h, **dict(kw))
        rc, info, coefs = _host_decode(lib, data)
        assert rc == 0, (w, h, kw, rc)
        np.testing.assert_array_equal(jpeg_ref.reconstruct(info, coefs), jpeg_ref.decode_pil(data), err_msg=str((w, h, kw)))


def test_jpeg_refusals_and_damaged_input(lib):
    """Files outside the decoder's scope (here: a four-component CMYK JPEG) are refused with FP_ERR_UNSUPPORTED (the caller
    decodes them on the host), non-JPEG bytes and truncated headers with FP_ERR_INVALID_ARG; a scan cut short or with garbage
    in it -- sequential and progressive -- never reads outside the buffer and returns either an error or a full-size
    coefficient set (libjpeg also pads a short scan)."""
    from PIL import Image
    b = io.BytesIO()
    Image.new("CMYK", (32, 32)).save(b, "JPEG")
    assert _host_decode(lib, b.getvalue())[0] == -3
    assert _host_decode(lib, b"\x89PNG\r\n\x1a\n" + bytes(64))[0] == -1
    good = open(os.path.join(JDIR, "ref_selfie3.jpeg"), "rb").read()
    assert _host_decode(lib, good[:40])[0] == -1
    rng = np.random.default_rng(1)
    for cut in (len(good) // 2, len(good) - 3, 700):
        rc, info, coefs = _host_decode(lib, good[:cut])
        assert rc in (0, -1)
    for src in (good, open(os.path.join(JDIR, "ref_selfie1_progressive.jpeg"), "rb").read()):
        for trial in range(4):
            noisy = bytearray(src)
            for i in rng.integers(700, len(src) - 2, 40):
                noisy[int(i)] = int(rng.integers(0, 256))
            assert _host_decode(lib, bytes(noisy))[0] in (0, -1, -3)
        assert _host_decode(lib, src[:len(src) // 3])[0] in (0, -1)


def test_jpeg_host_half_mutation_fuzz_under_asan_ubsan(tmp_path):
    """The parser and Huffman decoder read bytes nobody vetted: tools/fuzz/run_jpeg_fuzz.sh builds the host half of csrc/jpeg.hip
    with AddressSanitizer + UndefinedBehaviorSanitizer (CPU only, no device code) and feeds it truncated / bit-flipped / 0xff- and
    zero-stuffed variants of 50 small files (4:4:4 / 4:2:2 / 4:2:0 / gray, restart intervals, progressive), each input in an
    exact-size heap block: any read or write outside it, any shift or overflow the language leaves undefined, aborts the run.
    (1.5 M inputs of the same campaign ran clean when the decoder was written; this is the 40 k regression slice.)"""
    import shutil
    import subprocess
    if not (os.path.exists("/opt/rocm/bin/hipcc") and os.path.exists("/opt/rocm/lib/llvm/bin/clang++") and shutil.which("nm")):
        pytest.skip("needs hipcc + clang++ + nm")
    env = dict(os.environ, FUZZ_BUILD_DIR=str(tmp_path))
    for attempt in range(2):
        r = subprocess.run([os.path.join(ROOT, "tools", "fuzz", "run_jpeg_fuzz.sh"), "800", "3"], capture_output=True, text=True,
                           timeout=600, env=env)
        report = "Sanitizer" in r.stderr or "runtime error" in r.stderr
        if r.returncode == 0 or report:
            break
    if r.returncode != 0 and not report:
        pytest.skip("the sanitizer build of the fuzzer did not come up here: " + r.stderr[-300:])
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    words = r.stdout.strip().splitlines()[-1].split()
    n, ok, refused = int(words[0]), int(words[2]), int(words[4])
    assert n >= 39_000 and ok > n // 10 and refused > n // 10, r.stdout      # both outcomes are exercised


@pytest.mark.gpu
def test_jpeg_device_reconstruction_is_byte_identical_to_libjpeg_turbo(dev, lib):
    """decode_jpeg (host Huffman -> device dequantise + islow IDCT + fancy upsampling + YCbCr -> RGB) == Pillow's decode, byte
    for byte: the reference's own test images (the sha256 recorded in the build container too) and the synthetic set."""
    from face_detection_and_recognition_amd.modules.utils import jpeg as J
    for name in BASELINE:
        data = open(os.path.join(JDIR, name), "rb").read()
        got = J.decode_jpeg(data, dev, bgr=False).cpu().numpy()
        assert hashlib.sha256(got.tobytes()).hexdigest() == EXPECTED[name]["sha256_rgb"], name
        np.testing.assert_array_equal(got, jpeg_ref.decode_pil(data))
        bgr = J.decode_jpeg(data, dev).cpu().numpy()                       # cv2.imread's channel order
        np.testing.assert_array_equal(bgr[..., ::-1], got)
    rng = np.random.default_rng(0)
    for w, h, kw in SYNTH:
        data = _synthetic(rng, w, h, **dict(kw))
        got = J.decode_jpeg(data, dev, bgr=False).cpu().numpy()
        np.testing.assert_array_equal(got, jpeg_ref.decode_pil(data), err_msg=str((w, h, kw)))
        rc, info, coefs = _host_decode(lib, data)
        np.testing.assert_array_equal(got, jpeg_ref.reconstruct(info, coefs))


@pytest.mark.gpu
def test_jpeg_batch_imread_and_entry_point(dev, tmp_path):
    """decode_jpeg_batch (Huffman on a thread pool, frames of different sizes), imread (JPEGs on the device, anything the
    decoder refuses or that is not a JPEG through the host fallback: same pixels as Pillow either way) and the drop-in entry: inference_img on a JPEG path with
    a BlazeFaceModel decodes on the detector's device and returns what the host-decoded array returns."""
    from face_detection_and_recognition_amd import workload as W
    from face_detection_and_recognition_amd.modules.utils import jpeg as J
    from face_detection_and_recognition_amd.modules.utils.inference import inference_img, load_image
    names = BASELINE + BASELINE[:1]
    datas = [open(os.path.join(JDIR, n), "rb").read() for n in names]
    outs = J.decode_jpeg_batch(datas, dev, bgr=False, threads=3)
    for n, o in zip(names, outs):
        assert hashlib.sha256(o.cpu().numpy().tobytes()).hexdigest() == EXPECTED[n]["sha256_rgb"], n
    from PIL import Image
    cmyk = io.BytesIO()
    Image.new("CMYK", (32, 32)).save(cmyk, "JPEG")
    with pytest.raises(J.JpegUnsupported):
        J.decode_jpeg(cmyk.getvalue(), dev)
    for n in EXPECTED:
        path = os.path.join(JDIR, n)
        img = J.imread(path, dev)
        assert img.is_cuda and img.dtype == torch.uint8
        np.testing.assert_array_equal(img.cpu().numpy(), load_image(path))      # == the host (Pillow) BGR array
    same = [os.path.join(JDIR, "ref_test2_faces_3.jpg")] * 3
    stacked = J.imread_batch(same, dev)
    assert isinstance(stacked, torch.Tensor) and tuple(stacked.shape) == (3, 540, 720, 3)
    np.testing.assert_array_equal(stacked[2].cpu().numpy(), load_image(same[0]))
    png = str(tmp_path / "x.png")
    Image.fromarray(np.arange(24 * 16 * 3, dtype=np.uint8).reshape(24, 16, 3)).save(png)
    np.testing.assert_array_equal(J.imread(png, dev).cpu().numpy(), load_image(png))           # not a JPEG: host fallback
    mixed = J.imread_batch([os.path.join(JDIR, n) for n in sorted(EXPECTED)], dev)     # different sizes: a list
    assert isinstance(mixed, list) and [list(t.shape) for t in mixed] == [EXPECTED[n]["shape"] for n in sorted(EXPECTED)]
    det = W.build_detector(dev, W.make_frames(8, dev, seed=8), cand_per_frame=48)
    path = os.path.join(JDIR, "ref_test2_faces_3.jpg")
    a = inference_img(det, path)                                               # decoded on the device
    b = inference_img(det, load_image(path))                                   # host array, as the reference passes it
    np.testing.assert_array_equal(a.boxes, b.boxes)
    np.testing.assert_array_equal(a.bbox_confs, b.bbox_confs)
