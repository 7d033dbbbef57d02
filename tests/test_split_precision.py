"""CPU checks of the bf16x6 split arithmetic (csrc/split.h, plan.split3_bf16) -- no GPU needed.

The split-MFMA kernels compute an fp32 product a*b as six bf16 x bf16 products of the exact three-way splits of a and b,
accumulated in fp32.  Here the same arithmetic is emulated with torch on the CPU inside the Mobile-FaceNet oracle
(every groups == 1 conv and the linear layer) and compared with an fp64 run of the oracle: the claim "as accurate as the
fp32 fmaf chain" is a measured property, and the two-piece / three-product variant is shown to be two orders worse
(which is why it is not used)."""
import numpy as np
import torch
import torch.nn.functional as F

from face_detection_and_recognition_amd.modules.mobile_facenet.mobile_facenet import MobileFaceNet
from face_detection_and_recognition_amd.plan import split3_bf16
from face_detection_and_recognition_amd.synth import synth_state_dict
from oracle import mobilefacenet_ref


def _trunc_bf16(t):
    return (t.view(torch.int32) & ~0xFFFF).view(torch.float32)


def _split3(t):
    h = _trunc_bf16(t)
    r = t - h
    m = _trunc_bf16(r)
    return h, m, r - m


def test_split3_is_exact_and_three_bf16_pieces():
    rng = np.random.default_rng(0)
    w = np.concatenate([rng.normal(0, 1, 4096), rng.normal(0, 1e-20, 64), rng.normal(0, 1e20, 64), [0.0, -0.0, 1.0, -1.0]])
    w = w.astype(np.float32)
    p = split3_bf16(w)
    assert p.dtype == np.uint16 and p.shape == (3,) + w.shape
    pieces = (p.astype(np.uint32) << 16).view(np.float32)
    np.testing.assert_array_equal(pieces[0] + pieces[1] + pieces[2], w)            # exact, in this order of additions
    assert (np.abs(pieces[1]) <= np.abs(pieces[0]) * 2.0 ** -7).all()              # each piece carries the next 8 bits
    assert (np.abs(pieces[2]) <= np.abs(pieces[0]) * 2.0 ** -15).all()
    # the device-side split (fp_split_pair) is the same formula: truncation of the sign-magnitude pattern
    t = torch.from_numpy(w)
    h, m, l = _split3(t)
    np.testing.assert_array_equal(h.numpy(), pieces[0])
    np.testing.assert_array_equal(m.numpy(), pieces[1])
    np.testing.assert_array_equal(l.numpy(), pieces[2])


def test_bf16x6_is_as_accurate_as_the_fp32_chain_through_mobilefacenet(monkeypatch):
    sd = synth_state_dict(MobileFaceNet(512).state_dict(), 7)
    x = torch.from_numpy(np.random.default_rng(1).uniform(-1, 1, (3, 3, 112, 112)).astype(np.float32))
    conv, lin = F.conv2d, F.linear
    with torch.no_grad():
        e32 = mobilefacenet_ref.forward(sd, x)
        sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
        e64 = mobilefacenet_ref.forward(sd64, x.double()).float()

        def run(pairs):
            def conv_split(a, w, b=None, stride=1, padding=0, dilation=1, groups=1):
                if groups != 1:
                    return conv(a, w, b, stride, padding, dilation, groups)
                sa, sw = _split3(a), _split3(w)
                acc = None
                for i, j in pairs:          # smallest products first, as fp_mfma_x6 issues them
                    y = conv(sa[i], sw[j], None, stride, padding, dilation, groups)
                    acc = y if acc is None else acc + y
                return acc
            monkeypatch.setattr(F, "conv2d", conv_split)
            monkeypatch.setattr(F, "linear", lambda a, w, b=None: conv_split(a[:, :, None, None], w[:, :, None, None])[:, :, 0, 0])
            try:
                return mobilefacenet_ref.forward(sd, x)
            finally:
                monkeypatch.setattr(F, "conv2d", conv)
                monkeypatch.setattr(F, "linear", lin)

        e6 = run([(1, 1), (2, 0), (0, 2), (1, 0), (0, 1), (0, 0)])
        e3 = run([(1, 0), (0, 1), (0, 0)])
    err32 = (e32 - e64).abs().max().item()
    err6 = (e6 - e64).abs().max().item()
    err3 = (e3 - e64).abs().max().item()
    assert err6 <= 2.0 * err32 + 1e-7, (err6, err32)      # measured: 2.1e-7 against 3.3e-7
    assert (e6 - e32).abs().max().item() < 2e-6           # far inside the north_star's 1e-4
    assert err3 > 20 * err6                               # the two-piece split (three products) is NOT fp32-equivalent
