"""Oracle for Mobile-FaceNet (fde/modules/mobile_facenet/mobile_facenet.py).  TEST INFRASTRUCTURE ONLY."""
import torch
import torch.nn.functional as F


def _bn(sd, pre, x):
    return F.batch_norm(x, sd[pre + "running_mean"], sd[pre + "running_var"], sd[pre + "weight"], sd[pre + "bias"],
                        False, 0.0, 1e-5)


def _conv_block(sd, pre, x, stride, pad, groups, prelu=True):
    """Conv_block / Linear_block forward (mobile_facenet.py:47-51, 61-64)."""
    x = F.conv2d(x, sd[pre + "conv.weight"], None, stride=stride, padding=pad, groups=groups)
    x = _bn(sd, pre + "bn.", x)
    if prelu:
        x = F.prelu(x, sd[pre + "prelu.weight"])
    return x


def _depth_wise(sd, pre, x, stride, residual):
    """Depth_Wise.forward (mobile_facenet.py:77-88)."""
    g = sd[pre + "conv.conv.weight"].shape[0]
    y = _conv_block(sd, pre + "conv.", x, 1, 0, 1)
    y = _conv_block(sd, pre + "conv_dw.", y, stride, 1, g)
    y = _conv_block(sd, pre + "project.", y, 1, 0, 1, prelu=False)
    return x + y if residual else y


def forward(sd, x):
    """MobileFaceNet.forward (mobile_facenet.py:140-154): (b,3,112,112) -> (b,E) unit-norm."""
    sd = {k: torch.as_tensor(v) for k, v in sd.items()}
    out = _conv_block(sd, "conv1.", x, 2, 1, 1)
    out = _conv_block(sd, "conv2_dw.", out, 1, 1, 64)
    out = _depth_wise(sd, "conv_23.", out, 2, False)
    for i in range(4):
        out = _depth_wise(sd, f"conv_3.model.{i}.", out, 1, True)
    out = _depth_wise(sd, "conv_34.", out, 2, False)
    for i in range(6):
        out = _depth_wise(sd, f"conv_4.model.{i}.", out, 1, True)
    out = _depth_wise(sd, "conv_45.", out, 2, False)
    for i in range(2):
        out = _depth_wise(sd, f"conv_5.model.{i}.", out, 1, True)
    out = _conv_block(sd, "conv_6_sep.", out, 1, 0, 1)
    out = _conv_block(sd, "conv_6_dw.", out, 1, 0, 512, prelu=False)
    out = out.view(out.size(0), -1)
    out = F.linear(out, sd["linear.weight"])
    out = F.batch_norm(out, sd["bn.running_mean"], sd["bn.running_var"], sd["bn.weight"], sd["bn.bias"], False, 0.0,
                       1e-5)
    return out / torch.norm(out, 2, 1, True)                                      # l2_norm :30-33
