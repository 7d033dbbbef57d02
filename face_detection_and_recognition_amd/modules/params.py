"""Parameter containers with the reference's ``state_dict`` key names.

They hold weights only; calling them raises.  All arithmetic of the hot path
runs in the HIP plan built from these parameters (plan.py), never in torch.
"""
import math

import torch
import torch.nn as nn


class _NoCompute(nn.Module):
    def forward(self, *a, **k):
        raise RuntimeError(f"{type(self).__name__} is a parameter container; the network runs as a HIP plan "
                           "(libfacepath.so), not through torch modules")


class ConvParams(_NoCompute):
    """Keys: ``weight`` [O, I/g, k, k] (+ ``bias``) like nn.Conv2d."""

    def __init__(self, cin, cout, k, stride=1, padding=0, groups=1, bias=True):
        super().__init__()
        self.in_channels, self.out_channels = cin, cout
        self.kernel_size, self.stride, self.padding, self.groups = k, stride, padding, groups
        self.weight = nn.Parameter(torch.empty(cout, cin // groups, k, k), requires_grad=False)
        fan_in = (cin // groups) * k * k
        nn.init.normal_(self.weight, 0.0, math.sqrt(2.0 / fan_in))
        if bias:
            self.bias = nn.Parameter(torch.zeros(cout), requires_grad=False)
        else:
            self.register_parameter("bias", None)


class BNParams(_NoCompute):
    """Keys: weight, bias, running_mean, running_var, num_batches_tracked like nn.BatchNorm{1,2}d (eval mode)."""

    def __init__(self, c, eps=1e-5):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(c), requires_grad=False)
        self.bias = nn.Parameter(torch.zeros(c), requires_grad=False)
        self.register_buffer("running_mean", torch.zeros(c))
        self.register_buffer("running_var", torch.ones(c))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))


class PReLUParams(_NoCompute):
    def __init__(self, c):
        super().__init__()
        self.weight = nn.Parameter(torch.full((c,), 0.25), requires_grad=False)


class LinearParams(_NoCompute):
    def __init__(self, cin, cout, bias=False):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin), requires_grad=False)
        nn.init.normal_(self.weight, 0.0, math.sqrt(1.0 / cin))
        if bias:
            self.bias = nn.Parameter(torch.zeros(cout), requires_grad=False)
        else:
            self.register_parameter("bias", None)


def npy(t):
    return t.detach().cpu().numpy()
