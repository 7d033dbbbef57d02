"""Deterministic synthetic weights and inputs.

No model weights ship with the reference (GitHub release / Google Drive downloads), and
there is no network, so tests, goldens and the benchmark all run on seeded synthetic weights generated
here from the module's own ``state_dict`` layout.  numpy's PCG64 streams are stable across platforms,
so the same (keys, shapes, seed) reproduce bit-identical tensors in the build container (where the
goldens are made from the reference) and on the GPU box.
Scales are chosen so activations stay O(1) through ~50 layers (otherwise a 1e-4 tolerance is meaningless).
"""
import numpy as np
import torch


def synth_state_dict(template_sd, seed, conv_gain=1.0, residual_gain=None):
    """Fill a state_dict (name -> tensor, only names/shapes are used) with seeded values."""
    rng = np.random.default_rng(seed)
    out = {}
    for name, t in template_sd.items():
        shape = tuple(t.shape)
        leaf = name.rsplit(".", 1)[-1]
        if leaf == "num_batches_tracked":
            out[name] = torch.zeros(shape, dtype=torch.long)
            continue
        if leaf == "running_mean":
            v = rng.normal(0.0, 0.1, shape)
        elif leaf == "running_var":
            v = rng.uniform(0.8, 1.2, shape)
        elif leaf in ("anchors", "anchor_grid"):
            out[name] = t.clone()
            continue
        elif len(shape) == 4:   # conv weight [O, I/g, kh, kw]
            fan_in = shape[1] * shape[2] * shape[3]
            g = conv_gain
            if residual_gain is not None and ".convs." in name:
                g = residual_gain
            v = rng.normal(0.0, g * np.sqrt(2.0 / fan_in), shape)
        elif len(shape) == 2:   # linear weight [O, I]
            v = rng.normal(0.0, np.sqrt(1.0 / shape[1]), shape)
        elif len(shape) == 1:
            if ".bn" in name or name.startswith("bn.") or ".bn." in name or _is_bn_like(name, template_sd):
                v = rng.uniform(0.8, 1.2, shape) if leaf == "weight" else rng.normal(0.0, 0.1, shape)
            elif "prelu" in name:
                v = rng.uniform(0.1, 0.3, shape)
            elif leaf == "bias":
                v = rng.normal(0.0, 0.05, shape)
            else:
                v = rng.uniform(0.8, 1.2, shape)
        else:
            v = rng.normal(0.0, 1.0, shape)
        out[name] = torch.from_numpy(np.asarray(v, dtype=np.float32))
    return out


def _is_bn_like(name, sd):
    """A 1-D 'weight'/'bias' whose sibling 'running_mean' exists belongs to a BatchNorm."""
    prefix = name.rsplit(".", 1)[0]
    return (prefix + ".running_mean") in sd


def synth_frames(n, h, w, seed, blur=8):
    """uint8 [n, h, w, 3] BGR frames: seeded noise, box-blurred so bilinear resize is non-trivial
    (SURVEY 8d), plus two brighter textured patches per frame (the README's two-face video)."""
    rng = np.random.default_rng(seed)
    small = rng.integers(0, 256, size=(n, h // blur + 2, w // blur + 2, 3), dtype=np.uint8).astype(np.float32)
    # bilinear-ish upsample by repetition + one smoothing pass (cheap, deterministic)
    up = np.repeat(np.repeat(small, blur, axis=1), blur, axis=2)[:, :h, :w]
    k = blur // 2
    sm = (up + np.roll(up, k, axis=1) + np.roll(up, k, axis=2) + np.roll(np.roll(up, k, axis=1), k, axis=2)) / 4.0
    fine = rng.integers(0, 64, size=(n, h, w, 3), dtype=np.uint8).astype(np.float32)
    out = 0.6 * sm + fine
    return np.clip(out, 0, 255).astype(np.uint8)
