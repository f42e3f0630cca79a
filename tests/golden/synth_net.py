"""Seeded synthetic parameters and images for the composed-network fixture (net_forward.npz).

The R-50-FPN S2ANet holds ~41 M parameters (165 MB): too many to commit.  Instead BOTH sides — the
reference's own ``models.detector.S2ANet`` in ``make_golden.py`` (build container) and this repo's
``S2ANet`` in the tests — receive the same tensors, generated here from the fixture's ordered list of
(name, shape, dtype) of the REFERENCE ``state_dict`` and a seed.  Every entry has its own generator
(seed, position), so the values depend on position and shape only; the fixture also stores a float64
checksum per entry, which the tests compare before anything else.

Value rules (chosen so that a 50-layer random trunk keeps O(1) activations in eval mode):
  * conv filters (>= 4-D)            N(0, sqrt(2 / fan_in))
  * conv biases (1-D, no BN sibling) N(0, 0.05)
  * BN weight U(0.8, 1.2) (x0.25 for the last BN of a bottleneck), bias N(0, 0.1),
    running_mean N(0, 0.1), running_var U(0.5, 1.5), num_batches_tracked 0
  * non-float buffers (ORConv2d.indices) are not generated: they come from ``fixed``
"""
import collections

import numpy as np

SEED = 20261004


def _is_bn(name, names):
    return name.rsplit(".", 1)[0] + ".running_mean" in names


def synth_entry(i, name, shape, dtype, names, seed=SEED):
    rng = np.random.default_rng([seed, i])
    leaf = name.rsplit(".", 1)[1]
    shape = tuple(int(s) for s in shape)
    if leaf == "num_batches_tracked":
        return np.zeros(shape, np.int64)
    if not str(dtype).startswith("float"):
        return None
    if leaf == "running_var":
        return rng.uniform(0.5, 1.5, shape).astype(np.float32)
    if leaf == "running_mean":
        return (rng.standard_normal(shape) * 0.1).astype(np.float32)
    if _is_bn(name, names):
        if leaf == "weight":
            w = rng.uniform(0.8, 1.2, shape)
            return (w * (0.25 if ".bn3." in name else 1.0)).astype(np.float32)
        return (rng.standard_normal(shape) * 0.1).astype(np.float32)
    if len(shape) >= 4:
        fan_in = int(np.prod(shape[1:]))
        return (rng.standard_normal(shape) * np.sqrt(2.0 / fan_in)).astype(np.float32)
    return (rng.standard_normal(shape) * 0.05).astype(np.float32)


def synth_state(names, shapes, dtypes, fixed=None, scales=None, seed=SEED):
    """-> OrderedDict name -> ndarray in the given order.  fixed: name -> array (taken as is);
    scales: name -> factor applied to the generated values"""
    fixed, scales = fixed or {}, scales or {}
    nameset = set(names)
    out = collections.OrderedDict()
    for i, (n, s, d) in enumerate(zip(names, shapes, dtypes)):
        if n in fixed:
            out[n] = np.asarray(fixed[n])
            continue
        v = synth_entry(i, n, s, d, nameset, seed)
        assert v is not None, f"non-float entry {n} needs a fixed value"
        if n in scales:
            v = (v * np.float32(scales[n])).astype(np.float32)
        out[n] = v
    return out


def checksums(state):
    """float64 [n, 2]: (sum, sum of absolute values) per entry"""
    return np.array([[np.asarray(v, np.float64).sum(), np.abs(np.asarray(v, np.float64)).sum()]
                     for v in state.values()], np.float64)


def synth_images(batch, height, width, seed=SEED):
    """uint8 [B,3,H,W]: smooth blobs + noise (a white-noise image would leave nothing for the trunk to see)"""
    rng = np.random.default_rng([seed, 10 ** 6])
    yy, xx = np.mgrid[0:height, 0:width].astype(np.float32)
    imgs = np.empty((batch, 3, height, width), np.float32)
    for b in range(batch):
        for c in range(3):
            acc = np.zeros((height, width), np.float32)
            for _ in range(12):
                cx, cy = rng.uniform(0, width), rng.uniform(0, height)
                sx, sy = rng.uniform(8, 60, 2)
                th = rng.uniform(0, np.pi)
                u = (xx - cx) * np.cos(th) + (yy - cy) * np.sin(th)
                v = -(xx - cx) * np.sin(th) + (yy - cy) * np.cos(th)
                acc += rng.uniform(-1, 1) * np.exp(-0.5 * ((u / sx) ** 2 + (v / sy) ** 2))
            imgs[b, c] = 128 + 70 * acc + rng.normal(0, 12, (height, width))
    return np.clip(np.rint(imgs), 0, 255).astype(np.uint8)
