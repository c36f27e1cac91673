"""Deterministic parameter/buffer filler keyed by state-dict key.

Test infrastructure (see oracle/__init__.py).  Parameters are never stored in
fixtures: the reference model (in the build container) and the product model
(anywhere) are both filled from this closed-form generator, so weights
regenerate bit-identically and only *outputs* are committed as golden vectors.
"""
import zlib

import numpy as np
import torch


def _rng(key, seed):
    return np.random.default_rng([zlib.crc32(key.encode()), seed])


def fill_tensor(key, t, seed=123):
    """Return an ndarray for state-dict entry `key` shaped like tensor `t`."""
    shape = tuple(t.shape)
    r = _rng(key, seed)
    leaf = key.split('.')[-1]
    if leaf == 'num_batches_tracked':
        return np.zeros(shape, dtype=np.int64)
    if leaf == 'running_mean':
        return r.uniform(-0.2, 0.2, shape).astype(np.float32)
    if leaf == 'running_var':
        return r.uniform(0.5, 1.5, shape).astype(np.float32)
    if leaf == 'weight' and len(shape) == 4:          # conv weight, He-normal
        fan_in = shape[1] * shape[2] * shape[3]
        return (r.standard_normal(shape) * np.sqrt(2.0 / fan_in)).astype(np.float32)
    if leaf == 'weight':                              # BN gamma; ~10 % negative
        g = r.uniform(0.5, 1.5, shape)
        sign = np.where(r.uniform(0, 1, shape) < 0.1, -1.0, 1.0)
        return (g * sign).astype(np.float32)
    if leaf == 'bias' and key.split('.')[-2] in ('0', '3', 'conv') and 'outc' not in key \
            and _is_conv_bias(key):
        return r.uniform(-0.1, 0.1, shape).astype(np.float32)
    if leaf == 'bias':
        return r.uniform(-0.3, 0.3, shape).astype(np.float32)
    raise KeyError(key)


def _is_conv_bias(key):
    # conv layers sit at Sequential index 0 and 3 (models/unet_parts.py:12-19)
    idx = key.split('.')[-2]
    return idx in ('0', '3')


def fill_module(module, seed=123):
    """Overwrite every parameter and buffer of `module` in place."""
    sd = module.state_dict()
    new = {}
    for k, v in sd.items():
        kk = k[len('module.'):] if k.startswith('module.') else k
        new[k] = torch.from_numpy(fill_tensor(kk, v, seed)).to(v.dtype)
    module.load_state_dict(new)
    return module


def make_inputs(batch, channels, size, seed=0, different_dates=False, size_w=None):
    """Synthetic z-scored patch pair + sparse change labels."""
    size_w = size if size_w is None else size_w
    r = np.random.default_rng([seed, batch, channels, size, size_w])
    x1 = r.standard_normal((batch, channels, size, size_w)).astype(np.float32)
    if different_dates:
        x2 = (1.7 * r.standard_normal(x1.shape) + 0.8).astype(np.float32)
    else:
        x2 = (x1 + 0.3 * r.standard_normal(x1.shape)).astype(np.float32)
    lbl = (r.uniform(0, 1, (batch, size, size_w)) < 0.1).astype(np.uint8)
    return x1, x2, lbl
