"""Deterministic (numpy PCG64) weights and inputs for parity tests and bench.

The reference ships no checkpoint (``/root/reference/.MISSING_LARGE_BLOBS:1``), so every
parity statement is "reference model vs this build on identical seeded weights"
(SURVEY.md §8c).  The generator below is the single definition of those weights: the
golden-vector script (``tests/golden/make_golden.py``) loads them into the reference
model, the tests and ``bench.py`` load them into this build.

Key names / shapes follow the reference ``state_dict`` (printed from the live model:
``vad/models/self_attention.py:7-21``, ``vad/modeling/transformer.py:10-61,227-252,366-375``).
"""
from __future__ import annotations

from collections import OrderedDict

import numpy as np


def state_dict_spec(feature_size: int = 80, num_layers: int = 3, d_model: int = 128):
    """Ordered (key, shape, kind) list; kind in {"w","b","ln_w","ln_b"}."""
    d_ff = 4 * d_model  # vad/models/self_attention.py:10
    spec = [
        ("input_layer.0.weight", (d_model, feature_size), "w"),
        ("input_layer.0.bias", (d_model,), "b"),
    ]
    for l in range(num_layers):
        p = f"encoder.layers.{l}."
        for name in ("query", "key", "value", "final"):
            spec.append((p + f"self_attention.{name}_projection.weight", (d_model, d_model), "w"))
            spec.append((p + f"self_attention.{name}_projection.bias", (d_model,), "b"))
        spec.append((p + "self_attention_sublayer.layer_norm.weight", (d_model,), "ln_w"))
        spec.append((p + "self_attention_sublayer.layer_norm.bias", (d_model,), "ln_b"))
        spec.append((p + "feed_forward.feed_forward.0.weight", (d_ff, d_model), "w"))
        spec.append((p + "feed_forward.feed_forward.0.bias", (d_ff,), "b"))
        spec.append((p + "feed_forward.feed_forward.3.weight", (d_model, d_ff), "w"))
        spec.append((p + "feed_forward.feed_forward.3.bias", (d_model,), "b"))
        spec.append((p + "feed_forward_sublayer.layer_norm.weight", (d_model,), "ln_w"))
        spec.append((p + "feed_forward_sublayer.layer_norm.bias", (d_model,), "ln_b"))
    spec.append(("encoder.layer_norm.weight", (d_model,), "ln_w"))
    spec.append(("encoder.layer_norm.bias", (d_model,), "ln_b"))
    spec.append(("classifier.weight", (2, d_model), "w"))
    spec.append(("classifier.bias", (2,), "b"))
    return spec


def seeded_state_dict(seed: int = 1234, feature_size: int = 80, num_layers: int = 3,
                      d_model: int = 128, gain: float = 1.0) -> "OrderedDict[str, np.ndarray]":
    """Weights ~ U(-g/sqrt(fan_in), g/sqrt(fan_in)); LN weight 1 +- 0.1, LN bias +- 0.1.

    ``gain`` > 1 sharpens the attention softmax (used by the peaked-softmax parity cases).
    """
    rng = np.random.default_rng(seed)
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()
    last_fan_in = 1
    for key, shape, kind in state_dict_spec(feature_size, num_layers, d_model):
        if kind == "w":
            last_fan_in = shape[1]
            bound = gain / np.sqrt(last_fan_in)
            arr = rng.uniform(-bound, bound, size=shape)
        elif kind == "b":
            bound = 1.0 / np.sqrt(last_fan_in)
            arr = rng.uniform(-bound, bound, size=shape)
        elif kind == "ln_w":
            arr = 1.0 + rng.uniform(-0.1, 0.1, size=shape)
        else:
            arr = rng.uniform(-0.1, 0.1, size=shape)
        out[key] = arr.astype(np.float32)
    return out


def seeded_features(seed: int, shape, kind: str = "logmel") -> np.ndarray:
    """Synthetic mel input.  "logmel": U(-13.8, 4.2) (log(1e-6) = -13.8);
    "normal": N(0,1)."""
    rng = np.random.default_rng(seed)
    if kind == "logmel":
        return rng.uniform(-13.8, 4.2, size=shape).astype(np.float32)
    if kind == "normal":
        return rng.standard_normal(size=shape).astype(np.float32)
    raise ValueError(kind)
