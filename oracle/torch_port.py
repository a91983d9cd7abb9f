"""Stock-PyTorch CPU restatement of the hot path -- TEST / BASELINE INFRASTRUCTURE ONLY.

Used by bench.py's cpu_baseline leg: it dispatches the same ATen CPU ops the reference does
(addmm, bmm with a materialised [B,1,T,T] score tensor, softmax, layer_norm: SURVEY.md section 2),
so its timing on the GPU box's host cores stands in for "the reference's own CPU path", which
cannot travel.  Checked against the golden vectors in tests/test_oracle_golden.py.
Restates vad/models/self_attention.py:23-28 and vad/modeling/transformer.py:24-61,227-238,281-363,
366-382,392-414 functionally (no module classes).
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F


def positional_encoding(T: int, D: int) -> torch.Tensor:
    pe = torch.zeros(T, D)
    position = torch.arange(0, T, dtype=torch.float32).unsqueeze(1)
    div = torch.exp(torch.arange(0, D, 2, dtype=torch.float32) * -(math.log(10000.0) / D))
    pe[:, 0::2] = torch.sin(position * div)
    pe[:, 1::2] = torch.cos(position * div)
    return pe


@torch.no_grad()
def forward(state: dict, x: torch.Tensor) -> torch.Tensor:
    """state: key -> torch tensor (reference state_dict keys); x [B,T,F] fp32 -> log-probs [B,T,2]."""
    B, T, _ = x.shape
    D = state["input_layer.0.weight"].shape[0]
    L = 1 + max(int(k.split(".")[2]) for k in state if k.startswith("encoder.layers."))
    h = F.linear(x, state["input_layer.0.weight"], state["input_layer.0.bias"])
    h = h + positional_encoding(T, D).unsqueeze(0) / math.sqrt(D)
    for l in range(L):
        p = f"encoder.layers.{l}."
        n = F.layer_norm(h, (D,), state[p + "self_attention_sublayer.layer_norm.weight"],
                         state[p + "self_attention_sublayer.layer_norm.bias"])
        q, k, v = (F.linear(n, state[p + f"self_attention.{nm}_projection.weight"],
                            state[p + f"self_attention.{nm}_projection.bias"]).view(B, T, 1, D).transpose(1, 2)
                   for nm in ("query", "key", "value"))
        scores = torch.matmul(q, k.transpose(2, 3)) / np.sqrt(D)
        ctx = torch.matmul(torch.softmax(scores, dim=3), v).transpose(1, 2).contiguous().view(B, T, D)
        h = F.linear(ctx, state[p + "self_attention.final_projection.weight"],
                     state[p + "self_attention.final_projection.bias"]) + h
        n = F.layer_norm(h, (D,), state[p + "feed_forward_sublayer.layer_norm.weight"],
                         state[p + "feed_forward_sublayer.layer_norm.bias"])
        f = F.linear(torch.relu(F.linear(n, state[p + "feed_forward.feed_forward.0.weight"],
                                         state[p + "feed_forward.feed_forward.0.bias"])),
                     state[p + "feed_forward.feed_forward.3.weight"], state[p + "feed_forward.feed_forward.3.bias"])
        h = f + h
    h = F.layer_norm(h, (D,), state["encoder.layer_norm.weight"], state["encoder.layer_norm.bias"])
    return F.log_softmax(F.linear(h, state["classifier.weight"], state["classifier.bias"]), dim=2)
