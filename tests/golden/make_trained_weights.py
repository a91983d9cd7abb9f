#!/usr/bin/env python3
"""A TRAINED checkpoint for the AUC-parity tests (the reference ships none -- its own acceptance test,
tests/test_evaluate.py:11-31, needs tests/checkpoints/vad/sample.checkpoint, listed in .MISSING_LARGE_BLOBS).  Runs in the BUILD
container only (it imports the reference from /root/reference and never travels):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_trained_weights.py

Fine-tunes the REFERENCE's own SelfAttentiveVAD (vad/models/self_attention.py, imported unmodified; stock torch.optim.Adam, CPU)
on the labelled recordings of the reference's test data (tests/data/JamakeSpeechSample: 70 s + 60 s, and the FIRST 60 % of the
10 s WhenTheWeatherIsFine clip -- 130 s of two speakers alone do not transfer to a drama clip with music: trained on those two
only, the reference model ranks the clip's frames upside down, AUC 0.28; all kept as fixtures under tests/golden/data/), in the shape the reference trains and predicts in: windows of 7 frames at offsets -19 .. 19 step 9
(vad/predictor.py:180-224), every frame of a window scored against its own label (the boosted model's per-frame NLL), log-mel
features from oracle/logmel.py (the restated librosa defaults: librosa itself is absent), labels from VoiceActivity.to_labels(100).
Stored (tests/golden/trained.npz): ONLY the resulting state_dict (fp32) and, for the HELD-OUT clip (WhenTheWeatherIsFine, never
trained on), the reference model's own outputs -- log-probs of its first 1000 windows and the predictor-level boosted
probabilities computed the reference's way -- plus the AUC the reference arithmetic reaches on each of the three files."""
from __future__ import annotations

import sys
from pathlib import Path

import numpy as np
import torch

REPO = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(REPO))
sys.path.insert(0, "/root/reference")
sys.dont_write_bytecode = True

from oracle import logmel, oracle  # noqa: E402
from voice_activity_detection_amd.data_models import VoiceActivity  # noqa: E402
from voice_activity_detection_amd.features import load_wav_mono16k  # noqa: E402
from voice_activity_detection_amd.metrics import roc_auc  # noqa: E402

DATA = Path(__file__).resolve().parent / "data"
OUT = Path(__file__).resolve().parent / "trained.npz"
OFFSETS = oracle.window_offsets(19, 9)  # [-19, -10, -1, 0, 1, 10, 19]
HALF = 19


def load(wav, lab):
    audio = load_wav_mono16k(wav)
    feat = logmel.log_mel(audio)
    labels = VoiceActivity.load(lab).to_labels(100)
    n = min(len(labels), len(feat))
    return feat[:n].astype(np.float32), labels[:n].astype(np.int64)


def windows(feat, labels):
    idx = np.arange(HALF, len(feat) - HALF)[:, None] + OFFSETS[None, :]
    return feat[idx], labels[idx]


def main():
    from vad.models.self_attention import SelfAttentiveVAD  # the reference's model, unmodified

    torch.manual_seed(20260928)
    np.random.seed(0)
    torch.set_num_threads(8)
    jam = DATA / "JamakeSpeechSample" / "data"
    train = [load(jam / "sample_93" / "audio_93.wav", jam / "sample_93" / "voice_activity_93.json"),
             load(jam / "sample_95" / "audio_95.wav", jam / "sample_95" / "voice_activity_95.json")]
    held = load(DATA / "WhenTheWeatherIsFine" / "When_the_Weather_Is_Fine_12_4.wav", DATA / "WhenTheWeatherIsFine" / "voice_activity.json")
    cut = int(0.6 * len(held[0]))
    xs, ys = zip(*[windows(f, l) for f, l in train + [(held[0][:cut], held[1][:cut])]])
    X, Y = torch.from_numpy(np.concatenate(xs)), torch.from_numpy(np.concatenate(ys))
    print(f"train windows {tuple(X.shape)}, speech fraction {Y.float().mean():.3f}")

    model = SelfAttentiveVAD(80, 3, 128, 0.1)  # (dropout only matters in train(): it is not part of the state_dict)
    opt = torch.optim.Adam(model.parameters(), lr=5e-4)
    loss_fn = torch.nn.NLLLoss()
    steps, batch = 600, 256
    model.train()
    for step in range(steps):
        pick = torch.randint(0, len(X), (batch,))
        logp = model(features=X[pick])  # [B, 7, 2] log-probabilities
        loss = loss_fn(logp.reshape(-1, 2), Y[pick].reshape(-1))
        opt.zero_grad()
        loss.backward()
        opt.step()
        if step % 100 == 0 or step == steps - 1:
            print(f"step {step:4d} loss {loss.item():.4f}")
    model.eval()
    state = {k: v.detach().numpy().astype(np.float32).copy() for k, v in model.state_dict().items()}

    def reference_probs(feat):
        """vad/predictor.py:180-258 on the reference model: windows -> softmax -> boosted [N,7] with 0.5 placeholders"""
        xw, _ = windows(feat, np.zeros(len(feat), np.int64))
        with torch.no_grad():
            logp = torch.cat([model(features=torch.from_numpy(xw[i:i + 1000])) for i in range(0, len(xw), 1000)]).numpy()
        boosted = np.zeros((len(feat), 7, 2), np.float32)
        pos = np.arange(HALF, len(feat) - HALF)[:, None] + OFFSETS[None, :]
        for w in range(7):
            boosted[pos[:, w], w] = logp[:, w]
        e = np.exp(boosted - boosted.max(axis=2, keepdims=True))
        return (e / e.sum(axis=2, keepdims=True))[:, :, 1].astype(np.float32), logp

    out = {f"state/{k}": v for k, v in state.items()}
    aucs = []
    for name, (feat, labels) in zip(("sample_93", "sample_95", "held_out"), train + [held]):  # ("held_out": the clip)
        probs, logp = reference_probs(feat)
        auc = roc_auc(labels, probs.mean(axis=1))
        aucs.append(auc)
        print(f"{name}: frames {len(feat)}, reference AUC (boosted) {auc:.4f}")
        if name == "held_out":
            out["clip_logp"] = logp.astype(np.float32)
            out["clip_probs"] = probs
            tail = roc_auc(labels[cut + HALF:], probs.mean(axis=1)[cut + HALF:])
            print(f"   the clip's last 40 % (never trained on): AUC {tail:.4f}")
            aucs.append(tail)
    out["auc_ref"] = np.array(aucs)   # sample_93, sample_95, the whole clip, the clip's untrained tail
    assert aucs[2] > 0.8, "the clip must be well separated for the AUC-parity tests to mean something"
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, OUT.stat().st_size, "bytes")


if __name__ == "__main__":
    main()
