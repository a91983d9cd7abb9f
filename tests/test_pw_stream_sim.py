"""The generated instruction stream of attention_pw_kernel_bf16 (scripts/gen_attn_pw.py -> csrc/savad_attn_pw_bf16.inc) run on
the functional gfx950 model of scripts/gfx950_sim.py: CPU-side check of the stream's LOGIC -- item cursors (a sequence's tail
item in front of its first full group), ring addresses, counted waits, barrier pairing, key-split tail items and their
combine, reference moves -- against an fp64 attention on inputs built in the kernel's HBM layout.  The model raises on a read of
LDS bytes whose DMA piece is not landed-and-published, on a register read while its load is outstanding, on a write over
LDS another wave read in the same barrier epoch, and on unequal barrier counts.  (Timing hazards -- wait states -- are not
modelled; the GPU suite covers the same shapes on the chip.)"""
import sys
from pathlib import Path

import numpy as np
import pytest

sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "scripts"))

# (B, T, grid, scale, nan_pad): full groups + key-split tail of one block (264, 290) and of two (48, 33: sequences without a full
# group), an ordinary ragged tail item with idle waves (65), several items per workgroup and a stride of two (grid 16),
# scores that outrun their reference on most tiles (scale 3)
CASES = [(2, 264, 8, 0.3, False), (2, 48, 8, 0.3, True), (5, 33, 8, 0.3, True), (3, 65, 8, 0.3, False), (9, 264, 8, 3.0, True),
         (18, 290, 16, 0.3, False)]


@pytest.mark.parametrize("B,T,grid,scale,nan_pad", CASES)
def test_generated_stream_on_the_functional_model(B, T, grid, scale, nan_pad):
    import pw_sim

    r = pw_sim.simulate(B, T, grid=grid, scale=scale, seed=B + T, nan_pad=nan_pad, strict=True)
    assert r["finite"] and r["pad_zero"]          # every row written, rows past T exactly zero
    err = np.abs(r["ctx"] - r["ref"]).max()
    assert err < 0.03, err                         # bf16 probabilities and a bf16 context against fp64: ~0.015 here
    hits = {}
    for s in r["stats"]:
        assert len(set(s["barriers"])) == 1       # the four waves of a workgroup pass the same barriers
        for k, v in s["labels"].items():
            hits[k] = hits.get(k, 0) + v
    QB = (T + 31) // 32
    if QB % 8 in (1, 2):
        assert hits.get(".Lpw_ks_item", 0) == 4 * B        # one key-split item per sequence, four waves each
    if scale > 1:
        assert hits.get(".Lpw_coldmid_0_0_1", 0) > 0       # the key-split waves moved their references too


@pytest.mark.parametrize("B,T,grid,scale,nan_pad", [CASES[0], CASES[3], CASES[4]])
def test_packed_row_sum_variant_on_the_functional_model(tmp_path, B, T, grid, scale, nan_pad):
    """`gen_attn_pw.py --rowsum pk` (an experiment that waits for its GPU A/B: a tile's row sum as 7 v_pk_add_f32 + 1 add instead of 15
    sequential adds, 14 VALU instructions less per wave and step): the same checks as the product stream, and fewer instructions"""
    import subprocess

    import pw_sim

    inc = tmp_path / "pk.inc"
    subprocess.run([sys.executable, str(Path(pw_sim.__file__).with_name("gen_attn_pw.py")), "--rowsum", "pk", "--out", str(inc)], check=True)
    text = inc.read_text()
    assert text.count("v_pk_add_f32") > 200
    r = pw_sim.simulate(B, T, grid=grid, scale=scale, seed=B + T, nan_pad=nan_pad, strict=True, inc_text=text)
    assert r["finite"] and r["pad_zero"]
    assert np.abs(r["ctx"] - r["ref"]).max() < 0.03
    base = pw_sim.simulate(B, T, grid=grid, scale=scale, seed=B + T, nan_pad=nan_pad, strict=True)
    assert np.abs(r["ctx"] - base["ctx"]).max() < 4e-3          # one bf16 ulp of a context value at most: only the summation order moved
    assert sum(sum(s["instr"]) for s in r["stats"]) < sum(sum(s["instr"]) for s in base["stats"])


@pytest.mark.parametrize("B,T,grid,scale,nan_pad", [CASES[0], CASES[1], CASES[4]])
def test_no_split_stream_on_the_functional_model(B, T, grid, scale, nan_pad):
    """csrc/savad_attn_pw_bf16_nosplit.inc (savad_set_batch_invariant: every tail group an ORDINARY item, no key-split combine): the same
    checks, no key-split item ever entered, and -- outside the tail rows, whose summation order is the point of the variant -- the
    product stream's bits."""
    import pw_sim

    text = (Path(__file__).resolve().parent.parent / "voice_activity_detection_amd" / "csrc" / "savad_attn_pw_bf16_nosplit.inc").read_text()
    r = pw_sim.simulate(B, T, grid=grid, scale=scale, seed=B + T, nan_pad=nan_pad, strict=True, inc_text=text)
    assert r["finite"] and r["pad_zero"]
    assert np.abs(r["ctx"] - r["ref"]).max() < 0.03
    assert not any("_ks" in k for s in r["stats"] for k in s["labels"])
    base = pw_sim.simulate(B, T, grid=grid, scale=scale, seed=B + T, nan_pad=nan_pad, strict=True)
    full = 256 * (((T + 31) // 32) // 8)          # frames of a sequence that full groups cover
    if full:
        a, b = r["ctx"].reshape(B, -1, r["ctx"].shape[-1]), base["ctx"].reshape(B, -1, base["ctx"].shape[-1])
        assert np.array_equal(a[:, :full], b[:, :full])
    assert np.abs(r["ctx"] - base["ctx"]).max() <= 2.0 ** -7 * np.abs(base["ctx"]).max()   # the tail rows: the bf16 rounding of the context's range
