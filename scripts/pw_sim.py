#!/usr/bin/env python3
"""Runs the generated instruction stream of attention_pw_kernel_bf16 (scripts/gen_attn_pw.py) on the functional gfx950
model of scripts/gfx950_sim.py and compares the context it stores with an fp64 attention -- test infrastructure.

    python scripts/pw_sim.py B T [--grid G] [--scale S] [--inc FILE] [--seed N] [--nan-pad]

Inputs are built directly in the kernel's HBM layout (savad_kernels_bf16.h: fragment-major q / k / v^T / ctx in block
space): q is taken as already scaled by log2(e) / sqrt(D), so the reference is softmax in base 2 of q k^T.
"""
from __future__ import annotations

import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent))
import gfx950_sim as sim  # noqa: E402

D = 128
ROOT = Path(__file__).resolve().parent.parent
INC = ROOT / "voice_activity_detection_amd" / "csrc" / "savad_attn_pw_bf16.inc"
OPERANDS = {"%0": "s[0:1]", "%1": "s[2:3]", "%2": "s[4:5]", "%3": "s[6:7]", "%4": "s8", "%5": "s9", "%6": "s10", "%7": "s11",
            "%8": "s12", "%9": "s13", "%10": "s14", "%11": "s15", "%12": "s16", "%13": "s17", "%14": "s18", "%15": "v240",
            "%16": "s[20:21]"}


def feat_of(f, h, e):
    """feature held by element e of lane (., h) in K-step fragment f of a q / k / ctx block"""
    return 16 * f + 8 * (e >> 2) + 4 * h + (e & 3)


def pack_rows(x):
    """x [32 rows][128] fp32 -> 8 KiB block image: [ks 8][lane 64][8 bf16]"""
    out = np.zeros((8, 64, 8), dtype=np.uint16)
    for f in range(8):
        for h in range(2):
            for e in range(8):
                out[f, 32 * h:32 * h + 32, e] = sim.bf16_round(x[:, feat_of(f, h, e)]).astype(np.uint16)
    return out.reshape(-1).view(np.uint8)


def pack_vt(v):
    """v [32 keys][128] -> V^T block image: [nbd 4][j 2][lane (feature i, h)][8 keys]: key 16 j + 8 (e >> 2) + 4 h + (e & 3)"""
    out = np.zeros((4, 2, 64, 8), dtype=np.uint16)
    for nbd in range(4):
        for j in range(2):
            for h in range(2):
                for e in range(8):
                    key = 16 * j + 8 * (e >> 2) + 4 * h + (e & 3)
                    out[nbd, j, 32 * h:32 * h + 32, e] = sim.bf16_round(v[key, 32 * nbd:32 * nbd + 32]).astype(np.uint16)
    return out.reshape(-1).view(np.uint8)


def unpack_rows(img):
    """8 KiB block image -> [32 rows][128] fp32"""
    a = np.frombuffer(bytes(img), dtype=np.uint16).reshape(8, 64, 8)
    x = np.zeros((32, D), dtype=np.float32)
    for f in range(8):
        for h in range(2):
            for e in range(8):
                x[:, feat_of(f, h, e)] = sim.bf16_to_f32(a[f, 32 * h:32 * h + 32, e].astype(np.uint32))
    return x


def simulate(B, T, grid=8, scale=0.3, seed=0, inc_text=None, nan_pad=False, strict=True, verbose=False):
    """-> dict(ctx=[B][T][128] as the kernel stored it, ref=[B][T][128] fp64 attention, pad_ok, stats)"""
    assert grid % 8 == 0
    rng = np.random.default_rng(seed)
    QB = (T + 31) // 32
    nblk = B * QB
    q = (rng.standard_normal((B, QB * 32, D)) * scale).astype(np.float32)
    k = rng.standard_normal((B, QB * 32, D)).astype(np.float32)
    v = rng.standard_normal((B, QB * 32, D)).astype(np.float32)
    # slots past T: finite garbage in k / v^T (the producers keep them finite), anything in q
    if nan_pad and T % 32:
        q[:, T:, :] = np.nan
    base = 1 << 16
    size = nblk * 8192
    gmem = np.zeros(base + 4 * size + (1 << 16), dtype=np.uint8)
    qa, ka, va, ca = base, base + size, base + 2 * size, base + 3 * size
    for b in range(B):
        for qb in range(QB):
            blk = b * QB + qb
            rows = slice(32 * qb, 32 * qb + 32)
            gmem[qa + blk * 8192:qa + (blk + 1) * 8192] = pack_rows(q[b, rows])
            gmem[ka + blk * 8192:ka + (blk + 1) * 8192] = pack_rows(k[b, rows])
            gmem[va + blk * 8192:va + (blk + 1) * 8192] = pack_vt(v[b, rows])
    gmem[ca:ca + size] = 0xFF   # NaN-filled context: every slot must be written
    text = inc_text if inc_text is not None else INC.read_text()
    instrs, labels = sim.parse_program(text, OPERANDS)
    NGF = QB >> 3
    stride = grid >> 3
    stats = []
    t0 = time.time()
    for bid in range(grid):
        xcd, j = bid & 7, bid >> 3
        bi0 = g0 = dq = dr = 0
        if NGF:
            bi0, g0, dq, dr = j // NGF, j % NGF, stride // NGF, stride % NGF
        wg = sim.Workgroup(instrs, labels, gmem, strict=strict)
        for w in wg.waves:
            for reg, val in ((0, qa), (2, ka), (4, va), (6, ca)):
                w.s[reg], w.s[reg + 1] = val & 0xFFFFFFFF, val >> 32
            for reg, val in ((8, B), (9, T), (10, xcd), (11, j), (12, stride), (13, bi0), (14, g0), (15, dq), (16, dr), (17, w.wid), (18, 0)):
                w.s[reg] = val
            w.v[240] = (np.arange(64) * 16).astype(np.uint32)
            # everything else starts as NaN patterns: nothing may depend on a register's previous content
            w.v[:240] = 0x7FC12345
            w.a[:] = 0x7FC54321
        wg.run()
        stats.append({"bid": bid, "instr": [w.icount for w in wg.waves], "mfma": [w.counts.get("v_mfma_f32_32x32x16_bf16", 0) for w in wg.waves],
                      "barriers": [w.counts.get("s_barrier", 0) for w in wg.waves], "hazards": wg.hazards,
                      "labels": {k: v for k, v in wg.label_hits.items() if "cold" in k or "ksfirst" in k or k.endswith("_item")}})
        if verbose:
            print(f"  wg {bid}: instr {stats[-1]['instr']} mfma {stats[-1]['mfma']} barriers {stats[-1]['barriers'][0]} ({time.time() - t0:.1f} s)", flush=True)
    # decode + reference
    ctx = np.zeros((B, QB * 32, D), dtype=np.float32)
    for blk in range(nblk):
        b, qb = divmod(blk, QB)
        ctx[b, 32 * qb:32 * qb + 32] = unpack_rows(gmem[ca + blk * 8192:ca + (blk + 1) * 8192])
    qb16 = sim.bf16_to_f32(sim.bf16_round(q)).astype(np.float64)
    kb16 = sim.bf16_to_f32(sim.bf16_round(k)).astype(np.float64)
    vb16 = sim.bf16_to_f32(sim.bf16_round(v)).astype(np.float64)
    ref = np.zeros((B, T, D))
    for b in range(B):
        s = qb16[b, :T] @ kb16[b, :T].T
        p = np.exp2(s - s.max(axis=1, keepdims=True))
        ref[b] = (p @ vb16[b, :T]) / p.sum(axis=1, keepdims=True)
    pad = ctx[:, T:, :]
    return {"ctx": ctx[:, :T, :], "ref": ref, "pad_zero": bool((pad == 0).all()), "stats": stats,
            "finite": bool(np.isfinite(ctx[:, :T, :]).all())}


def main():
    args = sys.argv[1:]

    def opt(name, default, cast):
        if name in args:
            i = args.index(name)
            val = cast(args[i + 1])
            del args[i:i + 2]
            return val
        return default

    grid = opt("--grid", 8, int)
    scale = opt("--scale", 0.3, float)
    seed = opt("--seed", 0, int)
    inc = opt("--inc", None, str)
    nan_pad = "--nan-pad" in args
    if nan_pad:
        args.remove("--nan-pad")
    lax = "--lax" in args
    if lax:
        args.remove("--lax")
    B, T = int(args[0]), int(args[1])
    t0 = time.time()
    r = simulate(B, T, grid=grid, scale=scale, seed=seed, inc_text=Path(inc).read_text() if inc else None, nan_pad=nan_pad,
                 strict=not lax, verbose=True)
    err = np.abs(r["ctx"] - r["ref"])
    per_block = err.reshape(B, -1).max(axis=1)
    print(f"B={B} T={T} grid={grid}: max |ctx - ref| = {err.max():.4f} (per sequence {np.round(per_block, 4).tolist()}), "
          f"finite={r['finite']} pad rows zero={r['pad_zero']}  [{time.time() - t0:.1f} s]")
    hits = {}
    for s in r["stats"]:
        for h in s["hazards"]:
            print("HAZARD", s["bid"], h)
        for k, v in s["labels"].items():
            hits[k] = hits.get(k, 0) + v
    print("  label hits (waves):", dict(sorted(hits.items())))
    QB = (T + 31) // 32
    worst = err.max(axis=2).reshape(B, T)
    for b in range(B):
        blocks = [float(worst[b, 32 * qb:min(T, 32 * qb + 32)].max()) for qb in range(QB)]
        print(f"  seq {b}: per query block {np.round(blocks, 3).tolist()}")


if __name__ == "__main__":
    main()
