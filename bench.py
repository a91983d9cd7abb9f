#!/usr/bin/env python3
"""bench.py -- audio frames/sec of the self-attentive-VAD forward pass on MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch of synthetic mel frames: forward of
SelfAttentiveVAD on a device-resident [B, T, F] tensor -> device-resident [B, T, 2] log-probs
(N > 1: every rank runs its own B-sequence shard through voice_activity_detection_amd.distributed.ShardedPipeline, which issues every
collective of the run: by default (--gather final, what `value` is quoted on) every forward writes into a [K,B,T,2] send buffer and ONE
RCCL all_gather closes the K-step block -- north_star's "single RCCL gather at the end"; --gather step issues one all_gather of [B,T,2]
per forward, lagging behind the forwards in flight.  Both are measured on every N > 1 run.  `python bench.py --gpus N` without a
launcher starts N ranks itself (torch.distributed.run, one per GPU).  `--backend gloo --stub-forward` runs the same control
flow on CPU ranks with a stand-in forward: a dry run for tests/test_dist_gloo.py, never a measurement.)
Default workload = BASELINE.json configs[1]: [32, 800, 80] fp32 per GPU, seeded weights.

The K batches of a block are independent, so up to three of them are kept IN FLIGHT (voice_activity_detection_amd.PipelinedVAD: one HIP
stream, library handle and workspace per forward in flight; the same bits as one at a time): the warm-up times 1, 2 and 3 in flight and
the timed blocks use the fastest (--in-flight N fixes it).  Every step's work is complete inside its block (join + synchronize before the
clock stops).  The one-at-a-time figure is printed beside `value` (value_one_forward / ms_one_forward), and the per-kernel roofline block is
measured with ONE forward in flight, so that a launch's duration is the kernel's own.

Timing: W warm-up steps (at least 0.2 s of them, so the clocks have ramped), then BLOCKS
of exactly K steps, each bracketed by barrier + torch.cuda.synchronize() on both sides and by a pair of HIP events
on the launch stream, repeated until at least --min-seconds (0.5 s) of timed work has run (never fewer than 5
blocks).  `value` / `ms_per_step` come from the MEDIAN block (max over ranks); min and the HIP-event median are
reported beside it.  Prints ONE JSON line on rank 0 (contract: see the task statement).
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import re
import statistics
import sys
import time
from pathlib import Path

import numpy as np
import torch

REPO = Path(__file__).resolve().parent
sys.path.insert(0, str(REPO))

F_MEL, D_MODEL, N_LAYERS = 80, 128, 3
PEAK_FP32_MFMA_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense, spec
PEAK_BF16_MFMA_TFLOPS = 2500.0  # same guide: bf16 MFMA, dense (AMD's 5 PF figure includes 2:1 sparsity)
PEAK_HBM_TBPS = 8.0  # same guide: HBM3E


def flops_per_frame(T: int) -> float:
    """2FD + L(8D^2 + 4TD + 4D*d_ff) + 4D."""
    F, D, L = F_MEL, D_MODEL, N_LAYERS
    return 2 * F * D + L * (8 * D * D + 4 * T * D + 4 * D * 4 * D) + 4 * D


# Algorithmic work of one launch of each kernel of the forward (per frame unless noted):
# attention 4*T*D FLOP (QK^T + PV), row chain 393 216 FLOP (out-proj 32 768 + FFN 262 144 + next QKV 98 304), last row
# chain 295 424 (classifier instead of QKV), input stage 118 784.  Bytes: what the launch must read + write in HBM.
def launch_work(name: str, B: int, T: int, e: int):
    """name = launch label reported by the library -> (FLOP, algorithmic HBM bytes) of one launch; e = bytes per
    element of q/k/v/context (4 fp32, 2 bf16, 6 fp32s: three bf16 pieces).  The residual stream is fp32 in the fp32 and fp32s paths,
    fp16 in the bf16 path.  FLOPs are ALGORITHMIC (fp32s issues six bf16 MFMA products per algorithmic multiply-add)."""
    D, F = D_MODEL, F_MEL
    frames = B * T
    att = 4.0 * T * T * D * B
    hres = 2 if e == 2 else 4
    base = name.replace("_bf16", "").replace("_f32s", "")
    if base == "attention":
        return att, frames * 4 * D * e                                   # Q, K, V read + context written
    if base == "attention_row":
        return att + 393216.0 * frames, frames * (3 * D * e + D * hres) * 2   # q, k, v, h read; h, q', k', v' written
    if base == "attention_row_last":
        return att + 295424.0 * frames, frames * (3 * D * e + D * hres + 8)
    if base == "row":
        return 393216.0 * frames, frames * (D * e + 8 + D * hres + D * hres + 3 * D * e)
    if base == "row_last":
        return 295424.0 * frames, frames * (D * e + 8 + D * hres + 8)
    if base == "input_qkv":
        return 118784.0 * frames, frames * (F * (4 if e == 6 else e) + D * hres + 3 * D * e)
    if base == "packed_forward":  # whole forward in one launch (T <= 32): features in, log-probs out
        return flops_per_frame(T) * frames, frames * (F * 4 + 8)   # (the windows are fp32 features in both precisions)
    return 0.0, 0


def kernel_source_hash() -> str:
    """sha256 over the kernel sources: PMC traffic numbers under profiles/ are only quoted for the code they were
    measured on (scripts/summarize_profile.py stamps the same hash into *_traffic.json)."""
    h = hashlib.sha256()
    for f in sorted((REPO / "voice_activity_detection_amd" / "csrc").glob("*")):
        if f.suffix in (".h", ".hip", ".inc"):
            h.update(f.name.encode())
            h.update(f.read_bytes())
    return h.hexdigest()[:16]


def measured_traffic(name: str, precision: str, B: int, T: int):
    """HBM-side bytes per launch of kernel `name` from the committed rocprofv3 PMC passes
    (profiles/*_traffic.json, produced by scripts/profile_gpu.sh; FETCH_SIZE doubled per the gfx950
    correction of MI355X_MICROARCH.md).  PMC counters cannot be collected from inside this process, so
    the number is only reported when (a) the workload matches a profiled one -- fp32 [32,800] (the default
    bench line) or bf16 [256,800] -- and (b) the profile was taken on exactly the kernel sources that are
    running now (`csrc_hash` stored in the file); otherwise null."""
    if (precision, B, T) == ("fp32", 32, 800):
        files = [f for f in sorted((REPO / "profiles").glob("*_traffic.json")) if not any(t in f.name for t in ("bf16", "t7", "t50", "logmel", "fp32s"))]
    elif (precision, B, T) == ("fp32s", 32, 800):
        files = [f for f in sorted((REPO / "profiles").glob("*fp32s_traffic.json")) if "t7" not in f.name]
    elif (precision, B, T) == ("fp32s", 1000, 7):
        files = sorted((REPO / "profiles").glob("*t7_fp32s_traffic.json"))
    elif (precision, B, T) == ("bf16", 256, 800):
        files = sorted((REPO / "profiles").glob("*bf16_b256_traffic.json"))
    elif (precision, B, T) == ("fp32", 1000, 7):
        files = sorted((REPO / "profiles").glob("*t7_traffic.json"))
    elif (precision, B, T) == ("bf16", 1000, 7):
        files = sorted((REPO / "profiles").glob("*t7_bf16_traffic.json"))
    else:
        return None
    want = kernel_source_hash()
    files = [f for f in files if json.loads(f.read_text()).get("csrc_hash") == want]
    if not files:
        return None
    data = json.loads(files[-1].read_text())
    last = name.endswith("_last") or name.endswith("_last_bf16")
    stem = name.replace("_last", "")
    prefix = {"attention_row_f32s": "attention_row_kernel_f32s", "input_qkv_f32s": "input_qkv_kernel_f32s",
              "packed_forward_f32s": "packed_forward_kernel_f32s", "attention": "attention_kernel", "attention_row": "attention_row_kernel", "row": "row_kernel", "input_qkv": "input_qkv_kernel",
              "attention_bf16": "attention", "row_bf16": "row_kernel_bf16", "input_qkv_bf16": "input_qkv_kernel_bf16",
              "attention_row_bf16": "attention_row_kernel_bf16", "packed_forward": "packed_forward_kernel",
              "packed_forward_bf16": "packed_forward_kernel_bf16"}.get(stem)
    if prefix is None:
        return None
    for key, entry in data.items():
        if not isinstance(entry, dict):
            continue
        if key == prefix or key.startswith(prefix + "<") or key.startswith(prefix + "_m") or key.startswith(prefix + "_ns") or (stem == "attention_bf16" and key.startswith("attention") and "bf16" in key and "row" not in key):
            if "<" in key and prefix.startswith(("attention_row", "row")):
                if ("<true" in key) != last:
                    continue
            return round(entry["hbm_bytes_per_launch"])
    return None


def traffic_note(name, precision, B, T):
    """what the measured HBM-side bytes of the dominant kernel are made of, where they are far from the algorithmic ones"""
    if name in ("packed_forward", "packed_forward_bf16"):
        return ("the whole forward in one launch reads x and writes the log-probs (algorithmic) -- and every one of the 8 XCD L2s fetches the "
                "2.4 MB of packed weights once: 8 x 2.4 MB + x is the measured figure, 0.3 % of the HBM roof at this launch's duration; benign")
    if name == "attention_bf16" and (B, T) == (256, 800):
        return ("persistent kernel: q and ctx once, K / V^T once for a sequence's three full groups that run side by side in one XCD's L2, once more "
                "for the key-split tail item + the full group behind it, which run 0.55 item-times later than their siblings (DESIGN section 4b-3)")
    return None


def gpu_state():
    """sclk / mclk / power / power cap of this rank's GPU from rocm-smi (None when rocm-smi is missing or slow): recorded
    at the start and the end of every leg, so that a box-to-box or leg-to-leg clock difference is visible in the line."""
    import shutil
    import subprocess

    exe = shutil.which("rocm-smi") or "/opt/rocm/bin/rocm-smi"
    if not os.path.exists(exe) or os.environ.get("RANK", "0") != "0":  # (rank 0 reports; eight ranks polling would only add noise)
        return None
    try:
        out = subprocess.run([exe, "--showclocks", "--showpower", "--showmaxpower", "--json"], capture_output=True, text=True, timeout=8).stdout
        card = json.loads(out)
        card = card[sorted(card)[int(os.environ.get("LOCAL_RANK", "0")) % len(card)]]
    except Exception:
        return None
    keep = {}
    for k, v in card.items():  # e.g. "sclk clock speed:": "(2163Mhz)", "Current Socket Graphics Package Power (W)": "831.0"
        kl = k.lower()
        m = re.search(r"([0-9.]+)", str(v))
        if not m:
            continue
        if "clock speed" in kl:
            keep[kl.split()[0] + "_mhz"] = int(float(m.group(1)))
        elif "power" in kl:
            keep[("power_cap_w" if "max" in kl else "power_w")] = float(m.group(1))
    return keep or None


def rocprof_averages(precision, B, T):
    """{kernel short name: average ns} of the committed rocprofv3 --kernel-trace --stats run of this workload
    (profiles/*_kernel_avg.json, written by scripts/summarize_profile.py), only when taken on the running kernel sources."""
    if (precision, B, T) == ("fp32", 32, 800):
        files = [f for f in sorted((REPO / "profiles").glob("*_kernel_avg.json")) if not any(t in f.name for t in ("bf16", "_t7_", "_t50_", "logmel", "fp32s"))]
    elif (precision, B, T) == ("fp32s", 32, 800):
        files = [f for f in sorted((REPO / "profiles").glob("*fp32s_kernel_avg.json")) if "t7" not in f.name]
    elif (precision, B, T) == ("fp32s", 1000, 7):
        files = sorted((REPO / "profiles").glob("*t7_fp32s_kernel_avg.json"))
    elif (precision, B, T) == ("bf16", 256, 800):
        files = [f for f in sorted((REPO / "profiles").glob("*bf16_kernel_avg.json")) if "t7" not in f.name]
    elif (precision, B, T) == ("fp32", 1000, 7):
        files = sorted((REPO / "profiles").glob("*t7_kernel_avg.json"))
    elif (precision, B, T) == ("bf16", 1000, 7):
        files = sorted((REPO / "profiles").glob("*t7_bf16_kernel_avg.json"))
    else:
        return {}, None
    want = kernel_source_hash()
    for f in reversed(files):
        try:
            data = json.loads(f.read_text())
        except Exception:
            continue
        if data.get("csrc_hash") == want:
            return data.get("kernels", {}), f.name
    return {}, None


ROCPROF_NAMES = {  # bench launch label -> kernel short name in the rocprofv3 stats (fp32 M-split / fused regime, bf16 4-wave kernels)
    "attention_row_f32s": "attention_row_kernel_f32s<false, false>", "attention_row_last_f32s": "attention_row_kernel_f32s<true, false>",
    "input_qkv_f32s": "input_qkv_kernel_f32s", "packed_forward_f32s": ("packed_forward_kernel_f32s_ns", "packed_forward_kernel_f32s"),
    "attention_row": "attention_row_kernel<false>", "attention_row_last": "attention_row_kernel<true>",
    "input_qkv": "input_qkv_kernel_m", "packed_forward": "packed_forward_kernel",
    "attention_bf16": ("attention_pw_kernel_bf16", "attention_kernel_bf16<4>"), "row_bf16": "row_kernel_bf16<false, 4>", "row_last_bf16": "row_kernel_bf16<true, 4>",
    "input_qkv_bf16": ("input_qkv_kernel_bf16_p<__bf16, 8, 5>", "input_qkv_kernel_bf16<__bf16, 4>"),
    "packed_forward_bf16": ("packed_forward_kernel_bf16_ns", "packed_forward_kernel_bf16<4, 4, 4>", "packed_forward_kernel_bf16<4, 2, 0>"),
}


def cpu_baseline(state, B: int, T: int, seconds: float):
    """The reference's CPU path cannot travel; time its two stand-ins on this host's cores on the SAME workload
    shape ([B, T, 80] fp32, inputs default_rng(0).uniform(-13.8, 4.2): BASELINE.md section 3): (a) the stock-PyTorch
    port (the reference's ATen ops), (b) the C oracle (OpenMP, one sequence per thread).  Each: 2 warm-ups, then
    passes until `seconds`/2 of them have run (never fewer than 3, never more than 30); median and min reported, the
    faster median is `value`.  Bounded to ~`seconds` + the thread-count probe so that the GPU legs dominate the run."""
    from oracle import oracle, torch_port

    cores = os.cpu_count() or 1
    st = {k: torch.from_numpy(v) for k, v in state.items()}
    xn = np.random.default_rng(0).uniform(-13.8, 4.2, (B, T, F_MEL)).astype(np.float32)
    x = torch.from_numpy(xn)

    def passes(fn, budget):
        fn()
        fn()
        ts, t_start = [], time.perf_counter()
        while len(ts) < 3 or (time.perf_counter() - t_start < budget and len(ts) < 30):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        return ts

    # (a) stock-PyTorch port: pick the intra-op thread count with one pass each (oversubscription hurts), then time
    cands = sorted({c for c in (8, 32, cores) if c <= cores})
    best_nt, best_t = cands[0], float("inf")
    for nt in cands:
        torch.set_num_threads(nt)
        torch_port.forward(st, x)
        t0 = time.perf_counter()
        torch_port.forward(st, x)
        dt = time.perf_counter() - t0
        if dt < best_t:
            best_nt, best_t = nt, dt
    torch.set_num_threads(best_nt)
    tt = passes(lambda: torch_port.forward(st, x), seconds / 2)
    # (b) C oracle
    tc = passes(lambda: oracle.forward(state, xn, threads=cores), seconds / 2)
    frames = B * T
    t_med, t_min, c_med, c_min = statistics.median(tt), min(tt), statistics.median(tc), min(tc)
    torch_fps, c_fps = frames / t_med, frames / c_med
    used = best_nt if torch_fps >= c_fps else min(cores, B)
    return {
        "value": round(max(torch_fps, c_fps), 1), "unit": "frames/s", "cores": used, "host_cores": cores, "kind": "port",
        "torch_port": {"frames_per_s_median": round(torch_fps, 1), "frames_per_s_best": round(frames / t_min, 1),
                       "ms_median": round(t_med * 1e3, 2), "ms_min": round(t_min * 1e3, 2), "passes": len(tt), "threads": best_nt,
                       "torch": torch.__version__},
        "c_oracle": {"frames_per_s_median": round(c_fps, 1), "frames_per_s_best": round(frames / c_min, 1),
                     "ms_median": round(c_med * 1e3, 2), "ms_min": round(c_min * 1e3, 2), "passes": len(tc),
                     "threads": min(cores, B)},
        "sample": f"[{B},{T},{F_MEL}] fp32 (the GPU workload's shape), 2 warm-ups + {len(tt)} / {len(tc)} timed passes of the "
                  f"stock-PyTorch CPU port (thread count picked from {cands}) / the C oracle (OpenMP); median of the faster stand-in reported",
    }


def stub_forward(x: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
    """--stub-forward: a stand-in with the forward's signature and output shape (log-softmax over the first two features) for
    CPU dry runs of this file's control flow under gloo (tests/test_dist_gloo.py).  Nothing measured with it is a result."""
    out.copy_(torch.log_softmax(x[..., :2].float(), dim=-1))
    return out


class Clock:
    """HIP events on the current stream when there is a GPU, the host clock in a --stub-forward dry run"""

    def __init__(self, cuda: bool):
        self.cuda = cuda

    def sync(self):
        if self.cuda:
            torch.cuda.synchronize()

    def pair(self):
        return (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) if self.cuda else [0.0, 0.0]

    def mark(self, pair, i):
        if self.cuda:
            pair[i].record()
        else:
            pair[i] = time.perf_counter()

    def seconds(self, pair):
        return pair[0].elapsed_time(pair[1]) * 1e-3 if self.cuda else pair[1] - pair[0]


class Runner:
    """One workload on this rank: model + resident input, driven through voice_activity_detection_amd.distributed.ShardedPipeline
    (rank-local forwards in flight + the RCCL gather of their log-probs; without a process group: the plain pipeline).  Every
    collective of a run is issued by that module -- this file only decides when."""

    def __init__(self, state, B, T, precision, dev, rank, world, gather, slots, row_mode=0, splits=0, in_flight=0, stub=False):
        from voice_activity_detection_amd import distributed as vdist

        self.vdist = vdist
        self.B, self.T, self.precision, self.dev, self.world, self.rank = B, T, precision, dev, world, rank
        self.clock = Clock(dev.type == "cuda")
        self.stub = stub
        # each rank gets its own shard of the global batch (weak scaling: B per GPU fixed)
        x = torch.from_numpy(np.random.default_rng(rank).uniform(-13.8, 4.2, (B, T, F_MEL)).astype(np.float32)).to(dev)
        # bf16 features for the long-sequence configs; the reference pipeline's windows (T <= 32) are cut out of an fp32 log-mel matrix
        self.x = x.to(torch.bfloat16) if (precision == "bf16" and not stub and T > 32) else x
        self.in_flight_request = in_flight
        if stub:
            self.model = None
            self.sp = vdist.ShardedPipeline(forward=stub_forward, slots=slots, depth=max(in_flight, 1) if in_flight else 3, gather=gather)
        else:
            from voice_activity_detection_amd import SelfAttentiveVAD

            model = SelfAttentiveVAD(F_MEL, N_LAYERS, D_MODEL, 0.5)
            model.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()})
            self.model = model.to(dev).eval()
            self.model.attention_splits, self.model.row_mode, self.model.precision = splits, row_mode, precision
            self.model.reserve(T)
            # consecutive batches are independent: up to 3 forwards in flight (own stream / handle / workspace each,
            # voice_activity_detection_amd/pipeline.py); in_flight = 0 picks the fastest of 1 / 2 / 3 for this shape during warm-up
            self.sp = vdist.ShardedPipeline(self.model, slots=slots, depth=max(in_flight, 1) if in_flight else 3, gather=gather)
            self.sp.pipe.reserve(T)
        self.dist = self.sp.distributed
        self.tuning = None
        self.last = None

    def set_gather(self, gather):
        """'step': one all_gather of [B,T,2] per forward; 'final': every forward writes straight into its slot of a [K,B,T,2]
        send buffer and ONE all_gather of all K batches closes the block (both: ShardedPipeline)"""
        self.drain()
        self.sp.set_gather(gather)

    def step(self):
        if self.sp.pending >= self.sp.slots:   # warm-up loops run longer than one block
            self.drain()
        self.sp.submit(self.x)

    def drain(self):
        outs = self.sp.join()
        if outs:
            self.last = outs[-1]           # [world, B, T, 2]

    def tune_in_flight(self, steps=0):
        """pick the number of forwards in flight for this shape: time `steps` steps at 1, 2, 3 (untimed warm-up work) and keep the
        fastest; a fixed --in-flight N skips it.  One decision per job (rank 0's clock): every rank must issue the same collectives."""
        depth = self.sp.depth
        if self.in_flight_request or depth == 1:
            self.drain()
            self.sp.set_in_flight(depth)
            self.tuning = {"fixed": self.in_flight_request or 1}
            return
        res = {}
        for d in range(1, depth + 1):
            self.drain()
            self.sp.set_in_flight(d)
            n = int(self.vdist.agree(steps or 40, self.dev))
            for _ in range(10):
                self.step()
            self.drain()
            self.clock.sync()
            t0 = time.perf_counter()
            for _ in range(10):
                self.step()
            self.drain()
            self.clock.sync()
            if not steps:  # ~60 ms per candidate, at least 40 steps -- the same count on every rank
                n = int(self.vdist.agree(max(40, int(0.06 / max((time.perf_counter() - t0) / 10, 1e-6))), self.dev))
            t0 = time.perf_counter()
            for _ in range(n):
                self.step()
            self.drain()
            self.clock.sync()
            res[d] = (time.perf_counter() - t0) / n * 1e3
        best = int(self.vdist.agree(min(res, key=res.get), self.dev))
        self.drain()
        self.sp.set_in_flight(best)
        self.tuning = {str(k): round(v, 4) for k, v in res.items()}

    def timed_blocks(self, steps, warmup, min_seconds, max_blocks=400):
        """-> (per-block wall seconds [max over ranks], per-block HIP-event seconds on this rank)"""
        vdist, clock = self.vdist, self.clock
        t0 = time.perf_counter()
        n = 0
        # at least W steps and at least ~0.2 s of them (the clocks ramp for the first ~100 ms of load); with a process
        # group every rank must issue the same number of collectives, so the count is fixed there instead of timed
        while n < max(warmup, 1) or (n < 10000 and ((not self.dist and not self.stub and time.perf_counter() - t0 < 0.2) or (self.dist and n < 40))):
            self.step()
            n += 1
            if n % 8 == 0:
                clock.sync()
        self.drain()
        clock.sync()
        if self.tuning is None:
            self.tune_in_flight()
            self.drain()
            clock.sync()
        walls, evs = [], []
        total = 0.0
        while len(walls) < 5 or (total < min_seconds and len(walls) < max_blocks):
            ev = clock.pair()
            vdist.barrier()
            clock.sync()
            t0 = time.perf_counter()
            clock.mark(ev, 0)
            for _ in range(steps):
                self.step()
            self.drain()
            clock.mark(ev, 1)
            clock.sync()
            dt = time.perf_counter() - t0
            vdist.barrier()
            walls.append(dt)
            evs.append(clock.seconds(ev))
            total = vdist.agree(total + dt, self.dev)   # every rank must take the same number of blocks: rank 0's clock decides
        return vdist.max_over_ranks(walls, self.dev), evs

    def delivered(self):
        """the last gathered batch is finite and holds, in this rank's slot, exactly what a forward of this rank's shard gives"""
        if self.last is None:
            return False
        y = torch.empty((self.B, self.T, 2), dtype=torch.float32, device=self.dev)
        with torch.no_grad():
            if self.stub:
                stub_forward(self.x, y)
            else:
                self.model(features=self.x, out=y)
        self.clock.sync()
        return bool(torch.isfinite(self.last).all().item()) and bool(torch.equal(self.last[self.rank], y))

    def kernel_profile(self, steps, ms_hint=None):
        """the same K steps with every kernel bracketed by HIP events on the launch stream -> [(label, ms)]"""
        if self.stub:
            return []
        # creating the events idles the GPU; the first ~15 ms of kernels afterwards run at lower clocks (230 -> 207 us per
        # fused launch, seen in the rocprofv3 trace): ~0.15 s of un-recorded forwards first
        settle = max(steps, int(0.15 / (ms_hint * 1e-3))) if ms_hint else 200
        self.drain()
        torch.cuda.synchronize()
        out = torch.empty((self.B, self.T, 2), dtype=torch.float32, device=self.dev)
        self.model.set_profiling(steps, skip=settle)
        with torch.no_grad():
            for _ in range(settle + steps):   # ONE forward in flight: a launch's duration is the kernel's own
                self.model(features=self.x, out=out)
        torch.cuda.synchronize()
        kt = self.model.kernel_times()
        self.model.set_profiling(0)
        return kt


    def kernel_profile_in_flight(self, steps, ms_hint=None):
        """the per-launch durations in the mode `value` is quoted in: `in_flight` forwards on their own streams, every replica's
        launches bracketed by HIP events on ITS stream -> {label: mean ms over replicas and launches} (stretched by the overlap)"""
        pipe = getattr(self.sp, "pipe", None)
        if self.stub or pipe is None or pipe.active < 2:
            return None
        settle = max(steps, int(0.15 / (ms_hint * 1e-3))) if ms_hint else 200
        self.drain()
        torch.cuda.synchronize()
        reps = pipe._replicas[:pipe.active]
        outs = [torch.empty((self.B, self.T, 2), dtype=torch.float32, device=self.dev) for _ in reps]
        for rep in reps:
            pipe._follow(rep)
            rep.set_profiling(steps, skip=settle)
        with torch.no_grad():
            for _ in range(settle + steps):
                for k in range(len(reps)):
                    pipe.submit(self.x, out=outs[k])
            pipe.join()
        torch.cuda.synchronize()
        acc = {}
        for rep in reps:
            for n, t in rep.kernel_times():
                acc.setdefault(n, []).append(t)
            rep.set_profiling(0)
        return {n: round(sum(ts) / len(ts), 4) for n, ts in acc.items()}


def roofline_block(ktimes, precision, B, T, ms_forward, profiled_shape=False, ms_one_forward=None):
    """precision "fp32s" computes the fp32 arithmetic on the bf16 matrix pipe: every algorithmic multiply-add is SIX bf16 MFMA products
    (three-piece operands).  Its `achieved` / `frac` are the ISSUED bf16 FLOPs (6 x algorithmic) against the 2.5 PF bf16 peak -- the roof
    that binds it -- and `fp32_equivalent_tflops` = algorithmic FLOPs / time, which may exceed the 157.3 TF fp32-MFMA peak: that is the
    point of the mode.  A launch that ran the exact-fp32 kernels (fp32s picks them for small T <= 32 batches) is priced as fp32."""
    exact = precision == "fp32s" and ktimes and not any(n.endswith("_f32s") for n, _ in ktimes)
    if exact:
        precision = "fp32"
    peak = PEAK_FP32_MFMA_TFLOPS if precision == "fp32" else PEAK_BF16_MFMA_TFLOPS
    issue = 6.0 if precision == "fp32s" else 1.0   # MFMA FLOPs issued per algorithmic FLOP
    e = {"bf16": 2, "fp32s": 6}.get(precision, 4)
    fwd_tflops = issue * flops_per_frame(T) * B * T / (ms_forward * 1e-3) / 1e12
    one_tflops = issue * flops_per_frame(T) * B * T / ((ms_one_forward or ms_forward) * 1e-3) / 1e12
    if not ktimes:
        return {"bound": "mfma", "forward_achieved": round(fwd_tflops, 2), "forward_frac": round(fwd_tflops / peak, 4),
                "forward_frac_one_forward": round(one_tflops / peak, 4), "peak": peak, "unit": "TFLOP/s"}
    by_name = {}
    for n, t in ktimes:
        by_name.setdefault(n, []).append(t)
    # the dominant kernel = the launch label with the largest total time
    dom = max(by_name, key=lambda n: sum(by_name[n]))
    dom_ms = sum(by_name[dom]) / len(by_name[dom])
    dom_flops, dom_bytes = launch_work(dom, B, T, e)
    ach = issue * dom_flops / (dom_ms * 1e-3) / 1e12
    per_kernel = {}
    prof_avg, prof_file = rocprof_averages(precision, B, T) if profiled_shape else ({}, None)
    for n, ts in by_name.items():
        fl, by = launch_work(n, B, T, e)
        ms_k = sum(ts) / len(ts)
        per_kernel[n] = {"launches": len(ts), "ms": round(ms_k, 4), "tflops": round(issue * fl / (ms_k * 1e-3) / 1e12, 2),
                         "frac": round(issue * fl / (ms_k * 1e-3) / 1e12 / peak, 4),
                         "hbm_frac": round(by / (ms_k * 1e-3) / 1e12 / PEAK_HBM_TBPS, 4)}
        cands = ROCPROF_NAMES.get(n, ())
        ref_ns = next((prof_avg[c] for c in ((cands,) if isinstance(cands, str) else cands) if c in prof_avg), None)
        if ref_ns:  # this run's HIP-event duration over the committed rocprofv3 average of the same kernel on the same sources
            per_kernel[n]["rocprof_ms"] = round(ref_ns * 1e-6, 4)
            per_kernel[n]["event_over_rocprof"] = round(ms_k / (ref_ns * 1e-6), 3)
    f32s = {}
    if precision == "fp32s":
        f32s = {"issued_over_algorithmic_flops": 6,
                "fp32_equivalent_tflops": round(dom_flops / (dom_ms * 1e-3) / 1e12, 2),
                "fp32_equivalent_over_fp32_mfma_peak": round(dom_flops / (dom_ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4),
                "forward_fp32_equivalent_tflops": round(fwd_tflops / 6, 2), "forward_fp32_equivalent_tflops_one_forward": round(one_tflops / 6, 2),
                "note": "achieved / frac / peak: ISSUED bf16 MFMA FLOPs (six products per algorithmic multiply-add) against the dense bf16 peak; "
                        "fp32_equivalent_*: algorithmic FLOPs / time -- above the 157.3 TF fp32-MFMA peak by design"}
    elif exact:
        f32s = {"note": "precision fp32s ran the exact-fp32 kernels at this shape (small T <= 32 batch): priced against the fp32 MFMA peak"}
    return {
        "bound": "mfma", "kernel": f"{dom} ({len(by_name[dom])} launches per forward)",
        "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4), **f32s,
        "traffic": measured_traffic(dom, precision, B, T),
        "traffic_unit": "bytes/launch (rocprofv3 PMC pass under profiles/, quoted only when its csrc_hash matches the running kernels)",
        "traffic_note": traffic_note(dom, precision, B, T),
        "algorithmic_bytes": dom_bytes, "algorithmic_flops": dom_flops,
        # the same launch against the HBM roofline: the non-binding one in fp32 -- the fp32 MFMA
        # rate caps the attention stage at 9.8 % of 8 TB/s
        "hbm_achieved_TBps": round(dom_bytes / (dom_ms * 1e-3) / 1e12, 3), "hbm_peak_TBps": PEAK_HBM_TBPS,
        "hbm_frac": round(dom_bytes / (dom_ms * 1e-3) / 1e12 / PEAK_HBM_TBPS, 4),
        "ms_per_launch": round(dom_ms, 4),
        # whole forward: all launches' FLOPs over ms_per_step (the throughput figure: with several forwards in flight, theirs)
        "forward_achieved": round(fwd_tflops, 2), "forward_frac": round(fwd_tflops / peak, 4),
        # ... and over the wall time of ONE forward at a time (comparable with the per-kernel fractions; SURVEY section 8d's figure)
        "forward_achieved_one_forward": round(one_tflops, 2), "forward_frac_one_forward": round(one_tflops / peak, 4),
        "per_kernel": per_kernel, "rocprof_reference": prof_file,
        "kernels_ms": {f"{i}:{n}": round(t, 4) for i, (n, t) in enumerate(ktimes)},
    }


def clip_pipeline(state, dev, seconds, min_seconds):
    """The reference pipeline's own case (BASELINE configs[0]): `seconds` of 16 kHz audio resident on the device ->
    log-mel [N,80] -> N-6 windows of 7 frames -> SelfAttentiveVAD -> boosted probabilities [N,7] (vad/predictor.py:
    159-262), everything on the current stream.  Median over blocks of 50 calls, HIP events."""
    from voice_activity_detection_amd import SelfAttentiveVAD
    from voice_activity_detection_amd.features import log_mel
    from voice_activity_detection_amd.predictor import VADFromScratchPredictor

    model = SelfAttentiveVAD(F_MEL, N_LAYERS, D_MODEL, 0.5)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()})
    pred = VADFromScratchPredictor(model.to(dev).eval(), dev)
    audio = torch.from_numpy(np.random.default_rng(7).normal(0.0, 0.1, int(16000 * seconds)).astype(np.float32)).to(dev)

    def chain():
        return pred.predict_probabilities_device(log_mel(audio, dev))

    for _ in range(30):
        probs, _ = chain()
    torch.cuda.synchronize()
    per_call, spent = [], 0.0
    while len(per_call) < 5 or spent < min_seconds:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            chain()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        per_call.append(ms / 50)
        spent += ms * 1e-3
    med = statistics.median(per_call)
    # the same chain with bf16 / fp32s operands, and all three as a replayed HIP graph THROUGH THE PRODUCT API
    # (VADFromScratchPredictor(graph=True).predict_audio_device: four launches per clip -- at this size the host side is a visible
    # share of the call); the replayed results must be the eager call's bits
    extra = {}
    try:
        for prec in ("bf16", "fp32s"):
            model.precision = prec
            mp, _, _, _ = _event_blocks(chain, 50, min_seconds / 2, warm=30)
            extra[f"{prec}_ms_per_clip"] = round(mp, 4)
        pred_g = VADFromScratchPredictor(model, dev, graph=True)
        same = True
        for prec in ("fp32", "fp32s", "bf16"):
            model.precision = prec
            want, want_mean = chain()
            got, got_mean = pred_g.predict_audio_device(audio)
            same = same and bool(torch.equal(got, want)) and bool(torch.equal(got_mean, want_mean))
            mg, _, _, _ = _event_blocks(lambda: pred_g.predict_audio_device(audio), 50, min_seconds / 2, warm=5)
            extra[f"{prec}_graph_ms_per_clip"] = round(mg, 4)
        extra["graph_equals_eager_bits"] = same
        extra["graph_stats"] = dict(pred_g.graph_stats)
    except Exception as exc:   # (a capture problem must not take the leg down)
        extra["graph_error"] = f"{type(exc).__name__}: {exc}"[:200]
    finally:
        model.precision = "fp32"
    return {"workload": f"BASELINE configs[0]: {seconds:g} s of 16 kHz audio on the device -> log-mel -> {probs.shape[0] - 6} windows [7,80] -> forward -> boost -> probabilities {list(probs.shape)}",
            "ms_per_clip": round(med, 4), "ms_per_clip_min": round(min(per_call), 4), **extra, "frames": int(probs.shape[0]),
            "frames_per_s": round(probs.shape[0] / (med * 1e-3), 1), "real_time_factor": round(med * 1e-3 / seconds, 9),
            "finite": bool(torch.isfinite(probs).all().item()), "blocks": len(per_call)}


def _wall_blocks(fn, min_seconds, warm=2, max_calls=40):
    """median / min wall ms per call of fn() -- host clock around the call and a device synchronize on both sides: for paths that start
    in HOST memory (uploads are part of the call)"""
    for _ in range(warm):
        out = fn()
    torch.cuda.synchronize()
    per_call = []
    while len(per_call) < 5 or (sum(per_call) * 1e-3 < min_seconds and len(per_call) < max_calls):
        t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        per_call.append((time.perf_counter() - t0) * 1e3)
    return statistics.median(per_call), min(per_call), len(per_call), out


def _synthetic_pcm_hour(seconds, dev):
    """(pinned int16 PCM on the host, the same samples as float32 on the device, pure upload times): Gaussian noise at -20 dBFS"""
    pcm = np.clip(np.round(np.random.default_rng(0).standard_normal(16000 * seconds, dtype=np.float32) * (0.1 * 32768.0)), -32768, 32767).astype(np.int16)
    pinned = torch.from_numpy(pcm).pin_memory()
    audio = (pinned.to(dev, non_blocking=True).float() / 32768.0).contiguous()
    pinned_f32 = (torch.from_numpy(pcm.astype(np.float32)) / 32768.0).pin_memory()
    h2d16, _, _, _ = _wall_blocks(lambda: pinned.to(dev, non_blocking=True), 0.1, warm=2)
    h2d32, _, _, _ = _wall_blocks(lambda: pinned_f32.to(dev, non_blocking=True), 0.1, warm=2)
    return pinned, audio, {"h2d_pcm16_ms": round(h2d16, 4), "h2d_pcm16_GBps": round(pcm.nbytes / (h2d16 * 1e-3) / 1e9, 1),
                           "h2d_f32_ms": round(h2d32, 4), "h2d_f32_GBps": round(4 * pcm.size / (h2d32 * 1e-3) / 1e9, 1)}


def _event_blocks(fn, calls, min_seconds, warm=3):
    """median / min ms per call of fn() over blocks of `calls` calls (HIP events on the current stream, >= 5 blocks and
    >= min_seconds of timed work)"""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    per_call, spent = [], 0.0
    while len(per_call) < 5 or spent < min_seconds:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(calls):
            out = fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        per_call.append(ms / calls)
        spent += ms * 1e-3
    return statistics.median(per_call), min(per_call), len(per_call), out


def config3_global_one_gpu(state, dev, min_seconds):
    """BASELINE configs[3] at its stated GLOBAL size on ONE GPU: [2048, 800, 80] bf16 resident, evaluated as the eight
    256-sequence shards the eight ranks of a node would each see (voice_activity_detection_amd.distributed.shard_bounds),
    every shard's log-probs written into its slot of the gathered [2048, 800, 2] result -- the all_gather's layout."""
    from voice_activity_detection_amd import SelfAttentiveVAD
    from voice_activity_detection_amd.distributed import shard_bounds

    Bg, T, world = 2048, 800, 8
    model = SelfAttentiveVAD(F_MEL, N_LAYERS, D_MODEL, 0.5)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()})
    model = model.to(dev).eval()
    model.precision = "bf16"
    x = torch.empty((Bg, T, F_MEL), dtype=torch.bfloat16, device=dev)
    for r in range(world):  # the same per-rank seeding as the N-GPU run (Runner: default_rng(rank))
        lo, hi = shard_bounds(Bg, r, world)
        x[lo:hi] = torch.from_numpy(np.random.default_rng(r).uniform(-13.8, 4.2, (hi - lo, T, F_MEL)).astype(np.float32)).to(dev)
    y = torch.empty((Bg, T, 2), dtype=torch.float32, device=dev)
    model.reserve(T, max_batch=Bg // world)

    def one_pass():
        with torch.no_grad():
            for r in range(world):
                lo, hi = shard_bounds(Bg, r, world)
                model(features=x[lo:hi], out=y[lo:hi])
        return y

    med, mn, blocks, _ = _event_blocks(one_pass, 4, min_seconds)
    # the same eight shards two in flight (PipelinedVAD: the shards are independent batches)
    from voice_activity_detection_amd import PipelinedVAD
    pipe = PipelinedVAD(model, depth=2)
    pipe.reserve(T, max_batch=Bg // world)
    y2 = torch.empty_like(y)

    def one_pass_in_flight():
        for r in range(world):
            lo, hi = shard_bounds(Bg, r, world)
            pipe.submit(x[lo:hi], out=y2[lo:hi])
        pipe.join()
        return y2

    medp, mnp, _, _ = _event_blocks(one_pass_in_flight, 4, min_seconds / 2)
    same_p = bool(torch.equal(y2, y))
    with torch.no_grad():
        whole = model(features=x)  # the global batch as ONE forward (2.1 GB workspace)
    same = bool(torch.equal(whole, y))
    med1, mn1, _, _ = _event_blocks(lambda: model(features=x, out=whole), 4, min_seconds / 2, warm=1)
    frames = Bg * T
    return {"workload": "BASELINE configs[3] global batch on ONE GPU: synthetic [B=2048, T=800, F=80] bf16, eight 256-sequence shards back to back",
            "ms_per_pass": round(med, 4), "ms_per_pass_min": round(mn, 4), "frames_per_s": round(frames / (med * 1e-3), 1),
            "forward_frac_of_bf16_peak": round(flops_per_frame(T) * frames / (med * 1e-3) / 1e12 / PEAK_BF16_MFMA_TFLOPS, 4),
            "two_shards_in_flight_ms_per_pass": round(medp, 4), "two_shards_in_flight_ms_min": round(mnp, 4),
            "two_shards_in_flight_frames_per_s": round(frames / (medp * 1e-3), 1), "two_shards_in_flight_equals_sharded_bits": same_p,
            "single_forward_ms": round(med1, 4), "single_forward_ms_min": round(mn1, 4), "single_forward_equals_sharded_bits": same,
            "finite": bool(torch.isfinite(y).all().item()), "blocks": blocks, "unit": "frames/s"}


def logmel_roofline(n_samples, n_frames, ms):
    """The log-mel kernel against both roofs.  SURVEY.md section 8f#1 classifies it HBM-bound: algorithmic bytes = the samples read
    once + the [N,80] fp32 matrix written (4 B/sample + 320 B/frame).  Round 5's factored DFT (512 = 32 x 16) issues 624
    v_mfma_f32_32x32x2_f32 per 32 frames (4096 FLOP each) instead of 3584: what binds it now is the fp32 MFMA rate."""
    by = 4.0 * n_samples + 320.0 * n_frames
    fl = ((n_frames + 31) // 32) * 624 * 4096.0
    t = ms * 1e-3
    return {"algorithmic_bytes": by, "hbm_achieved_TBps": round(by / t / 1e12, 3), "hbm_frac": round(by / t / 1e12 / PEAK_HBM_TBPS, 4),
            "issued_flops": fl, "mfma_achieved_TFLOPs": round(fl / t / 1e12, 1), "mfma_frac": round(fl / t / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4),
            "bound": "mfma (fp32)", "traffic": profile_traffic("logmel", "logmel_fft_kernel")}


def profile_traffic(tag, kernel):
    """HBM-side bytes per launch of `kernel` from profiles/*_{tag}_traffic.json, quoted only when taken on the running sources"""
    want = kernel_source_hash()
    for f in reversed(sorted((REPO / "profiles").glob(f"*_{tag}_traffic.json"))):
        try:
            data = json.loads(f.read_text())
        except Exception:
            continue
        if data.get("csrc_hash") != want:
            continue
        for key, entry in data.items():
            if isinstance(entry, dict) and key.startswith(kernel):
                return round(entry["hbm_bytes_per_launch"])
    return None


def reference_mode_hour(state, dev, min_seconds):
    """The reference's OWN mode at configs[4]'s size: an hour of audio = 360 001 feature frames -> 359 963 windows of 7 frames
    (vad/predictor.py:169-224 cuts N - 38 of them, 1000 per forward) -> forward -> boosted probabilities [N,7] (:238-258), one
    library call (savad_predict_probabilities) on a device-resident feature matrix; fp32 and bf16 operands."""
    from voice_activity_detection_amd import SelfAttentiveVAD, VADFromScratchPredictor
    from voice_activity_detection_amd.features import log_mel

    seconds = 3600
    model = SelfAttentiveVAD(F_MEL, N_LAYERS, D_MODEL, 0.5)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()})
    model = model.to(dev).eval()
    pinned_pcm, audio, h2d = _synthetic_pcm_hour(seconds, dev)
    feat = log_mel(audio, dev)
    N = int(feat.shape[0])
    windows = N - 38
    res = {"workload": f"reference mode, {seconds} s of audio on ONE GPU: feature matrix [{N},80] -> {windows} windows of 7 frames "
                       "(vad/predictor.py:169-224) -> forward -> boosted probabilities [N,7] (:238-258); from_host_ms: the same from pinned "
                       "16-bit PCM in host memory (VADFromScratchPredictor.predict_audio_host: chunked upload on a copy stream under the "
                       "previous chunk's log-mel + forwards)", "audio_seconds": seconds,
           "frames": N, "windows": windows, "flops": windows * 7 * flops_per_frame(7), **h2d}
    ref32 = None
    for prec in ("fp32", "fp32s", "bf16"):
        model.precision = prec
        pred = VADFromScratchPredictor(model, dev)
        med, mn, blocks, out = _event_blocks(lambda: pred.predict_probabilities_device(feat), 1 if prec == "fp32" else 4, min_seconds, warm=2)
        probs = out[0]
        peak = PEAK_FP32_MFMA_TFLOPS if prec == "fp32" else PEAK_BF16_MFMA_TFLOPS
        issue = 6.0 if prec == "fp32s" else 1.0   # fp32s: six bf16 MFMA products per algorithmic multiply-add
        tf = issue * res["flops"] / (med * 1e-3) / 1e12
        res[prec] = {"ms_per_hour_of_audio": round(med, 4), "ms_min": round(mn, 4), "blocks": blocks, "rtf": round(med * 1e-3 / seconds, 10),
                     "windows_per_s": round(windows / (med * 1e-3), 1), "window_frames_per_s": round(windows * 7 / (med * 1e-3), 1),
                     "audio_frames_per_s": round(N / (med * 1e-3), 1),
                     "roofline": {"bound": "mfma", "achieved": round(tf, 1), "peak": peak, "unit": "TFLOP/s", "frac": round(tf / peak, 4)},
                     "finite": bool(torch.isfinite(probs).all().item())}
        if prec != "fp32":   # end to end from host memory (the exact-fp32 hour is 30 ms of kernels: nothing to learn from its upload)
            dev_ms, _, _, dev_out = _event_blocks(lambda: pred.predict_audio_device(audio), 4, min_seconds / 2, warm=1)
            host_ms, host_min, _, host_out = _wall_blocks(lambda: pred.predict_audio_host(pinned_pcm), min_seconds / 2, warm=1)
            res[prec].update({"from_audio_ms": round(dev_ms, 4), "from_host_ms": round(host_ms, 4), "from_host_ms_min": round(host_min, 4),
                              "from_host_equals_device_bits": bool(torch.equal(host_out[0], dev_out[0]) and torch.equal(host_out[1], dev_out[1])),
                              "from_host_over_max_of_h2d_and_device": round(host_ms / max(h2d["h2d_pcm16_ms"], dev_ms), 3),
                              "rtf_from_host": round(host_ms * 1e-3 / seconds, 10)})
        if prec == "fp32":
            ref32 = probs.clone()
        else:   # against the exact-fp32 run of the same call (fp32s: the fp32 bar; bf16: reported)
            res[prec]["max_abs_dprob_vs_fp32"] = float((probs - ref32).abs().max())
        if prec == "fp32s":
            res[prec]["roofline"]["fp32_equivalent_tflops"] = round(tf / 6, 1)
            res[prec]["within_1e-4_of_fp32"] = res[prec]["max_abs_dprob_vs_fp32"] < 1e-4
    model.precision = "fp32"
    return res


def stream_one_hour_sharded(state, dev, rank, world, min_seconds):
    """BASELINE configs[4] on N GPUs: 1 h of synthetic audio (the same recording on every rank's HOST) -> every rank uploads and
    transforms ITS samples only (StreamingPredictor.audio_shard_plan: its windows, their frames, the samples those read), runs its
    windows in place, ONE all_gather of the log-probs, overlap merge -> per-frame probabilities on every rank; bf16 and fp32
    operands; wall time per call bracketed by barriers (max over ranks), real-time factor end to end from host audio."""
    from voice_activity_detection_amd import SelfAttentiveVAD, StreamingPredictor
    from voice_activity_detection_amd import distributed as vdist

    seconds = 3600
    model = SelfAttentiveVAD(F_MEL, N_LAYERS, D_MODEL, 0.5)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()})
    model = model.to(dev).eval()
    # 16-bit PCM on every rank's host (AudioData's source format): a rank uploads its own samples as they are, 2 bytes each
    audio = np.clip(np.round(np.random.default_rng(0).standard_normal(16000 * seconds, dtype=np.float32) * (0.1 * 32768.0)), -32768, 32767).astype(np.int16)
    plan = StreamingPredictor.audio_shard_plan(len(audio), 800, 400, rank, world)
    res = {"workload": f"BASELINE configs[4] on {world} GPU(s): {seconds} s of 16 kHz audio on the host -> per rank: its samples -> log-mel of its frames -> its "
                       "windows T=800 hop=400 in place -> forward; one all_gather; overlap merge", "audio_seconds": seconds,
           "rank0_share": {"windows": [int(plan[1]), int(plan[2])], "frames": [int(plan[3]), int(plan[4])], "samples": int(plan[6])},
           "note": "wall time from HOST audio (16-bit PCM, pageable): includes each rank's host -> device copy of its own samples (115 MB / world over PCIe); the "
                   "device-resident figure of one GPU is secondary.configs4_stream_1h.*.from_audio_ms of the N = 1 line"}
    for prec in ("bf16", "fp32"):
        model.precision = prec
        sp = StreamingPredictor(model, dev, 800, 400, max_batch=256)
        for _ in range(2):
            probs = sp.predict_audio_device(audio)
        torch.cuda.synchronize()
        walls = []
        while len(walls) < 5 or (vdist.agree(sum(walls), dev) < min_seconds and len(walls) < 50):   # rank 0's clock decides: equal counts of collectives
            vdist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            probs = sp.predict_audio_device(audio)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            vdist.barrier()
            walls.append(dt)
        walls = vdist.max_over_ranks(walls, dev)
        med = statistics.median(walls)
        res[prec] = {"ms_per_hour_of_audio_from_host_audio": round(med * 1e3, 4), "ms_min": round(min(walls) * 1e3, 4), "calls": len(walls),
                     "rtf": round(med / seconds, 10), "finite": bool(torch.isfinite(probs).all().item())}
    model.precision = "fp32"
    return res


def stream_one_hour(state, dev, min_seconds):
    """BASELINE configs[4] at its full size on one GPU: 1 h of synthetic 16 kHz audio resident on the device -> log-mel
    [360001, 80] -> 900 sliding windows T=800 hop=400 (read in place: savad_forward_strided) -> forward -> overlap merge ->
    per-frame probabilities; fp32 and bf16 operands; real-time factor with and without the log-mel front-end, and through the
    audio-level entry point of the sharded run (StreamingPredictor.predict_audio_device)."""
    from voice_activity_detection_amd import SelfAttentiveVAD, StreamingPredictor
    from voice_activity_detection_amd.features import log_mel

    seconds = 3600
    model = SelfAttentiveVAD(F_MEL, N_LAYERS, D_MODEL, 0.5)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()})
    model = model.to(dev).eval()
    pinned_pcm, audio, h2d = _synthetic_pcm_hour(seconds, dev)
    mel_med, mel_min, _, feat = _event_blocks(lambda: log_mel(audio, dev), 4, min_seconds / 2, warm=2)
    N = int(feat.shape[0])
    res = {"workload": f"BASELINE configs[4] on ONE GPU: {seconds} s of 16 kHz audio -> log-mel [{N},80] -> 900 windows T=800 hop=400 -> "
                       "forward -> overlap merge -> probabilities; from_host_ms: end to end from pinned 16-bit PCM in host memory "
                       "(StreamingPredictor.predict_audio_host: spans of 256 windows, span c + 1 uploaded on a copy stream under span c's "
                       "log-mel + forwards; wall clock)", "audio_seconds": seconds, "frames": N, **h2d,
           "logmel_ms": round(mel_med, 4), "logmel_ms_min": round(mel_min, 4), "logmel_roofline": logmel_roofline(audio.numel(), N, mel_med)}
    for prec in ("fp32", "fp32s", "bf16"):
        model.precision = prec
        sp = StreamingPredictor(model, dev, 800, 400, max_batch=256)
        med, mn, blocks, probs = _event_blocks(lambda: sp.predict_device(feat), 2 if prec == "fp32" else 8, min_seconds, warm=2)
        e2e, e2e_min, _, probs2 = _event_blocks(lambda: sp.predict_audio_device(audio), 2 if prec == "fp32" else 8, min_seconds / 2, warm=1)
        res[prec] = {"ms_per_hour_of_audio": round(med, 4), "ms_min": round(mn, 4), "blocks": blocks,
                     # the audio-level entry point (what a rank of the sharded run executes for its span): log-mel + windows in place + merge
                     "from_audio_ms": round(e2e, 4), "from_audio_ms_min": round(e2e_min, 4), "rtf_from_audio": round(e2e * 1e-3 / seconds, 10),
                     "from_audio_equals_two_steps": bool(torch.equal(probs2, probs)),
                     "rtf_without_logmel": round(med * 1e-3 / seconds, 10), "rtf_with_logmel": round((med + mel_med) * 1e-3 / seconds, 10),
                     "frames_per_s": round(N / (med * 1e-3), 1), "finite": bool(torch.isfinite(probs).all().item()),
                     "in_unit_interval": bool(((probs >= 0) & (probs <= 1)).all().item())}
        host_ms, host_min, _, probs3 = _wall_blocks(lambda: sp.predict_audio_host(pinned_pcm), min_seconds / 2, warm=1)
        res[prec].update({"from_host_ms": round(host_ms, 4), "from_host_ms_min": round(host_min, 4),
                          "from_host_equals_device_bits": bool(torch.equal(probs3, probs2)),
                          "from_host_over_max_of_h2d_and_device": round(host_ms / max(h2d["h2d_pcm16_ms"], e2e), 3),
                          "rtf_from_host": round(host_ms * 1e-3 / seconds, 10)})
    model.precision = "fp32"
    return res


def workload_label(precision, B, T):
    if (precision, B, T) == ("fp32", 32, 800):
        tag = "BASELINE configs[1], exact-fp32 MFMA"
    elif (precision, B, T) == ("fp32s", 32, 800):
        tag = "BASELINE configs[1], fp32 arithmetic as split-bf16 operands (fp32s)"
    elif (precision, B, T) == ("bf16", 256, 800):
        tag = "BASELINE configs[2] (= the per-GPU shard of configs[3])"
    elif precision in ("fp32", "fp32s") and T == 7:
        tag = "the reference pipeline's window shape (vad/predictor.py:180-224)"
    else:
        tag = "custom shape"
    return f"{tag}: synthetic [B={B}, T={T}, F={F_MEL}] {precision} per GPU, SelfAttentiveVAD(80, 3, 128) forward -> log-probs [B,T,2]"


DTYPE_LABEL = {"fp32": "f32", "fp32s": "f32 (bf16x6 split operands, f32 accumulate)", "bf16": "bf16 operands, f32 accumulate"}


def collect_flags(obj, path=""):
    """every self-check flag in a (nested) result: keys `finite`, `in_unit_interval`, `within_*` and `*equals*` -> {path: bool}"""
    flags = {}
    if isinstance(obj, dict):
        for k, v in obj.items():
            here = f"{path}.{k}" if path else k
            if isinstance(v, bool) and (k in ("finite", "in_unit_interval") or "equals" in k or k.startswith("within_")):
                flags[here] = v
            elif isinstance(v, dict):
                flags.update(collect_flags(v, here))
    return flags


def bench_summary(line, flags):
    """{leg: {ms, frac}} for the headline and every secondary leg + the self-check flags, compact enough for a log's tail"""
    def pair(ms, frac):
        return {"ms": ms, "frac": frac}

    rl = line.get("roofline", {})
    out = {"headline": pair(line.get("ms_per_step"), rl.get("forward_frac")),
           "headline_one_forward": pair(line.get("ms_one_forward"), rl.get("forward_frac_one_forward")),
           "headline_dominant_kernel": pair(rl.get("ms_per_launch"), rl.get("frac"))}
    if "fp32_equivalent_tflops" in rl:
        out["headline_fp32_equivalent_tflops"] = {"forward": rl.get("forward_fp32_equivalent_tflops"), "one_forward": rl.get("forward_fp32_equivalent_tflops_one_forward"),
                                                  "dominant_kernel": rl.get("fp32_equivalent_tflops")}
    legs = dict(line.get("secondary", {}))
    for k in ("config3", "config4"):
        if k in line:
            legs[k] = line[k]
    for k, v in legs.items():
        if not isinstance(v, dict):
            continue
        if "error" in v:
            out[k] = {"error": str(v["error"])[:80]}
        elif "ms_per_step" in v:
            r2 = v.get("roofline", {})
            out[k] = pair(v["ms_per_step"], r2.get("forward_frac"))
            out[k]["ms_one_forward"] = v.get("ms_per_step_one_in_flight")
        elif "ms_per_pass" in v:
            out[k] = pair(v["ms_per_pass"], v.get("forward_frac_of_bf16_peak"))
        elif "ms_per_clip" in v:
            out[k] = {"ms": v["ms_per_clip"], **{kk: vv for kk, vv in v.items() if kk.endswith("_ms_per_clip")}}
        else:   # per-precision legs (the reference-mode hour, the streamed hour)
            sub = {}
            for prec in ("fp32", "fp32s", "bf16"):
                if isinstance(v.get(prec), dict):
                    d = v[prec]
                    ms = d.get("ms_per_hour_of_audio", d.get("ms_per_hour_of_audio_from_host_audio"))
                    sub[prec] = pair(ms, d.get("roofline", {}).get("frac"))
                    for kk in ("from_audio_ms", "from_host_ms"):
                        if kk in d:
                            sub[prec][kk] = d[kk]
            if "logmel_ms" in v:
                sub["logmel"] = pair(v["logmel_ms"], v.get("logmel_roofline", {}).get("hbm_frac"))
            if sub:
                out[k] = sub
    false = sorted(k for k, ok_ in flags.items() if not ok_)
    out["flags_checked"] = len(flags)
    out["false_flags"] = false
    out["all_flags_true"] = not false
    return out


def summarize(walls, evs, steps, frames_per_step):
    med, mn = statistics.median(walls), min(walls)
    return {"ms_per_step": round(med / steps * 1e3, 4), "ms_per_step_min": round(mn / steps * 1e3, 4),
            "ms_per_step_hip_events_median": round(statistics.median(evs) / steps * 1e3, 4),
            "blocks": len(walls), "timed_seconds": round(sum(walls), 3),
            "value": round(frames_per_step * steps / med, 1)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=32, help="sequences per GPU")
    ap.add_argument("--frames", type=int, default=800, help="T")
    ap.add_argument("--splits", type=int, default=0, help="attention key splits (0 = auto)")
    ap.add_argument("--row-mode", type=int, default=0, help="launch schedule knob (include/savad.h: savad_set_row_mode), 0 = automatic")
    ap.add_argument("--precision", default="fp32s", choices=["fp32", "fp32s", "bf16"],
                    help="fp32s (default) = BASELINE configs[1], the fp32 arithmetic on the bf16 matrix pipe (three-piece operands, six MFMA products, "
                         "fp32 accumulate: the same 1e-4 parity bar as fp32); fp32 = the same config on the exact-fp32 MFMA; bf16 = configs[2]: "
                         "bf16 MFMA operands, bf16 features")
    ap.add_argument("--plant-false-flag", action="store_true", help=argparse.SUPPRESS)  # tests: a planted failing self-check must fail the run
    ap.add_argument("--min-seconds", type=float, default=0.5, help="timed work per measurement, in K-step blocks")
    ap.add_argument("--cpu-seconds", type=float, default=8.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-events", action="store_true", help="skip the per-kernel HIP-event pass (no roofline per kernel)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the configs[2] / T=7 / configs[0] / configs[3] / configs[4] legs")
    ap.add_argument("--legs", default="all", help="comma-separated secondary legs to run (default: all)")
    ap.add_argument("--in-flight", type=int, default=0,
                    help="independent forwards kept in flight (own HIP stream / library handle / workspace each): 0 = pick the fastest "
                         "of 1, 2, 3 for the shape during warm-up (default), N = exactly N")
    ap.add_argument("--gather", default="final", choices=["step", "final"],
                    help="multi-GPU: which gather mode `value` is quoted on (both are always measured and printed): 'final' (default) = every "
                         "forward writes into a [K,B,T,2] send buffer and ONE all_gather closes the K-step block -- north_star's 'a single RCCL "
                         "gather over xGMI at the end'; 'step' = one all_gather of the [B,T,2] log-probs per forward, issued while the newer "
                         "forwards run (both: voice_activity_detection_amd.distributed.ShardedPipeline)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend: nccl (= RCCL, default); gloo only together with --stub-forward")
    ap.add_argument("--stub-forward", action="store_true",
                    help="CPU dry run of this file's control flow (no GPU, no library): a stand-in forward; the line it prints is marked "
                         "and carries no measurement (tests/test_dist_gloo.py runs it with two gloo ranks)")
    ap.add_argument("--config3-shape", default="256,800", help="per-rank [B,T] of the config3 leg (dry runs use a small one)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if "WORLD_SIZE" not in os.environ and args.gpus > 1:
            # `python bench.py --gpus N` on its own: become the launcher (one rank per GPU under torch.distributed.run, this file again)
            import socket

            with socket.socket() as sock:
                sock.bind(("127.0.0.1", 0))
                port = sock.getsockname()[1]
            os.execv(sys.executable, [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
                                      "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve()), *sys.argv[1:]])
        args.gpus = world
    stub = args.stub_forward
    if args.backend == "gloo" and not stub:
        sys.exit("--backend gloo is for --stub-forward dry runs: the product runs on HIP devices over RCCL")
    if stub:
        dev = torch.device("cpu")
        torch.set_num_threads(1)
    else:
        if not torch.cuda.is_available():
            sys.exit("bench.py needs a HIP device")
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or os.environ.get("SAVAD_BENCH_FORCE_DIST") == "1"  # the env var lets a 1-GPU box exercise RCCL
    if use_dist:
        import torch.distributed as dist  # "nccl" = RCCL on ROCm

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29512")
        if stub:
            dist.init_process_group(args.backend, rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from voice_activity_detection_amd import distributed as vdist
    from voice_activity_detection_amd import seeded_state_dict

    B, T, K = args.batch, args.frames, args.steps
    state = None if stub else seeded_state_dict(1234)
    main_run = Runner(state, B, T, args.precision, dev, rank, world, args.gather, K, args.row_mode, args.splits, args.in_flight, stub)
    frames_per_step = world * B * T

    # ---- the headline measurement
    clocks0 = None if stub else gpu_state()
    walls, evs = main_run.timed_blocks(K, args.warmup, args.min_seconds)
    head = summarize(walls, evs, K, frames_per_step)
    ok = main_run.delivered()
    gather_modes = None
    if use_dist:  # the OTHER gather mode, for comparison
        other = "final" if args.gather == "step" else "step"
        main_run.set_gather(other)
        w2, e2 = main_run.timed_blocks(K, 2, args.min_seconds / 2)
        alt = summarize(w2, e2, K, frames_per_step)
        ok = ok and main_run.delivered()
        gather_modes = {f"gather_{args.gather}_ms": head["ms_per_step"], f"gather_{other}_ms": alt["ms_per_step"],
                        f"gather_{other}_value": alt["value"],
                        "note": "step = one RCCL all_gather of [B,T,2] log-probs per forward, issued while the newer forwards run; final = each "
                                "forward writes into a [K,B,T,2] send buffer and ONE all_gather closes the K-step block "
                                "(voice_activity_detection_amd.distributed.ShardedPipeline; `value` is quoted on gather_" + args.gather + ")"}
        main_run.set_gather(args.gather)
    # the same K-step blocks with ONE forward in flight (wall time of one forward; the kernels' own durations are measured this way)
    one = None
    if main_run.sp.in_flight > 1:
        tuned = main_run.sp.in_flight
        main_run.drain()
        main_run.sp.set_in_flight(1)
        w1, e1 = main_run.timed_blocks(K, 2, args.min_seconds / 2)
        one = summarize(w1, e1, K, frames_per_step)
        main_run.drain()
        main_run.sp.set_in_flight(tuned)
    ktimes = [] if (args.no_events or stub) else main_run.kernel_profile(min(K, 20), (one or head)["ms_per_step"])
    ktimes_in_flight = None if (args.no_events or stub) else main_run.kernel_profile_in_flight(min(K, 20), head["ms_per_step"])
    clocks1 = None if stub else gpu_state()

    # ---- secondary legs, measured in the same run so that they are driver-witnessed
    secondary = {}
    want = None if args.legs == "all" else set(args.legs.split(","))

    def leg(key, fn):
        """run one secondary leg; it must never take the headline line down with it"""
        if want is not None and key not in want:
            return
        c0 = None if stub else gpu_state()
        try:
            res = fn()
        except Exception as exc:
            if use_dist:
                raise   # a rank that skips collectives its peers still issue would hang the job: fail loudly instead
            res = {"error": f"{type(exc).__name__}: {exc}"}
        res["clocks"] = {"start": c0, "end": None if stub else gpu_state()}
        secondary[key] = res
        if not stub:
            torch.cuda.empty_cache()

    def shape_leg(prec, b2, t2, gm):
        def run():
            k2 = 20 if t2 > 32 else 50
            r = Runner(state, b2, t2, prec, dev, rank, world, gm or "step", k2, in_flight=args.in_flight, stub=stub)
            w, e = r.timed_blocks(k2, 5, args.min_seconds)
            s = summarize(w, e, k2, world * b2 * t2)
            s1 = s
            tuned = r.sp.in_flight
            if tuned > 1:   # and one forward at a time
                r.drain()
                r.sp.set_in_flight(1)
                w1, e1 = r.timed_blocks(k2, 2, args.min_seconds / 2)
                s1 = summarize(w1, e1, k2, world * b2 * t2)
                r.drain()
                r.sp.set_in_flight(tuned)
            kt = [] if args.no_events else r.kernel_profile(10, s1["ms_per_step"])
            delivered = r.delivered()   # BEFORE the batch-invariant timing below: that mode's last output differs from a default forward by design
            if prec == "bf16" and t2 > 32 and not stub and r.model is not None:
                # the price of model.batch_invariant (savad_set_batch_invariant: the persistent attention kernel without key-split tail
                # items, every batching the same bits), one forward at a time, same run
                try:
                    r.drain()
                    r.sp.set_in_flight(1)
                    r.model.batch_invariant = True
                    wi, ei = r.timed_blocks(k2, 2, args.min_seconds / 2)
                    s["batch_invariant_ms_per_step_one_in_flight"] = summarize(wi, ei, k2, world * b2 * t2)["ms_per_step"]
                finally:
                    r.drain()
                    r.model.batch_invariant = False
                    r.sp.set_in_flight(tuned)
            s.update({"workload": workload_label(prec, b2, t2), "global_batch": world * b2, "unit": "frames/s",
                      "in_flight": r.sp.in_flight, "in_flight_tuning_ms": r.tuning,
                      "ms_per_step_one_in_flight": s1["ms_per_step"], "value_one_in_flight": s1["value"],
                      "finite": delivered,
                      "roofline": roofline_block(kt, prec, b2, t2, s["ms_per_step"], profiled_shape=True, ms_one_forward=s1["ms_per_step"])})
            if gm:
                s["parallelism"] = (f"batch-shard x{world} + 1 RCCL all_gather of [{b2},{t2},2] f32 per forward" if gm == "step" else
                                    f"batch-shard x{world} + 1 RCCL all_gather of all K batches' [{b2},{t2},2] f32 per block")
            return s
        return run

    if not args.no_secondary:
        if world == 1 and not use_dist and not stub:
            if (args.precision, B, T) != ("fp32", 32, 800):
                leg("configs1_exact_fp32_b32_t800", shape_leg("fp32", 32, 800, None))   # the same config on the exact-fp32 MFMA (rounds 1-5's headline)
            if (args.precision, B, T) != ("fp32s", 32, 800):
                leg("configs1_fp32s_b32_t800", shape_leg("fp32s", 32, 800, None))
            if (args.precision, B, T) != ("bf16", 256, 800):
                leg("configs2_bf16_b256_t800", shape_leg("bf16", 256, 800, None))
            if (args.precision, B, T) != ("fp32", 1000, 7):
                leg("pipeline_fp32_b1000_t7", shape_leg("fp32", 1000, 7, None))
            if (args.precision, B, T) != ("fp32s", 1000, 7):
                leg("pipeline_fp32s_b1000_t7", shape_leg("fp32s", 1000, 7, None))   # the reference's chunk of windows: the fp32s single launch, latency variant
            leg("pipeline_fp32s_b16384_t7", shape_leg("fp32s", 16384, 7, None))   # the predictor's default chunk of windows: the fp32s single launch, a wave per block
            leg("pipeline_fp32_b16384_t7", shape_leg("fp32", 16384, 7, None))
            if (args.precision, B, T) != ("bf16", 1000, 7):
                leg("pipeline_bf16_b1000_t7", shape_leg("bf16", 1000, 7, None))
            leg("reference_mode_1h", lambda: reference_mode_hour(state, dev, args.min_seconds))
            leg("configs0_clip10s_audio_to_probabilities", lambda: clip_pipeline(state, dev, 10.0, args.min_seconds))
            leg("configs3_global_b2048_one_gpu", lambda: config3_global_one_gpu(state, dev, args.min_seconds))
            leg("configs4_stream_1h", lambda: stream_one_hour(state, dev, args.min_seconds))
        elif use_dist:
            b3, t3 = (int(v) for v in args.config3_shape.split(","))
            leg("config3", shape_leg("bf16", b3, t3, args.gather))  # configs[3]: [256 x world, 800, 80] bf16, batch-sharded
            if not stub:
                leg("config4", lambda: stream_one_hour_sharded(state, dev, rank, world, args.min_seconds))

    # every rank's collective counts (they must agree: a mismatch is a hang waiting to happen)
    counts = vdist.collective_counts()
    all_counts = [counts]
    if use_dist:
        all_counts = [None] * world
        dist.all_gather_object(all_counts, counts)

    all_ok = True
    if rank == 0:
        one_fwd = one or head
        line = {
            "metric": "audio frames/sec (whole node)", "value": head["value"],
            # SURVEY section 8d's own definition (wall time of ONE forward at a time) right beside the throughput figure `value`
            "value_one_forward": one_fwd["value"], "ms_one_forward": one_fwd["ms_per_step"], "unit": "frames/s",
            "n_gpus": world, "steps": K, "warmup": args.warmup, "ms_per_step": head["ms_per_step"],
            "ms_per_step_min": head["ms_per_step_min"], "ms_per_step_hip_events_median": head["ms_per_step_hip_events_median"],
            "timing": {"statistic": "median over blocks of exactly K steps (barrier + synchronize on both sides of every block; max over ranks)",
                       "blocks": head["blocks"], "timed_seconds": head["timed_seconds"],
                       "value_is": f"throughput with `in_flight` independent batches in flight; value_one_forward / ms_one_forward = one forward at a time"},
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": DTYPE_LABEL[args.precision],
            "data": "synthetic (seeded U(-13.8,4.2) mel frames, seeded random-init weights)",
            "config": {"workload": workload_label(args.precision, B, T) + "; `value` = throughput with `in_flight` independent batches in flight, "
                                   "`value_one_forward` / `ms_one_forward` = SURVEY.md section 8d's own definition (wall time of ONE forward at a time)",
                       "global_batch": world * B, "frames_per_sequence": T,
                       "parallelism": f"batch-shard x{world}" + ((" + 1 RCCL all_gather of the [B,T,2] log-probs per forward" if args.gather == "step"
                                                                   else " + 1 RCCL all_gather of all K batches' log-probs per block") if use_dist else "")},
            "finite": ok,
            "in_flight": main_run.sp.in_flight,
            "in_flight_note": "consecutive batches are independent: `in_flight` forwards are kept in flight, each on its own HIP stream with its own "
                              "library handle and workspace (voice_activity_detection_amd.PipelinedVAD; same bits as one at a time); picked during "
                              "warm-up from the ms per step in in_flight_tuning; roofline.per_kernel, roofline.frac and roofline.forward_frac_one_forward "
                              "are measured with ONE in flight, roofline.forward_frac with `in_flight`",
            "in_flight_tuning_ms": main_run.tuning,
            "ms_per_step_one_in_flight": one_fwd["ms_per_step"], "value_one_in_flight": one_fwd["value"],
            "roofline": roofline_block(ktimes, args.precision, B, T, head["ms_per_step"], profiled_shape=True, ms_one_forward=one_fwd["ms_per_step"]),
            "clocks": {"start": clocks0, "end": clocks1},
            "collective_counts": all_counts,
        }
        if ktimes_in_flight:   # the launches' durations in the mode `value` is quoted in (stretched by the overlap; sum / in_flight ~ ms_per_step)
            line["roofline"]["kernels_ms_in_flight"] = ktimes_in_flight
            line["roofline"]["kernels_ms_in_flight_note"] = (f"mean launch durations with {main_run.sp.in_flight} forwards in flight, HIP events on each "
                                                            "forward's own stream; per_kernel / kernels_ms / frac are one forward at a time")
        if os.environ.get("SAVAD_LIB"):   # a kernel experiment: never to be mistaken for the product library
            line["library_override"] = os.environ["SAVAD_LIB"]
        if stub:
            line.update({"stub_forward": True, "data": "STUB forward on the CPU (control-flow dry run): no number in this line is a measurement",
                         "dtype": "none (stub)"})
        if gather_modes:
            line.update(gather_modes)
        for k, v in secondary.items():
            if k in ("config3", "config4"):
                line[k] = v
        rest = {k: v for k, v in secondary.items() if k not in ("config3", "config4")}
        if rest:
            line["secondary"] = rest
        if world == 1 and not args.no_cpu_baseline and not stub:
            line["cpu_baseline"] = cpu_baseline(state, B, T, args.cpu_seconds)
            line["speedup_vs_cpu_baseline"] = round(line["value"] / line["cpu_baseline"]["value"], 1)
        # ---- every leg's self-checks folded into the top-level flag, and a compact summary as the LAST key of the line (the tail of the
        # line is what a truncated log still shows): one {ms, frac} pair per BASELINE config / leg, every flag by name when it is false
        if args.plant_false_flag:
            line.setdefault("secondary", {})["planted"] = {"finite": False}
        flags = collect_flags({k: v for k, v in line.items() if k != "finite"})
        flags["headline.finite"] = bool(ok)
        line["finite"] = all_ok = all(flags.values())
        line["summary"] = bench_summary(line, flags)
        print(json.dumps(line), flush=True)
    if use_dist:
        vdist.barrier()
        dist.destroy_process_group()
    if rank == 0 and not all_ok:
        sys.exit(3)   # a self-check read false: the line above says which (summary.false_flags)


if __name__ == "__main__":
    main()
