#!/usr/bin/env python3
"""bench.py -- audio frames/sec of the self-attentive-VAD forward pass on MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch of synthetic mel frames: forward of
SelfAttentiveVAD on a device-resident [B, T, F] fp32 tensor -> device-resident [B, T, 2] log-probs
(N > 1: every rank runs its own B-sequence shard, then ONE RCCL all_gather of the log-probs).
Default workload = BASELINE.json configs[1]: [32, 800, 80] fp32 per GPU, seeded weights.
Prints ONE JSON line on rank 0 (contract: see the task statement).
"""
from __future__ import annotations

import argparse
import json
import os
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
    element of q/k/v/context (4 fp32, 2 bf16).  The residual stream is fp32 in the fp32 path, fp16 in the bf16 path."""
    D, F = D_MODEL, F_MEL
    frames = B * T
    att = 4.0 * T * T * D * B
    hres = 4 if e == 4 else 2
    base = name.replace("_bf16", "")
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
        return 118784.0 * frames, frames * (F * e + D * hres + 3 * D * e)
    return 0.0, 0


def measured_traffic(name: str, precision: str, B: int, T: int):
    """HBM-side bytes per launch of kernel `name` from the committed rocprofv3 PMC passes
    (profiles/*_traffic.json, produced by scripts/profile_gpu.sh; FETCH_SIZE doubled per the gfx950
    correction of MI355X_MICROARCH.md).  PMC counters cannot be collected from inside this process, so
    the number is only reported when the workload matches a profiled one: fp32 [32,800] (the default
    bench line) or bf16 [256,800]."""
    if (precision, B, T) == ("fp32", 32, 800):
        files = [f for f in sorted((REPO / "profiles").glob("*_traffic.json")) if "bf16" not in f.name]
    elif (precision, B, T) == ("bf16", 256, 800):
        files = sorted((REPO / "profiles").glob("*bf16_b256_traffic.json"))
    else:
        return None
    if not files:
        return None
    data = json.loads(files[-1].read_text())
    last = name.endswith("_last") or name.endswith("_last_bf16")
    stem = name.replace("_last", "")
    prefix = {"attention": "attention_kernel", "attention_row": "attention_row_kernel", "row": "row_kernel", "input_qkv": "input_qkv_kernel",
              "attention_bf16": "attention_kernel_bf16", "row_bf16": "row_kernel_bf16", "input_qkv_bf16": "input_qkv_kernel_bf16"}.get(stem)
    if prefix is None:
        return None
    for key, entry in data.items():
        if key == prefix or key.startswith(prefix + "<") or key.startswith(prefix + "_m"):
            if "<" in key and prefix.startswith(("attention_row", "row")):
                if ("<true" in key) != last:
                    continue
            return round(entry["hbm_bytes_per_launch"])
    return None


def cpu_baseline(state, T: int, seconds: float):
    """The reference's CPU path cannot travel; time its stand-ins on this host's cores on a bounded
    sample of the same workload: (a) the stock-PyTorch port (same ATen ops as the reference),
    (b) the C oracle.  Report the faster one."""
    from oracle import oracle, torch_port

    cores = os.cpu_count() or 1
    st = {k: torch.from_numpy(v) for k, v in state.items()}
    rng = np.random.default_rng(0)
    # (a) stock-PyTorch port: sweep the intra-op thread count (oversubscription hurts small batches)
    Bt = 16
    x = torch.from_numpy(rng.uniform(-13.8, 4.2, (Bt, T, F_MEL)).astype(np.float32))
    torch_fps, torch_threads, it_total = 0.0, 1, 0
    budget = seconds * 0.6
    cands = sorted({c for c in (8, 16, 32, 64, cores) if c <= cores})
    t_start = time.perf_counter()
    for nt in cands:
        torch.set_num_threads(nt)
        torch_port.forward(st, x)  # warm-up
        t0 = time.perf_counter()
        it = 0
        while True:
            torch_port.forward(st, x)
            it += 1
            dt = time.perf_counter() - t0
            if dt > budget / len(cands) or it >= 20:
                break
        it_total += it
        if it * Bt * T / dt > torch_fps:
            torch_fps, torch_threads = it * Bt * T / dt, nt
        if time.perf_counter() - t_start > budget:
            break
    # (b) C oracle: one sequence per OpenMP thread
    Bc = min(cores, 128)
    xn = rng.uniform(-13.8, 4.2, (Bc, T, F_MEL)).astype(np.float32)
    oracle.forward(state, xn[:1])
    t0 = time.perf_counter()
    oracle.forward(state, xn, threads=cores)
    c_fps = Bc * T / (time.perf_counter() - t0)
    best = max(torch_fps, c_fps)
    used = torch_threads if torch_fps >= c_fps else cores  # threads of the run that produced `value`
    return {
        "value": round(best, 1), "unit": "frames/s", "cores": used, "host_cores": cores, "kind": "port",
        "sample": f"stock-PyTorch CPU port (the reference's ATen ops) on [{Bt},{T},{F_MEL}] fp32, {it_total} forwards, "
                  f"best of thread counts {cands}: {torch_fps:.0f} frames/s at {torch_threads} threads (torch "
                  f"{torch.__version__}); C oracle on [{Bc},{T},{F_MEL}], 1 pass, OpenMP {cores} threads: "
                  f"{c_fps:.0f} frames/s; faster one reported",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=32, help="sequences per GPU")
    ap.add_argument("--frames", type=int, default=800, help="T")
    ap.add_argument("--splits", type=int, default=0, help="attention key splits (0 = auto)")
    ap.add_argument("--row-mode", type=int, default=0, help="0 auto, 1 N-split 32-row tiles, 2 M-split 128-row tiles (separate attention / row launches), 3 M-split fused with attention, 4 N-split with the T<=32 attention as its own launch, 5 fused with helper waves (experimental)")
    ap.add_argument("--precision", default="fp32", choices=["fp32", "bf16"],
                    help="fp32 = BASELINE configs[1] (default); bf16 = configs[2]: bf16 MFMA operands, bf16 features")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-events", action="store_true", help="do not bracket kernels with HIP events")
    ap.add_argument("--gather", default="final", choices=["final", "step"],
                    help="multi-GPU: one all_gather of all K batches' log-probs at the end of the timed region, or one per batch")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit(f"--gpus {args.gpus} needs torch.distributed.run with --nproc-per-node {args.gpus}")
        args.gpus = world
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a HIP device")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    use_dist = world > 1 or os.environ.get("SAVAD_BENCH_FORCE_DIST") == "1"  # the env var lets a 1-GPU box exercise RCCL
    if use_dist:
        import torch.distributed as dist  # RCCL ("nccl" backend on ROCm)

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from voice_activity_detection_amd import SelfAttentiveVAD, seeded_state_dict

    B, T = args.batch, args.frames
    state = seeded_state_dict(1234)
    model = SelfAttentiveVAD(F_MEL, N_LAYERS, D_MODEL, 0.5)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()})
    model = model.to(dev).eval()
    model.attention_splits = args.splits
    model.row_mode = args.row_mode
    model.precision = args.precision
    # each rank gets its own shard of the global batch (weak scaling: B per GPU fixed)
    x = torch.from_numpy(np.random.default_rng(rank).uniform(-13.8, 4.2, (B, T, F_MEL)).astype(np.float32)).to(dev)
    if args.precision == "bf16":
        x = x.to(torch.bfloat16)
    # The single collective of the path (north_star: "utterance batches shard embarrassingly across the 8 GPUs of
    # one node with a single RCCL gather over xGMI at the end"): every rank keeps the log-probs of its K batches on
    # the device and ONE all_gather of all of them ([K,B,T,2] per rank: 10 MB at the default sizes) closes the
    # timed region.  `--gather step` gathers after every batch instead (stream-ordered; measured +2 us per step on
    # 1 GPU through RCCL; an async double-buffered variant measured +45 us per step and was dropped).
    final = use_dist and args.gather == "final"
    n_keep = max(args.steps, args.warmup, 1)
    gathered = torch.empty((world, B, T, 2), dtype=torch.float32, device=dev) if use_dist else None
    local_all = torch.empty((n_keep, B, T, 2), dtype=torch.float32, device=dev) if final else None  # this rank's K batches
    gathered_all = torch.empty((world, n_keep, B, T, 2), dtype=torch.float32, device=dev) if final else None
    done = [0]

    def step():
        with torch.no_grad():
            if final:  # the forward writes straight into this batch's slot of the gather's send buffer
                y = model(features=x, out=local_all[done[0] % n_keep])
                done[0] += 1
            else:
                y = model(features=x)
                if use_dist:
                    dist.all_gather_into_tensor(gathered, y)
        return y

    def drain():
        if final and done[0]:
            dist.all_gather_into_tensor(gathered_all, local_all)
            done[0] = 0

    for _ in range(max(args.warmup, 1)):
        step()
    drain()
    torch.cuda.synchronize()
    def timed_region():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            y = step()
        drain()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0  # this rank's K steps (+ the gather, which completes only when every rank has contributed)
        if use_dist:
            dist.barrier()  # closing bracket; its own latency (a host-synchronised RCCL all_reduce) is not part of the K steps
        if use_dist:
            tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = float(tmax.item())
        return dt, y

    # Region A: exactly K steps, nothing but the hot path on the stream -> `value`.
    elapsed, y = timed_region()
    # Region B: the same K steps again with every kernel bracketed by HIP events on the launch stream (for the
    # roofline block).  Event records put barrier packets between the kernels (+~40 us per step), which is why
    # they are kept out of region A; both step times are reported.
    elapsed_ev = None
    if not args.no_events:
        model.set_profiling(args.steps)
        elapsed_ev, y = timed_region()
    ktimes = [] if args.no_events else model.kernel_times()
    ok = bool(torch.isfinite(y).all().item())
    if use_dist:  # the last gather really delivered this rank's shard
        got = gathered[rank] if args.gather == "step" else gathered_all[rank, args.steps - 1]
        ok = ok and bool(torch.equal(got, y))

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        frames = world * B * T
        value = frames * args.steps / elapsed
        fwd_tflops = flops_per_frame(T) * B * T / (elapsed / args.steps) / 1e12  # per GPU
        roof = None
        if ktimes:
            peak = PEAK_BF16_MFMA_TFLOPS if args.precision == "bf16" else PEAK_FP32_MFMA_TFLOPS
            e = 2 if args.precision == "bf16" else 4
            # the dominant kernel = the launch label with the largest total time (the fused attention + row chain
            # when the library fuses them, otherwise the attention stage)
            by_name = {}
            for n, t in ktimes:
                by_name.setdefault(n, []).append(t)
            dom = max(by_name, key=lambda n: sum(by_name[n]))
            dom_ms = sum(by_name[dom]) / len(by_name[dom])
            dom_flops, dom_bytes = launch_work(dom, B, T, e)
            ach = dom_flops / (dom_ms * 1e-3) / 1e12
            per_kernel = {}
            for n, ts in by_name.items():
                fl, by = launch_work(n, B, T, e)
                ms_k = sum(ts) / len(ts)
                per_kernel[n] = {"launches": len(ts), "ms": round(ms_k, 4), "tflops": round(fl / (ms_k * 1e-3) / 1e12, 2),
                                 "frac": round(fl / (ms_k * 1e-3) / 1e12 / peak, 4)}
            roof = {
                "bound": "mfma", "kernel": f"{dom} ({len(by_name[dom])} launches per forward)",
                "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s",
                "frac": round(ach / peak, 4),
                "traffic": measured_traffic(dom, args.precision, B, T), "traffic_unit": "bytes/launch (rocprofv3 PMC, profiles/)",
                "algorithmic_bytes": dom_bytes, "algorithmic_flops": dom_flops,
                # the same launch against the HBM roofline: the non-binding one -- the fp32 MFMA
                # rate caps the attention stage at 9.8 % of 8 TB/s
                "hbm_achieved_TBps": round(dom_bytes / (dom_ms * 1e-3) / 1e12, 3), "hbm_peak_TBps": PEAK_HBM_TBPS,
                "hbm_frac": round(dom_bytes / (dom_ms * 1e-3) / 1e12 / PEAK_HBM_TBPS, 4),
                "ms_per_launch": round(dom_ms, 4),
                "forward_achieved": round(fwd_tflops, 2), "forward_frac": round(fwd_tflops / peak, 4),
                "per_kernel": per_kernel,
                "kernels_ms": {f"{i}:{n}": round(t, 4) for i, (n, t) in enumerate(ktimes)},
            }
        line = {
            "metric": "audio frames/sec (whole node)", "value": round(value, 1), "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 4),
            "ms_per_step_with_kernel_events": round(elapsed_ev / args.steps * 1e3, 4) if elapsed_ev else None,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32" if args.precision == "fp32" else "bf16 operands, f32 accumulate",
            "data": "synthetic (seeded U(-13.8,4.2) mel frames, seeded random-init weights)",
            "config": {"workload": f"BASELINE configs[{1 if args.precision == 'fp32' else 2}]: synthetic [B={B}, T={T}, F={F_MEL}] {args.precision} per GPU, "
                                   f"SelfAttentiveVAD(80, 3, 128) forward -> log-probs [B,T,2]",
                       "global_batch": world * B, "frames_per_sequence": T,
                       "parallelism": f"batch-shard x{world}" + ((" + 1 RCCL all_gather of all K batches' log-probs at the end" if args.gather == "final"
                                                                   else " + 1 RCCL all_gather per batch") if world > 1 else "")},
            "finite": ok,
            "roofline": roof,
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(state, T, args.cpu_seconds)
            line["speedup_vs_cpu_baseline"] = round(value / line["cpu_baseline"]["value"], 1)
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
