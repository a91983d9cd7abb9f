"""``python -m voice_activity_detection_amd predict AUDIO CHECKPOINT [options]`` and
``... evaluate EVAL_LIST CHECKPOINT [options]`` -- the reference's ``python main.py predict`` / ``evaluate``
(``main.py:8-10``, ``vad/predict.py:10-50``, ``vad/evaluate.py:20-29``) on the MI355X path: same positional
arguments, same options, JSON v0.3 (predict) / metric lines (evaluate) to ``--output-path`` or stdout.
``train`` is out of scope."""
from __future__ import annotations

import argparse
import json
import sys
from pathlib import Path


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(prog="voice_activity_detection_amd")
    sub = ap.add_subparsers(dest="command", required=True)
    p = sub.add_parser("predict", help="voice activity of one 16 kHz WAV file (vad/predict.py:10-25)")
    p.add_argument("audio_path", type=Path)
    p.add_argument("checkpoint_path", type=Path)
    p.add_argument("--output-path", type=Path, default=None, help="Path to store output. Default to stdout.")
    p.add_argument("--split-max-seconds", type=float, default=None, help="Chunk size to split audio in seconds.")
    p.add_argument("--activity-max-sec", type=int, default=None, help="Maximum length of voice activity in seconds")
    p.add_argument("--threshold", type=float, default=0.5)
    p.add_argument("--min-vally-ms", type=int, default=0)
    p.add_argument("--min-hill-ms", type=int, default=0)
    p.add_argument("--hang-before-ms", type=int, default=0)
    p.add_argument("--hang-over-ms", type=int, default=0)
    p.add_argument("--return-probs", action="store_true")
    p.add_argument("--probs-sample-rate", type=int, default=None)
    p.add_argument("--device", default="cuda")
    p.add_argument("--precision", choices=("fp32", "fp32s", "bf16"), default="fp32",
                   help="not in the reference: fp32 = exact-fp32 matrix cores; fp32s = the same results to fp32 rounding on the bf16 matrix "
                        "pipe (split operands, ~1.8x faster on long sequences); bf16 = bf16 operands (AUC parity, log-probs to ~1e-2). "
                        "bf16 results depend on how windows are batched (<= 3e-3 in log-probs) unless --batch-invariant is given.")
    p.add_argument("--batch-invariant", action="store_true",
                   help="bf16 only: the same window gives the same bits in every batching (+4 %% on large forwards); off by default.")
    p.add_argument("--graph", action="store_true",
                   help="replay clip-sized chunks (<= 120 s) as a captured HIP graph: pays when many chunks have the same length "
                        "(--split-max-seconds), same results.")
    e = sub.add_parser("evaluate", help="frame metrics over a labelled data list (vad/evaluate.py:20-29)")
    e.add_argument("eval_path", type=Path)
    e.add_argument("checkpoint_path", type=Path)
    e.add_argument("--output-path", type=Path, default=None, help="Path to store output. Default to stdout.")
    e.add_argument("--data-dir", type=Path, default=None)
    e.add_argument("--threshold", type=float, default=0.5)
    e.add_argument("--shuffle", action="store_true")
    e.add_argument("--limit", type=int, default=None)
    e.add_argument("--random-seed", type=int, default=0)
    e.add_argument("--device", default="cuda")
    args = ap.parse_args(argv)

    if args.command == "evaluate":
        from .evaluate import evaluate_vad_from_scratch

        evaluate_vad_from_scratch(args.eval_path, args.checkpoint_path, args.output_path, args.data_dir, args.threshold,
                                  args.shuffle, args.limit, args.random_seed, args.device)
        return 0

    from .predictor import VADFromScratchPredictor, VADPredictParameters

    predictor = VADFromScratchPredictor.from_checkpoint(args.checkpoint_path, args.device)
    predictor.model.precision, predictor.model.batch_invariant, predictor.graph = args.precision, args.batch_invariant, args.graph
    voice_activity = predictor.predict_from_path(
        args.audio_path,
        VADPredictParameters(args.split_max_seconds, args.threshold, args.min_vally_ms, args.min_hill_ms, args.hang_before_ms,
                             args.hang_over_ms, args.activity_max_sec, args.return_probs, args.probs_sample_rate, True))
    if args.output_path:
        args.output_path.parent.mkdir(parents=True, exist_ok=True)
        voice_activity.save(args.output_path)
    else:
        print(json.dumps(voice_activity.to_json(), ensure_ascii=False, indent=4))
    return 0


if __name__ == "__main__":
    sys.exit(main())
