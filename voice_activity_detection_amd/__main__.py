"""``python -m voice_activity_detection_amd predict AUDIO CHECKPOINT [options]`` -- the reference's
``python main.py predict`` (``main.py:9``, ``vad/predict.py:10-50``) on the MI355X path: same positional
arguments, same options, JSON v0.3 to ``--output-path`` or a printed summary."""
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
    args = ap.parse_args(argv)

    from .predictor import VADFromScratchPredictor, VADPredictParameters

    predictor = VADFromScratchPredictor.from_checkpoint(args.checkpoint_path, args.device)
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
