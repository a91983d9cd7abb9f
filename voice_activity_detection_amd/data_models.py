"""``Activity`` / ``VoiceActivity`` of the reference (``vad/data_models/voice_activity.py:30-162``) with the
JSON v0.3 format ``main.py predict`` writes, and the timecode helpers of ``vad/util/time_utils.py:6-32``."""
from __future__ import annotations

import json
from dataclasses import dataclass
from datetime import datetime, timedelta
from pathlib import Path
from typing import List, Optional


def format_timedelta_to_timecode(t: timedelta) -> str:
    """vad/util/time_utils.py:17-32 -- HH:MM:SS.mmm with milliseconds = round(microseconds / 1000)
    (Python's round-half-even; can print ".1000", kept)."""
    total_seconds = int(t.total_seconds())
    hours = total_seconds // 3600
    minutes = total_seconds % 3600 // 60
    seconds = total_seconds % 60
    milliseconds = round(t.microseconds / 1000)
    return f"{hours:02d}:{minutes:02d}:{seconds:02d}.{milliseconds:03d}"


def parse_timecode_to_timedelta(timecode: str) -> timedelta:
    """vad/util/time_utils.py:6-8"""
    return datetime.strptime(timecode, "%H:%M:%S.%f") - datetime(year=1900, month=1, day=1)


@dataclass
class Activity:
    start: timedelta
    end: timedelta


@dataclass
class VoiceActivity:
    duration: timedelta
    activities: List[Activity]
    probs_sample_rate: Optional[int]
    probs: Optional[List[float]]

    def to_json(self) -> dict:
        """JSON v0.3 (vad/data_models/voice_activity.py:146-159)."""
        return {
            "version": "v0.3",
            "duration": format_timedelta_to_timecode(self.duration),
            "activities": [{"start": format_timedelta_to_timecode(a.start), "end": format_timedelta_to_timecode(a.end)}
                           for a in self.activities],
            "probs_sample_rate": self.probs_sample_rate,
            "probs": self.probs,
        }

    def save(self, path: Path):
        with Path(path).open("w") as file:  # same dump settings as voice_activity.py:111-114
            json.dump(self.to_json(), file, ensure_ascii=False, indent=4)

    def to_labels(self, sample_rate: int):
        """vad/data_models/voice_activity.py:239-246: 0/1 label per 1/sample_rate second."""
        import numpy as np

        labels = np.zeros(int(self.duration.total_seconds() * sample_rate), dtype=np.int64)
        for a in self.activities:
            labels[int(a.start.total_seconds() * sample_rate):int(a.end.total_seconds() * sample_rate)] = 1
        return labels

    @classmethod
    def from_json(cls, data: dict) -> "VoiceActivity":
        if data.get("version") != "v0.3":
            raise NotImplementedError("only the v0.3 format is restated")
        return cls(duration=parse_timecode_to_timedelta(data["duration"]),
                   activities=[Activity(parse_timecode_to_timedelta(a["start"]), parse_timecode_to_timedelta(a["end"]))
                               for a in data["activities"]],
                   probs_sample_rate=data.get("probs_sample_rate"), probs=data.get("probs"))

    @classmethod
    def load(cls, path: Path) -> "VoiceActivity":
        with Path(path).open() as file:
            return cls.from_json(json.load(file))


def merge_voice_activities(voice_activities: List[VoiceActivity]) -> VoiceActivity:
    """vad/predictor.py:283-304"""
    offset = timedelta(0)
    new_activities = []
    for va in voice_activities:
        for a in va.activities:
            new_activities.append(Activity(start=a.start + offset, end=a.end + offset))
        offset += va.duration
    new_probs = None
    if voice_activities[0].probs:
        new_probs = [p for va in voice_activities for p in va.probs]
    return VoiceActivity(duration=sum([va.duration for va in voice_activities], timedelta(0)), activities=new_activities,
                         probs_sample_rate=voice_activities[0].probs_sample_rate, probs=new_probs)
