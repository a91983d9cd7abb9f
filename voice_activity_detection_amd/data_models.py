"""``Activity`` / ``VoiceActivity`` of the reference (``vad/data_models/voice_activity.py:30-162``) with the
JSON formats (v0.3 = what ``main.py predict`` writes; v0.1 / v0.2 / milliseconds readers and writers), and the timecode helpers of ``vad/util/time_utils.py:6-32``."""
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


def format_timedelta_to_milliseconds(t: timedelta) -> int:
    """vad/util/time_utils.py:35-36"""
    return int(t.total_seconds() * 1000)


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

    def to_json(self, version: str = "v0.3") -> dict:
        """vad/data_models/voice_activity.py:116-162: JSON v0.3 (what ``main.py predict`` writes) and the two older
        timecode layouts (v0.1; v0.2 adds ``time_format``)."""
        tc = format_timedelta_to_timecode
        if version == "v0.3":
            return {
                "version": "v0.3",
                "duration": tc(self.duration),
                "activities": [{"start": tc(a.start), "end": tc(a.end)} for a in self.activities],
                "probs_sample_rate": self.probs_sample_rate,
                "probs": self.probs,
            }
        if version not in ("v0.1", "v0.2"):
            raise NotImplementedError(version)
        out = {"version": version, "duration": tc(self.duration)}
        if version == "v0.2":
            out["time_format"] = "timecode"
        out["voice_activity"] = [{"start_time": tc(a.start), "end_time": tc(a.end)} for a in self.activities]
        out["probs_sample_rate"] = self.probs_sample_rate
        out["probs"] = self.probs
        return out

    def to_milliseconds(self, version: str = "v0.3") -> dict:
        """vad/data_models/voice_activity.py:164-205 (integer milliseconds, truncated: time_utils.py:35-36)."""
        ms = format_timedelta_to_milliseconds
        if version == "v0.2":
            return {"version": version, "duration": ms(self.duration), "time_format": "millisecond",
                    "voice_activity": [{"start_time": ms(a.start), "end_time": ms(a.end)} for a in self.activities],
                    "probs_sample_rate": self.probs_sample_rate, "probs": self.probs}
        if version == "v0.3":
            return {"version": version, "duration": {"total_milliseconds": ms(self.duration)},
                    "activities": [{"start": {"total_milliseconds": ms(a.start)}, "end": {"total_milliseconds": ms(a.end)}}
                                   for a in self.activities],
                    "probs_sample_rate": self.probs_sample_rate, "probs": self.probs}
        raise NotImplementedError(version)

    def save(self, path: Path, version: str = "v0.3"):
        with Path(path).open("w") as file:  # same dump settings as voice_activity.py:111-114
            json.dump(self.to_json(version), file, ensure_ascii=False, indent=4)

    def to_labels(self, sample_rate: int):
        """vad/data_models/voice_activity.py:239-246: 0/1 label per 1/sample_rate second."""
        import numpy as np

        labels = np.zeros(int(self.duration.total_seconds() * sample_rate), dtype=np.int64)
        for a in self.activities:
            labels[int(a.start.total_seconds() * sample_rate):int(a.end.total_seconds() * sample_rate)] = 1
        return labels

    @classmethod
    def from_json(cls, data: dict) -> "VoiceActivity":
        """vad/data_models/voice_activity.py:49-109: v0.3, v0.1, and v0.2 in its timecode / millisecond flavours."""
        version = data["version"]
        tc = parse_timecode_to_timedelta
        if version == "v0.3":
            duration = tc(data["duration"])
            acts = [Activity(tc(a["start"]), tc(a["end"])) for a in data["activities"]]
        elif version == "v0.1" or (version == "v0.2" and data["time_format"] == "timecode"):
            duration = tc(data["duration"])
            acts = [Activity(tc(a["start_time"]), tc(a["end_time"])) for a in data["voice_activity"]]
        elif version == "v0.2" and data["time_format"] == "millisecond":
            duration = timedelta(milliseconds=data["duration"])
            acts = [Activity(timedelta(milliseconds=a["start_time"]), timedelta(milliseconds=a["end_time"]))
                    for a in data["voice_activity"]]
        else:
            raise NotImplementedError(f"VoiceActivity version {version!r}")
        return cls(duration=duration, activities=acts, probs_sample_rate=data.get("probs_sample_rate"), probs=data.get("probs"))

    @classmethod
    def from_milliseconds(cls, data: dict) -> "VoiceActivity":
        """vad/data_models/voice_activity.py:207-237"""
        version = data["version"]
        if version == "v0.2":
            duration = timedelta(milliseconds=data["duration"])
            acts = [Activity(timedelta(milliseconds=a["start_time"]), timedelta(milliseconds=a["end_time"]))
                    for a in data["voice_activity"]]
        elif version == "v0.3":
            duration = timedelta(milliseconds=data["duration"]["total_milliseconds"])
            acts = [Activity(timedelta(milliseconds=a["start"]["total_milliseconds"]),
                             timedelta(milliseconds=a["end"]["total_milliseconds"])) for a in data["activities"]]
        else:
            raise NotImplementedError(f"VoiceActivity milliseconds version {version!r}")
        return cls(duration=duration, activities=acts, probs_sample_rate=data.get("probs_sample_rate"), probs=data.get("probs"))

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
