"""Shot-boundary scoring for the benchmark harness (tools/bbc_harness.py, bench.py --workload bbc) where the reference
checkout is not at hand (the GPU box): the TRECVID-SBD convention of the reference's ``benchmark/evaluator.py`` --

* a prediction inside a ground-truth FADE interval is taken by that interval: the first one is the interval's match, every
  further one in the same interval a false positive; an interval nobody hits is missed (evaluator.py:268-300);
* what is left is matched 1-to-1 against the HARD cuts within ``tolerance`` frames, nearest pairs first
  (evaluator.py:227-265: all candidate pairs ordered by (distance, prediction index, truth index), each taking its two ends
  if both are still free);
* precision / recall / F1 from the counts, the mean absolute offset over the hard matches, sums (not means of means)
  across videos (evaluator.py:78-123, 152-215).

Same public names and fields as that module, so the reference's own ``tests/test_benchmark_evaluator.py`` runs against this
file (tests/test_reference_own_tests.py aliases ``benchmark.evaluator`` to it).  Bench / harness infrastructure: nothing in
``pyscenedetect_amd/`` imports it.  The pairing below does not enumerate all predictions x truths: both lists are walked in
frame order with a window of ``tolerance`` (BBC episodes hold hundreds of cuts, sweeps score thousands of runs).
"""
from __future__ import annotations

import bisect
import math
from dataclasses import dataclass, field
from pathlib import Path

Frames = int


@dataclass(frozen=True)
class EventInterval:
    """A gradual transition: frames ``start .. end``, both included."""

    start: Frames
    end: Frames

    def contains(self, frame: Frames) -> bool:
        return not (frame < self.start or frame > self.end)


@dataclass
class GroundTruth:
    hard_cuts: list[Frames]
    fades: list[EventInterval] = field(default_factory=list)
    category: str | None = None


@dataclass
class Prediction:
    predicted_cuts: list[Frames]
    ground_truth: GroundTruth
    elapsed: float


def _ratio(num: float, den: float) -> float:
    return num / den if den else 0.0


@dataclass
class EventMetrics:
    """Counts of one kind of event (hard cuts or fades) and what follows from them."""

    matched: int = 0
    false_positives: int = 0
    missed: int = 0

    @property
    def precision(self) -> float:
        return _ratio(self.matched, self.matched + self.false_positives)

    @property
    def recall(self) -> float:
        return _ratio(self.matched, self.matched + self.missed)

    @property
    def f1(self) -> float:
        return _ratio(2 * self.precision * self.recall, self.precision + self.recall)

    def __add__(self, other: "EventMetrics") -> "EventMetrics":
        return EventMetrics(self.matched + other.matched, self.false_positives + other.false_positives, self.missed + other.missed)

    def to_dict(self) -> dict:
        out = {"matched": self.matched, "false_positives": self.false_positives, "missed": self.missed}
        for name in ("precision", "recall", "f1"):
            out[name] = round(getattr(self, name) * 100, 4)
        return out


@dataclass
class VideoMetrics:
    elapsed: float
    category: str | None
    hard_cuts: EventMetrics
    fades: EventMetrics
    hard_offset: tuple[float, int]          # (sum of |prediction - truth| over the hard matches, number of matches)

    @property
    def mean_abs_offset(self) -> float:
        total, count = self.hard_offset
        return total / count if count else math.nan

    def to_dict(self) -> dict:
        return {"elapsed": self.elapsed, "category": self.category, "hard_cuts": self.hard_cuts.to_dict(), "fades": self.fades.to_dict(),
                "mean_abs_offset_hard_cuts": self.mean_abs_offset}


@dataclass
class BenchmarkResult:
    per_video: dict[Path, VideoMetrics]
    tolerance: Frames

    def _total(self, which: str) -> EventMetrics:
        total = EventMetrics()
        for video in self.per_video.values():
            total = total + getattr(video, which)
        return total

    @property
    def hard_cuts(self) -> EventMetrics:
        return self._total("hard_cuts")

    @property
    def fades(self) -> EventMetrics:
        return self._total("fades")

    @property
    def mean_abs_offset_hard_cuts(self) -> float:
        total = sum(v.hard_offset[0] for v in self.per_video.values())
        count = sum(v.hard_offset[1] for v in self.per_video.values())
        return total / count if count else math.nan

    @property
    def elapsed_total(self) -> float:
        return sum(v.elapsed for v in self.per_video.values())

    @property
    def elapsed_mean(self) -> float:
        return self.elapsed_total / len(self.per_video) if self.per_video else 0.0

    def by_category(self) -> dict[str, "BenchmarkResult"]:
        groups: dict[str, dict[Path, VideoMetrics]] = {}
        for path, video in self.per_video.items():
            groups.setdefault(video.category or "unknown", {})[path] = video
        return {name: BenchmarkResult(per_video=videos, tolerance=self.tolerance) for name, videos in groups.items()}

    def to_dict(self, root: Path | None = None) -> dict:
        def shown(path: Path) -> str:
            if root is not None:
                try:
                    return path.relative_to(root).as_posix()
                except ValueError:
                    pass
            return path.as_posix()

        return {"tolerance": self.tolerance,
                "aggregate": {"hard_cuts": self.hard_cuts.to_dict(), "mean_abs_offset_hard_cuts": self.mean_abs_offset_hard_cuts,
                              "fades": self.fades.to_dict(), "elapsed_total": self.elapsed_total, "elapsed_mean": self.elapsed_mean,
                              "video_count": len(self.per_video)},
                "per_video": {shown(path): video.to_dict() for path, video in self.per_video.items()}}


def _score_hard_cuts(predicted_cuts, ground_truth_cuts, tolerance: Frames) -> tuple[EventMetrics, list[Frames]]:
    """(counts, absolute offsets of the matches).  Every pair within ``tolerance`` is a candidate; candidates are taken in the
    order (distance, position of the prediction, position of the truth), a candidate wins if neither end is taken yet."""
    predicted, truth = list(predicted_cuts), list(ground_truth_cuts)
    # truths in frame order (with their positions in the caller's list): a prediction's candidates are one bisect window
    by_frame = sorted(range(len(truth)), key=lambda j: (truth[j], j))
    frames = [truth[j] for j in by_frame]
    candidates = []
    for i, p in enumerate(predicted):
        lo, hi = bisect.bisect_left(frames, p - tolerance), bisect.bisect_right(frames, p + tolerance)
        candidates.extend((abs(p - frames[k]), i, by_frame[k]) for k in range(lo, hi))
    candidates.sort()
    taken_p, taken_t, offsets = set(), set(), []
    for distance, i, j in candidates:
        if i not in taken_p and j not in taken_t:
            taken_p.add(i)
            taken_t.add(j)
            offsets.append(distance)
    hits = len(offsets)
    return EventMetrics(matched=hits, false_positives=len(predicted) - hits, missed=len(truth) - hits), offsets


def _score_fade_transitions(predicted_cuts, intervals) -> tuple[EventMetrics, set[int]]:
    """(counts, positions in ``predicted_cuts`` that an interval took).  A prediction belongs to the FIRST listed interval that
    contains it."""
    predicted, spans = list(predicted_cuts), list(intervals)
    owner_hits: dict[EventInterval, int] = {}
    taken: set[int] = set()
    for k, frame in enumerate(predicted):
        owner = next((span for span in spans if span.contains(frame)), None)
        if owner is not None:
            taken.add(k)
            owner_hits[owner] = owner_hits.get(owner, 0) + 1
    matched = len(owner_hits)
    extra = sum(owner_hits.values()) - matched
    return EventMetrics(matched=matched, false_positives=extra, missed=len(spans) - matched), taken


def score_video(predicted_cuts, ground_truth: GroundTruth, tolerance: Frames, elapsed: float) -> VideoMetrics:
    """Fades first (they take the predictions inside them), the rest against the hard cuts."""
    predicted = list(predicted_cuts)
    fades, taken = _score_fade_transitions(predicted, ground_truth.fades)
    hard, offsets = _score_hard_cuts([p for k, p in enumerate(predicted) if k not in taken], ground_truth.hard_cuts, tolerance)
    return VideoMetrics(elapsed=elapsed, category=ground_truth.category, hard_cuts=hard, fades=fades,
                        hard_offset=(float(sum(offsets)), len(offsets)))


def evaluate(predictions: dict[Path, Prediction], tolerance: Frames) -> BenchmarkResult:
    if not predictions:
        raise AssertionError("predictions must not be empty")
    return BenchmarkResult(per_video={path: score_video(p.predicted_cuts, p.ground_truth, tolerance, p.elapsed)
                                      for path, p in predictions.items()}, tolerance=tolerance)
