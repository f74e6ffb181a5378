"""Per-frame metric store (reference ``scenedetect/stats_manager.py:85-314``).

Detectors write the same metric keys as the reference (``content_val``, ``delta_hue`` ...,
``adaptive_ratio (w=N)``, ``hist_diff [bins=N]``, ``average_rgb``); ``ThresholdDetector`` reads
``average_rgb`` back if it is already present (``threshold_detector.py:122-125``).
"""

import csv
import logging
import os

from pyscenedetect_amd.timecode import FrameTimecode, _is_foreign_timecode

logger = logging.getLogger("pyscenedetect")

COLUMN_NAME_FRAME_NUMBER = "Frame Number"
COLUMN_NAME_TIMECODE = "Timecode"


class FrameMetricRegistered(Exception):
    """A metric key was registered twice."""

    def __init__(self, metric_key: str, message: str = "Attempted to re-register frame metric key."):
        super().__init__(message)
        self.metric_key = metric_key


class FrameMetricNotRegistered(Exception):
    """Kept for code that still names it: the reference no longer raises it (``stats_manager.py:60-66``, deprecated)."""


class StatsFileCorrupt(Exception):
    """The stats file is not one this class wrote (reference ``stats_manager.py:60-70``)."""

    def __init__(self, message: str = "Could not load frame metric data data from passed CSV file."):
        super().__init__(message)


class StatsManager:
    def __init__(self, base_timecode: FrameTimecode | None = None):
        # keyed by an int frame number or a FrameTimecode: both hash / compare to the same slot (FrameTimecode.__hash__ is the frame
        # number), and the slot keeps the key object it was created with (reference stats_manager.py:106-110)
        self._frame_metrics: dict = {}
        self._metric_keys: set[str] = set()
        self._metrics_updated = False
        self._base_timecode = base_timecode

    @property
    def metric_keys(self):
        return self._metric_keys

    def register_metrics(self, metric_keys) -> None:
        self._metric_keys = self._metric_keys.union(set(metric_keys))

    # (argument names and semantics as in the reference, stats_manager.py:126-153 and 300-314: `timecode` is a frame number or a
    #  FrameTimecode; setting an EMPTY dict creates nothing and does not mark the manager dirty; an empty key list "exists" anywhere)
    def get_metrics(self, timecode, metric_keys) -> list:
        if _is_foreign_timecode(timecode):      # another library's timecode as the key (a detector written against the reference's ABC)
            timecode = FrameTimecode(timecode)
        row = self._frame_metrics.get(timecode)
        return [row.get(key) if row is not None else None for key in metric_keys]

    def set_metrics(self, timecode, metric_kv_dict: dict) -> None:
        if _is_foreign_timecode(timecode):
            timecode = FrameTimecode(timecode)
        if not metric_kv_dict:
            return
        self._metrics_updated = True
        row = self._frame_metrics.get(timecode)          # (one lookup for the whole dict: every lookup hashes and compares a timecode)
        if row is None:
            row = self._frame_metrics[timecode] = {}
        row.update(metric_kv_dict)

    def metrics_exist(self, timecode, metric_keys) -> bool:
        if _is_foreign_timecode(timecode):
            timecode = FrameTimecode(timecode)
        row = self._frame_metrics.get(timecode)
        return all([row is not None and key in row for key in metric_keys])

    def is_save_required(self) -> bool:
        return self._metrics_updated

    def save_to_csv(self, csv_file, force_save: bool = True, base_timecode: FrameTimecode | None = None) -> None:
        """``Frame Number, Timecode, <sorted metric keys>`` with 1-based frame numbers
        (reference ``stats_manager.py:164-203``: ``save_to_csv(csv_file, force_save=True)``).  ``base_timecode`` (keyword, not in
        the reference) overrides the time base ``detect_scenes`` left behind, for metrics keyed by plain frame numbers."""
        if not (force_save or self.is_save_required()):
            logger.info("No metrics to write.")
            return
        close = False
        if isinstance(csv_file, (str, bytes, os.PathLike)):      # a path in any spelling (reference stats_manager.py:187)
            csv_file = open(csv_file, "w", newline="")
            close = True
        try:
            writer = csv.writer(csv_file, lineterminator="\n")
            skipped = 0
            keys = sorted(self._metric_keys)
            writer.writerow([COLUMN_NAME_FRAME_NUMBER, COLUMN_NAME_TIMECODE] + keys)
            logger.info("Writing %d frames to CSV...", len(self._frame_metrics))
            for key in sorted(self._frame_metrics.keys()):      # (the keys' own order, like the reference: two timestamps that round
                                                                  #  to one frame number are two rows, the earlier first)
                # A row whose key is a bare frame number -- set through the public API with an int, or read back by the deprecated
                # load_from_csv -- has no timecode to print: the reference skips it (stats_manager.py:196-199; a later set_metrics
                # with a FrameTimecode for the same frame updates the row but the dict keeps the int key), unless the caller names
                # the time base (`base_timecode`, a keyword the reference does not have).
                if isinstance(key, FrameTimecode):
                    tc = key
                elif base_timecode is not None:
                    tc = base_timecode + int(key)
                else:
                    skipped += 1
                    continue
                row = self._frame_metrics[key]
                writer.writerow([int(key) + 1, tc.get_timecode()] + [str(row.get(k, "None")) for k in keys])
            if skipped:
                # (on this package's OWN logger: the reference skips these rows without a word, and the records of its logger,
                #  "pyscenedetect", are compared line by line with the mirror's by the differential fuzzers)
                logging.getLogger("pyscenedetect_amd").info(
                    "save_to_csv: %d row(s) keyed by a bare frame number were not written (no timecode to print; pass base_timecode=... "
                    "to write them)", skipped)
        finally:
            if close:
                csv_file.close()

    @staticmethod
    def valid_header(row) -> bool:
        """``Frame Number, Timecode, ...`` (reference ``stats_manager.py:205-216``)."""
        return bool(row) and len(row) >= 2 and row[0] == COLUMN_NAME_FRAME_NUMBER and row[1] == COLUMN_NAME_TIMECODE

    def load_from_csv(self, csv_file):
        """Read metrics written by :meth:`save_to_csv` back (reference ``stats_manager.py:220-296``, deprecated
        there but still what lets ``ThresholdDetector`` reuse a cached ``average_rgb``).  Rows are keyed by 0-based
        frame number.  Returns the number of rows, or None for a missing / empty file; raises
        :class:`StatsFileCorrupt` for anything that is not a stats file."""
        # (logged once per call and once more for the recursion on the opened file, like the reference, stats_manager.py:241-248)
        logger.warning("load_from_csv() is deprecated and will be removed in a future release.")
        if isinstance(csv_file, (str, bytes, os.PathLike)):
            if not os.path.exists(csv_file):
                return None
            with open(csv_file) as handle:
                return self.load_from_csv(handle)
        reader = csv.reader(csv_file, lineterminator="\n")
        try:
            row = next(reader)
            if not self.valid_header(row):   # older files carried one extra header line
                row = next(reader)
        except StopIteration:
            return None
        if not self.valid_header(row):
            raise StatsFileCorrupt()
        num_cols = len(row)
        if num_cols - 2 <= 0:
            raise StatsFileCorrupt("No metrics defined in CSV file.")
        keys = list(row[2:])
        num_frames = 0
        for row in reader:
            if len(row) != num_cols:
                raise StatsFileCorrupt("Wrong number of columns detected in stats file row.")
            frame_number = int(row[0])
            if frame_number > 0:
                frame_number -= 1
            # (a row of "None"s creates nothing; a file that turns out corrupt half-way leaves what was read before it, the keys
            #  unregistered and the manager dirty -- reference stats_manager.py:276-296)
            for key, text in zip(keys, row[2:]):
                if text and text != "None":
                    try:
                        self.set_metrics(frame_number, {key: float(text)})
                    except ValueError:
                        raise StatsFileCorrupt(f"Corrupted value in stats file: {text}") from ValueError
            num_frames += 1
        self.register_metrics(keys)
        logger.info("Loaded %d metrics for %d frames.", num_cols - 2, num_frames)
        self._metrics_updated = False
        return num_frames
