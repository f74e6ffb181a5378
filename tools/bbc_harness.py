"""The reference's benchmark run (benchmark/__main__.py:44-61, `_run_predictions`) with the detectors of this package.

For every sample of a dataset: open a frame source, run ``SceneManager.detect_scenes`` with one default-constructed
detector, take ``[scene[1].frame_num for scene in scene_list]`` as the predicted cuts (the reference's convention: the
end of every scene, the last one included) and the wall-clock of the run.  The records are written as JSON;
``tools/bbc_evaluate.py`` scores them with the reference's own ``benchmark/evaluator.py`` (TRECVID-SBD precision /
recall / F1 at a frame tolerance), which needs the reference checkout and no GPU.

There is no video decoder in this image, so the stock frame source is ``NpyVideoStream`` (frames dumped to
``uint8[N,H,W,3]`` .npy files, memory-mapped); ``open_stream`` may return ANY object with the VideoStream members
``SceneManager`` uses -- the reference's ``VideoStreamCv2`` / ``VideoStreamAv`` included -- so the day a decoder and the
BBC files (benchmark/README.md:58-68, layout ``BBC/videos/bbc_<id>.mp4`` + ``BBC/fixed/<id>-scenes.txt``,
benchmark/dataset.py:77-106) are present this runs unchanged.

    python tools/bbc_harness.py --dataset-dir BBC --detector detect-adaptive --out predictions.json
"""
import argparse
import glob
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import pyscenedetect_amd as psd  # noqa: E402

#: the reference's table (benchmark/_common.py:37-43) on this package's classes
DETECTORS = {
    "detect-adaptive": psd.AdaptiveDetector,
    "detect-content": psd.ContentDetector,
    "detect-hash": psd.HashDetector,
    "detect-hist": psd.HistogramDetector,
    "detect-threshold": psd.ThresholdDetector,
}


class NpyVideoStream(psd.ArrayVideoStream):
    """Frames of one video dumped to a ``uint8[N,H,W,3]`` .npy file (memory-mapped, so only what is read is paged in)."""

    def __init__(self, path: str, fps: float = 25.0):
        super().__init__(np.load(path, mmap_mode="r"), fps, name=os.path.basename(path))


def read_tab_separated_cuts(scene_file: str) -> list[int]:
    """BBC / AutoShot annotations exactly as the reference reads them (benchmark/dataset.py:66-74): second column =
    0-based frame index of a hard cut, returned 1-based."""
    with open(scene_file) as f:
        return [int(line.strip().split("\t")[1]) + 1 for line in f if line.strip()]


def bbc_samples(dataset_dir: str) -> list[dict]:
    """``videos/*`` paired with ``fixed/*-scenes.txt`` by id, like ``BBCDataset`` (benchmark/dataset.py:87-106)."""
    videos = sorted(glob.glob(os.path.join(dataset_dir, "videos", "*")))
    scenes = sorted(glob.glob(os.path.join(dataset_dir, "fixed", "*.txt")))
    if len(videos) != len(scenes):
        raise ValueError(f"BBC dataset at {dataset_dir!r}: {len(videos)} videos but {len(scenes)} annotation files.")
    out = []
    for v, s in zip(videos, scenes):
        vid = os.path.basename(v).replace("bbc_", "").split(".")[0]
        if vid != os.path.basename(s).split("-")[0]:
            raise ValueError(f"BBC id mismatch: {v} vs {s}")
        out.append({"video_file": v, "hard_cuts": read_tab_separated_cuts(s)})
    return out


def run_predictions(samples, detector_name: str, open_stream=None, engine=None) -> list[dict]:
    """One default-constructed detector per video through ``SceneManager.detect_scenes`` (the reference's ``detect()``:
    scenedetect/__init__.py:150-230); returns one record per sample."""
    open_stream = open_stream or (lambda sample: NpyVideoStream(sample["video_file"]))
    records = []
    for sample in samples:
        start = time.time()
        kw = {"engine": engine} if engine is not None else {}
        sm = psd.SceneManager(**kw)
        sm.add_detector(DETECTORS[detector_name](**kw))
        sm.detect_scenes(open_stream(sample))
        scene_list = sm.get_scene_list()
        records.append({"video_file": str(sample["video_file"]), "predicted_cuts": [scene[1].frame_num for scene in scene_list],
                        "hard_cuts": [int(c) for c in sample["hard_cuts"]], "elapsed": time.time() - start})
    return records


#: the reference's table entries the packed flow decides (corpus.DETECTORS; HashDetector is scored per video)
PACKED = {"detect-adaptive": "adaptive", "detect-content": "content", "detect-hist": "hist", "detect-threshold": "threshold"}


def run_predictions_packed(samples, detector_name: str, engine, open_frames=None, fps: float = 25.0) -> list[dict]:
    """The same records from the PACKED flow (``corpus.detect_corpus``: every video of the dataset -- or of this rank's shard
    under ``torch.distributed`` -- in shared device batches behind the reference's default downscale,
    ``psd_score_segments_downscaled_device``, native decisions): what ``run_predictions`` computes one SceneManager at a time.
    ``elapsed`` is the whole call divided by the number of videos (the videos are not scored one after the other).
    ``open_frames(sample)`` returns ``uint8[N,H,W,3]`` (default: the sample's .npy file, memory-mapped)."""
    from pyscenedetect_amd import corpus

    name = PACKED[detector_name]
    open_frames = open_frames or (lambda sample: np.load(sample["video_file"], mmap_mode="r"))
    start = time.time()
    clips = [open_frames(sample) for sample in samples]
    results = corpus.detect_corpus(engine, clips, fps, {name: {}})
    each = (time.time() - start) / max(1, len(samples))
    records = []
    for sample, clip, res in zip(samples, clips, results):
        cuts = [int(c) for c in res[name]]
        # detect() returns the scene list with start_in_scene=False: no cuts, no scenes (scene_manager.py:397-400); otherwise every
        # scene's end, the last one the end of the video (:394-396)
        ends = cuts + [int(clip.shape[0])] if cuts else []
        records.append({"video_file": str(sample["video_file"]), "predicted_cuts": ends, "hard_cuts": [int(c) for c in sample["hard_cuts"]],
                        "elapsed": each})
    return records


def dump(records, path: str, **meta) -> None:
    with open(path, "w") as f:
        json.dump({"format": "psd-benchmark-predictions/1", **meta, "videos": records}, f)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--dataset-dir", required=True)
    ap.add_argument("--detector", default="detect-adaptive", choices=sorted(DETECTORS))
    ap.add_argument("--out", default="predictions.json")
    ap.add_argument("--packed", action="store_true", help="all videos in shared device batches (corpus.detect_corpus) instead of one "
                                                          "SceneManager per video; the same predictions")
    a = ap.parse_args()
    if a.packed:
        recs = run_predictions_packed(bbc_samples(a.dataset_dir), a.detector, psd.engine.default_engine())
    else:
        recs = run_predictions(bbc_samples(a.dataset_dir), a.detector)
    dump(recs, a.out, detector=a.detector, dataset=a.dataset_dir)
    print("wrote", a.out, len(recs), "videos")
