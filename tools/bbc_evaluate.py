"""Score benchmark predictions (tools/bbc_harness.py / tools/bbc_standin.py --dump) with the REFERENCE's evaluator:
``benchmark.evaluator.evaluate`` (benchmark/evaluator.py:334-346; greedy 1-to-1 matching within a frame tolerance,
TRECVID-SBD precision / recall / F1, mean absolute offset), imported from the reference checkout, unmodified -- or, where
there is no checkout (the GPU box), with tools/bbc_scoring.py, which restates it and passes the reference's own tests for it
(tests/test_reference_own_tests.py).

    python tools/bbc_evaluate.py predictions.json [--tolerances 0,1,2] [--reference /root/reference]
"""
import argparse
import json
import sys
from pathlib import Path

ap = argparse.ArgumentParser()
ap.add_argument("predictions")
ap.add_argument("--tolerances", default="0,1,2")
ap.add_argument("--reference", default="/root/reference")
a = ap.parse_args()
import os  # noqa: E402

if os.path.isdir(os.path.join(a.reference, "benchmark")):
    sys.path.insert(0, a.reference)
    from benchmark.evaluator import GroundTruth, Prediction, evaluate  # noqa: E402  (the reference's module)
    which = "benchmark/evaluator.py of the reference checkout (%s)" % a.reference
else:
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from bbc_scoring import GroundTruth, Prediction, evaluate  # noqa: E402
    which = "tools/bbc_scoring.py (no reference checkout at %s)" % a.reference

d = json.load(open(a.predictions))
predictions = {Path(v["video_file"]): Prediction(predicted_cuts=v["predicted_cuts"], ground_truth=GroundTruth(hard_cuts=v["hard_cuts"]),
                                                 elapsed=v["elapsed"]) for v in d["videos"]}
out = {k: v for k, v in d.items() if k != "videos"}
out["evaluator"] = which
out["results"] = []
for tol in (int(x) for x in a.tolerances.split(",") if x.strip()):
    r = evaluate(predictions, tol)
    hc = r.hard_cuts
    out["results"].append({"tolerance": tol, "precision": round(hc.precision, 4), "recall": round(hc.recall, 4), "f1": round(hc.f1, 4),
                           "matched": hc.matched, "false_positives": hc.false_positives, "missed": hc.missed,
                           "mean_abs_offset": r.mean_abs_offset_hard_cuts, "elapsed_total_s": round(r.elapsed_total, 4),
                           "videos": len(predictions)})
print(json.dumps(out))
