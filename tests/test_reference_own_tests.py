"""CPU: the REFERENCE'S OWN video-free tests, executed against the mirror.

``/root/reference/tests/test_timecode.py``, ``test_stats_manager.py``, the video-free cases of ``test_scene_manager.py`` and
``test_benchmark_evaluator.py`` hold the reference's assertions for SURVEY.md 8 rows a13 / a14 / a15 / f2 (FrameTimecode /
Timecode, StatsManager, SceneManager's crop and scene-list helpers, the benchmark's TRECVID-style scorer).  They are run unmodified, from where they lie, in a pytest subprocess whose ``scenedetect`` package is
``tests/ref_alias/scenedetect`` -- import aliases onto ``pyscenedetect_amd`` -- with ``--noconftest`` (the reference's conftest
imports cv2 and video fixtures) and importlib import mode (so that the reference checkout never lands on ``sys.path``).
Cases that open a video file are deselected BY NAME below; everything else must pass.  These are reference-held
assertions -- stronger than the differential goldens of test_timecode.py / test_host_golden.py.  (First run of this file found
four gaps: the deprecated ``get_framerate`` / ``equal_framerate`` / ``framerate`` names, ``Timecode`` as a minimum scene
length, and ``pathlib.Path`` stats files.)  ``test_api.py`` -- the reference's "common workflow patterns" -- runs too, on a synthetic
clip (see NEEDS_A_VIDEO below).  Skipped where the reference checkout does not exist (the GPU box)."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_TESTS = "/root/reference/tests"

NEEDS_A_VIDEO = {
    "test_stats_manager.py": ["test_detector_metrics", "test_save_load_from_video"],
    "test_scene_manager.py": ["test_scene_list", "test_get_scene_list_start_in_scene", "test_detect_scenes_callback",
                              "test_detect_scenes_callback_adaptive", "test_detect_scenes_crop"],
    "test_timecode.py": [],
    "test_benchmark_evaluator.py": [],      # (against tools/bbc_scoring.py, the harness's scorer on boxes without the reference)
}
# test_api.py: "common workflow patterns used when integrating the PySceneDetect API" -- detect() with start / end times and a stats
# file, SceneManager with seek / end_time, StatsManager.save_to_csv, a callback.  They assert nothing about WHERE the cuts are, so they
# run on the alias package's synthetic clip (tests/ref_alias/workflow_plugin.py provides the fixture the reference's conftest would);
# left out: open_video's deprecated keyword, cv2.VideoCapture, the deprecated import shims (decoders / packaging, out of scope).
NEEDS_A_VIDEO["test_api.py"] = ["test_api_open_video_framerate_legacy_alias", "test_api_device_callback", "test_deprecated_modules"]
WITH_PLUGIN = {"test_api.py"}
EXPECT_AT_LEAST = {"test_api.py": 8, "test_timecode.py": 35, "test_stats_manager.py": 5, "test_scene_manager.py": 5, "test_benchmark_evaluator.py": 24}


@pytest.mark.skipif(not os.path.isdir(REF_TESTS), reason="the reference checkout is only in the build container")
@pytest.mark.parametrize("name", sorted(NEEDS_A_VIDEO))
def test_reference_test_file_passes_against_the_mirror(name, tmp_path):
    env = dict(os.environ)
    env["PYTHONPATH"] = os.pathsep.join([os.path.join(ROOT, "tests", "ref_alias"), ROOT])
    cmd = [sys.executable, "-m", "pytest", "--noconftest", "--import-mode=importlib", "-p", "no:cacheprovider", "-q",
           os.path.join(REF_TESTS, name)]
    if name in WITH_PLUGIN:
        cmd[3:3] = ["-p", "workflow_plugin"]
    if NEEDS_A_VIDEO[name]:      # (exact names: none of them is a prefix of a case that stays)
        cmd += ["-k", "not (" + " or ".join(NEEDS_A_VIDEO[name]) + ")"]
    run = subprocess.run(cmd, cwd=tmp_path, env=env, capture_output=True, text=True, timeout=300)
    tail = run.stdout[-3000:] + run.stderr[-1000:]
    assert run.returncode == 0, tail
    m = re.search(r"(\d+) passed", run.stdout)
    assert m and int(m.group(1)) >= EXPECT_AT_LEAST[name], tail
    assert "failed" not in run.stdout.splitlines()[-1] and "error" not in run.stdout.splitlines()[-1], tail
    # the module under test really was the mirror
    probe = subprocess.run([sys.executable, "-c", "import scenedetect.common as c, scenedetect.stats_manager as s, benchmark.evaluator as b; "
                            "print(c.FrameTimecode.__module__, s.StatsManager.__module__, b.score_video.__module__)"], cwd=tmp_path, env=env,
                           capture_output=True, text=True, timeout=120)
    assert probe.stdout.split() == ["pyscenedetect_amd.timecode", "pyscenedetect_amd.stats_manager", "bbc_scoring"], probe.stdout + probe.stderr
