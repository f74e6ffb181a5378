"""CPU: the GPU tests that stay at the SceneManager / detector level, run with ``hip_engine`` replaced by the host-memory stand-in of the
device engine (``tools/sim_as_hip.py``: poisoned buffers, batched row uploads that land at the fence, slots, halo, resident per-frame
buffers).  What they assert about the mirror's behaviour -- goldens through the device feeder, scenarios, the carried frame, the
per-frame API, plug-in detectors behind a downscale -- is then checked on every CPU run: a change to the host side that breaks one of
their expectations shows HERE, not at the next GPU run.  (It did once: when the predecessor of a manager's first frame became what its
detectors saw last -- the reference's semantics -- ``test_one_manager_on_two_videos_carries_the_last_frame`` still asserted that
``clear()`` drops it for detectors that are added again.)  Kernels are not involved; the GPU box runs the same tests over the real engine."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

OVER_THE_SIM = [
    "tests/test_gpu_feed_rows.py::test_one_manager_on_two_videos_carries_the_last_frame",
    "tests/test_gpu_feed_rows.py::test_scene_manager_feeds_tap_rows_and_decides_like_the_oracle",
    "tests/test_gpu_feed_rows.py::test_reference_runs_on_larger_frames_through_the_row_feeder",
    "tests/test_gpu_feed_rows.py::test_a_manager_of_plug_in_detectors_only_hands_them_downscaled_frames",
    "tests/test_gpu_feed_rows.py::test_callback_of_a_later_call_gets_the_downscaled_frame_an_earlier_call_buffered",
    "tests/test_gpu_parity.py::test_reference_golden_runs_through_hip",
    "tests/test_gpu_parity.py::test_auto_downscale_golden_through_hip",
    "tests/test_gpu_parity.py::test_downscale_interpolation_modes_golden_through_hip",
    "tests/test_gpu_parity.py::test_per_frame_api_on_gpu",
    "tests/test_gpu_parity.py::test_hash_detector_per_frame_api_and_mixed_pass",
    "tests/test_gpu_flows.py::test_benchmark_harness_through_hip",
    "tests/test_scene_manager.py::test_scenarios_match_reference_on_gpu",
    "tests/test_scene_manager.py::test_detectors_do_not_depend_on_co_registered_detectors_gpu",
]


def test_manager_level_gpu_tests_pass_over_the_simulated_device_engine():
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([os.path.join(ROOT, "tools"), ROOT]))
    run = subprocess.run([sys.executable, "-m", "pytest", "-p", "sim_as_hip", "-q", "-p", "no:cacheprovider", "-m", "gpu"] + OVER_THE_SIM,
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    tail = run.stdout[-3000:] + run.stderr[-1500:]
    assert run.returncode == 0, tail
    last = run.stdout.strip().splitlines()[-1]
    assert "failed" not in last and "error" not in last, tail
    counted = sum(int(n) for n in re.findall(r"(\d+) (?:passed|xpassed)", last))
    assert counted >= len(OVER_THE_SIM), tail
