"""SceneManager's small surface as scripted cases, shared by oracle/gen_api_golden.py (which runs them against the
REFERENCE's SceneManager and stores the outcomes in tests/golden/api_cases.json["manager_ops"]) and tests/test_host_golden.py
(which replays them against pyscenedetect_amd).  No import of either package here.

TEST INFRASTRUCTURE ONLY.
"""


def outcome(fn):
    try:
        return {"ok": fn()}
    except Exception as ex:  # noqa: BLE001
        return {"raises": type(ex).__name__}


# ---- SceneManager's small surface: property setters, argument checks of detect_scenes, clear(), crop at the frame border
# (scene_manager.py:278-335, 358-372, 446-530).  Each case is a list of steps run against ONE manager; a step's outcome is its
# value or the exception type, plus the WARNING-and-above records the "pyscenedetect" logger saw during the step.
MANAGER_CASES = {
    "downscale_values": [["set", "auto_downscale", False], ["set", "downscale", 0], ["set", "downscale", -3], ["set", "downscale", 1],
                         ["get", "downscale"], ["set", "downscale", 2], ["get", "downscale"], ["set", "downscale", 2.7],
                         ["get", "downscale"]],
    "downscale_while_auto": [["get", "auto_downscale"], ["set", "downscale", 3], ["get", "downscale"]],
    "crop_values": [["set", "crop", [1, 2, 3]], ["set", "crop", [0, 0, 5, 5.0]], ["set", "crop", [-1, 0, 5, 5]],
                    ["set", "crop", [30, 20, 10, 5]], ["get", "crop"], ["set", "crop", None], ["get", "crop"]],
    "detect_argument_errors": [["detect", {"duration": -1}], ["detect", {"end_time": -2}], ["detect", {"duration": 5, "end_time": 9}],
                               ["detect", {"video": None}]],
    "frame_skip_with_stats": [["detect", {"frame_skip": 1, "stats": True}]],
    "lists_before_detect": [["cuts"], ["scenes"], ["scenes_start_in_scene"], ["num_detectors"]],
    "clear_after_detect": [["set", "auto_downscale", False], ["detect", {}], ["cuts"], ["num_detectors"], ["clear"], ["cuts"], ["scenes"],
                           ["num_detectors"]],
    "crop_past_the_border": [["set", "auto_downscale", False], ["set", "crop", [100, 50, 400, 300]], ["detect", {}], ["cuts"]],
    "crop_outside": [["set", "auto_downscale", False], ["set", "crop", [200, 10, 260, 40]], ["detect", {}]],
    "crop_inside": [["set", "auto_downscale", False], ["set", "crop", [8, 4, 100, 60]], ["detect", {}], ["cuts"]],
    "duration_and_end_time": [["set", "auto_downscale", False], ["detect", {"duration": 40}], ["cuts"], ["scenes"]],
    "end_time_seconds": [["set", "auto_downscale", False], ["detect", {"end_time": 3.0}], ["cuts"]],
}


def manager_cases(frames_of, make_manager, make_stream, make_detector, logger_name="pyscenedetect"):
    """Run MANAGER_CASES with the given factories (the reference's here, pyscenedetect_amd's in the test)."""
    import logging

    class Catch(logging.Handler):
        def __init__(self):
            super().__init__(logging.WARNING)
            self.seen = []

        def emit(self, record):
            self.seen.append([record.levelname, record.getMessage()])

    out = {}
    for name, steps in MANAGER_CASES.items():
        sm = None
        results = []
        for step in steps:
            catch = Catch()
            log = logging.getLogger(logger_name)
            log.addHandler(catch)
            try:
                if sm is None:
                    with_stats = any(s[0] == "detect" and s[1].get("stats") for s in steps)
                    sm = make_manager(with_stats)
                    sm.add_detector(make_detector())
                op = step[0]
                if op == "set":
                    value = tuple(step[2]) if isinstance(step[2], list) else step[2]
                    res = outcome(lambda: setattr(sm, step[1], value))
                elif op == "get":
                    res = outcome(lambda: (list(getattr(sm, step[1])) if isinstance(getattr(sm, step[1]), tuple) else getattr(sm, step[1])))
                elif op == "detect":
                    kw = {k: v for k, v in step[1].items() if k != "stats"}
                    if "video" not in kw:
                        kw["video"] = make_stream(frames_of())
                    res = outcome(lambda: sm.detect_scenes(**kw))
                elif op == "cuts":
                    res = outcome(lambda: [c.frame_num for c in sm.get_cut_list()])
                elif op in ("scenes", "scenes_start_in_scene"):
                    res = outcome(lambda: [[a.frame_num, b.frame_num] for a, b in sm.get_scene_list(start_in_scene=op != "scenes")])
                elif op == "num_detectors":
                    res = outcome(lambda: sm.get_num_detectors())
                elif op == "clear":
                    res = outcome(lambda: sm.clear())
                else:
                    raise AssertionError(op)
            finally:
                log.removeHandler(catch)
            res["log"] = catch.seen
            results.append(res)
        out[name] = results
    return out
