"""CPU: bench.py's control flow (2-deep pipelining, the score-vector exchange, max-over-ranks
timing, exactly one JSON line on stdout) run for real over gloo with world_size 1 and 2, with a
stand-in engine.  The numbers mean nothing here; the N > 1 code path must simply work."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER = os.path.join(ROOT, "tests", "_bench_dry_driver.py")
ARGS = ["--steps", "3", "--warmup", "1", "--frames", "12", "--height", "36", "--width", "64", "--cpu-sample", "8"]
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
        "dtype", "data", "config", "roofline", "cpu_baseline"}


def _check(stdout: str, n_gpus: int):
    lines = [ln for ln in stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, f"stdout must be exactly one JSON line, got: {lines!r}"
    j = json.loads(lines[0])
    assert KEYS <= set(j)
    assert j["n_gpus"] == n_gpus and j["steps"] == 3 and j["warmup"] == 1 and j["value"] > 0
    assert j["scaling"] == "weak" and j["higher_is_better"] is True and j["vs_baseline"] is None and j["dtype"] == "u8"
    assert set(j["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"}
    assert "workload" in j["config"] and "model" not in j["config"]
    return j


@pytest.mark.timeout(300)
def test_single_process():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, DRIVER, *ARGS], capture_output=True, text=True, env=env, cwd=ROOT, timeout=280)
    assert out.returncode == 0, out.stderr[-2000:]
    j = _check(out.stdout, 1)
    assert j["cpu_baseline"]["kind"] == "port" and "identical" in j["parity_sample"]


@pytest.mark.timeout(300)
def test_two_ranks_over_gloo():
    port = 29600 + os.getpid() % 300
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), DRIVER, "--gpus", "2", *ARGS]
    out = subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT, timeout=280)
    assert out.returncode == 0, out.stderr[-3000:]
    j = _check(out.stdout, 2)
    assert j["cpu_baseline"] is None and "RCCL all-gather" in j["config"]["parallelism"]
    assert j["config"]["exchange"].startswith("one all-gather of all 3 timed steps")


@pytest.mark.timeout(300)
def test_two_ranks_over_gloo_with_the_exchange_every_step():
    """--exchange step: the round-1..3 form (an all-gather of the score vectors every step) stays runnable."""
    port = 29900 + os.getpid() % 90
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), DRIVER, "--gpus", "2", "--exchange", "step", *ARGS]
    out = subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT, timeout=280)
    assert out.returncode == 0, out.stderr[-3000:]
    j = _check(out.stdout, 2)
    assert j["config"]["exchange"].startswith("every step")


def _env_without_launcher():
    return {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT",
                                                             "LOCAL_WORLD_SIZE", "GROUP_RANK", "TORCHELASTIC_RUN_ID")}


@pytest.mark.timeout(300)
def test_gpus_2_without_a_launcher_starts_two_ranks():
    """`python bench.py --gpus 2` on its own (no torchrun around it) must start the two ranks itself and still print
    exactly one JSON line that says n_gpus = 2 (round-2 review: the flag was parsed and ignored)."""
    out = subprocess.run([sys.executable, DRIVER, "--gpus", "2", *ARGS], capture_output=True, text=True, env=_env_without_launcher(),
                         cwd=ROOT, timeout=280)
    assert out.returncode == 0, out.stderr[-3000:]
    j = _check(out.stdout, 2)
    assert j["config"]["ranks_seen"] == 2 and "RCCL all-gather" in j["config"]["parallelism"]


@pytest.mark.timeout(300)
@pytest.mark.parametrize("workload,gpus,size,pipeline", [("corpus", 1, (36, 64), "default"), ("corpus", 2, (90, 160), "default"),
                                                         ("bbc", 2, (90, 320), "default"), ("bbc", 1, (90, 320), "full")])
def test_flow_workloads(workload, gpus, size, pipeline):
    """BASELINE configs[3] / [4] as bench workloads: same JSON contract, parity sample against the oracle, sharded by
    clip over gloo when N = 2 (self-launched).  By default the flows score what the reference's detect() scores (frames wider
    than 256 pixels behind its auto-downscale: 320 x 90 -> 256 x 72; the corpus' second resolution 320 x 180 -> 256 x 144);
    --flow-pipeline full is the labelled full-resolution form."""
    extra = ["--workload", workload, "--corpus-frames", "48", "--bbc-frames", "40", "--cpu-sample", "64", "--flow-pipeline", pipeline]
    args = [a for a in ARGS]
    args[args.index("--height") + 1], args[args.index("--width") + 1] = str(size[0]), str(size[1])
    out = subprocess.run([sys.executable, DRIVER, "--gpus", str(gpus), *args, *extra], capture_output=True, text=True,
                         env=_env_without_launcher(), cwd=ROOT, timeout=280)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    j = json.loads(lines[0])
    assert KEYS <= set(j) and j["n_gpus"] == gpus and j["config"]["ranks_seen"] == gpus and j["value"] > 0
    assert j["scaling"] == ("weak" if workload == "corpus" else "strong")
    assert j["config"]["clips"] == (4 * gpus if workload == "corpus" else 11)
    if pipeline == "default":
        assert j["parity_sample"].startswith("the reference pipeline's records and cut lists"), j["parity_sample"]
        assert j["pipeline"].startswith("the reference's default") and "default downscale" in j["config"]["workload"]
        assert ("-> 256x" in j["parity_sample"]) == (size != (36, 64))
    else:
        assert j["parity_sample"].startswith("records and cut lists of FULL-resolution frames identical"), j["parity_sample"]
        assert j["pipeline"].startswith("full resolution") and "-> 256x" not in j["parity_sample"]
    assert j["cpu_baseline"]["kind"] == "port" and j["roofline"]["bound"] == "hbm"
