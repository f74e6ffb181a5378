"""The CPU oracle (``oracle/cv2_restate.c``: the checker every parity claim rests on) under AddressSanitizer and
UndefinedBehaviorSanitizer: an out-of-bounds read or a signed overflow in the checker could make it wrong in a way its own
self-consistency tests would not show.  Builds the C file with ``-fsanitize=address,undefined`` into /tmp, points ``oracle.lib`` at that
build and runs the oracle's tests, the published-value tests and the golden-run tests through it.  Build container; no GPU.

    python tools/sanitize/run_oracle_sanitized.py      (re-executes itself under LD_PRELOAD=libasan)
"""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
OUT = "/tmp/psd_sanitize"
LIB = os.path.join(OUT, "liboracle_san.so")


def main():
    if os.environ.get("PSD_SANITIZED") != "1":
        os.makedirs(OUT, exist_ok=True)
        subprocess.check_call(["gcc", "-O1", "-g", "-msse4.2", "-fPIC", "-std=c11", "-fno-fast-math", "-ffp-contract=off",
                               "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-shared", "-o", LIB,
                               os.path.join(ROOT, "oracle", "cv2_restate.c"), "-lm"])
        abi = os.path.join(OUT, "libpsd_oracle_abi_san.so")      # the CPU build of the C-ABI (oracle/abi_cpu.c) as well
        subprocess.check_call(["gcc", "-O1", "-g", "-msse4.2", "-fPIC", "-std=c11", "-fno-fast-math", "-ffp-contract=off",
                               "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-shared", "-o", abi,
                               os.path.join(ROOT, "oracle", "abi_cpu.c"), os.path.join(ROOT, "oracle", "cv2_restate.c"), "-lm"])
        asan = sorted(glob.glob("/usr/lib/x86_64-linux-gnu/libasan.so.*"))[0]
        env = dict(os.environ, PSD_SANITIZED="1", PSD_ORACLE_ABI_LIB=abi, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1",
                   UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
        raise SystemExit(subprocess.call([sys.executable, os.path.abspath(__file__)], env=env))
    sys.path[:0] = [os.path.join(ROOT, "oracle", "cv2_shim"), ROOT]
    from oracle import lib as orc

    orc._SO = LIB                      # (lib() loads _SO on first use; _ensure_built only looks at the in-tree file)
    loaded = orc.lib()
    assert loaded._name == LIB, loaded._name
    print("oracle library under the sanitizers:", loaded._name, flush=True)
    import pytest

    rc = pytest.main(["-q", "-x", "-p", "no:cacheprovider", os.path.join(ROOT, "tests", "test_oracle.py"),
                      os.path.join(ROOT, "tests", "test_published_values.py"), os.path.join(ROOT, "tests", "test_host_golden.py"),
                      os.path.join(ROOT, "tests", "test_scene_manager.py"), os.path.join(ROOT, "tests", "test_reference_binding.py")])
    if rc != 0:
        raise SystemExit(rc)
    if os.path.isdir("/root/reference/scenedetect"):      # random shapes, crops, downscales, thumbnails, kernels through the same build
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import fuzz_host_vs_reference as F

        sys.argv = ["fuzz_host_vs_reference.py", "--seconds", os.environ.get("PSD_SANITIZE_FUZZ_SECONDS", "120"), "--seed", "78", "--wide"]
        F.main()
    print("sanitizers: clean")


if __name__ == "__main__":
    main()
