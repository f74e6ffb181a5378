"""The host epilogues (``psd_epilogue.cpp``: what ``bench.py``'s timed region and ``detect_corpus`` decide with) under AddressSanitizer and
UndefinedBehaviorSanitizer.  ``psd_epilogue.cpp`` is plain C++, so g++ builds it on its own with ``-fsanitize=address,undefined``; the
entry points of the loaded product library are then swapped for the sanitized ones in THIS process and the epilogue tests plus a
time-boxed differential fuzz against the reference run through them.  Build container; no GPU.

    python tools/sanitize/run_epilogues_sanitized.py [--seconds 120]      (re-executes itself under LD_PRELOAD=libasan)
"""
import argparse
import ctypes
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
OUT = "/tmp/psd_sanitize"
LIB = os.path.join(OUT, "libpsd_epilogue_san.so")


def build():
    os.makedirs(OUT, exist_ok=True)
    stub = os.path.join(OUT, "set_error_stub.cpp")
    with open(stub, "w") as f:
        f.write('#include <cstdio>\nextern "C" void psd_set_error(const char* fmt, ...) { (void)fmt; }\n')
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fPIC", "-shared", "-fsanitize=address,undefined",
                           "-fno-sanitize-recover=undefined", "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"),
                           "-I" + os.path.join(ROOT, "pyscenedetect_amd", "csrc"),
                           os.path.join(ROOT, "pyscenedetect_amd", "csrc", "psd_epilogue.cpp"), stub, "-o", LIB, "-lpthread"])


def swap_in():
    sys.path[:0] = [os.path.join(ROOT, "oracle", "cv2_shim"), ROOT]
    from pyscenedetect_amd import _native

    lib, san = _native.load(), ctypes.CDLL(LIB)
    swapped = []
    for name, (restype, argtypes) in _native.SYMBOLS.items():
        if name.startswith("psd_epilogue_"):
            fn = getattr(san, name)
            fn.restype, fn.argtypes = restype, argtypes
            setattr(lib, name, fn)
            swapped.append(name)
    return swapped


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120.0)
    args = ap.parse_args()
    if os.environ.get("PSD_SANITIZED") != "1":
        build()
        asan = sorted(glob.glob("/usr/lib/x86_64-linux-gnu/libasan.so.*"))[0]
        env = dict(os.environ, PSD_SANITIZED="1", LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1",
                   UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
        raise SystemExit(subprocess.call([sys.executable, os.path.abspath(__file__), "--seconds", str(args.seconds)], env=env))
    swapped = swap_in()
    print("sanitized entry points:", ", ".join(swapped), flush=True)
    import pytest

    rc = pytest.main(["-q", "-x", "-p", "no:cacheprovider", os.path.join(ROOT, "tests", "test_epilogue_fuzz.py"),
                      os.path.join(ROOT, "tests", "test_host_golden.py"), "-k", "not subprocess"])
    if rc != 0:
        raise SystemExit(rc)
    if os.path.isdir("/root/reference/scenedetect"):
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import fuzz_epilogue_vs_reference as E

        for extra in ([], ["--wide", "--tiny"]):
            sys.argv = ["fuzz_epilogue_vs_reference.py", "--seconds", str(args.seconds / 2), "--seed", "77"] + extra
            E.main()
    print("sanitizers: clean")


if __name__ == "__main__":
    main()
