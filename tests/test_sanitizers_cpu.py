"""The sanitizer builds: the reference runs with Vulkan validation layers whenever NDEBUG is not defined
(engine/include/engine/core/GPUContext.h:84-90, engine/src/engine/core/GPUContext.cpp:102-133) -- a debug build that checks the host's
use of the API.  Here: the library's host code, the C++ host mirror with its logic test, the host-only C-ABI entry points and the CPU
oracle under AddressSanitizer + UndefinedBehaviorSanitizer, and the loopback hub's thread rendezvous (host-memory mode, 1 / 2 / 3 / 8
ranks as threads, every collective of the multi-GPU step and a rank that breaks the rules) under ThreadSanitizer.  No GPU needed; the
GPU side of it is tools/asan_fuzz.sh (profiles/r06_asan_fuzz.txt).  First build about 50 s per flavour, then incremental."""
import shutil
import subprocess

import pytest

from vkradixsort_amd import build as b

pytestmark = pytest.mark.skipif(shutil.which("hipcc") is None and not b.Path("/opt/rocm/bin/hipcc").exists(), reason="needs the ROCm toolchain")

REPORTS = ("ERROR: AddressSanitizer", "runtime error:", "WARNING: ThreadSanitizer", "FAILED")


def _run(kind, exe, *args):
    proc = subprocess.run([str(exe), *args], capture_output=True, text=True, env=b.sanitizer_env(kind), cwd=str(b.REPO_ROOT), timeout=600)
    text = proc.stdout + proc.stderr
    assert proc.returncode == 0 and not any(t in text for t in REPORTS), text[-6000:]
    return text


def test_address_and_undefined_behaviour_sanitizers_are_clean():
    exes = b.build_sanitized("asan")
    assert set(exes) == {"host_logic_test", "capi_host_sanity"}
    assert "capi_host_sanity: ok" in _run("asan", exes["capi_host_sanity"])
    _run("asan", exes["host_logic_test"])
    # the CPU checker (test infrastructure: built by its own Makefile, never by the product package)
    made = subprocess.run(["make", "-C", str(b.REPO_ROOT / "oracle"), "selftest-asan"], capture_output=True, text=True)
    assert made.returncode == 0, made.stdout + made.stderr
    assert "oracle_selftest: ok" in _run("asan", b.REPO_ROOT / "oracle" / "_san" / "oracle_selftest")
    # the instrumented library is the product's sources, every one of them, and really instrumented
    lib = b.san_dir("asan") / "libvkradixsort_amd.so"
    syms = subprocess.run(["nm", "-D", "--undefined-only", str(lib)], capture_output=True, text=True).stdout
    assert "__asan_init" in syms and "__ubsan_handle" in syms


def test_thread_sanitizer_on_the_loopback_hub_is_clean():
    exes = b.build_sanitized("tsan")
    assert "capi_host_sanity: ok" in _run("tsan", exes["capi_host_sanity"], "hub")
    syms = subprocess.run(["nm", "-D", "--undefined-only", str(b.san_dir("tsan") / "libvkradixsort_amd.so")], capture_output=True, text=True).stdout
    assert "__tsan_init" in syms or "__tsan_func_entry" in syms
