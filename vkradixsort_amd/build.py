"""In-tree builds: the gfx950 C-ABI library and the C++ host mirror + example binaries.

Everything is compiled with explicit hipcc / g++ command lines (no JIT cache: the built files must
travel with the repo snapshot to the GPU box).
"""
from __future__ import annotations

import os
import shutil
import subprocess
from pathlib import Path

PKG_DIR = Path(__file__).resolve().parent
REPO_ROOT = PKG_DIR.parent
CSRC = PKG_DIR / "csrc"
HOST = PKG_DIR / "host"
INCLUDE = REPO_ROOT / "include"
LIB_PATH = PKG_DIR / "libvkradixsort_amd.so"

HIP_SOURCES = [CSRC / name for name in ("vrs_contract.hip", "vrs_one_call.hip", "vrs_msd_hybrid.hip", "vrs_msd_pool.hip", "vrs_msd_pool_local.hip", "vrs_pool_shape.hip",
                                        "vrs_capi.hip", "vrs_capi_contract.hip", "vrs_capi_sort.hip", "vrs_capi_pool.hip", "vrs_capi_msd.hip", "vrs_dist.hip")]
HIP_HEADERS = sorted(CSRC.glob("*.h")) + sorted(CSRC.glob("*.hpp")) + [INCLUDE / "vkradixsort_amd.h"]  # every object is rebuilt when any header is newer
ARCH = "gfx950"


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found (ROCm toolchain is required to build vkradixsort_amd)")


def _stale(target: Path, sources) -> bool:
    if not target.exists():
        return True
    t = target.stat().st_mtime
    return any(Path(s).stat().st_mtime > t for s in sources)


def _run(cmd, cwd=None) -> None:
    proc = subprocess.run([str(c) for c in cmd], cwd=cwd, capture_output=True, text=True)
    if proc.returncode != 0:
        raise RuntimeError("build failed: %s\n%s\n%s" % (" ".join(map(str, cmd)), proc.stdout, proc.stderr))


def build_library(force: bool = False) -> Path:
    """hipcc --offload-arch=gfx950 -> vkradixsort_amd/libvkradixsort_amd.so (kernels + C ABI): one object per source
    (compiled side by side, only the stale ones), then one link."""
    from concurrent.futures import ThreadPoolExecutor
    obj_dir = PKG_DIR / "_build"
    obj_dir.mkdir(exist_ok=True)
    objs = [obj_dir / (src.stem + ".o") for src in HIP_SOURCES]
    stale = [(src, obj) for src, obj in zip(HIP_SOURCES, objs) if force or _stale(obj, [src] + HIP_HEADERS)]
    with ThreadPoolExecutor(max_workers=max(len(stale), 1)) as pool:
        list(pool.map(lambda so: _run([_hipcc(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-c",
                                       f"-I{INCLUDE}", f"-I{CSRC}", so[0], "-o", so[1]]), stale))
    if stale or force or _stale(LIB_PATH, objs):
        _run([_hipcc(), f"--offload-arch={ARCH}", "-fPIC", "-shared", *objs, "-ldl", "-o", LIB_PATH])
    return LIB_PATH


def host_sources():
    return sorted((HOST / "src").glob("*.cpp"))


def build_host(force: bool = False):
    """g++ -> the C++ host mirror (engine::GPUContext/Buffer/.../MultiRadixSort) as
    libvkradixsort_host.so plus the two example executables.  Links only the C ABI."""
    build_library(force)
    out_lib = PKG_DIR / "libvkradixsort_host.so"
    hdrs = list((HOST / "include").rglob("*.h")) + [INCLUDE / "vkradixsort_amd.h"]
    srcs = host_sources()
    cxx = shutil.which("g++") or "g++"
    common = ["-O2", "-std=c++20", "-fPIC", f"-I{HOST / 'include'}", f"-I{INCLUDE}"]
    link = [f"-L{PKG_DIR}", "-lvkradixsort_amd", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath,/opt/rocm/lib"]
    if force or _stale(out_lib, srcs + hdrs + [LIB_PATH]):
        _run([cxx, *common, "-shared", *srcs, "-o", out_lib, *link])
    exes = []
    for name in ("MultiRadixSortExample", "SingleRadixSortExample"):
        src = HOST / "bin" / f"{name}.cpp"
        exe = PKG_DIR / name.lower()
        if force or _stale(exe, [src, out_lib] + hdrs):
            _run([cxx, *common, src, "-o", exe, f"-L{PKG_DIR}", "-lvkradixsort_host", *link])
        exes.append(exe)
    build_dist_example(force)
    return out_lib, exes


def build_dist_example(force: bool = False) -> Path:
    """g++ -> distsortexample: the multi-GPU step from a C++ host, one thread per rank over the in-process transport (C ABI only)."""
    build_library(force)
    src = HOST / "bin" / "DistSortExample.cpp"
    exe = PKG_DIR / "distsortexample"
    if force or _stale(exe, [src, LIB_PATH, INCLUDE / "vkradixsort_amd.h"]):
        cxx = shutil.which("g++") or "g++"
        _run([cxx, "-O2", "-std=c++20", "-pthread", f"-I{INCLUDE}", src, "-o", exe, f"-L{PKG_DIR}", "-lvkradixsort_amd",
              "-Wl,-rpath,$ORIGIN", "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def build_host_logic_test(force: bool = False) -> Path:
    """g++ -> the CPU-only unit test of the C++ host mirror's logic (vkradixsort_amd/host/test)."""
    out_lib, _ = build_host(force)
    src = HOST / "test" / "host_logic_test.cpp"
    exe = PKG_DIR / "host_logic_test"
    hdrs = list((HOST / "include").rglob("*.h")) + [INCLUDE / "vkradixsort_amd.h"]
    if force or _stale(exe, [src, out_lib] + hdrs):
        cxx = shutil.which("g++") or "g++"
        _run([cxx, "-O2", "-std=c++20", f"-I{HOST / 'include'}", f"-I{INCLUDE}", src, "-o", exe, f"-L{PKG_DIR}",
              "-lvkradixsort_host", "-lvkradixsort_amd", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath,/opt/rocm/lib"])
    return exe


# ---------------------------------------------------------------------------------------------
# Sanitizer builds (the reference runs with Vulkan validation layers whenever NDEBUG is not defined,
# engine/include/engine/core/GPUContext.h:84-90 -- here: the host code of the library, the C++ host mirror and the host-only entry points
# under AddressSanitizer + UndefinedBehaviorSanitizer, the loopback hub's thread rendezvous under ThreadSanitizer).
# Everything is compiled by ONE compiler (ROCm's clang, which hipcc is) so that one sanitizer runtime serves the process; device code is
# left as it is (-fno-gpu-sanitize).  Output: vkradixsort_amd/_build/san_<kind>/.
SAN_FLAGS = {
    "asan": ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer", "-g", "-O1"],
    "tsan": ["-fsanitize=thread", "-fno-omit-frame-pointer", "-g", "-O1"],
}


def _clangxx() -> str:
    for cand in (os.environ.get("VRS_CLANGXX"), "/opt/rocm/lib/llvm/bin/clang++", shutil.which("amdclang++"), shutil.which("clang++")):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("clang++ not found (the sanitizer builds use ROCm's clang for host and device code alike)")


def san_dir(kind: str) -> Path:
    d = PKG_DIR / "_build" / f"san_{kind}"
    d.mkdir(parents=True, exist_ok=True)
    return d


def build_sanitized(kind: str = "asan", force: bool = False) -> dict:
    """The library, the C++ host mirror with host_logic_test and capi_host_sanity, instrumented.  (The CPU checker under tests/ has a
    sanitizer target of its own in its Makefile: the product package never touches it.)
    Returns {name: path} of the executables to run (their rpath finds the instrumented libraries and the sanitizer runtime)."""
    from concurrent.futures import ThreadPoolExecutor
    flags = SAN_FLAGS[kind]
    out = san_dir(kind)
    lib = out / "libvkradixsort_amd.so"
    objs = [out / (src.stem + ".o") for src in HIP_SOURCES]
    stale = [(src, obj) for src, obj in zip(HIP_SOURCES, objs) if force or _stale(obj, [src] + HIP_HEADERS)]
    with ThreadPoolExecutor(max_workers=max(len(stale), 1)) as pool:
        list(pool.map(lambda so: _run([_hipcc(), f"--offload-arch={ARCH}", "-std=c++17", "-fPIC", "-fno-gpu-sanitize", "-shared-libsan", *flags, "-c",
                                       f"-I{INCLUDE}", f"-I{CSRC}", so[0], "-o", so[1]]), stale))
    if stale or force or _stale(lib, objs):
        _run([_hipcc(), f"--offload-arch={ARCH}", "-fPIC", "-shared", "-fno-gpu-sanitize", "-shared-libsan", *flags, *objs, "-ldl", "-o", lib])
    cxx = _clangxx()
    rt_dir = subprocess.run([cxx, "-print-runtime-dir"], capture_output=True, text=True).stdout.strip()
    rt_alt = str(Path(rt_dir).parent / "linux")  # (where ROCm's clang keeps libclang_rt.*-x86_64.so)
    common = ["-std=c++20", "-fPIC", "-pthread", "-shared-libsan", *flags, f"-I{HOST / 'include'}", f"-I{INCLUDE}"]
    link = [f"-L{out}", "-lvkradixsort_amd", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath,/opt/rocm/lib", f"-Wl,-rpath,{rt_dir}", f"-Wl,-rpath,{rt_alt}"]
    hdrs = list((HOST / "include").rglob("*.h")) + [INCLUDE / "vkradixsort_amd.h"]
    exes = {}
    # (the host mirror's sources go INTO the test executable: as a second instrumented shared library its globals were registered twice by
    #  the runtime -- reported as a violation of the one-definition rule although nothing is defined twice)
    # (-asan-globals=0 for the host mirror only: its class templates are instantiated for uint32 and uint64 keys, every string literal inside them is
    #  emitted per instantiation, the linker folds the identical copies, and the runtime then finds two instrumented globals at one -- misaligned --
    #  address and aborts with "odr-violation" before main; heap, stack and use-after-free checks are unaffected)
    no_globals = ["-mllvm", "-asan-globals=0"] if kind == "asan" else []
    for name, srcs, extra in (("host_logic_test", [HOST / "test" / "host_logic_test.cpp", *host_sources()], no_globals),
                              ("capi_host_sanity", [HOST / "test" / "capi_host_sanity.cpp"], [])):
        exe = out / name
        if force or _stale(exe, [*srcs, lib] + hdrs):
            _run([cxx, *common, *srcs, "-o", exe, *extra, *link])
        exes[name] = exe
    return exes


def sanitizer_env(kind: str) -> dict:
    """The environment the instrumented executables (and a python that preloads the runtime) run in."""
    env = dict(os.environ)
    if kind == "asan":
        # (leaks: the HIP runtime keeps what it allocates until exit; the link order check: python itself is not instrumented)
        env["ASAN_OPTIONS"] = "detect_leaks=0:verify_asan_link_order=0:abort_on_error=0:halt_on_error=1"
        env["UBSAN_OPTIONS"] = "print_stacktrace=1:halt_on_error=1"
    else:
        env["TSAN_OPTIONS"] = "halt_on_error=1:second_deadlock_stack=1"
    return env


def sanitizer_runtime(kind: str) -> Path:
    """The shared sanitizer runtime to LD_PRELOAD into an uninstrumented host (python + ctypes loading the instrumented library)."""
    cxx = _clangxx()
    rt_dir = Path(subprocess.run([cxx, "-print-runtime-dir"], capture_output=True, text=True).stdout.strip())
    name = f"libclang_rt.{kind}-x86_64.so" if kind != "asan" else "libclang_rt.asan-x86_64.so"
    for d in (rt_dir, rt_dir.parent / "linux"):
        if (d / name).exists():
            return d / name
    raise RuntimeError(f"{name} not found under {rt_dir}")


def run_sanitized(kind: str = "asan", force: bool = False) -> int:
    """Builds and runs every instrumented executable; returns the number that failed (their output is printed)."""
    failed = 0
    exes = build_sanitized(kind, force)
    runs = [(name, [str(exe)]) for name, exe in exes.items()]
    if kind == "tsan":
        runs = [("capi_host_sanity hub", [str(exes["capi_host_sanity"]), "hub"])]
    for name, cmd in runs:
        proc = subprocess.run(cmd, capture_output=True, text=True, env=sanitizer_env(kind), cwd=str(REPO_ROOT))
        bad = proc.returncode != 0 or any(t in proc.stdout + proc.stderr for t in ("ERROR: AddressSanitizer", "runtime error:", "WARNING: ThreadSanitizer"))
        print(f"[{kind}] {name}: {'FAILED' if bad else 'clean'} (exit {proc.returncode})")
        if bad:
            failed += 1
            print(proc.stdout[-4000:], proc.stderr[-8000:])
    return failed


if __name__ == "__main__":
    import argparse
    import sys
    ap = argparse.ArgumentParser(description="in-tree builds of vkradixsort_amd")
    ap.add_argument("--asan", action="store_true", help="build and run the AddressSanitizer + UndefinedBehaviorSanitizer targets")
    ap.add_argument("--tsan", action="store_true", help="build and run the ThreadSanitizer target (the loopback hub's rendezvous)")
    ap.add_argument("--force", action="store_true")
    args = ap.parse_args()
    if not (args.asan or args.tsan):
        build_library(args.force)
        build_host(args.force)
        build_host_logic_test(args.force)
        sys.exit(0)
    bad = (run_sanitized("asan", args.force) if args.asan else 0) + (run_sanitized("tsan", args.force) if args.tsan else 0)
    sys.exit(1 if bad else 0)
