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

HIP_SOURCES = [CSRC / "vrs_contract.hip", CSRC / "vrs_one_call.hip", CSRC / "vrs_msd_hybrid.hip", CSRC / "vrs_msd_pool.hip", CSRC / "vrs_capi.hip", CSRC / "vrs_dist.hip"]
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
