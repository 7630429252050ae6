"""In-tree native builds (no hipify, no torch JIT cache).

* ``libblades_host.so``  — g++,  ``csrc/host.cpp``  (CPU; clustering, partitioner)
* ``libblades_hip.so``   — hipcc ``--offload-arch=gfx950``, every ``csrc/*.hip``

Both expose a plain C ABI and are loaded with ctypes, so neither needs the
PyTorch headers: a full rebuild is seconds, and hipcc cross-compiles gfx950
code objects on a machine without a GPU.
"""
import os
import subprocess
import sys
from pathlib import Path

HERE = Path(__file__).resolve().parent
CSRC = HERE / "csrc"
HOST_LIB = HERE / "libblades_host.so"
HIP_LIB = HERE / "libblades_hip.so"
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"


def _stale(target: Path, sources) -> bool:
    if not target.exists():
        return True
    t = target.stat().st_mtime
    return any(Path(s).stat().st_mtime > t for s in sources)


def build_host(force: bool = False, verbose: bool = False) -> Path:
    srcs = [CSRC / "host.cpp"]
    if force or _stale(HOST_LIB, srcs):
        cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", str(HOST_LIB)] + [str(s) for s in srcs]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return HOST_LIB


def hip_sources():
    return sorted(CSRC.glob("*.hip"))


def build_hip(force: bool = False, verbose: bool = False, jobs: int = 4) -> Path:
    srcs = hip_sources()
    hdrs = list(CSRC.glob("*.h")) + list(CSRC.glob("*.hpp"))
    objdir = HERE / "_build"
    objdir.mkdir(exist_ok=True)
    procs = []
    objs = []
    for s in srcs:
        o = objdir / (s.stem + ".o")
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            cmd = [HIPCC, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-c", str(s), "-o", str(o),
                   "-I", str(CSRC), "-Wno-unused-result"]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((s, subprocess.Popen(cmd)))
            if len(procs) >= jobs:
                s0, p0 = procs.pop(0)
                if p0.wait() != 0:
                    raise RuntimeError(f"hipcc failed on {s0}")
    for s0, p0 in procs:
        if p0.wait() != 0:
            raise RuntimeError(f"hipcc failed on {s0}")
    if force or _stale(HIP_LIB, objs):
        cmd = [HIPCC, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", str(HIP_LIB)] + [str(o) for o in objs]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return HIP_LIB


def build_all(force: bool = False, verbose: bool = True):
    build_host(force, verbose)
    build_hip(force, verbose)


if __name__ == "__main__":
    build_all(force="--force" in sys.argv)
    print("built", HOST_LIB, HIP_LIB)
