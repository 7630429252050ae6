"""Host-side native code under AddressSanitizer + UndefinedBehaviorSanitizer (the reference has no
race/sanitizer story; GPU sanitizers are unavailable on the target pool, so the C++ that runs on the CPU — clustering,
Dirichlet partitioner, rank placement, the process launcher — is what gets sanitized)."""
import os
import shutil
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN = ["-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-static-libasan"]

pytestmark = pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not available")


def _build(out, sources):
    cmd = ["g++"] + SAN + ["-o", str(out)] + [os.path.join(REPO, s) for s in sources]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    if r.returncode != 0 and "asan" in r.stderr.lower() and "cannot find" in r.stderr.lower():
        pytest.skip("static ASan runtime not installed")
    assert r.returncode == 0, r.stderr[-3000:]


def test_host_library_selftest_under_asan_ubsan(tmp_path):
    exe = tmp_path / "host_selftest"
    _build(exe, ["tests/native/host_selftest.cpp", "blades_amd/ops/csrc/host.cpp"])
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "HOST_SELFTEST_OK" in r.stdout, r.stdout + r.stderr[-3000:]


def test_native_launcher_under_asan_ubsan(tmp_path):
    exe = tmp_path / "blades_launch_san"
    _build(exe, ["blades_amd/parallel/csrc/launch.cpp"])
    # success path: every rank sees its own RANK / WORLD_SIZE
    r = subprocess.run([str(exe), "--nproc", "3", "--master-port", "29761", "--", sys.executable, "-c",
                        "import os; print('R', os.environ['RANK'], os.environ['WORLD_SIZE'])"],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr[-3000:]
    assert sorted(l for l in r.stdout.splitlines() if l.startswith("R ")) == ["R 0 3", "R 1 3", "R 2 3"]
    # failure path: one rank dies -> the others are torn down and the exit code is that rank's
    r = subprocess.run([str(exe), "--nproc", "2", "--grace", "2", "--master-port", "29762", "--", sys.executable, "-c",
                        "import os, sys, time\nif os.environ['RANK'] == '1': sys.exit(7)\ntime.sleep(60)"],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 7, (r.returncode, r.stdout, r.stderr[-2000:])
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr
    # usage error
    r = subprocess.run([str(exe), "--nproc", "0", "--", "true"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 2
