"""setup.py: metadata resolves, the package list covers the framework and the reference-name shims, and the native
build is a setup command (reference: setup.py at the repository root, find_packages + requirements only)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _setup(*args):
    return subprocess.run([sys.executable, "setup.py", *args], cwd=ROOT, capture_output=True, text=True, timeout=120)


def test_setup_metadata_and_commands():
    r = _setup("--name", "--version")
    assert r.returncode == 0, r.stderr
    assert r.stdout.split() == ["blades_amd", "0.3.0"]
    r = _setup("--help-commands")
    assert r.returncode == 0 and "build_native" in r.stdout


def test_package_list_covers_framework_and_shims():
    from setuptools import find_packages

    pk = set(find_packages(ROOT, include=["blades_amd", "blades_amd.*", "blades", "blades.*", "fllib", "fllib.*"]))
    for name in ("blades_amd", "blades_amd.ops", "blades_amd.engine", "blades_amd.aggregators", "blades_amd.adversaries",
                 "blades_amd.parallel", "blades_amd.models", "blades_amd.datasets", "blades_amd.tune", "blades", "fllib",
                 "blades.adversaries", "fllib.aggregators"):
        assert name in pk, name
