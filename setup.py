"""Packaging for blades_amd.

``pip install -e .`` (or ``python setup.py build_native``) compiles the native libraries in-tree —
``libblades_hip.so`` (hipcc, gfx950), ``libblades_host.so`` (g++) and the ``blades_launch`` process launcher — with
the same builder the tests and ``__graft_entry__.build()`` use (``blades_amd/ops/build.py``: plain C ABI, ctypes, no
PyTorch headers), and ships them as package data.  The ``blades`` / ``fllib`` packages are import shims for code
written against the reference (``from blades.algorithms.fedavg import FedavgConfig``).
"""
import os
import sys

from setuptools import Command, find_packages, setup
from setuptools.command.build_py import build_py
from setuptools.command.develop import develop

ROOT = os.path.dirname(os.path.abspath(__file__))


def _build_native():
    sys.path.insert(0, ROOT)
    from blades_amd.ops import build as b
    from blades_amd.parallel import launcher

    b.build_all(force=False, verbose=True)
    launcher.build_native(verbose=True)


class build_native(Command):
    description = "compile libblades_hip.so (gfx950), libblades_host.so and blades_launch in-tree"
    user_options = []

    def initialize_options(self):
        pass

    def finalize_options(self):
        pass

    def run(self):
        _build_native()


class build_py_native(build_py):
    def run(self):
        _build_native()
        super().run()


class develop_native(develop):
    def run(self):
        _build_native()
        super().run()


setup(
    name="blades_amd",
    version="0.3.0",
    description="Byzantine-robust federated-learning simulator, native to AMD Instinct MI355X (gfx950)",
    python_requires=">=3.9",
    license="Apache License 2.0",
    zip_safe=False,
    packages=find_packages(include=["blades_amd", "blades_amd.*", "blades", "blades.*", "fllib", "fllib.*"]),
    package_data={
        "blades_amd.ops": ["*.so", "csrc/*"],
        "blades_amd.parallel": ["blades_launch", "csrc/*", "*.cpp"],
        "blades_amd.tuned_examples": ["*.yaml"],
    },
    install_requires=["torch>=2.0", "numpy", "pyyaml"],
    extras_require={"cli": ["typer"], "plots": ["matplotlib"], "tests": ["pytest"]},
    entry_points={"console_scripts": ["blades-amd-train=blades_amd.train:main"]},
    cmdclass={"build_native": build_native, "build_py": build_py_native, "develop": develop_native},
    keywords="federated learning, byzantine robustness, ROCm, HIP, MI355X",
)
