"""Build-time guard: the FAST conv_wgrad variants keep in-flight (compiler-untracked) loads in registers, so
they must not spill VGPRs.  Compiles conv_wgrad.hip to assembly and checks the kernel metadata.

usage: python tools/check_wgrad_spills.py      (exit status 1 on a spilling FAST variant)
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "blades_amd", "ops", "csrc", "conv_wgrad.hip")


def main():
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "w.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S",
                               "--cuda-device-only", "-I", os.path.dirname(SRC), SRC, "-o", out],
                              stderr=subprocess.DEVNULL)
        text = open(out).read()
    bad = 0
    for m in re.finditer(r"\.name:\s+(\S*conv_wgrad8?m?_kernel\S*)\n(?:.*\n)*?\s+\.vgpr_spill_count:\s+(\d+)", text):
        name, spills = m.group(1), int(m.group(2))
        # FAST variants of the four-wave kernel (template argument 5 = true) and every eight-wave variant keep
        # untracked loads in flight in registers
        fast = "conv_wgrad8_kernel" in name or "conv_wgrad8m_kernel" in name or re.search(r"ELb1ELi\dELb[01]EEEvNS_11WgradParamsE$", name) is not None
        print(f"{'FAST' if fast else 'table'} {name}: {spills} VGPR spills")
        if fast and spills:
            bad += 1
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
