"""The shipped library's device code is the one that was last checked on a GPU: host-only edits of kernels.hip and of the headers
must not change the gfx950 code objects (tools/device_code_hash.sh), and a deliberate change has to come with a GPU run and a
new line in tests/golden/device_code_hash.txt."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_device_code_is_the_gpu_checked_one(built):
    objcopy = "/opt/rocm/lib/llvm/bin/llvm-objcopy"
    if not os.path.exists(objcopy):
        pytest.skip("no llvm-objcopy in this image")
    want = [l.strip() for l in open(os.path.join(ROOT, "tests", "golden", "device_code_hash.txt")) if l.strip() and not l.startswith("#")][-1]
    got = subprocess.run([os.path.join(ROOT, "tools", "device_code_hash.sh")], capture_output=True, text=True, check=True).stdout.strip()
    assert got == want, ("the gfx950 code objects of libtrayhip.so changed: run tools/quick_gpu_check.py (and pytest -m gpu) on an MI355X, "
                         "then record the new hash in tests/golden/device_code_hash.txt")
