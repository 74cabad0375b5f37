"""The shipped library's device code is the one that was last checked on a GPU: host-only edits of kernels.hip and of the headers
must not change the gfx950 code objects (tools/device_code_hash.sh), and a deliberate change has to come with a GPU run and a
new line in tests/golden/device_code_hash.txt (tools/record_device_hash.sh). The hash is tied to the compiler that produced it:
with another hipcc the code objects differ although no source changed, and the test only says so."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_device_code_is_the_gpu_checked_one(built):
    objcopy, hipcc = "/opt/rocm/lib/llvm/bin/llvm-objcopy", "/opt/rocm/bin/hipcc"
    if not os.path.exists(objcopy) or not os.path.exists(hipcc):
        pytest.skip("no llvm-objcopy / hipcc in this image")
    last = [l.strip() for l in open(os.path.join(ROOT, "tests", "golden", "device_code_hash.txt")) if l.strip() and not l.startswith("#")][-1]
    want, compiler = [x.strip() for x in last.split("|")][:2]
    have = subprocess.run([hipcc, "--version"], capture_output=True, text=True, check=True).stdout.splitlines()[0].strip()
    if have != compiler:
        pytest.skip(f"the recorded hash belongs to '{compiler}', this image has '{have}': re-record it from a GPU run")
    got = subprocess.run([os.path.join(ROOT, "tools", "device_code_hash.sh")], capture_output=True, text=True, check=True).stdout.strip()
    assert got == want, ("the gfx950 code objects of libtrayhip.so changed: run pytest -m gpu on an MI355X, "
                         "then tools/record_device_hash.sh '<what ran>'")
