#!/bin/bash
# Appends the hash of the current libtrayhip.so's device code + the compiler to tests/golden/device_code_hash.txt.
# Run it only after this very build passed `pytest -m gpu` on an MI355X:   tools/record_device_hash.sh "pytest -m gpu 44 passed (gpurun_out/...)"
set -e
cd "$(dirname "$0")/.."
echo "$(tools/device_code_hash.sh) | $(/opt/rocm/bin/hipcc --version | head -1) | ${1:-GPU run}" >> tests/golden/device_code_hash.txt
tail -1 tests/golden/device_code_hash.txt
