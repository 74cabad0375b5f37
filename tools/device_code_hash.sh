#!/bin/bash
# md5 of the gfx950 code objects inside a built libtrayhip.so (the .hip_fatbin section). Host-only changes of kernels.hip must
# leave it unchanged: that is how a rebuild without a GPU at hand is shown to run the device code the GPU suite validated.
#   tools/device_code_hash.sh [path/to/libtrayhip.so]
set -e
LIB=${1:-$(dirname "$0")/../tray_rust_amd/libtrayhip.so}
TMP=$(mktemp)
/opt/rocm/lib/llvm/bin/llvm-objcopy -O binary --only-section=.hip_fatbin "$LIB" "$TMP"
md5sum "$TMP" | cut -d' ' -f1
rm -f "$TMP"
