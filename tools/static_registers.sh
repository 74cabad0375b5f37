#!/bin/bash
# Static register / scratch / LDS report of the kernels inside the built libtrayhip.so (no GPU needed), as text and -- for the DEFAULT
# library only -- as profiles/static_registers_latest.json (read by bench.py for the compute view of the roofline, which uses it only when
# the device-code hash recorded in it equals the running library's).   tools/static_registers.sh [lib]
# A variant build (an explicit lib argument, TRAYHIP_LIB or EXTRA_HIPFLAGS in the environment) is reported as text and never becomes "latest".
set -e
cd "$(dirname "$0")/.."
LIB=${1:-tray_rust_amd/libtrayhip.so}
LATEST=profiles/static_registers_latest.json
if [ -n "$1" ] || [ -n "$TRAYHIP_LIB" ] || [ -n "$EXTRA_HIPFLAGS" ]; then LATEST=/dev/null; echo "(variant build: profiles/static_registers_latest.json is left alone)" >&2; fi
T=$(mktemp -d)
for co in $(tools/code_objects.sh "$LIB" $T); do /opt/rocm/lib/llvm/bin/llvm-readelf --notes $co; done | awk -v latest="$LATEST" -v hash="$(tools/device_code_hash.sh $LIB)" '
  /\.name:/ {name=$2}
  /\.private_segment_fixed_size:/ {scr=$2}
  /\.sgpr_spill_count:/ {ss=$2}
  /\.group_segment_fixed_size:/ {lds=$2}
  /\.sgpr_count:/ {sg=$2}
  /\.vgpr_count:/ {v=$2}
  /\.vgpr_spill_count:/ {sp=$2; if (name ~ /k_path_tilesILi0ELi0ELi0ELb0E/) { printf "{\"device_code_hash\": \"%s\", \"kernel\": \"k_path_tiles<0,0>\", \"vgprs\": %d, \"spilled_vgprs\": %d, \"scratch_bytes_per_lane\": %d, \"sgprs\": %d, \"sgprs_spilled_to_vgpr_lanes\": %d, \"static_lds_bytes\": %d, \"waves_per_simd_by_vgprs\": %d}\n", hash, v, sp, scr, sg, ss, lds, (v <= 64 ? 8 : (v <= 72 ? 7 : (v <= 80 ? 6 : (v <= 96 ? 5 : (v <= 128 ? 4 : (v <= 168 ? 3 : (v <= 256 ? 2 : 1))))))) > latest }
    if (name ~ /k_path_tiles|k_wf_/) printf "%-60s vgprs %3d  spilled %3d  scratch %4d B  sgprs %3d (%3d spilled to lanes)  lds %d\n", substr(name,1,60), v, sp, scr, sg, ss, lds}'
[ "$LATEST" = /dev/null ] || cat $LATEST
rm -rf $T
