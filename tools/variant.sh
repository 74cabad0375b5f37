#!/bin/bash
# Build a variant of the library next to the default one: tools/variant.sh <name> <extra hipcc flags...>  ->  tray_rust_amd/libtrayhip_<name>.so
# (A/B on the GPU: tools/ab.sh <tag> libtrayhip.so libtrayhip_<name>.so ...; delete the variant .so afterwards, it travels with every gpurun push.)
# The build runs in a SNAPSHOT of csrc/ and include/ under /tmp (the sources as they are when the script starts), so the tree can be edited
# while a variant compiles -- hipcc maps its inputs and dies with a bus error when one of them is rewritten under it.
# VARIANT_HIPFLAGS="..." replaces the Makefile's whole HIPFLAGS line (to REMOVE a default flag such as -fno-slp-vectorize).
# VARIANT_MAKEARGS="'GROUPFLAGS_12=-mllvm -x=y' 'GROUPFLAGS_3=...'" (each assignment in single quotes) passes make variables (flags of single kernel groups: csrc/Makefile).
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
name=$1; shift
SNAP=/tmp/variant_$name
rm -rf $SNAP; mkdir -p $SNAP/tray_rust_amd
cp -r $ROOT/include $SNAP/include
cp -r $ROOT/tray_rust_amd/csrc $SNAP/tray_rust_amd/csrc
find $SNAP -name "*.o" -delete
cd $SNAP/tray_rust_amd/csrc
if [ -n "$VARIANT_HIPFLAGS" ]; then
  eval make -s -j${VARIANT_JOBS:-8} OUT=../libtrayhip_$name.so 'HIPFLAGS="$VARIANT_HIPFLAGS $*"' $VARIANT_MAKEARGS 2>&1 | grep -E "error|Error" || true
else
  eval make -s -j${VARIANT_JOBS:-8} OUT=../libtrayhip_$name.so 'EXTRA_HIPFLAGS="$*"' $VARIANT_MAKEARGS 2>&1 | grep -E "error|Error" || true
fi
cp ../libtrayhip_$name.so $ROOT/tray_rust_amd/libtrayhip_$name.so
ls -la $ROOT/tray_rust_amd/libtrayhip_$name.so
rm -rf $SNAP
