#!/bin/bash
# The gfx950 code objects inside a built library or object file (one per translation unit: csrc/hip/kernel_list.h), extracted into <dir>
# as co_0, co_1, ...; prints their paths.   tools/code_objects.sh <lib-or-object> <dir>
set -e
IN=$1; DIR=$2
mkdir -p "$DIR"
cp "$IN" "$DIR/in.bin"
( cd "$DIR" && /opt/rocm/lib/llvm/bin/llvm-objdump --offloading in.bin > /dev/null )
n=0
for f in $(ls "$DIR"/in.bin.*.hipv4-amdgcn-amd-amdhsa--gfx950 2>/dev/null | sort -t. -k3 -n); do mv "$f" "$DIR/co_$n"; echo "$DIR/co_$n"; n=$((n + 1)); done
rm -f "$DIR"/in.bin "$DIR"/in.bin.*.host-*
[ $n -gt 0 ] || { echo "no gfx950 code object in $IN" >&2; exit 1; }
