#!/bin/bash
# Build a variant of the library next to the default one: tools/variant.sh <name> <extra hipcc flags...>  ->  tray_rust_amd/libtrayhip_<name>.so
# (A/B on the GPU: tools/ab_round2.sh <name> ...; delete the variant .so / .o afterwards, they travel with every gpurun push)
cd "$(dirname "$0")/../tray_rust_amd/csrc" || exit 1
name=$1; shift
make -s OUT=../libtrayhip_$name.so KOBJ=hip/kernels_$name.o EXTRA_HIPFLAGS="$*" 2>&1 | grep -E "error|Error" ; ls -la ../libtrayhip_$name.so
