#!/bin/bash
# The DEVICE source under the sanitizers: tests/emu (the kernels compiled for the host) rebuilt with UBSan, then with ASan, and the emulation
# tests run against those builds. UBSan runs everything (fibers included); ASan only the one-lane launches (swapcontext confuses its stack
# tracking). The regular libtrayemu.so is put back afterwards.     bash tools/sanitize_emulation.sh
set -e
cd "$(dirname "$0")/.."
E=tests/emu; KEEP=$(mktemp); cp $E/libtrayemu.so $KEEP
trap 'cp $KEEP $E/libtrayemu.so; rm -f $KEEP' EXIT
FLAGS="-O1 -g -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-attributes -shared"
g++ $FLAGS -fsanitize=undefined -fno-sanitize-recover=undefined -o $E/libtrayemu.so $E/emu_kernels.cpp
UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 python -m pytest tests/test_device_emulation.py tests/test_samplers.py tests/test_animated_mesh.py -x -q | tail -2
g++ $FLAGS -fsanitize=address -o $E/libtrayemu.so $E/emu_kernels.cpp
LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 python -m pytest tests/test_device_emulation.py -x -q \
  -k "debug_intersect or per_sample_radiance or bsdf_eval or wavefront_traversal_kernel or axis_parallel or spline_stacks or random_scene_sweep" | tail -2
