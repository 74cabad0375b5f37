// One group of the library's kernel instantiations as a translation unit of its own (kernel_list.h says which and why):
//   hipcc -DTR_INST_GROUP=<g> -c kernel_group.hip -o kgroup_<g>.o
// kernels.hip with TR_DEVICE_TU defined is the kernel templates and device functions without the host side.
#ifndef TR_INST_GROUP
#error "compile with -DTR_INST_GROUP=<0 .. TR_INST_GROUPS - 1> (csrc/Makefile)"
#endif
#define TR_DEVICE_TU
#include "kernels.hip"
#include "kernel_list.h"
#if TR_INST_GROUP >= TR_INST_GROUPS
#error "no such kernel group"
#endif
