// Every instantiation of the kernel templates the library launches, in GROUPS that compile as translation units of their own
// (kernel_group.hip, one object per group: hipcc spends ~5 s per instantiation of the tile kernel and compiles a translation unit on one
// core -- as ONE unit the 83 kernels took six minutes, as eight they take one on the build container's eight cores (thirteen groups now), and a change to one
// header recompiles all of them side by side).
//   * kernel_group.hip defines TR_INST_GROUP = g and gets the explicit instantiation DEFINITIONS of group g: device code + host stub;
//   * kernels.hip (the host side: scene upload, launches) gets explicit instantiation DECLARATIONS of all groups (`extern template`), so its
//     launch sites name the kernels as before and no device code is generated for them there; the host stubs resolve inside the library.
// `template __global__ decltype(k<..>) k<..>;` names the specialisation without repeating its signature.
// The non-template kernels (k_xf_table_build / _check, k_sampler_decide, k_debug_bsdf) are compiled with kernels.hip itself.
// A kernel launched without an entry here fails to link (undefined host stub), it cannot silently fall out of the library.
#pragma once

#ifdef TR_INST_EXTERN
#define TR_K(...) extern template __global__ decltype(__VA_ARGS__) __VA_ARGS__;
#define TR_GROUP(g) 1
#else
#define TR_K(...) template __global__ decltype(__VA_ARGS__) __VA_ARGS__;
#define TR_GROUP(g) (TR_INST_GROUP == (g))
#endif

#define TR_K_TILES_LF(A, F) TR_K(k_path_tiles<A, F, TRAY_INTEGRATOR_PATH, false>) TR_K(k_path_tiles<A, F, TRAY_INTEGRATOR_PATH, true>)

#if TR_GROUP(0)   // static scenes, the bench's feature sets: cornell_box (none; this group alone is compiled with another scheduling strategy: csrc/Makefile) ...
TR_K_TILES_LF(0, FEAT_NONE)
#endif
#if TR_GROUP(12)   // ... smallpt (specular), the dragon (MERL)
TR_K_TILES_LF(0, FEAT_SPEC)
#endif
#if TR_GROUP(1)
TR_K_TILES_LF(0, FEAT_MERL)
TR_K_TILES_LF(0, FEAT_MERL | FEAT_SPEC)
#endif
#if TR_GROUP(2)   // static scenes, every lobe / textures / Whitted
TR_K_TILES_LF(0, FEAT_ALL)
TR_K_TILES_LF(0, FEAT_ALL | FEAT_TEX)
TR_K(k_path_tiles<0, FEAT_ALL | FEAT_TEX, TRAY_INTEGRATOR_WHITTED>)
#endif
#if TR_GROUP(3)   // moving scenes
TR_K_TILES_LF(1, FEAT_NONE)
TR_K_TILES_LF(1, FEAT_MERL)
TR_K_TILES_LF(1, FEAT_SPEC)
#endif
#if TR_GROUP(4)
TR_K_TILES_LF(1, FEAT_MERL | FEAT_SPEC)
TR_K_TILES_LF(1, FEAT_ALL)
#endif
#if TR_GROUP(5)
TR_K_TILES_LF(1, FEAT_ALL | FEAT_TEX)
TR_K(k_path_tiles<1, FEAT_ALL | FEAT_TEX, TRAY_INTEGRATOR_WHITTED>)
TR_K(k_debug_intersect<0>) TR_K(k_debug_intersect<2>) TR_K(k_debug_intersect<3>)
#endif
#if TR_GROUP(6)   // the wavefront schedule, traversal side
TR_K(tr::k_wf_trace_dyn<0, 0>) TR_K(tr::k_wf_trace_dyn<1, 0>) TR_K(tr::k_wf_trace_dyn<2, 0>)
TR_K(tr::k_wf_trace_dyn<0, 1>) TR_K(tr::k_wf_trace_dyn<1, 1>) TR_K(tr::k_wf_trace_dyn<2, 1>)
TR_K(tr::k_wf_trace_fallback<0, 0>) TR_K(tr::k_wf_trace_fallback<1, 0>) TR_K(tr::k_wf_trace_fallback<2, 0>)
TR_K(tr::k_wf_trace_fallback<0, 1>) TR_K(tr::k_wf_trace_fallback<1, 1>) TR_K(tr::k_wf_trace_fallback<2, 1>)
TR_K(tr::k_wf_advance<0>) TR_K(tr::k_wf_advance<1>) TR_K(tr::k_wf_regen<0>) TR_K(tr::k_wf_regen<1>)
TR_K(tr::k_wf_bin_hist<0>) TR_K(tr::k_wf_bin_hist<1>) TR_K(tr::k_wf_bin_scatter<0>) TR_K(tr::k_wf_bin_scatter<1>) TR_K(tr::k_wf_sort<0>)
#endif
#if TR_GROUP(7)   // ... shading side, static scenes
TR_K(tr::k_wf_begin<0>)
TR_K(tr::k_wf_query_kind<0, TRAY_MAT_MATTE>) TR_K(tr::k_wf_query_kind<0, TRAY_MAT_PLASTIC>) TR_K(tr::k_wf_query_kind<0, TRAY_MAT_METAL>) TR_K(tr::k_wf_query_kind<0, TRAY_MAT_GLASS>)
TR_K(tr::k_wf_query_kind<0, TRAY_MAT_ROUGH_GLASS>) TR_K(tr::k_wf_query_kind<0, TRAY_MAT_SPECULAR_METAL>) TR_K(tr::k_wf_query_kind<0, TRAY_MAT_MERL>)
TR_K(tr::k_wf_query<0, FEAT_ALL | FEAT_TEX>)
TR_K(k_debug_sample_radiance<0>)
#endif
#if TR_GROUP(10)   // ... fused shading (round 6), static scenes
TR_K(tr::k_wf_shade_kind<0, TRAY_MAT_MATTE>) TR_K(tr::k_wf_shade_kind<0, TRAY_MAT_PLASTIC>) TR_K(tr::k_wf_shade_kind<0, TRAY_MAT_METAL>) TR_K(tr::k_wf_shade_kind<0, TRAY_MAT_GLASS>)
TR_K(tr::k_wf_shade_kind<0, TRAY_MAT_ROUGH_GLASS>) TR_K(tr::k_wf_shade_kind<0, TRAY_MAT_SPECULAR_METAL>) TR_K(tr::k_wf_shade_kind<0, TRAY_MAT_MERL>)
#endif
#if TR_GROUP(11)   // ... fused shading, moving scenes
TR_K(tr::k_wf_shade_kind<1, TRAY_MAT_MATTE>) TR_K(tr::k_wf_shade_kind<1, TRAY_MAT_PLASTIC>) TR_K(tr::k_wf_shade_kind<1, TRAY_MAT_METAL>) TR_K(tr::k_wf_shade_kind<1, TRAY_MAT_GLASS>)
TR_K(tr::k_wf_shade_kind<1, TRAY_MAT_ROUGH_GLASS>) TR_K(tr::k_wf_shade_kind<1, TRAY_MAT_SPECULAR_METAL>) TR_K(tr::k_wf_shade_kind<1, TRAY_MAT_MERL>)
#endif
#if TR_GROUP(8)   // ... shading side, moving scenes
TR_K(tr::k_wf_begin<1>)
TR_K(tr::k_wf_query_kind<1, TRAY_MAT_MATTE>) TR_K(tr::k_wf_query_kind<1, TRAY_MAT_PLASTIC>) TR_K(tr::k_wf_query_kind<1, TRAY_MAT_METAL>) TR_K(tr::k_wf_query_kind<1, TRAY_MAT_GLASS>)
TR_K(tr::k_wf_query_kind<1, TRAY_MAT_ROUGH_GLASS>) TR_K(tr::k_wf_query_kind<1, TRAY_MAT_SPECULAR_METAL>) TR_K(tr::k_wf_query_kind<1, TRAY_MAT_MERL>)
TR_K(tr::k_wf_query<1, FEAT_ALL | FEAT_TEX>)
TR_K(k_debug_sample_radiance<2>) TR_K(k_debug_sample_radiance<3>)
#endif
#if TR_GROUP(9)   // the Uniform / Adaptive samplers and AnimatedMesh
TR_K(k_sampler_pass<0, FEAT_NONE>) TR_K(k_sampler_pass<0, FEAT_ALL | FEAT_TEX>)
TR_K(k_sampler_pass<2, FEAT_NONE>) TR_K(k_sampler_pass<2, FEAT_ALL | FEAT_TEX>)
TR_K(k_sampler_pass<3, FEAT_NONE>) TR_K(k_sampler_pass<3, FEAT_ALL | FEAT_TEX>)
#endif
#define TR_INST_GROUPS 13

#undef TR_K_TILES_LF
#undef TR_K
#undef TR_GROUP
