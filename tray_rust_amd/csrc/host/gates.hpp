// Derived per-instance / per-triangle data of the flat instance loop (hip/dev_geom.h: trace_flat, mesh_leaf_coop), built from a
// TrayFlatScene at tray_scene_create: the BVH<Instance> leaf node that holds each instance, and for the meshes small enough for
// the cooperative leaf test the BVH<Triangle> leaf node of each triangle. The reference reaches an instance / a triangle only
// through those boxes (bvh.rs:89-98). Shared with the host emulation of the device source (tests/emu).
#pragma once
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../../include/trayhip.h"

namespace tray {

inline void flat_loop_gates(const TrayFlatScene* f, uint32_t coop_max_tris, std::vector<TrayBvhNode>& inst_leaf, std::vector<uint8_t>& tri_leaf) {
    inst_leaf.assign(f->n_instances, TrayBvhNode{});
    for (uint32_t nd = 0; nd < f->n_top_nodes; ++nd)
        for (uint32_t k = 0; k < f->top_nodes[nd].count; ++k) {
            const uint32_t slot = f->top_nodes[nd].offset + k;
            if (slot < f->n_top_order && f->top_order[slot] < f->n_instances) inst_leaf[f->top_order[slot]] = f->top_nodes[nd];
        }
    tri_leaf.assign(f->n_tris, 0u);
    for (uint32_t m = 0; m < f->n_meshes; ++m) {
        const TrayMesh& me = f->meshes[m];
        if (me.tri_count > coop_max_tris || me.node_count > 255u) continue;
        const TrayBvhNode* tree = f->mesh_nodes + me.node_offset;
        for (uint32_t nd = 0; nd < me.node_count; ++nd)
            for (uint32_t k = 0; k < tree[nd].count; ++k)
                if (tree[nd].offset + k < me.tri_count) tri_leaf[me.tri_offset + tree[nd].offset + k] = (uint8_t)nd;
    }
}

}  // namespace tray
