// Derived data of the flat instance loop (hip/dev_geom.h: trace_flat, mesh_leaf_coop), built from a TrayFlatScene at
// tray_scene_create and shared with the host emulation of the device source (tests/emu):
//  * FlatLeaf / FlatInst: the instances regrouped by the BVH<Instance> leaf that holds them (the reference reaches an instance
//    only through that leaf's box, bvh.rs:89-98), each as one 128-byte record the wave-uniform loop reads with scalar loads:
//    rows of `inv`, geometry parameters, and the instance's OWN world bounding box, inflated -- a cull that is not part of the
//    reference's traversal and therefore must never reject a ray its primitive test would accept: the box is the exact bounds
//    of the transformed geometry grown by 1e-4 of the scene's scale (the f32 error of a hit point is ~1e-6 of it), and a
//    NaN in the slab arithmetic passes;
//  * tri_leaf: for the meshes small enough for the cooperative leaf test, the BVH<Triangle> leaf node of each triangle.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../../include/trayhip.h"

namespace tray {

struct FlatLeaf {   // 32 B
    float bmin[3], bmax[3];   // the BVH<Instance> leaf's box, as the reference tests it
    uint32_t first, count;    // its instances in FlatInst order
};
struct FlatInst {   // 128 B
    float inv[16];            // world -> object (row 3 matters only for quirk Q5)
    float lo[3], hi[3];       // own world bounds, inflated (conservative cull)
    float gp0, gp1;
    uint32_t geom_type, mesh_id, inst, pad[5];
};
static_assert(sizeof(FlatLeaf) == 32 && sizeof(FlatInst) == 128, "records are read with aligned scalar loads");

inline void flat_loop_gates(const TrayFlatScene* f, uint32_t coop_max_tris, std::vector<FlatLeaf>& leaves, std::vector<FlatInst>& insts,
                            std::vector<uint8_t>& tri_leaf) {
    leaves.clear(); insts.clear();
    for (uint32_t nd = 0; nd < f->n_top_nodes; ++nd) {
        const TrayBvhNode& node = f->top_nodes[nd];
        if (node.count == 0) continue;
        FlatLeaf lf{};
        for (int k = 0; k < 3; ++k) { lf.bmin[k] = node.bmin[k]; lf.bmax[k] = node.bmax[k]; }
        lf.first = (uint32_t)insts.size();
        for (uint32_t k = 0; k < node.count; ++k) {
            const uint32_t slot = node.offset + k;
            if (slot >= f->n_top_order || f->top_order[slot] >= f->n_instances) continue;
            const uint32_t i = f->top_order[slot];
            const TrayInstance& in = f->instances[i];
            if (in.kind == TRAY_INST_POINT_EMITTER) continue;   // never intersects (emitter.rs:120)
            FlatInst fi{};
            std::memcpy(fi.inv, in.inv, sizeof fi.inv);
            fi.gp0 = in.geom_params[0]; fi.gp1 = in.geom_params[1];
            fi.geom_type = in.geom_type; fi.mesh_id = in.mesh_id; fi.inst = i;
            // object-space bounds of the geometry
            float olo[3] = {0, 0, 0}, ohi[3] = {0, 0, 0};
            if (in.geom_type == TRAY_GEOM_RECT) { olo[0] = -0.5f * std::fabs(in.geom_params[0]); ohi[0] = -olo[0]; olo[1] = -0.5f * std::fabs(in.geom_params[1]); ohi[1] = -olo[1]; }
            else if (in.geom_type == TRAY_GEOM_SPHERE) { for (int k = 0; k < 3; ++k) { olo[k] = -std::fabs(in.geom_params[0]); ohi[k] = std::fabs(in.geom_params[0]); } }
            else if (in.geom_type == TRAY_GEOM_DISK) { for (int k = 0; k < 2; ++k) { olo[k] = -std::fabs(in.geom_params[0]); ohi[k] = std::fabs(in.geom_params[0]); } }
            else if (in.geom_type == TRAY_GEOM_MESH && in.mesh_id < f->n_meshes && f->meshes[in.mesh_id].node_count) {
                const TrayBvhNode& root = f->mesh_nodes[f->meshes[in.mesh_id].node_offset];
                for (int k = 0; k < 3; ++k) { olo[k] = root.bmin[k]; ohi[k] = root.bmax[k]; }
            }
            double wlo[3] = {INFINITY, INFINITY, INFINITY}, whi[3] = {-INFINITY, -INFINITY, -INFINITY};
            bool finite = true;
            for (int c = 0; c < 8; ++c) {
                const double p[3] = {(c & 1) ? ohi[0] : olo[0], (c & 2) ? ohi[1] : olo[1], (c & 4) ? ohi[2] : olo[2]};
                for (int r = 0; r < 3; ++r) {
                    const double v = (double)in.mat[4 * r] * p[0] + (double)in.mat[4 * r + 1] * p[1] + (double)in.mat[4 * r + 2] * p[2] + (double)in.mat[4 * r + 3];
                    finite = finite && std::isfinite(v);
                    wlo[r] = std::min(wlo[r], v); whi[r] = std::max(whi[r], v);
                }
            }
            // row 3 other than (0 0 0 1) would make the transform projective: no cull then
            finite = finite && in.mat[12] == 0.0f && in.mat[13] == 0.0f && in.mat[14] == 0.0f && in.mat[15] == 1.0f;
            double scale = 0.0;
            for (int r = 0; r < 3; ++r) scale = std::max({scale, std::fabs(wlo[r]), std::fabs(whi[r])});
            const double margin = 1e-4 * scale + 1e-6;
            for (int r = 0; r < 3; ++r) {
                fi.lo[r] = finite ? (float)(wlo[r] - margin) : -INFINITY;
                fi.hi[r] = finite ? (float)(whi[r] + margin) : INFINITY;
            }
            insts.push_back(fi);
        }
        lf.count = (uint32_t)insts.size() - lf.first;
        if (lf.count) leaves.push_back(lf);
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
