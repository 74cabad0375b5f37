// Derived data of the flat instance loop (hip/dev_geom.h: trace_flat, mesh_leaf_coop), built from a TrayFlatScene at
// tray_scene_create and shared with the host emulation of the device source (tests/emu):
//  * FlatLeaf / FlatInst: the instances regrouped by the BVH<Instance> leaf that holds them (the reference reaches an instance
//    only through that leaf's box, bvh.rs:89-98), each as one 128-byte record the wave-uniform loop reads with scalar loads:
//    rows of `inv`, geometry parameters, and the instance's OWN world bounding box, inflated -- a cull that is not part of the
//    reference's traversal and therefore must never reject a ray its primitive test would accept: the box is the exact bounds
//    of the transformed geometry grown by 1e-4 of the scene's scale (the f32 error of a hit point is ~1e-6 of it), and a
//    NaN in the slab arithmetic passes;
//  * tri_leaf: for the meshes small enough for the cooperative leaf test, the BVH<Triangle> leaf node of each triangle
//    (an index into the mesh's tree in DEVICE order, below);
//  * PairedTrees: the device's order of the two kinds of tree. The reference flattens a BVH in preorder (bvh.rs:278-295: the first
//    child follows its parent, the second lies wherever the first one's subtree ends). On the device the children of a node are
//    NEIGHBOURS -- one 64-byte aligned record per sibling pair, an interior node's `offset` = index of its first child, the second
//    child at offset + 1; the root is node 0 and its twin (node 1) is an empty box nothing refers to. Same nodes, same boxes, same
//    leaves: only where a node lies changes, so every traversal visits what the reference visits in the reference's order, and
//    expanding a node -- testing the boxes of both its children -- is ONE fetch that needs nothing but the node's `offset`.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstddef>
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

struct PairedTrees {
    std::vector<TrayBvhNode> top, mesh;   // BVH<Instance>, all BVH<Triangle> in device order
    std::vector<TrayMesh> meshes;         // f->meshes with node_offset / node_count of `mesh`
    bool narrow = true;                   // every offset fits the descriptor's 23 bits (trees of < 8.4 M nodes / triangles)
};
static_assert(offsetof(TrayBvhNode, count) == 28 && sizeof(TrayBvhNode) == 32, "the descriptor is word 7");
// appends the tree ref[0 .. n) in device order (n + 1 nodes); false if ref is not a tree of n nodes
inline bool pair_tree(const TrayBvhNode* ref, uint32_t n, std::vector<TrayBvhNode>& out, bool& narrow) {
    if (n == 0u) return true;
    const size_t base = out.size();
    out.reserve(base + n + 1u);
    out.push_back(ref[0]);
    TrayBvhNode twin{};
    for (int k = 0; k < 3; ++k) { twin.bmin[k] = INFINITY; twin.bmax[k] = -INFINITY; }
    twin.count = 1;   // an empty leaf behind a box no ray enters
    out.push_back(twin);
    std::vector<std::pair<uint32_t, uint32_t>> st;   // (index in ref, index in out relative to base)
    st.push_back({0u, 0u});
    uint32_t placed = 1u;
    while (!st.empty()) {
        const auto [i, j] = st.back();
        st.pop_back();
        out[base + j] = ref[i];
        if (ref[i].count != 0) continue;
        const uint32_t first = i + 1u, second = ref[i].offset;
        placed += 2u;
        if (first >= n || second >= n || placed > n) return false;
        const uint32_t c = (uint32_t)(out.size() - base);
        out.resize(out.size() + 2u);
        out[base + j].offset = c;
        st.push_back({second, c + 1u});
        st.push_back({first, c});      // (popped first: the first child's subtree follows its pair, as in the reference's order)
    }
    if (placed != n) return false;
    // word 7 of a device node (count / axis / pad in the ABI's struct) becomes the node's DESCRIPTOR: offset, count and axis in one
    // word whose top two bits are 0 (hip/dev_geom.h: nd_offset / nd_count / nd_axis) -- what a traversal keeps of a node whose box
    // it has tested; word 6 stays the full offset
    for (size_t j = base; j < out.size(); ++j) {
        TrayBvhNode& nd = out[j];
        if (nd.offset > 0x7fffffu || nd.count > 31u) narrow = false;
        const uint32_t desc = (nd.offset & 0x7fffffu) | ((uint32_t)std::min<uint32_t>(nd.count, 31u) << 23) | (((uint32_t)nd.axis & 3u) << 28);
        std::memcpy(&nd.count, &desc, sizeof desc);
    }
    return true;
}
inline bool pair_trees(const TrayFlatScene* f, PairedTrees& p, bool top_only = false) {
    p.top.clear();
    if (!pair_tree(f->top_nodes, f->n_top_nodes, p.top, p.narrow)) return false;
    if (top_only) return true;
    p.mesh.clear();
    p.meshes.assign(f->meshes, f->meshes + f->n_meshes);
    for (uint32_t m = 0; m < f->n_meshes; ++m) {
        const TrayMesh& me = f->meshes[m];
        p.meshes[m].node_offset = (uint32_t)p.mesh.size();
        if ((uint64_t)me.node_offset + me.node_count > f->n_mesh_nodes || !pair_tree(f->mesh_nodes + me.node_offset, me.node_count, p.mesh, p.narrow)) return false;
        p.meshes[m].node_count = (uint32_t)p.mesh.size() - p.meshes[m].node_offset;
    }
    return true;
}

// One 64-byte record per BVH<Instance> leaf slot (top_order), for the wavefront traversal (hip/wavefront.h): everything an instance
// entry needs -- rows 0..2 of world -> object, what the instance is, and where its tree / what its parameters are -- in ONE
// fetch addressed by the stack entry itself, instead of the chain top_order -> TrayInstance (kind, animated, geom_type, inv at four
// places of a 224-byte record) -> TrayMesh. Moving instances take their rows from the per-path transform cache (slot in `flags`);
// a transform whose row 3 is not (0 0 0 1) (never produced by the loader: instance transforms are products of TRS keyframes) keeps
// the flag WI_AFFINE clear and is read from the TrayInstance with the projective quirk of xf_point.
struct WfInst {
    float inv[12];
    uint32_t flags;   // bits 0-2 geom_type, WI_* bits, moving_slot << 8
    uint32_t inst;    // index in instances[]
    union { float gp[2]; uint32_t tree[2]; };   // geometry parameters | first node of the BVH<Triangle> in mesh_nodes, first triangle in tri_verts
};
static_assert(sizeof(WfInst) == 64, "one record = one 64-byte fetch");
enum : uint32_t { WI_POINT = 1u << 3, WI_ANIMATED = 1u << 4, WI_AFFINE = 1u << 5 };
inline void wf_inst_records(const TrayFlatScene* f, const std::vector<TrayMesh>& paired_meshes, std::vector<WfInst>& out) {
    out.assign(f->n_top_order, WfInst{});
    for (uint32_t k = 0; k < f->n_top_order; ++k) {
        WfInst& r = out[k];
        const uint32_t i = f->top_order[k];
        r.inst = i;
        if (i >= f->n_instances) { r.flags = WI_POINT; continue; }   // (validated away; never intersects)
        const TrayInstance& in = f->instances[i];
        std::memcpy(r.inv, in.inv, sizeof r.inv);
        r.flags = in.geom_type & 7u;
        if (in.kind == TRAY_INST_POINT_EMITTER) r.flags |= WI_POINT;
        if (in.animated) r.flags |= WI_ANIMATED | (in.moving_slot << 8);
        if (in.inv[12] == 0.0f && in.inv[13] == 0.0f && in.inv[14] == 0.0f && in.inv[15] == 1.0f) r.flags |= WI_AFFINE;
        if (in.geom_type == TRAY_GEOM_MESH && in.mesh_id < paired_meshes.size()) {
            r.tree[0] = paired_meshes[in.mesh_id].node_offset;
            r.tree[1] = paired_meshes[in.mesh_id].tri_offset;
        } else {
            r.gp[0] = in.geom_params[0]; r.gp[1] = in.geom_params[1];
        }
    }
}

inline void flat_loop_gates(const TrayFlatScene* f, const PairedTrees& paired, uint32_t coop_max_tris, std::vector<FlatLeaf>& leaves, std::vector<FlatInst>& insts,
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
    for (uint32_t m = 0; m < f->n_meshes && m < paired.meshes.size(); ++m) {
        const TrayMesh& me = paired.meshes[m];
        if (me.tri_count > coop_max_tris || me.node_count > 256u) continue;
        const TrayBvhNode* tree = paired.mesh.data() + me.node_offset;
        for (uint32_t nd = 0; nd < me.node_count; ++nd) {
            uint32_t desc;
            std::memcpy(&desc, &tree[nd].count, sizeof desc);
            const uint32_t count = (desc >> 23) & 31u;
            for (uint32_t k = 0; nd != 1u && k < count; ++k)   // (node 1 is the root's empty twin)
                if (tree[nd].offset + k < me.tri_count) tri_leaf[me.tri_offset + tree[nd].offset + k] = (uint8_t)nd;
        }
    }
}

}  // namespace tray
