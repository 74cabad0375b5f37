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
#include <cstdlib>
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
    uint32_t geom_type, mesh_id, inst;
    uint32_t animated;        // the instance moves while the shutter is open: its transform is the path's (per-path cache), `inv` above is not used
    uint32_t leaf;            // index of its FlatLeaf (the per-lane pass of trace_flat re-derives the gate's entry distance from that box)
    uint32_t lane_pass;       // 1: tested in trace_flat's per-lane pass -- a rectangle or disk alone in a FLAT leaf box (thinnest side <= 1 % of the longest),
                              // which a ray passes only where it crosses the surface; everything else stays in the wave-uniform loop
    uint32_t pad[2];
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

// QuadTrees: the wavefront traversal's view of the same trees (hip/wavefront.h: k_wf_trace_dyn) -- TWO levels of the binary tree per
// 128-byte record, so a ray's chain of dependent fetches is half as long (a divergent lane's step costs one round trip whatever the
// record's size once it comes out of HBM: profiles/r03_ubench_gather.txt). The record of an interior node N with children A, B holds
// up to four SLOTS -- (box, descriptor) pairs -- in the fixed order [A-group | B-group]:
//     a leaf child C             -> (C)                      its own box, descriptor = (first primitive, count)
//     an interior child C        -> (C.first, C.second)      the grandchildren's boxes; C's own box is NOT stored
// and the split axes of N, A and B (meta word 0), from which follows the order in which the reference would reach the slots: near
// child of N first (bvh.rs:105-119), inside a group the near child of A (of B) -- tabulated per direction octant in meta word 1. The
// descriptor of an interior slot X is the index of X's own record. Unused slots hold a box no ray enters (every plane at +inf).
// Why C's own box may be skipped: the reference tests it before it looks at C.first / C.second (bvh.rs:89-92), and a BVH node's box is
// the exact union of its children's (min / max of floats are exact), so C.first's box lies inside C's componentwise; for a ray whose
// reciprocal direction is finite and nonzero in every component the slab arithmetic of fast_intersect (bbox.rs:75-104: subtract,
// multiply, compare -- each monotone, no NaN can arise) then gives tmin(C) <= tmin(C.first) and tmax(C) >= tmax(C.first) axis by axis,
// hence hit(C.first, max_t) implies hit(C, max_t'), for every max_t' >= max_t: C's test can only fail where both grandchildren's fail.
// The traversal sends rays with a zero / denormal / non-finite direction component (where -0 * inf and 0 * inf break that argument and
// the reference's own behaviour is erratic) to the binary trees instead (k_wf_trace_fallback -> trace_bvh). The builder checks the
// containment for every collapsed child and keeps a child whose box does NOT contain its children's (a caller-built BVH) as a slot of
// its own with an explicit test, so the result is the reference's for any tree validate_flat_scene admits.
// Record order: the top `bfs_levels` levels of a tree breadth-first (dense, cache resident), below them depth-first with the records
// of a node's interior slots adjacent (as pair_tree).
struct QuadNode {   // 128 B = 8 float4: lo.x, lo.y, lo.z, hi.x, hi.y, hi.z of the 4 slots, 4 descriptors, meta
    float lo[3][4], hi[3][4];
    uint32_t desc[4];
    uint32_t meta[4];   // [0]: axis of N (bits 0-1), of the A-group's node (2-3), of the B-group's node (4-5)
                        // [1]: the visiting order per direction octant o = (d.x < 0) | (d.y < 0) << 1 | (d.z < 0) << 2, three bits at 3 * o:
                        //      bit 0 = N's second child first, bit 1 / 2 = the A / B group's second slot first (bvh.rs:105-119 on both levels)
};
static_assert(sizeof(QuadNode) == 128, "one record = eight 16-byte loads of one 128-byte line");
struct QuadTrees {
    std::vector<QuadNode> top, mesh;     // BVH<Instance>; all BVH<Triangle>s
    std::vector<uint32_t> mesh_first;    // per mesh: index of its entry record in `mesh`
    bool narrow = true;                  // every descriptor field fits (23-bit offsets, 5-bit counts)
    bool ordered = true;                 // every box has min <= max on every axis (no NaN): the traversal's form of the slab test assumes it
    uint32_t top_pend = 0, mesh_pend = 0;   // most node entries (descriptor + entry distance pairs) a traversal of BVH<Instance> / of one BVH<Triangle> can have pending
};
#ifndef TRAY_QUAD_BFS_LEVELS
#define TRAY_QUAD_BFS_LEVELS 5
#endif
inline QuadNode quad_empty_record() {
    QuadNode q{};
    // every plane at +inf: whatever the signs of a (regular) ray's direction, the slot's entry distance is +inf or its exit distance -inf
    for (int a = 0; a < 3; ++a) for (int s = 0; s < 4; ++s) { q.lo[a][s] = INFINITY; q.hi[a][s] = INFINITY; }
    for (int s = 0; s < 4; ++s) q.desc[s] = 1u << 23;   // (a leaf of one primitive behind a box nothing enters)
    return q;
}
// appends the quad records of the reference-order tree ref[0 .. n) (a tree: pair_tree has accepted it); the entry record is the
// first one appended. Returns the largest number of node entries a traversal of this tree can have pending at once.
// `prim_box` (BVH<Instance> only; [lo x y z, hi x y z] per ordered primitive, or null): a leaf becomes one more record whose slots are the leaf's
// primitives, in leaf order, each behind a CONSERVATIVE box of its own (instance_gate_box) -- the reference enters every instance of a leaf it reaches
// (bvh.rs:96-104), the ones whose box the ray misses cannot be hit, and skipping them here saves the traversal an instance entry each (record fetch, ray
// transform, three divisions, the mesh's entry record). The leaf's own box stays in its parent's slot: the reference tests it.
inline uint32_t quad_tree(const TrayBvhNode* ref, uint32_t n, std::vector<QuadNode>& out, bool& narrow, bool& ordered, uint32_t bfs_levels = TRAY_QUAD_BFS_LEVELS,
                          const float (*prim_box)[6] = nullptr) {
    if (n == 0u) { out.push_back(quad_empty_record()); return 1u; }
    const size_t base = out.size();
    const uint32_t none = 0xffffffffu;
    auto is_leaf = [&](uint32_t i) { return ref[i].count != 0; };
    auto inside = [&](uint32_t c, uint32_t p) {   // box of c within box of p, componentwise (false for NaNs)
        for (int a = 0; a < 3; ++a) if (!(ref[c].bmin[a] >= ref[p].bmin[a] && ref[c].bmax[a] <= ref[p].bmax[a])) return false;
        return true;
    };
    auto collapsible = [&](uint32_t i) {   // an interior node whose own box test is implied by its children's
        return !is_leaf(i) && i + 1u < n && ref[i].offset < n && inside(i + 1u, i) && inside(ref[i].offset, i);
    };
    // a record to fill: the slots of interior node `node` (wrap: `node` itself as the only slot -- a leaf root, or a root whose box
    // does not contain its children's); pend: node entries pending on the stack when the traversal expands this record
    struct Job { uint32_t node, rec, level, pend; bool wrap; };
    uint32_t worst = 0u;
    auto alloc = [&] { out.push_back(quad_empty_record()); return (uint32_t)(out.size() - base - 1u); };
    // fills j's record; the records of its interior slots are allocated adjacently in slot order and their jobs appended to `next`
    // (reverse: last slot first, for a stack whose top is processed next)
    auto fill = [&](const Job j, std::vector<Job>& next, bool reverse) {
        uint32_t slot_node[4] = {none, none, none, none};
        uint32_t axes = 0u;
        if (j.wrap) slot_node[0] = j.node;
        else {
            const uint32_t c[2] = {j.node + 1u, ref[j.node].offset};
            axes = (uint32_t)ref[j.node].axis & 3u;
            for (int g = 0; g < 2; ++g) {
                if (collapsible(c[g])) {
                    slot_node[2 * g] = c[g] + 1u; slot_node[2 * g + 1] = ref[c[g]].offset;
                    axes |= ((uint32_t)ref[c[g]].axis & 3u) << (2 + 2 * g);
                } else slot_node[2 * g] = c[g];
            }
        }
        uint32_t used = 0u;
        for (int s = 0; s < 4; ++s) used += slot_node[s] != none ? 1u : 0u;
        // of the slots hit, all but the one the traversal goes on with are pushed
        const uint32_t pend = j.pend + used - 1u;
        worst = std::max(worst, pend);
        Job kids[4];
        int n_kids = 0;
        for (int s = 0; s < 4; ++s) {
            const uint32_t x = slot_node[s];
            if (x == none) continue;
            uint32_t desc;
            if (is_leaf(x) && prim_box && ref[x].count <= 4u && ref[x].offset <= 0x7fffffu - 4u) {
                const uint32_t r = alloc();   // the leaf's primitives, one slot each, visited in slot order whatever the ray's direction (order word 0)
                if (r > 0x7fffffu) narrow = false;
                desc = r & 0x7fffffu;
                QuadNode& lq = out[base + r];
                for (uint32_t k = 0; k < ref[x].count; ++k) {
                    const float* b = prim_box[ref[x].offset + k];
                    for (int a = 0; a < 3; ++a) { lq.lo[a][k] = b[a]; lq.hi[a][k] = b[3 + a]; if (!(b[a] <= b[3 + a])) ordered = false; }
                    lq.desc[k] = (ref[x].offset + k) | (1u << 23);
                }
                lq.meta[0] = 0u; lq.meta[1] = 0u;
                worst = std::max(worst, pend + ref[x].count - 1u);
            } else if (is_leaf(x)) {
                if (ref[x].offset > 0x7fffffu || ref[x].count > 31u) narrow = false;
                desc = (ref[x].offset & 0x7fffffu) | ((uint32_t)std::min<uint32_t>(ref[x].count, 31u) << 23);
            } else {
                const uint32_t r = alloc();   // (may move `out`: the record is addressed by index below)
                if (r > 0x7fffffu) narrow = false;
                desc = r & 0x7fffffu;
                kids[n_kids++] = Job{x, r, j.level + 1u, pend, false};
            }
            QuadNode& q = out[base + j.rec];
            for (int a = 0; a < 3; ++a) {
                q.lo[a][s] = ref[x].bmin[a]; q.hi[a][s] = ref[x].bmax[a];
                if (!(ref[x].bmin[a] <= ref[x].bmax[a])) ordered = false;
            }
            q.desc[s] = desc;
        }
        out[base + j.rec].meta[0] = axes;
        uint32_t order = 0u;
        for (uint32_t oct = 0; oct < 8u; ++oct) {
            const uint32_t neg_n = (oct >> (axes & 3u)) & 1u, neg_a = (oct >> ((axes >> 2) & 3u)) & 1u, neg_b = (oct >> ((axes >> 4) & 3u)) & 1u;
            order |= (neg_n | (neg_a << 1) | (neg_b << 2)) << (3u * oct);
        }
        out[base + j.rec].meta[1] = order;
        if (reverse) for (int k = n_kids - 1; k >= 0; --k) next.push_back(kids[k]);
        else for (int k = 0; k < n_kids; ++k) next.push_back(kids[k]);
    };
    // breadth-first for the top levels ...
    std::vector<Job> bfs, dfs_roots;
    bfs.push_back(Job{0u, alloc(), 0u, 0u, is_leaf(0u) || !collapsible(0u)});
    for (size_t head = 0; head < bfs.size(); ++head) {
        const Job j = bfs[head];
        if (!j.wrap && j.level >= bfs_levels) dfs_roots.push_back(j);
        else fill(j, bfs, false);
    }
    // ... depth-first below them (a slot's subtree follows the records of its siblings)
    std::vector<Job> st;
    for (const Job& root : dfs_roots) {
        st.push_back(root);
        while (!st.empty()) {
            const Job j = st.back();
            st.pop_back();
            fill(j, st, true);
        }
    }
    return worst + 1u;
}
inline void instance_gate_box(const TrayFlatScene* f, const TrayInstance& in, float lo[3], float hi[3]);   // (below)
inline void quad_trees(const TrayFlatScene* f, QuadTrees& q, bool top_only = false, uint32_t bfs_levels = TRAY_QUAD_BFS_LEVELS) {
    q.top.clear();
    std::vector<float> boxes((size_t)f->n_top_order * 6u, 0.0f);   // per BVH<Instance> leaf slot: the instance's conservative box (TRAYHIP_NO_INSTANCE_BOXES: leaves as the reference has them)
    for (uint32_t k = 0; k < f->n_top_order; ++k) {
        float* b = &boxes[(size_t)k * 6u];
        if (f->top_order[k] < f->n_instances) instance_gate_box(f, f->instances[f->top_order[k]], b, b + 3);
        else for (int a = 0; a < 3; ++a) { b[a] = -INFINITY; b[3 + a] = INFINITY; }
    }
    const bool per_instance = !getenv("TRAYHIP_NO_INSTANCE_BOXES") && f->n_top_order != 0u;
    q.top_pend = quad_tree(f->top_nodes, f->n_top_nodes, q.top, q.narrow, q.ordered, bfs_levels, per_instance ? reinterpret_cast<const float (*)[6]>(boxes.data()) : nullptr);
    if (top_only) return;
    q.mesh.clear();
    q.mesh_first.assign(f->n_meshes, 0u);
    q.mesh_pend = 0u;
    for (uint32_t m = 0; m < f->n_meshes; ++m) {
        const TrayMesh& me = f->meshes[m];
        q.mesh_first[m] = (uint32_t)q.mesh.size();
        q.mesh_pend = std::max(q.mesh_pend, quad_tree(f->mesh_nodes + me.node_offset, me.node_count, q.mesh, q.narrow, q.ordered, bfs_levels));
    }
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
    union { float gp[2]; uint32_t tree[2]; };   // geometry parameters | entry record of the BVH<Triangle> in mesh_quads (QuadTrees), first triangle in tri_verts
};
static_assert(sizeof(WfInst) == 64, "one record = one 64-byte fetch");
enum : uint32_t { WI_POINT = 1u << 3, WI_ANIMATED = 1u << 4, WI_AFFINE = 1u << 5 };
inline void wf_inst_records(const TrayFlatScene* f, const std::vector<uint32_t>& quad_first, std::vector<WfInst>& out) {
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
        if (in.geom_type == TRAY_GEOM_MESH && in.mesh_id < quad_first.size() && in.mesh_id < f->n_meshes) {
            r.tree[0] = quad_first[in.mesh_id];
            r.tree[1] = f->meshes[in.mesh_id].tri_offset;
        } else {
            r.gp[0] = in.geom_params[0]; r.gp[1] = in.geom_params[1];
        }
    }
}

// A conservative world-space box of an instance's geometry at this frame: the object-space bounds through `mat` in double precision, widened by
// 1e-4 of the scene's scale (far above what f32 rounding of the ray transform and of the intersection tests can move a hit) -- a ray that misses it
// cannot hit the instance. Infinite for an instance that moves within the frame and for a projective transform (no cull).
inline void instance_gate_box(const TrayFlatScene* f, const TrayInstance& in, float lo[3], float hi[3]) {
    // object-space bounds of the geometry
    float olo[3] = {0, 0, 0}, ohi[3] = {0, 0, 0};
    if (in.geom_type == TRAY_GEOM_RECT) { olo[0] = -0.5f * std::fabs(in.geom_params[0]); ohi[0] = -olo[0]; olo[1] = -0.5f * std::fabs(in.geom_params[1]); ohi[1] = -olo[1]; }
    else if (in.geom_type == TRAY_GEOM_SPHERE) { for (int k = 0; k < 3; ++k) { olo[k] = -std::fabs(in.geom_params[0]); ohi[k] = std::fabs(in.geom_params[0]); } }
    else if (in.geom_type == TRAY_GEOM_DISK) { for (int k = 0; k < 2; ++k) { olo[k] = -std::fabs(in.geom_params[0]); ohi[k] = std::fabs(in.geom_params[0]); } }
    else if ((in.geom_type == TRAY_GEOM_MESH || in.geom_type == TRAY_GEOM_ANIMATED_MESH) && in.mesh_id < f->n_meshes && f->meshes[in.mesh_id].node_count) {   // (an AnimatedMesh: the root of its one tree -- nothing outside it is ever reached, quirk Q13)
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
    // row 3 other than (0 0 0 1) would make the transform projective: no cull then; nor for an instance that moves within the frame
    // (its own box would have to be its swept one; the gate that counts -- the BVH<Instance> leaf's box -- is the reference's either way)
    finite = finite && in.mat[12] == 0.0f && in.mat[13] == 0.0f && in.mat[14] == 0.0f && in.mat[15] == 1.0f && !in.animated;
    double scale = 0.0;
    for (int r = 0; r < 3; ++r) scale = std::max({scale, std::fabs(wlo[r]), std::fabs(whi[r])});
    const double margin = 1e-4 * scale + 1e-6;
    for (int r = 0; r < 3; ++r) {
        lo[r] = finite ? (float)(wlo[r] - margin) : -INFINITY;
        hi[r] = finite ? (float)(whi[r] + margin) : INFINITY;
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
            fi.geom_type = in.geom_type; fi.mesh_id = in.mesh_id; fi.inst = i; fi.animated = in.animated ? 1u : 0u;
            instance_gate_box(f, in, fi.lo, fi.hi);
            fi.leaf = (uint32_t)leaves.size();   // (this leaf is pushed below: it holds at least this instance)
            {
                float emin = INFINITY, emax = 0.0f;
                for (int c = 0; c < 3; ++c) { const float e = node.bmax[c] - node.bmin[c]; emin = std::min(emin, e); emax = std::max(emax, e); }
                const bool flat_box = std::isfinite(emax) && emin >= 0.0f && emin <= 0.01f * emax;
                fi.lane_pass = (node.count == 1 && flat_box && !in.animated && (in.geom_type == TRAY_GEOM_RECT || in.geom_type == TRAY_GEOM_DISK)) ? 1u : 0u;
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
