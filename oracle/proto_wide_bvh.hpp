// TEST INFRASTRUCTURE ONLY -- DESIGN PROTOTYPE, not part of the oracle's restatement of the reference and not used by the
// product. It answers, on the CPU, two questions the next device traversal depends on (DESIGN.md "Next", C5):
//   1. is a 4-wide collapse of the reference's binary BVH<Triangle> EXACT (same accepted candidates, same order, hence the same
//      hit record bit for bit) if a popped child is re-tested against the ray's current max_t?
//   2. how many dependent node fetches per ray does it save?
// Collapse: the wide node of a binary interior node N holds its grandchildren (a child that is a leaf stays as it is), slots in
// binary order [L.first, L.second, R.first, R.second] with each slot's own box. Traversal: all slot boxes are tested when the
// wide node is reached (a box missed with the current max_t is missed with any later, smaller one), the hit ones are pushed in
// reverse visiting order together with their entry distance tmin; on pop the reference's box test with the CURRENT max_t is
// `tmin < max_t` (the only clause of BBox::fast_intersect that involves max_t). The tests of the skipped intermediate nodes L
// and R are implied: a box that contains a hit box is hit (every step of the slab test is monotone in the box).
#pragma once
#include <cstring>
#include <vector>

#include "oracle_scene.hpp"

namespace orc {

struct WideNode {
    float bmin[4][3], bmax[4][3];
    uint32_t ref[4];    // kind 1: index of the child's wide node; kind 2: index of the binary leaf node
    uint8_t kind[4];    // 0 empty, 1 interior, 2 leaf
    uint8_t axis_top, axis_l, axis_r, pad;
};
struct WideBvh {
    std::vector<WideNode> nodes;   // nodes[0] is the root's wide node when the binary root is interior
};
struct ProtoCounters { unsigned long long binary_fetches = 0, wide_fetches = 0, leaf_visits_binary = 0, leaf_visits_wide = 0; };

// Third question: what does it cost to store the slot boxes QUANTISED (qbits per coordinate, relative to the union of the four
// slots, rounded outwards)? The dequantised box contains the exact one, so no visit is lost; the extra visits (and the rare
// candidate the exact box would have culled by rounding) are what the test measures. quantise_wide() rewrites a WideBvh in place.
inline void quantise_wide(WideBvh& wb, int qbits) {
    const float levels = (float)((1u << qbits) - 1u);
    for (WideNode& w : wb.nodes) {
        float lo[3] = {INF, INF, INF}, hi[3] = {-INF, -INF, -INF};
        for (int s = 0; s < 4; ++s)
            if (w.kind[s])
                for (int k = 0; k < 3; ++k) { lo[k] = std::fmin(lo[k], w.bmin[s][k]); hi[k] = std::fmax(hi[k], w.bmax[s][k]); }
        for (int k = 0; k < 3; ++k) {
            float scale = (hi[k] - lo[k]) / levels;
            if (!(scale > 0.0f)) scale = 1e-30f;
            for (int s = 0; s < 4; ++s) {
                if (!w.kind[s]) continue;
                float qmin = std::floor((w.bmin[s][k] - lo[k]) / scale), qmax = std::ceil((w.bmax[s][k] - lo[k]) / scale);
                qmin = std::fmax(0.0f, std::fmin(levels, qmin)); qmax = std::fmax(0.0f, std::fmin(levels, qmax));
                float dmin = lo[k] + qmin * scale, dmax = lo[k] + qmax * scale;
                while (dmin > w.bmin[s][k] && qmin > 0.0f) { qmin -= 1.0f; dmin = lo[k] + qmin * scale; }   // rounding of the dequantisation itself
                while (dmax < w.bmax[s][k] && qmax < levels) { qmax += 1.0f; dmax = lo[k] + qmax * scale; }
                if (dmin > w.bmin[s][k]) dmin = w.bmin[s][k];
                if (dmax < w.bmax[s][k]) dmax = w.bmax[s][k];
                w.bmin[s][k] = dmin; w.bmax[s][k] = dmax;
            }
        }
    }
}

inline uint32_t wide_build(const TrayBvhNode* tree, uint32_t n, WideBvh& out) {
    uint32_t self = (uint32_t)out.nodes.size();
    out.nodes.push_back(WideNode{});
    const TrayBvhNode& N = tree[n];
    const uint32_t kids[2] = {n + 1u, N.offset};
    uint8_t axes[2] = {0, 0};
    uint32_t slot_node[4] = {0, 0, 0, 0};
    uint8_t slot_kind[4] = {0, 0, 0, 0};
    for (int c = 0; c < 2; ++c) {
        const TrayBvhNode& K = tree[kids[c]];
        if (K.count > 0) { slot_node[2 * c] = kids[c]; slot_kind[2 * c] = 2; }
        else {
            axes[c] = K.axis;
            slot_node[2 * c] = kids[c] + 1u; slot_node[2 * c + 1] = K.offset;
            slot_kind[2 * c] = tree[kids[c] + 1u].count > 0 ? 2 : 1;
            slot_kind[2 * c + 1] = tree[K.offset].count > 0 ? 2 : 1;
        }
    }
    for (int sidx = 0; sidx < 4; ++sidx) {
        if (!slot_kind[sidx]) continue;
        const TrayBvhNode& S = tree[slot_node[sidx]];
        uint32_t ref = slot_kind[sidx] == 1 ? wide_build(tree, slot_node[sidx], out) : slot_node[sidx];
        WideNode& w = out.nodes[self];   // (re-fetch: the vector may have grown)
        for (int k = 0; k < 3; ++k) { w.bmin[sidx][k] = S.bmin[k]; w.bmax[sidx][k] = S.bmax[k]; }
        w.ref[sidx] = ref; w.kind[sidx] = slot_kind[sidx];
    }
    WideNode& w = out.nodes[self];
    w.axis_top = N.axis; w.axis_l = axes[0]; w.axis_r = axes[1];
    return self;
}

// the slab test of BBox::fast_intersect (bbox.rs:75-104) that also returns the entry distance it compares with max_t
inline bool slab_tmin(const float bmin[3], const float bmax[3], const Ray& r, Vec3 inv_dir, const int neg_dir[3], float& tmin_out) {
    const float* b[2] = {bmin, bmax};
    float tmin = (b[neg_dir[0]][0] - r.o.x) * inv_dir.x;
    float tmax = (b[1 - neg_dir[0]][0] - r.o.x) * inv_dir.x;
    float tymin = (b[neg_dir[1]][1] - r.o.y) * inv_dir.y;
    float tymax = (b[1 - neg_dir[1]][1] - r.o.y) * inv_dir.y;
    if (tmin > tymax || tymin > tmax) return false;
    if (tymin > tmin) tmin = tymin;
    if (tymax < tmax) tmax = tymax;
    float tzmin = (b[neg_dir[2]][2] - r.o.z) * inv_dir.z;
    float tzmax = (b[1 - neg_dir[2]][2] - r.o.z) * inv_dir.z;
    if (tmin > tzmax || tzmin > tmax) return false;
    if (tzmin > tmin) tmin = tzmin;
    if (tzmax < tmax) tmax = tzmax;
    tmin_out = tmin;
    return tmin < r.max_t && tmax > r.min_t;
}

template <class LeafFn>
inline void wide_traverse(const TrayBvhNode* tree, const WideBvh& wb, Ray& ray, LeafFn&& leaf, ProtoCounters* pc) {
    Vec3 inv_dir(1.0f / ray.d.x, 1.0f / ray.d.y, 1.0f / ray.d.z);
    int neg_dir[3] = {ray.d.x < 0.0f, ray.d.y < 0.0f, ray.d.z < 0.0f};
    if (pc) pc->wide_fetches++;   // the root record
    if (!bbox_fast_intersect(tree[0], ray, inv_dir, neg_dir)) return;
    if (tree[0].count > 0) { if (pc) pc->leaf_visits_wide++; leaf(tree[0].offset, (uint32_t)tree[0].count); return; }
    struct Entry { uint32_t ref; uint8_t kind; float tmin; };
    Entry stack[128];
    int sp = 0;
    uint32_t current = 0;   // wide node whose binary node's box is known to be hit
    for (;;) {
        const WideNode& w = wb.nodes[current];
        if (pc) pc->wide_fetches++;
        // reference visiting order of the four slots: near child of N first, inside a child its near child first
        int order[4];
        const int first_c = neg_dir[w.axis_top] ? 1 : 0;
        for (int c = 0; c < 2; ++c) {
            const int child = c == 0 ? first_c : 1 - first_c;
            const int ax = child == 0 ? w.axis_l : w.axis_r;
            const bool leaf_child = w.kind[2 * child + 1] == 0;   // the child itself is a leaf: one slot
            const int near_s = (!leaf_child && neg_dir[ax]) ? 1 : 0;
            order[2 * c] = 2 * child + near_s;
            order[2 * c + 1] = 2 * child + (1 - near_s);
        }
        for (int k = 3; k >= 0; --k) {   // push in reverse visiting order
            const int s = order[k];
            if (!w.kind[s]) continue;
            float tmin;
            if (slab_tmin(w.bmin[s], w.bmax[s], ray, inv_dir, neg_dir, tmin)) stack[sp++] = Entry{w.ref[s], w.kind[s], tmin};
        }
        bool have = false;
        while (sp > 0) {
            const Entry e = stack[--sp];
            if (!(e.tmin < ray.max_t)) continue;   // the reference's test of this node at this moment
            if (e.kind == 2) {
                if (pc) { pc->wide_fetches++; pc->leaf_visits_wide++; }   // the leaf's triangle records
                leaf(tree[e.ref].offset, (uint32_t)tree[e.ref].count);
                continue;
            }
            current = e.ref; have = true;
            break;
        }
        if (!have) break;
    }
}

// Fourth question: does the PRODUCT's packed node buffer (tray_debug_wide_nodes: what the device reads) reproduce the binary
// traversal when it is walked the way the device kernel walks it (hip/wavefront_wide.h: all four slots tested when the node is
// reached, the first hit slot in visiting order entered directly, the later ones pushed with their entry distance and re-tested
// as `tmin < max_t` when popped)? The decode below restates the device's arithmetic: one rounded multiply and one rounded add
// per quantised coordinate (this file is compiled with -ffp-contract=off like the library).
struct PackedWide {
    const uint32_t* words = nullptr;   // all meshes, concatenated
    const uint64_t* mesh_first = nullptr;   // first word of each mesh
    const uint32_t* mesh_root = nullptr;    // wide node of each mesh's root (0xffffffff: the root is a leaf)
    int quantised = 0;
};
inline float packed_f(uint32_t w) { float f; std::memcpy(&f, &w, sizeof f); return f; }
inline float packed_dequant(float lo, uint32_t word, int s, float scale) {
    const float step = (float)((word >> (8 * s)) & 0xffu) * scale;
    return lo + step;
}

template <class LeafFn>
inline void packed_traverse(const TrayBvhNode* tree, const uint32_t* words, int quantised, uint32_t root, Ray& ray, LeafFn&& leaf, ProtoCounters* pc) {
    Vec3 inv_dir(1.0f / ray.d.x, 1.0f / ray.d.y, 1.0f / ray.d.z);
    int neg_dir[3] = {ray.d.x < 0.0f, ray.d.y < 0.0f, ray.d.z < 0.0f};
    if (pc) pc->wide_fetches++;   // the root record
    if (!bbox_fast_intersect(tree[0], ray, inv_dir, neg_dir)) return;
    if (tree[0].count > 0) { if (pc) pc->leaf_visits_wide++; leaf(tree[0].offset, (uint32_t)tree[0].count); return; }
    struct Entry { uint32_t ref; float tmin; };
    Entry stack[192];
    int sp = 0;
    uint32_t current = root;
    auto visit_leaf = [&](uint32_t ref) {
        if (pc) { pc->wide_fetches++; pc->leaf_visits_wide++; }
        leaf(ref & 0xffffffu, (ref >> 24) & 0x1fu);
    };
    for (;;) {
        if (pc) pc->wide_fetches++;
        float bmin[4][3], bmax[4][3];
        uint32_t ref[4], meta;
        if (quantised) {
            const uint32_t* w = words + (size_t)current * 16u;
            const float lo[3] = {packed_f(w[0]), packed_f(w[1]), packed_f(w[2])}, sc[3] = {packed_f(w[3]), packed_f(w[4]), packed_f(w[5])};
            for (int s = 0; s < 4; ++s) {
                ref[s] = w[12 + s];
                for (int k = 0; k < 3; ++k) { bmin[s][k] = packed_dequant(lo[k], w[6 + k], s, sc[k]); bmax[s][k] = packed_dequant(lo[k], w[9 + k], s, sc[k]); }
            }
            meta = (w[3] & 3u) | ((w[4] & 3u) << 2) | ((w[5] & 3u) << 4);
        } else {
            const uint32_t* w = words + (size_t)current * 32u;
            for (int s = 0; s < 4; ++s) {
                ref[s] = w[24 + s];
                for (int k = 0; k < 3; ++k) { bmin[s][k] = packed_f(w[4 * k + s]); bmax[s][k] = packed_f(w[12 + 4 * k + s]); }
            }
            meta = w[28];
        }
        bool hit[4];
        float tmin[4] = {0, 0, 0, 0};
        for (int s = 0; s < 4; ++s) hit[s] = ref[s] != 0xffffffffu && slab_tmin(bmin[s], bmax[s], ray, inv_dir, neg_dir, tmin[s]);
        const bool neg_top = neg_dir[meta & 3u] != 0, neg_l = neg_dir[(meta >> 2) & 3u] != 0, neg_r = neg_dir[(meta >> 4) & 3u] != 0;
        const int near_l = (ref[1] != 0xffffffffu && neg_l) ? 1 : 0, near_r = (ref[3] != 0xffffffffu && neg_r) ? 1 : 0;
        int ord[4];
        if (!neg_top) { ord[0] = near_l; ord[1] = 1 - near_l; ord[2] = 2 + near_r; ord[3] = 3 - near_r; }
        else { ord[0] = 2 + near_r; ord[1] = 3 - near_r; ord[2] = near_l; ord[3] = 1 - near_l; }
        int kfirst = 4;
        for (int k = 3; k >= 0; --k) if (hit[ord[k]]) kfirst = k;
        for (int k = 3; k >= 1; --k) if (hit[ord[k]] && k > kfirst) stack[sp++] = Entry{ref[ord[k]], tmin[ord[k]]};
        bool have = false;
        if (kfirst < 4) {
            const uint32_t r = ref[ord[kfirst]];
            if (r & 0x80000000u) visit_leaf(r);
            else { current = r; have = true; }
        }
        while (!have && sp > 0) {
            const Entry e = stack[--sp];
            if (!(e.tmin < ray.max_t)) continue;
            if (e.ref & 0x80000000u) { visit_leaf(e.ref); continue; }
            current = e.ref; have = true;
        }
        if (!have) break;
    }
}

}  // namespace orc
