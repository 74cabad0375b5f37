// SAH BVH2 builder producing the flattened node array the kernels traverse.
// Follows BVH::build / build_leaf / flatten_tree (src/geometry/bvh.rs:139-268): 12 SAH buckets,
// cost 0.125 + (nL*A_L + nR*A_R)/A, median split below 5 primitives, depth-first flattening with
// the first child adjacent to its parent. The node order matters for parity only through exact-t
// ties, but matching the reference's tree keeps even those identical.
#pragma once
#include <algorithm>
#include <future>
#include <memory>
#include <stdexcept>
#include <vector>

#include "../../../include/trayhip.h"
#include "linalg.hpp"

namespace trayh {

struct BvhBuild {
    std::vector<TrayBvhNode> nodes;
    std::vector<uint32_t> ordered;   // ordered_geom (bvh.rs:19-21)
};

namespace detail {

struct GeomInfo {   // bvh.rs:299-317
    uint32_t idx;
    V3 center;
    BBox bounds;
};

struct BuildNode {
    BBox bounds;
    std::unique_ptr<BuildNode> child[2];
    int axis = 0;
    uint32_t ngeom = 0, geom_offset = 0;
};

// partition.rs:9-39 (std::partition for a double ended iterator): returns index of the first
// element of the false group
template <class Pred>
inline size_t partition_ref(GeomInfo* a, size_t n, Pred pred) {
    size_t split = 0, lo = 0, hi = n;   // [lo, hi) is the unconsumed middle
    for (;;) {
        GeomInfo* front = nullptr;
        GeomInfo* back = nullptr;
        while (lo < hi) {
            GeomInfo* f = &a[lo++];
            if (!pred(*f)) { front = f; break; }
            ++split;
        }
        while (lo < hi) {
            GeomInfo* b = &a[--hi];
            if (pred(*b)) { back = b; break; }
        }
        if (front && back) { std::swap(*front, *back); ++split; }
        else break;
    }
    return split;
}

// A leaf covers a contiguous range of the (partitioned in place) info array, and the reference appends leaves to
// ordered_geom in depth-first order, i.e. in the order of those ranges: geom_offset is the range's start, ordered_geom
// is the final info order. That makes sibling subtrees independent, so large ones are built on separate threads.
inline std::unique_ptr<BuildNode> make_leaf(GeomInfo* info, size_t n, const GeomInfo* base, const BBox& bounds) {
    auto node = std::make_unique<BuildNode>();
    node->bounds = bounds;
    node->geom_offset = (uint32_t)(info - base);
    node->ngeom = (uint32_t)n;
    return node;
}

inline std::unique_ptr<BuildNode> make_interior(std::unique_ptr<BuildNode> l, std::unique_ptr<BuildNode> r, int axis) {
    auto node = std::make_unique<BuildNode>();
    node->bounds = l->bounds.box_union(r->bounds);
    node->axis = axis;
    node->child[0] = std::move(l);
    node->child[1] = std::move(r);
    return node;
}

inline std::unique_ptr<BuildNode> build(GeomInfo* info, size_t ngeom, const GeomInfo* base, size_t max_geom, int par_levels);

// build both halves of a split; the left one on its own thread while the range is large and levels remain
inline std::unique_ptr<BuildNode> build_children(GeomInfo* info, size_t ngeom, size_t mid, const GeomInfo* base, size_t max_geom, int par_levels, int axis) {
    std::unique_ptr<BuildNode> l, r;
    if (par_levels > 0 && ngeom >= 65536) {
        auto fl = std::async(std::launch::async, [=] { return build(info, mid, base, max_geom, par_levels - 1); });
        r = build(info + mid, ngeom - mid, base, max_geom, par_levels - 1);
        l = fl.get();
    } else {
        l = build(info, mid, base, max_geom, 0);
        r = build(info + mid, ngeom - mid, base, max_geom, 0);
    }
    return make_interior(std::move(l), std::move(r), axis);
}

inline std::unique_ptr<BuildNode> build(GeomInfo* info, size_t ngeom, const GeomInfo* base, size_t max_geom, int par_levels) {
    BBox bounds;
    for (size_t i = 0; i < ngeom; ++i) bounds = bounds.box_union(info[i].bounds);
    if (ngeom == 1) return make_leaf(info, ngeom, base, bounds);
    BBox centroids;
    for (size_t i = 0; i < ngeom; ++i) centroids = centroids.point_union(info[i].center);
    int axis = centroids.max_extent();
    size_t mid = ngeom / 2;
    if (std::fabs(centroids.mx[axis] - centroids.mn[axis]) < kEps) {   // bvh.rs:155-165
        if (ngeom < max_geom) return make_leaf(info, ngeom, base, bounds);
        return build_children(info, ngeom, mid, base, max_geom, par_levels, axis);
    }
    if (ngeom < 5) {
        // slice::sort_by is a stable merge sort
        std::stable_sort(info, info + ngeom, [axis](const GeomInfo& a, const GeomInfo& b) { return a.center[axis] < b.center[axis]; });
    } else {
        const int NB = 12;
        struct Bucket { size_t count = 0; BBox bounds; } buckets[NB];
        auto bucket_of = [&](const GeomInfo& g) {
            // `as usize` saturates: negative / NaN -> 0
            float f = (g.center[axis] - centroids.mn[axis]) / (centroids.mx[axis] - centroids.mn[axis]) * (float)NB;
            int b = f > 0.0f ? (f >= 4294967296.0f ? NB : (int)(long long)f) : 0;
            if (b >= NB) b = NB - 1;   // reference only remaps b == NB; larger values cannot occur
            return b;
        };
        for (size_t i = 0; i < ngeom; ++i) {
            int b = bucket_of(info[i]);
            buckets[b].count += 1;
            buckets[b].bounds = buckets[b].bounds.box_union(info[i].bounds);
        }
        float cost[NB - 1];
        for (int i = 0; i < NB - 1; ++i) {
            Bucket left, right;
            for (int k = 0; k <= i; ++k) { left.bounds = left.bounds.box_union(buckets[k].bounds); left.count += buckets[k].count; }
            for (int k = i + 1; k < NB; ++k) { right.bounds = right.bounds.box_union(buckets[k].bounds); right.count += buckets[k].count; }
            cost[i] = 0.125f + ((float)left.count * left.bounds.surface_area() + (float)right.count * right.bounds.surface_area())
                                   / bounds.surface_area();
        }
        int min_bucket = 0;
        float min_cost = INFINITY;
        for (int i = 0; i < NB - 1; ++i)
            if (cost[i] < min_cost) { min_bucket = i; min_cost = cost[i]; }
        if (ngeom > max_geom || min_cost < (float)ngeom) {
            mid = partition_ref(info, ngeom, [&](const GeomInfo& g) { return bucket_of(g) <= min_bucket; });
        } else {
            return make_leaf(info, ngeom, base, bounds);
        }
    }
    if (mid == 0 || mid == ngeom) throw std::runtime_error("BVH build: degenerate split (reference asserts mid != 0 && mid != len)");
    return build_children(info, ngeom, mid, base, max_geom, par_levels, axis);
}

inline uint32_t flatten(const BuildNode& n, std::vector<TrayBvhNode>& out) {
    uint32_t offset = (uint32_t)out.size();
    TrayBvhNode f{};
    for (int i = 0; i < 3; ++i) { f.bmin[i] = n.bounds.mn[i]; f.bmax[i] = n.bounds.mx[i]; }
    if (n.child[0]) {
        f.count = 0;
        f.axis = (uint8_t)n.axis;
        out.push_back(f);
        flatten(*n.child[0], out);
        uint32_t second = flatten(*n.child[1], out);
        out[offset].offset = second;
    } else {
        if (n.ngeom > 0xffff) throw std::runtime_error("BVH leaf too large");
        f.count = (uint16_t)n.ngeom;
        f.offset = n.geom_offset;
        out.push_back(f);
    }
    return offset;
}

}  // namespace detail

inline BvhBuild build_bvh(const std::vector<BBox>& bounds, size_t max_geom) {
    if (bounds.empty()) throw std::runtime_error("BVH build: no geometry");
    std::vector<detail::GeomInfo> info(bounds.size());
    for (size_t i = 0; i < bounds.size(); ++i) info[i] = {(uint32_t)i, bounds[i].center(), bounds[i]};
    BvhBuild out;
    auto root = detail::build(info.data(), info.size(), info.data(), max_geom, 4);
    out.ordered.resize(info.size());
    for (size_t i = 0; i < info.size(); ++i) out.ordered[i] = info[i].idx;
    out.nodes.reserve(2 * info.size());
    detail::flatten(*root, out.nodes);
    return out;
}

}  // namespace trayh
