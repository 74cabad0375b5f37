// 4-wide collapse of one BVH<Triangle> for the wavefront traversal (hip/wavefront_wide.h). Host only.
// The wide node of a binary interior node N holds N's grandchildren -- a child that is a leaf stays one slot -- in binary
// order [L.first, L.second, R.first, R.second]; visiting order and exactness argument: wavefront_wide.h.
// Slot reference: WIDE_EMPTY, or bit 31 set = leaf (count << 24 | first triangle of the mesh), else index of the child's wide node.
//
// Two node formats:
//   build_wide_nodes   32 words (128 B): bminx[4] bminy[4] bminz[4] bmaxx[4] bmaxy[4] bmaxz[4] ref[4] meta pad[3], exact boxes
//   build_qwide_nodes  16 words (64 B): slot boxes quantised OUTWARDS to 8 bits per coordinate relative to the union of the
//                      slots. The dequantised box  lo + float(q) * scale  (one rounded multiply, one rounded add, as the device
//                      computes it) contains the exact box -- checked coordinate by coordinate while packing -- so the traversal
//                      visits a superset of the reference's nodes and can only lose nothing.
//        word 0-2  lo.xyz (f32)             word 3   scale.x (f32; low 2 mantissa bits = split axis of N)
//        word 4    scale.y (low 2 bits = split axis of N's first child)    word 5   scale.z (low 2 bits = axis of the second child)
//        word 6-8  qmin x / y / z, one byte per slot (slot s in bits 8s..8s+7)   word 9-11  qmax x / y / z
//        word 12-15 ref[4]
//      The axis bits are part of the scale the boxes are quantised with, so they cost nothing but 2 bits of scale precision.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../../include/trayhip.h"

namespace tray {

constexpr uint32_t WIDE_EMPTY = 0xffffffffu, WIDE_LEAF = 0x80000000u;
constexpr uint32_t WIDE_WORDS = 32u, QWIDE_WORDS = 16u;

struct WideSlots {   // the (up to) four slots of the wide node of binary node n
    uint32_t node[4] = {0u, 0u, 0u, 0u};
    bool used[4] = {false, false, false, false};
    uint32_t axis_top = 0u, axis_child[2] = {0u, 0u};
};
inline WideSlots wide_slots(const TrayBvhNode* tree, uint32_t n) {
    WideSlots w;
    const TrayBvhNode& N = tree[n];
    const uint32_t kids[2] = {n + 1u, N.offset};
    w.axis_top = N.axis;
    for (int c = 0; c < 2; ++c) {
        const TrayBvhNode& K = tree[kids[c]];
        if (K.count > 0) { w.node[2 * c] = kids[c]; w.used[2 * c] = true; }
        else { w.axis_child[c] = K.axis; w.node[2 * c] = kids[c] + 1u; w.node[2 * c + 1] = K.offset; w.used[2 * c] = w.used[2 * c + 1] = true; }
    }
    return w;
}
inline uint32_t wide_leaf_ref(const TrayBvhNode& leaf) { return WIDE_LEAF | ((uint32_t)leaf.count << 24) | leaf.offset; }

inline uint32_t build_wide_nodes(const TrayBvhNode* tree, uint32_t n, std::vector<float>& out) {
    const uint32_t self = (uint32_t)(out.size() / WIDE_WORDS);
    out.resize(out.size() + WIDE_WORDS, 0.0f);
    const WideSlots ws = wide_slots(tree, n);
    for (int sidx = 0; sidx < 4; ++sidx) {
        uint32_t ref = WIDE_EMPTY;
        float bmin[3] = {0, 0, 0}, bmax[3] = {0, 0, 0};
        if (ws.used[sidx]) {
            const TrayBvhNode& S = tree[ws.node[sidx]];
            for (int k = 0; k < 3; ++k) { bmin[k] = S.bmin[k]; bmax[k] = S.bmax[k]; }
            ref = S.count > 0 ? wide_leaf_ref(S) : build_wide_nodes(tree, ws.node[sidx], out);
        }
        float* w = out.data() + (size_t)self * WIDE_WORDS;   // (re-derive: the vector may have grown)
        for (int k = 0; k < 3; ++k) { w[4 * k + sidx] = bmin[k]; w[12 + 4 * k + sidx] = bmax[k]; }
        std::memcpy(w + 24 + sidx, &ref, sizeof ref);
    }
    const uint32_t meta = ws.axis_top | (ws.axis_child[0] << 2) | (ws.axis_child[1] << 4);
    std::memcpy(out.data() + (size_t)self * WIDE_WORDS + 28, &meta, sizeof meta);
    return self;
}

// The device's dequantisation, operation for operation (two roundings; the library is built with -ffp-contract=off)
inline float qwide_dequant(float lo, uint32_t q, float scale) {
    const float step = (float)q * scale;
    return lo + step;
}

// Returns the index of the wide node built for binary node n, or WIDE_EMPTY if some box could not be enclosed (non-finite
// bounds); the caller then keeps the binary traversal.
inline uint32_t build_qwide_nodes(const TrayBvhNode* tree, uint32_t n, std::vector<uint32_t>& out) {
    const uint32_t self = (uint32_t)(out.size() / QWIDE_WORDS);
    out.resize(out.size() + QWIDE_WORDS, 0u);
    const WideSlots ws = wide_slots(tree, n);
    float lo[3], hi[3];
    for (int k = 0; k < 3; ++k) { lo[k] = INFINITY; hi[k] = -INFINITY; }
    for (int s = 0; s < 4; ++s)
        if (ws.used[s])
            for (int k = 0; k < 3; ++k) { lo[k] = std::fmin(lo[k], tree[ws.node[s]].bmin[k]); hi[k] = std::fmax(hi[k], tree[ws.node[s]].bmax[k]); }
    const uint32_t axis_bits[3] = {ws.axis_top & 3u, ws.axis_child[0] & 3u, ws.axis_child[1] & 3u};
    float scale[3];
    uint32_t qmin[3] = {0u, 0u, 0u}, qmax[3] = {0u, 0u, 0u};
    for (int k = 0; k < 3; ++k) {
        if (!std::isfinite(lo[k]) || !std::isfinite(hi[k])) return WIDE_EMPTY;
        // a scale a few ulps above extent / 255, its low two mantissa bits replaced by the axis code
        float s0 = (float)(((double)hi[k] - (double)lo[k]) / 255.0);
        if (!(s0 > 1e-30f)) s0 = 1e-30f;
        uint32_t bits;
        std::memcpy(&bits, &s0, sizeof bits);
        bits = ((bits + 8u) & ~3u) | axis_bits[k];
        std::memcpy(&scale[k], &bits, sizeof bits);
        if (!std::isfinite(scale[k])) return WIDE_EMPTY;
        for (int s = 0; s < 4; ++s) {
            if (!ws.used[s]) continue;
            const TrayBvhNode& S = tree[ws.node[s]];
            double a = std::floor(((double)S.bmin[k] - (double)lo[k]) / (double)scale[k]);
            double b = std::ceil(((double)S.bmax[k] - (double)lo[k]) / (double)scale[k]);
            int qa = (int)std::fmin(255.0, std::fmax(0.0, a)), qb = (int)std::fmin(255.0, std::fmax(0.0, b));
            while (qa > 0 && qwide_dequant(lo[k], (uint32_t)qa, scale[k]) > S.bmin[k]) --qa;     // the roundings of the dequantisation itself
            while (qb < 255 && qwide_dequant(lo[k], (uint32_t)qb, scale[k]) < S.bmax[k]) ++qb;
            if (qwide_dequant(lo[k], (uint32_t)qa, scale[k]) > S.bmin[k] || qwide_dequant(lo[k], (uint32_t)qb, scale[k]) < S.bmax[k]) return WIDE_EMPTY;
            qmin[k] |= (uint32_t)qa << (8 * s);
            qmax[k] |= (uint32_t)qb << (8 * s);
        }
    }
    uint32_t refs[4];
    for (int s = 0; s < 4; ++s) {
        refs[s] = WIDE_EMPTY;
        if (!ws.used[s]) continue;
        const TrayBvhNode& S = tree[ws.node[s]];
        refs[s] = S.count > 0 ? wide_leaf_ref(S) : build_qwide_nodes(tree, ws.node[s], out);
        if (refs[s] == WIDE_EMPTY) return WIDE_EMPTY;
    }
    uint32_t* w = out.data() + (size_t)self * QWIDE_WORDS;   // (derived after the recursion: the vector may have grown)
    std::memcpy(w + 0, lo, 3 * sizeof(float));
    std::memcpy(w + 3, &scale[0], sizeof(float));
    std::memcpy(w + 4, &scale[1], sizeof(float));
    std::memcpy(w + 5, &scale[2], sizeof(float));
    for (int k = 0; k < 3; ++k) { w[6 + k] = qmin[k]; w[9 + k] = qmax[k]; }
    for (int s = 0; s < 4; ++s) w[12 + s] = refs[s];
    return self;
}

}  // namespace tray
