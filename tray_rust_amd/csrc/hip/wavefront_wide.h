// 4-wide traversal of BVH<Triangle> inside the persistent dynamic-fetch kernel (TRAYHIP_WF_WIDE=1; off by default until it has
// been through the whole GPU suite). The binary tree of the reference is collapsed two levels at a time on the host
// (build_wide_nodes, kernels.hip): a wide node holds the boxes of the four grandchildren (a child that is a leaf stays one
// slot), 128 B = one fetch instead of up to three dependent ones. Exactness: every slot is tested when its wide node is
// reached, pushed with its entry distance tmin, and re-tested as `tmin < max_t` when popped -- the only clause of
// BBox::fast_intersect that depends on max_t -- so the accepted candidates and their order are the binary traversal's
// (checked bit for bit on the CPU by the prototype that tests/test_proto_wide_bvh.py exercises; 0.36x the dependent fetches).
// The top level (BVH<Instance>) keeps the two-children step of k_wf_trace_dyn. Stack entries are two words (ref, tmin).
#pragma once

namespace tr {

// BBox::fast_intersect (bbox.rs:75-104) on separate bounds, also returning the entry distance it compares with max_t
TR_DEV bool bbox_hit_tmin(float bminx, float bminy, float bminz, float bmaxx, float bmaxy, float bmaxz, const f3 o, const f3 inv_dir,
                          const bool nx, const bool ny, const bool nz, float min_t, float max_t, float& tmin_out) {
    float tmin = ((nx ? bmaxx : bminx) - o.x) * inv_dir.x;
    float tmax = ((nx ? bminx : bmaxx) - o.x) * inv_dir.x;
    float tymin = ((ny ? bmaxy : bminy) - o.y) * inv_dir.y;
    float tymax = ((ny ? bminy : bmaxy) - o.y) * inv_dir.y;
    if (tmin > tymax || tymin > tmax) return false;
    if (tymin > tmin) tmin = tymin;
    if (tymax < tmax) tmax = tymax;
    float tzmin = ((nz ? bmaxz : bminz) - o.z) * inv_dir.z;
    float tzmax = ((nz ? bminz : bmaxz) - o.z) * inv_dir.z;
    if (tmin > tzmax || tzmin > tmax) return false;
    if (tzmin > tmin) tmin = tzmin;
    if (tzmax < tmax) tmax = tzmax;
    tmin_out = tmin;
    return tmin < max_t && tmax > min_t;
}

// coordinate of slot `s` of a quantised node (host/wide_nodes.hpp: qwide_dequant -- one rounded multiply, one rounded add)
TR_DEV float qwide_dequant(float lo, uint32_t word, uint32_t s, float scale) {
    const float step = (float)((word >> (8u * s)) & 0xffu) * scale;
    return lo + step;
}

template <int STAGE, int ANIM>
__global__ __launch_bounds__(TR_BLOCK, WF_TRACE_WAVES) void k_wf_trace_wide(const DevScene scv, WfPool pool, const uint32_t* __restrict__ queue,
                                                           uint32_t* __restrict__ qctl, DevStats* __restrict__ stats, uint32_t lds_depth,
                                                           uint32_t* __restrict__ overflow) {
    const DevScene& sc = scv;
    TR_DYN_LDS(uint32_t, s_stack);   // stack_depth x TR_BLOCK entries
    uint32_t* __restrict__ stack = s_stack + threadIdx.x;
    // entries past lds_depth live in a per-thread column of `overflow` (HBM): the LDS part is sized for the occupancy the
    // kernel is compiled for, the rarely reached deep levels of the largest meshes must not cost every workgroup its LDS
    const uint32_t ovf_stride = gridDim.x * TR_BLOCK;
    uint32_t* __restrict__ ovf = overflow + (blockIdx.x * TR_BLOCK + threadIdx.x);
#define WF_PUSH(v) do { const uint32_t v_ = (v); if ((uint32_t)sp < lds_depth) stack[sp * TR_BLOCK] = v_; else ovf[(size_t)((uint32_t)sp - lds_depth) * ovf_stride] = v_; ++sp; } while (0)
#define WF_POP() (--sp, (uint32_t)sp < lds_depth ? stack[sp * TR_BLOCK] : ovf[(size_t)((uint32_t)sp - lds_depth) * ovf_stride])
    const uint32_t n = qctl[STAGE];
    uint32_t* __restrict__ cursor = qctl + 3 + STAGE;
    const uint32_t lane = threadIdx.x & 63u;
    const bool any_hit = STAGE == 1;
    bool active = false, exhausted = false;
    uint32_t slot = 0u, n_rays = 0u;
#ifdef WF_TRACE_STATS
    uint32_t c_iter = 0u, c_visit = 0u, c_expand = 0u, c_inst = 0u, c_tri = 0u;
#define WF_COUNT(x) (++(x))
#else
#define WF_COUNT(x) ((void)0)
#endif
    // traversal state (trace_bvh)
    f3 wo = mk(0, 0, 0), wd = mk(0, 0, 0), o = wo, d = wd, inv_dir = wo;
    bool nx = false, ny = false, nz = false, in_mesh = false, any = false;
    float min_t = 0.0f, max_t = 0.0f, time = 0.0f;
    int sp = 0;
    uint32_t node_a = 0u, node_b = 0xffffffffu, cur_inst = 0u, tri_base = 0u, cur_offset = 0u, cur_count = 0u, wnode = 0u;
    enum : uint32_t { TM_NODE = 0u, TM_LEAF = 1u, TM_POP = 2u, TM_WNODE = 3u, WF_NO_NODE = 0xffffffffu };
#ifndef TR_QWIDE
    const float4* __restrict__ wide = reinterpret_cast<const float4*>(sc.wide_nodes);
#endif
    const uint32_t no_tmin = __float_as_uint(-TR_INF);   // entries of the top level carry no entry distance
    uint32_t mode = TM_NODE;
    const TrayBvhNode* __restrict__ tree = sc.top_nodes;
    const TrayTriVerts* __restrict__ tris = nullptr;
    HitRec rec;
    rec.t = 0.0f; rec.inst = 0xffffffffu; rec.prim = 0u; rec.b1 = 0.0f; rec.b2 = 0.0f;
    for (;;) {
        // ---- refill idle lanes from the queue
        if (!exhausted) {
            const unsigned long long idle = __ballot(!active);
            const uint32_t n_idle = (uint32_t)__popcll(idle);
            if (n_idle >= WF_REFILL_MIN || n_idle == (uint32_t)__popcll(__ballot(1))) {
                const uint32_t leader = (uint32_t)__ffsll((long long)idle) - 1u;
                uint32_t base = 0u;
                if (lane == leader) base = atomicAdd(cursor, n_idle);
                base = __shfl(base, (int)leader);
                if (base + n_idle >= n) exhausted = true;
                if (!active) {
                    const uint32_t q = base + (uint32_t)__popcll(idle & ((1ull << lane) - 1ull));
                    if (q < n) {
                        slot = queue[q];
                        if (STAGE == 0) {
                            wo = ld3(pool, F_O, slot); wd = ld3(pool, F_D, slot);
                            min_t = pu(pool, F_BOUNCE, slot) == 0u ? 0.0f : 0.001f; max_t = TR_INF;
                        } else {
                            wo = ld3(pool, F_P, slot); wd = ld3(pool, F_AUX, slot);
                            min_t = 0.001f; max_t = STAGE == 1 ? 0.999f : TR_INF;
                        }
                        if (ANIM) time = pf(pool, F_TIME, slot);
                        o = wo; d = wd;
                        inv_dir = mk(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
                        nx = d.x < 0.0f; ny = d.y < 0.0f; nz = d.z < 0.0f;
                        tree = sc.top_nodes; node_a = 0u; node_b = WF_NO_NODE; sp = 0; in_mesh = false; any = false; mode = TM_NODE;
                        rec.t = 0.0f; rec.inst = 0xffffffffu; rec.prim = 0u; rec.b1 = 0.0f; rec.b2 = 0.0f;
                        active = true;
                        ++n_rays;
                    }
                }
            }
        }
        if (!__any(active)) break;
        // ---- traversal, while-while form. A lane is in one of three modes:
        //   TM_NODE  has one or two nodes to test: a popped node (the box test the reference runs when it reaches a node,
        //            with the ray's current max_t) or BOTH children of a node whose box was hit. Fetching the children
        //            together means a child whose box is missed never costs a dependent fetch of its own; the far child is
        //            pushed only if its box is hit now (a box missed with the current max_t is missed with any later, smaller
        //            one) and is re-tested when popped, so the accepted candidates and their order are exactly the
        //            reference's (bvh.rs:89-127).
        //   TM_LEAF  reached a leaf: triangles of a BVH<Triangle> leaf, or the instances of a BVH<Instance> leaf
        //   TM_POP   needs the next stack entry (instance entry, primitive tests, leaving a mesh)
        // The node phase repeats while enough lanes have node work, so the (much longer) leaf / pop code runs once per
        // several node steps instead of once per step for whichever few lanes happen to need it.
        bool finished = false;
#pragma nounroll
        for (int it = 0; it < WF_NODE_STEPS; ++it) {
            const bool in_node = active && (mode == TM_NODE || mode == TM_WNODE);
            const uint32_t n_node = (uint32_t)__popcll(__ballot(in_node));
            if (n_node == 0u) break;
            if (it > 0 && n_node < WF_NODE_MIN && __any(active && mode != TM_NODE && mode != TM_WNODE)) break;
            if (in_node && mode == TM_WNODE) {
                // 4-wide node of a BVH<Triangle> (collapse of two binary levels, wavefront_wide.h): one 128-B fetch, four slab tests
                WF_COUNT(c_iter); WF_COUNT(c_expand);
                float t0, t1, t2, t3;
#ifdef TR_QWIDE   // 64-B node, 8-bit slot boxes rounded outwards (host/wide_nodes.hpp): the dequantised box contains the exact one
                const uint4* q = reinterpret_cast<const uint4*>(sc.wide_nodes) + (size_t)wnode * 4u;
                const uint4 w0 = q[0], w1 = q[1], w2 = q[2], w3 = q[3];
                const float lox = __uint_as_float(w0.x), loy = __uint_as_float(w0.y), loz = __uint_as_float(w0.z);
                const float qsx = __uint_as_float(w0.w), qsy = __uint_as_float(w1.x), qsz = __uint_as_float(w1.y);   // (axis bits included: the host quantised with exactly these)
                const uint32_t r0 = w3.x, r1 = w3.y, r2 = w3.z, r3 = w3.w;
                const uint32_t meta = (w0.w & 3u) | ((w1.x & 3u) << 2) | ((w1.y & 3u) << 4);
#define QW_HIT(S, T) bbox_hit_tmin(qwide_dequant(lox, w1.z, S, qsx), qwide_dequant(loy, w1.w, S, qsy), qwide_dequant(loz, w2.x, S, qsz), \
                                   qwide_dequant(lox, w2.y, S, qsx), qwide_dequant(loy, w2.z, S, qsy), qwide_dequant(loz, w2.w, S, qsz), \
                                   o, inv_dir, nx, ny, nz, min_t, max_t, T)
                const bool h0 = r0 != 0xffffffffu && QW_HIT(0u, t0);
                const bool h1 = r1 != 0xffffffffu && QW_HIT(1u, t1);
                const bool h2 = r2 != 0xffffffffu && QW_HIT(2u, t2);
                const bool h3 = r3 != 0xffffffffu && QW_HIT(3u, t3);
#undef QW_HIT
#else
                const float4* q = wide + (size_t)wnode * 8u;
                const float4 mnx = q[0], mny = q[1], mnz = q[2], mxx = q[3], mxy = q[4], mxz = q[5], rf = q[6], mt = q[7];
                const uint32_t r0 = __float_as_uint(rf.x), r1 = __float_as_uint(rf.y), r2 = __float_as_uint(rf.z), r3 = __float_as_uint(rf.w);
                const uint32_t meta = __float_as_uint(mt.x);
                const bool h0 = r0 != 0xffffffffu && bbox_hit_tmin(mnx.x, mny.x, mnz.x, mxx.x, mxy.x, mxz.x, o, inv_dir, nx, ny, nz, min_t, max_t, t0);
                const bool h1 = r1 != 0xffffffffu && bbox_hit_tmin(mnx.y, mny.y, mnz.y, mxx.y, mxy.y, mxz.y, o, inv_dir, nx, ny, nz, min_t, max_t, t1);
                const bool h2 = r2 != 0xffffffffu && bbox_hit_tmin(mnx.z, mny.z, mnz.z, mxx.z, mxy.z, mxz.z, o, inv_dir, nx, ny, nz, min_t, max_t, t2);
                const bool h3 = r3 != 0xffffffffu && bbox_hit_tmin(mnx.w, mny.w, mnz.w, mxx.w, mxy.w, mxz.w, o, inv_dir, nx, ny, nz, min_t, max_t, t3);
#endif
                // reference visiting order of the slots: near child of the collapsed node first, inside a child its near child first
                const uint32_t ax_top = meta & 3u, ax_l = (meta >> 2) & 3u, ax_r = (meta >> 4) & 3u;
                const bool neg_top = ax_top == 0u ? nx : (ax_top == 1u ? ny : nz);
                const bool neg_l = ax_l == 0u ? nx : (ax_l == 1u ? ny : nz);
                const bool neg_r = ax_r == 0u ? nx : (ax_r == 1u ? ny : nz);
                const uint32_t near_l = (r1 != 0xffffffffu && neg_l) ? 1u : 0u, near_r = (r3 != 0xffffffffu && neg_r) ? 1u : 0u;
                uint32_t ord[4];
                if (!neg_top) { ord[0] = near_l; ord[1] = 1u - near_l; ord[2] = 2u + near_r; ord[3] = 3u - near_r; }
                else { ord[0] = 2u + near_r; ord[1] = 3u - near_r; ord[2] = near_l; ord[3] = 1u - near_l; }
                // the first hit slot in visiting order is entered directly (the reference tests it right here, with this max_t);
                // the later ones are pushed in reverse visiting order with their entry distance
                int kfirst = 4;
#pragma unroll
                for (int kk = 3; kk >= 0; --kk) {
                    const uint32_t s = ord[kk];
                    if (s == 0u ? h0 : (s == 1u ? h1 : (s == 2u ? h2 : h3))) kfirst = kk;
                }
#pragma unroll
                for (int kk = 3; kk >= 1; --kk) {
                    const uint32_t s = ord[kk];
                    const bool hs = s == 0u ? h0 : (s == 1u ? h1 : (s == 2u ? h2 : h3));
                    if (hs && kk > kfirst) {
                        const uint32_t rs = s == 0u ? r0 : (s == 1u ? r1 : (s == 2u ? r2 : r3));
                        const float ts = s == 0u ? t0 : (s == 1u ? t1 : (s == 2u ? t2 : t3));
                        WF_PUSH(rs); WF_PUSH(__float_as_uint(ts));
                    }
                }
                if (kfirst < 4) {
                    const uint32_t s = kfirst == 0 ? ord[0] : (kfirst == 1 ? ord[1] : (kfirst == 2 ? ord[2] : ord[3]));
                    const uint32_t rs = s == 0u ? r0 : (s == 1u ? r1 : (s == 2u ? r2 : r3));
                    if (rs & 0x80000000u) { cur_offset = rs & 0xffffffu; cur_count = (rs >> 24) & 0x1fu; mode = TM_LEAF; }
                    else wnode = rs;   // stays TM_WNODE
                } else {
                    mode = TM_POP;
                }
            } else if (in_node) {
                WF_COUNT(c_iter);
                const bool two = node_b != WF_NO_NODE;
                const float4* qa = reinterpret_cast<const float4*>(tree + node_a);
                const float4* qb = reinterpret_cast<const float4*>(tree + (two ? node_b : node_a));
                const float4 alo = qa[0], ahi = qa[1], blo = qb[0], bhi = qb[1];
                const bool ha = bbox_hit(alo, ahi, o, inv_dir, nx, ny, nz, min_t, max_t);
                const bool hb = two && bbox_hit(blo, bhi, o, inv_dir, nx, ny, nz, min_t, max_t);
                if (two) WF_COUNT(c_expand); else WF_COUNT(c_visit);
                if (ha || hb) {
                    if (ha && hb) { WF_PUSH(node_b); WF_PUSH(no_tmin); }
                    const uint32_t cur = ha ? node_a : node_b;
                    cur_offset = __float_as_uint(ha ? ahi.z : bhi.z);
                    const uint32_t meta = __float_as_uint(ha ? ahi.w : bhi.w);
                    cur_count = meta & 0xffffu;
                    if (cur_count == 0u) {   // interior: near child first by the sign of the split axis (bvh.rs:105-119)
                        const uint32_t axis = (meta >> 16) & 0xffu;
                        const bool neg = axis == 0u ? nx : (axis == 1u ? ny : nz);
                        node_a = neg ? cur_offset : cur + 1u;
                        node_b = neg ? cur + 1u : cur_offset;
                    } else {
                        mode = TM_LEAF;
                    }
                } else {
                    mode = TM_POP;
                }
            }
        }
        if (active && mode == TM_LEAF) {
            if (in_mesh) {   // BVH<Triangle> leaf (<= 16 triangles), tested in order
                for (uint32_t k = 0; k < cur_count; ++k) {
                    float t, bb1, bb2;
                    WF_COUNT(c_tri);
                    if (triangle_test(tris + cur_offset + k, o, d, min_t, max_t, t, bb1, bb2)) {
                        max_t = t;
                        rec.t = t; rec.inst = cur_inst; rec.prim = tri_base + cur_offset + k; rec.b1 = bb1; rec.b2 = bb2;
                        any = true;
                        if (any_hit) { finished = true; break; }
                    }
                }
            } else {   // BVH<Instance> leaf (<= 4 instances): queue them so that pops come in leaf order
                for (uint32_t k = cur_count; k > 0u; --k) {
                    WF_PUSH(STK_INSTANCE | (cur_offset + k - 1u)); WF_PUSH(no_tmin);
                }
            }
            mode = TM_POP;
        }
        if (active && mode == TM_POP && !finished) {
            bool have_node = false;
            while (sp > 0) {
                const float e_tmin = __uint_as_float(WF_POP());
                uint32_t e = WF_POP();
                if (in_mesh && e != STK_EXIT_MESH) {   // slot of a 4-wide node: the reference's box test of this node at this moment
                    if (!(e_tmin < max_t)) continue;
                    if (e & 0x80000000u) { cur_offset = e & 0xffffffu; cur_count = (e >> 24) & 0x1fu; mode = TM_LEAF; }
                    else { wnode = e; mode = TM_WNODE; }
                    have_node = true;
                    break;
                }
                uint32_t kind = e & STK_KIND_MASK;
                if (kind == STK_NODE) { node_a = e; node_b = WF_NO_NODE; mode = TM_NODE; have_node = true; break; }
                if (kind == STK_EXIT_MESH) {   // back to world space and the top-level tree
                    in_mesh = false;
                    tree = sc.top_nodes;
                    o = wo; d = wd;
                    inv_dir = mk(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
                    nx = d.x < 0.0f; ny = d.y < 0.0f; nz = d.z < 0.0f;
                    continue;
                }
                // Instance::intersect (receiver.rs:29-35): world ray -> object ray by `inv`, direction not renormalised
                uint32_t i = sc.top_order[e & ~STK_KIND_MASK];
                const TrayInstance* __restrict__ in = sc.instances + i;
                if (in->kind == TRAY_INST_POINT_EMITTER) continue;   // emitter.rs:120
                WF_COUNT(c_inst);
                f3 lo_, ld;
                if (ANIM && in->animated) {   // the path's transform of a moving instance, from the per-slot cache
                    float x[24];
                    instance_inv_at<ANIM>(sc, in, time, slot, x);
                    lo_ = xf_point_affine(x + 12, wo);
                    ld = xf_vector(x + 12, wd);
                } else {
                    lo_ = xf_point(in->inv, wo);
                    ld = xf_vector(in->inv, wd);
                }
                uint32_t gt = in->geom_type;
                if (gt == TRAY_GEOM_MESH) {
                    const TrayMesh m = sc.meshes[in->mesh_id];
                    WF_PUSH(STK_EXIT_MESH); WF_PUSH(no_tmin);
                    in_mesh = true;
                    cur_inst = i;
                    tris = sc.tri_verts + m.tri_offset;
                    tri_base = m.tri_offset;
                    o = lo_; d = ld;
                    inv_dir = mk(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
                    nx = d.x < 0.0f; ny = d.y < 0.0f; nz = d.z < 0.0f;
                    // the mesh's root: the reference's first box test inside BVH<Triangle>::intersect
                    const float4* rq = reinterpret_cast<const float4*>(sc.mesh_nodes + m.node_offset);
                    const float4 rlo = rq[0], rhi = rq[1];
                    if (!bbox_hit(rlo, rhi, o, inv_dir, nx, ny, nz, min_t, max_t)) continue;   // next pop is the exit-mesh sentinel
                    const uint32_t rmeta = __float_as_uint(rhi.w);
                    if ((rmeta & 0xffffu) != 0u) { cur_offset = __float_as_uint(rhi.z); cur_count = rmeta & 0xffffu; mode = TM_LEAF; }
                    else { wnode = sc.mesh_wide_root[in->mesh_id]; mode = TM_WNODE; }
                    have_node = true;
                    break;
                }
                float t;
                bool hit;
                if (gt == TRAY_GEOM_RECT) hit = rect_test(in->geom_params[0], in->geom_params[1], lo_, ld, min_t, max_t, t);
                else if (gt == TRAY_GEOM_SPHERE) hit = sphere_test(in->geom_params[0], lo_, ld, min_t, max_t, t);
                else hit = disk_test(in->geom_params[0], in->geom_params[1], lo_, ld, min_t, max_t, t);
                if (hit) {
                    max_t = t;
                    rec.t = t; rec.inst = i; rec.prim = 0u; rec.b1 = 0.0f; rec.b2 = 0.0f;
                    any = true;
                    if (any_hit) { finished = true; break; }
                }
            }
            if (!have_node) finished = true;
        }
        if (finished) {   // write the result to the ray's own slot
            uint32_t flags = pu(pool, F_FLAGS, slot);
            if (STAGE == 1) {
                flags = any ? (flags | WF_OCCLUDED) : (flags & ~WF_OCCLUDED);
            } else {
                const uint32_t bit = STAGE == 0 ? WF_HIT_A : WF_HIT_C;
                flags = any ? (flags | bit) : (flags & ~bit);
                if (any) {
                    pf(pool, F_REC_T, slot) = rec.t; pu(pool, F_REC_INST, slot) = rec.inst; pu(pool, F_REC_PRIM, slot) = rec.prim;
                    pf(pool, F_REC_B1, slot) = rec.b1; pf(pool, F_REC_B2, slot) = rec.b2;
                }
            }
            pu(pool, F_FLAGS, slot) = flags;
            active = false;
        }
    }
    // one counter update per wave
    for (int off = 32; off > 0; off >>= 1) n_rays += __shfl_down(n_rays, off);
    if (lane == 0u && n_rays) atomicAdd(&stats[blockIdx.x & (WF_STAT_SLOTS - 1)].rays, (unsigned long long)n_rays);
#ifdef WF_TRACE_STATS
    uint32_t cs[6] = {c_iter, c_visit, c_expand, c_inst, c_tri, 0u};
    for (int k = 0; k < 5; ++k) {
        uint32_t v = cs[k];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
        if (lane == 0u && v) atomicAdd(&stats[blockIdx.x & (WF_STAT_SLOTS - 1)].trav[STAGE * 6 + k], (unsigned long long)v);
    }
    if (lane == 0u && n_rays) atomicAdd(&stats[blockIdx.x & (WF_STAT_SLOTS - 1)].trav[STAGE * 6 + 5], (unsigned long long)n_rays);
#endif
#undef WF_COUNT
#undef WF_PUSH
#undef WF_POP
}

}  // namespace tr
