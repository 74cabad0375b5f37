// Structural validation of a TrayFlatScene before anything of it is uploaded: every index the device kernels follow without a
// bounds check (BVH child / leaf ranges, ordered-instance lists, light list, mesh and triangle ranges, MERL tables, spline
// stacks, emission keys) must stay inside its array. tray_host_scene_flatten() always produces valid scenes; this is for callers
// that build or patch the POD themselves through the C ABI. Returns "" when the scene is consistent, else what is wrong.
#pragma once
#include <cstdint>
#include <string>

#include "../../../include/trayhip.h"

namespace tray {

// one flattened BVH2: n nodes, leaves refer to [offset, offset + count) of `n_prims` ordered primitives, at most `max_leaf` each
inline std::string validate_bvh(const TrayBvhNode* nodes, uint64_t n, uint64_t n_prims, uint32_t max_leaf, const char* what) {
    if (n == 0) return "";
    if (!nodes) return std::string(what) + ": null node array";
    for (uint64_t i = 0; i < n; ++i) {
        const TrayBvhNode& nd = nodes[i];
        if (nd.count > 0) {
            if (nd.count > max_leaf || (uint64_t)nd.offset + nd.count > n_prims)
                return std::string(what) + ": leaf " + std::to_string(i) + " refers to primitives outside the ordered list";
        } else {
            // interior: first child is the next node, second child `offset` lies behind the whole first subtree
            if (i + 1 >= n || nd.offset <= i + 1 || nd.offset >= n || nd.axis > 2)
                return std::string(what) + ": interior node " + std::to_string(i) + " has a child outside the node array";
        }
    }
    return "";
}

// the walk validate_flat_scene(f, false) left out
inline std::string validate_mesh_trees(const TrayFlatScene* f) {
    for (uint32_t m = 0; m < f->n_meshes; ++m) {
        const TrayMesh& me = f->meshes[m];
        if (std::string e = validate_bvh(f->mesh_nodes + me.node_offset, me.node_count, me.tri_count, 16u, "BVH<Triangle>"); !e.empty())
            return "mesh " + std::to_string(m) + ": " + e;
    }
    return "";
}

// `walk_mesh_trees = false` leaves out the node-by-node walk of the BVH<Triangle>s (millions of nodes): for a caller that has checked these very
// arrays before (the loader's later frames; a frame update that keeps the device's trees and never reads the new ones).
inline std::string validate_flat_scene(const TrayFlatScene* f, bool walk_mesh_trees = true) {
    auto need = [](const void* p, uint64_t n) { return n == 0 || p != nullptr; };
    if (!need(f->instances, f->n_instances) || !need(f->top_nodes, f->n_top_nodes) || !need(f->top_order, f->n_top_order) ||
        !need(f->meshes, f->n_meshes) || !need(f->mesh_nodes, f->n_mesh_nodes) || !need(f->tri_verts, f->n_tris) ||
        !need(f->tri_attrs, f->n_tris) || !need(f->materials, f->n_materials) || !need(f->merl_tables, f->n_merl) ||
        !need(f->merl_data, f->n_merl_floats) || !need(f->lights, f->n_lights) || !need(f->xf_levels, f->n_xf_levels) ||
        !need(f->keyframes, f->n_keyframes) || !need(f->knots, f->n_knots) || !need(f->color_keys, f->n_color_keys) ||
        !need(f->mesh_keys, f->n_mesh_keys) || !need(f->key_times, f->n_key_times))
        return "an array with a non-zero count is null";
    if (f->n_mesh_keys != 0 && f->n_mesh_keys != f->n_meshes) return "mesh_keys must have one entry per mesh (or none)";
    if (f->n_instances == 0) return "the scene has no instances";
    if (f->min_depth > f->max_depth) return "integrator min_depth > max_depth";
    // BVH<Instance>
    if (f->n_top_nodes == 0) return "the top-level BVH is empty";
    if (std::string e = validate_bvh(f->top_nodes, f->n_top_nodes, f->n_top_order, 4u, "BVH<Instance>"); !e.empty()) return e;
    for (uint32_t i = 0; i < f->n_top_order; ++i)
        if (f->top_order[i] >= f->n_instances) return "top_order refers to a missing instance";
    // meshes and their BVH<Triangle>
    for (uint32_t m = 0; m < f->n_meshes; ++m) {
        const TrayMesh& me = f->meshes[m];
        const uint64_t n_keys = f->n_mesh_keys ? f->mesh_keys[m].n_keys : 1u;   // (keyframe k of the mesh: tri_offset + k * tri_count)
        if (f->n_mesh_keys) {
            const TrayMeshKeys& mk = f->mesh_keys[m];
            if (mk.n_keys == 0) return "mesh " + std::to_string(m) + " has no keyframe";
            if (mk.n_keys > 1) {
                if ((uint64_t)mk.time_first + mk.n_keys > f->n_key_times) return "mesh " + std::to_string(m) + " refers to keyframe times outside key_times";
                for (uint32_t k = 1; k < mk.n_keys; ++k)
                    if (!(f->key_times[mk.time_first + k] > f->key_times[mk.time_first + k - 1])) return "mesh " + std::to_string(m) + ": keyframe times must ascend";
            }
        }
        if ((uint64_t)me.node_offset + me.node_count > f->n_mesh_nodes || (uint64_t)me.tri_offset + n_keys * me.tri_count > f->n_tris)
            return "mesh " + std::to_string(m) + " refers to nodes or triangles outside the arrays";
        if (me.node_count == 0 || me.tri_count == 0) return "mesh " + std::to_string(m) + " is empty";
        if (!walk_mesh_trees) continue;
        if (std::string e = validate_bvh(f->mesh_nodes + me.node_offset, me.node_count, me.tri_count, 16u, "BVH<Triangle>"); !e.empty())
            return "mesh " + std::to_string(m) + ": " + e;
    }
    // materials and MERL tables
    for (uint32_t t = 0; t < f->n_merl; ++t) {
        const TrayMerlTable& mt = f->merl_tables[t];
        const uint64_t floats = (uint64_t)mt.n_theta_h * mt.n_theta_d * mt.n_phi_d * 3u;
        if (mt.n_theta_h == 0 || mt.n_theta_d == 0 || mt.n_phi_d == 0 || mt.offset > f->n_merl_floats || floats > f->n_merl_floats - mt.offset)
            return "MERL table " + std::to_string(t) + " lies outside merl_data";
    }
    if (!need(f->textures, f->n_textures) || !need(f->tex_frames, f->n_tex_frames) || !need(f->tex_data, f->n_tex_bytes)) return "a texture array with a non-zero count is null";
    for (uint32_t t = 0; t < f->n_textures; ++t) {
        const TrayTexture& tx = f->textures[t];
        if (tx.n_frames == 0 || (uint64_t)tx.first_frame + tx.n_frames > f->n_tex_frames) return "texture " + std::to_string(t) + " refers to frames outside tex_frames";
        for (uint32_t k = 0; k < tx.n_frames; ++k) {
            const TrayTexFrame& fr = f->tex_frames[tx.first_frame + k];
            const uint64_t bytes = (uint64_t)fr.width * fr.height * 4u;
            if (fr.width == 0 || fr.height == 0 || (fr.offset & 3u) || fr.offset > f->n_tex_bytes || bytes > f->n_tex_bytes - fr.offset)
                return "texture " + std::to_string(t) + " has a frame outside tex_data";
            if (k > 0 && !(fr.time >= f->tex_frames[tx.first_frame + k - 1].time)) return "texture " + std::to_string(t) + " has keyframe times that do not increase";
        }
    }
    for (uint32_t i = 0; i < f->n_materials; ++i) {
        const TrayMaterial& ma = f->materials[i];
        if (ma.kind > TRAY_MAT_MERL) return "material " + std::to_string(i) + " has an unknown kind";
        if (ma.kind == TRAY_MAT_MERL && ma.table >= f->n_merl) return "material references a missing MERL table";
        for (uint32_t tex : {ma.tex_c0, ma.tex_c1, ma.tex_f0, ma.tex_f1})
            if (tex != TRAY_NO_TEXTURE && tex >= f->n_textures) return "material " + std::to_string(i) + " references a missing texture";
    }
    // instances
    auto stack_in_range = [&](uint32_t first, uint32_t count) { return (uint64_t)first + count <= f->n_xf_levels; };
    for (uint32_t i = 0; i < f->n_instances; ++i) {
        const TrayInstance& in = f->instances[i];
        const std::string who = "instance " + std::to_string(i);
        if (in.kind > TRAY_INST_POINT_EMITTER) return who + " has an unknown kind";
        if (in.kind == TRAY_INST_POINT_EMITTER) {
            if (in.geom_type != TRAY_GEOM_NONE) return who + ": a point emitter has no geometry";
        } else {
            const bool any_mesh = in.geom_type == TRAY_GEOM_MESH || in.geom_type == TRAY_GEOM_ANIMATED_MESH;
            if (in.geom_type > TRAY_GEOM_MESH && in.geom_type != TRAY_GEOM_ANIMATED_MESH) return who + " has an unknown geometry type";
            if (in.material_id >= f->n_materials) return "instance references a missing material";
            if (any_mesh && in.mesh_id >= f->n_meshes) return "instance references a missing mesh";
            if (in.kind == TRAY_INST_AREA_EMITTER && any_mesh) return who + ": area lights are spheres, disks or rectangles (scene.rs:584-654)";
            // a Mesh has one keyframe, an AnimatedMesh at least two (AnimatedMesh::new reads times[1], animated_mesh.rs:125)
            const uint32_t n_keys = any_mesh && f->n_mesh_keys ? f->mesh_keys[in.mesh_id].n_keys : 1u;
            if (in.geom_type == TRAY_GEOM_MESH && n_keys != 1u) return who + ": a mesh instance refers to an animated mesh's triangles";
            if (in.geom_type == TRAY_GEOM_ANIMATED_MESH && n_keys < 2u) return who + ": an animated mesh needs at least two keyframes";
        }
        if (!stack_in_range(in.xf_first, in.xf_count)) return who + " refers to spline levels outside xf_levels";
        if (in.emis_count && (uint64_t)in.emis_first + in.emis_count > f->n_color_keys) return "instance references missing colour keys";
        if (in.kind != TRAY_INST_RECEIVER && (in.light_index >= f->n_lights || f->lights[in.light_index] != i)) return who + " is an emitter that the light list does not hold";
    }
    {   // moving_slot indexes the per-path transform cache (dev_geom.h: instance_xf_at): the slots of the animated instances must be
        // a permutation of 0 .. n_animated-1
        uint32_t n_animated = 0;
        for (uint32_t i = 0; i < f->n_instances; ++i) n_animated += f->instances[i].animated != 0 ? 1u : 0u;
        std::string seen(n_animated, '\0');
        for (uint32_t i = 0; i < f->n_instances; ++i) {
            const TrayInstance& in = f->instances[i];
            if (!in.animated) continue;
            if (in.moving_slot >= n_animated) return "instance " + std::to_string(i) + " is animated but its moving_slot is not below the number of animated instances";
            if (seen[in.moving_slot]) return "instance " + std::to_string(i) + " shares its moving_slot with another animated instance";
            seen[in.moving_slot] = 1;
        }
    }
    if (!stack_in_range(f->camera.xf_first, f->camera.xf_count)) return "the camera refers to spline levels outside xf_levels";
    for (uint32_t l = 0; l < f->n_lights; ++l)
        if (f->lights[l] >= f->n_instances || f->instances[f->lights[l]].kind == TRAY_INST_RECEIVER) return "light " + std::to_string(l) + " is not an emitter instance";
    for (uint32_t l = 0; l < f->n_xf_levels; ++l) {
        const TrayXformLevel& lv = f->xf_levels[l];
        if (lv.kf_count == 0 || (uint64_t)lv.kf_first + lv.kf_count > f->n_keyframes || (uint64_t)lv.knot_first + lv.knot_count > f->n_knots)
            return "spline level " + std::to_string(l) + " refers to keyframes or knots outside the arrays";
        if (lv.kf_count >= 2) {   // spline_point (dev_anim.h) searches the knots: they must be finite and non-decreasing, or the span index wraps
            if (lv.knot_count != lv.kf_count + lv.degree + 1) return "spline level " + std::to_string(l) + " has a knot vector that does not fit its keyframes and degree";
            for (uint32_t k = 0; k < lv.knot_count; ++k) {
                const float kv = f->knots[lv.knot_first + k];
                if (!(kv - kv == 0.0f)) return "spline level " + std::to_string(l) + " has a knot that is not finite";
                if (k > 0 && kv < f->knots[lv.knot_first + k - 1]) return "spline level " + std::to_string(l) + " has decreasing knots";
            }
        }
    }
    return "";
}

}  // namespace tray
