// Host-only C-ABI entry points: error channel, Morton tile queue, spp rounding, sRGB8 resolve.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "../../../include/trayhip.h"

namespace trayh {

static thread_local std::string g_error;
void set_error(const std::string& msg) { g_error = msg; }

// sampler/morton.rs:5-20
static uint32_t part1_by1(uint32_t x) {
    x &= 0x0000ffffu;
    x = (x ^ (x << 8)) & 0x00ff00ffu;
    x = (x ^ (x << 4)) & 0x0f0f0f0fu;
    x = (x ^ (x << 2)) & 0x33333333u;
    return (x ^ (x << 1)) & 0x55555555u;
}
static uint32_t morton2(uint32_t x, uint32_t y) { return (part1_by1(y) << 1) + part1_by1(x); }

}  // namespace trayh

using namespace trayh;

extern "C" {

const char* tray_last_error(void) { return g_error.c_str(); }
const char* tray_version(void) { return "trayhip 0.4 (abi 4, gfx950)"; }

uint32_t tray_abi_sizeof(const char* name) {
    if (!name) return 0;
    const std::string n(name);
#define TRAY_SZ(T) if (n == #T) return (uint32_t)sizeof(T);
    TRAY_SZ(TrayBvhNode) TRAY_SZ(TrayTriVerts) TRAY_SZ(TrayTriAttrs) TRAY_SZ(TrayMesh) TRAY_SZ(TrayInstance) TRAY_SZ(TrayKeyframe)
    TRAY_SZ(TrayXformLevel) TRAY_SZ(TrayColorKey) TRAY_SZ(TrayMaterial) TRAY_SZ(TrayMerlTable) TRAY_SZ(TrayCamera) TRAY_SZ(TrayFilm)
    TRAY_SZ(TrayFlatScene) TRAY_SZ(TrayMeshKeys) TRAY_SZ(TrayTexture) TRAY_SZ(TrayTexFrame) TRAY_SZ(TraySceneInfo) TRAY_SZ(TrayKernelTiming) TRAY_SZ(TrayScheduleInfo) TRAY_SZ(TrayRay) TRAY_SZ(TrayHit)
#undef TRAY_SZ
    return 0;
}

int tray_block_queue(uint32_t width, uint32_t height, uint32_t select_start, uint32_t select_count,
                     uint32_t* xy, uint32_t cap, uint32_t* n_out) {
    if (!n_out) { set_error("tray_block_queue: null n_out"); return TRAY_E_INVALID; }
    if (width == 0 || height == 0 || width % 8 != 0 || height % 8 != 0) {   // block_queue.rs:29-31
        set_error("Image with dimension (" + std::to_string(width) + ", " + std::to_string(height) + ") not evenly divided by blocks of (8, 8)");
        return TRAY_E_INVALID;
    }
    uint32_t nx = width / 8, ny = height / 8;
    std::vector<std::pair<uint32_t, uint32_t>> blocks(size_t(nx) * ny);
    for (uint32_t i = 0; i < nx * ny; ++i) blocks[i] = {i % nx, i / nx};
    std::stable_sort(blocks.begin(), blocks.end(), [](const std::pair<uint32_t, uint32_t>& a, const std::pair<uint32_t, uint32_t>& b) {
        return morton2(a.first, a.second) < morton2(b.first, b.second);
    });
    size_t first = 0, count = blocks.size();
    if (select_count > 0) {   // skip(start).take(count), block_queue.rs:39-41
        first = std::min<size_t>(select_start, blocks.size());
        count = std::min<size_t>(select_count, blocks.size() - first);
    }
    *n_out = (uint32_t)count;
    if (xy) {
        if (cap < count) { set_error("tray_block_queue: output capacity too small"); return TRAY_E_INVALID; }
        for (size_t i = 0; i < count; ++i) { xy[2 * i] = blocks[first + i].first; xy[2 * i + 1] = blocks[first + i].second; }
    }
    return TRAY_OK;
}

int tray_shard_tiles(uint32_t n_tiles, uint32_t shard, uint32_t n_shards, uint32_t chunk_tiles, uint32_t* out, uint32_t cap, uint32_t* n_out) {
    if (!n_out || n_shards == 0 || shard >= n_shards || chunk_tiles == 0) { set_error("tray_shard_tiles: bad arguments"); return TRAY_E_INVALID; }
    uint32_t n = 0;
    for (uint32_t c = shard; (uint64_t)c * chunk_tiles < n_tiles; c += n_shards)
        for (uint32_t k = 0; k < chunk_tiles && (uint64_t)c * chunk_tiles + k < n_tiles; ++k) {
            if (out) {
                if (n >= cap) { set_error("tray_shard_tiles: output capacity too small"); return TRAY_E_INVALID; }
                out[n] = c * chunk_tiles + k;
            }
            ++n;
        }
    *n_out = n;
    return TRAY_OK;
}

uint32_t tray_round_spp(uint32_t spp) {   // ld.rs:22-25 (usize::next_power_of_two; 0 -> 1)
    uint32_t p = 1;
    while (p < spp && p < 0x80000000u) p <<= 1;
    return p;
}

uint32_t tray_adaptive_step(uint32_t min_spp, uint32_t max_spp) {   // adaptive.rs:36-48
    const uint32_t lo = tray_round_spp(min_spp), hi = tray_round_spp(max_spp);
    return tray_round_spp(hi > lo ? (hi - lo) / 5u : 0u);
}

int tray_resolve_srgb8(const float* rgbw, uint32_t width, uint32_t height, uint8_t* rgb8) {   // render_target.rs:185-210, color.rs:36-71
    if (!rgbw || !rgb8) { set_error("tray_resolve_srgb8: null argument"); return TRAY_E_INVALID; }
    const size_t n = size_t(width) * height;
    std::memset(rgb8, 0, n * 3);
    for (size_t i = 0; i < n; ++i) {
        const float* c = rgbw + 4 * i;
        if (!(c[3] > 0.0f)) continue;
        for (int k = 0; k < 3; ++k) {
            float v = c[k] / c[3];
            v = v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v);
            float s = v <= 0.0031308f ? 12.92f * v : (1.0f + 0.055f) * std::pow(v, 1.0f / 2.4f) - 0.055f;
            float b = s * 255.0f;
            rgb8[3 * i + k] = b != b ? 0 : (b <= 0.0f ? 0 : (b >= 255.0f ? 255 : (uint8_t)b));   // saturating `as u8`
        }
    }
    return TRAY_OK;
}

}  // extern "C"
