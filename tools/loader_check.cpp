// Standalone harness for the host loader (csrc/host/scene.cpp + capi_host.cpp: JSON, OBJ, MERL, textures, BVH builds, flattening), meant to
// be built with the sanitizers -- no HIP involved:
//   g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all -Iinclude tools/loader_check.cpp \
//       tray_rust_amd/csrc/host/scene.cpp tray_rust_amd/csrc/host/capi_host.cpp -pthread -o /tmp/loader_check
//   /tmp/loader_check scene.json...      prints "ok" or "err <code> <message>" per file (every frame 0 flattened); memory errors abort
#include <cstdio>
#include "../include/trayhip.h"

int main(int argc, char** argv) {
    for (int i = 1; i < argc; ++i) {
        TrayHostScene* h = nullptr;
        int rc = tray_scene_load_file(argv[i], &h);
        if (rc == TRAY_OK) {
            const TrayFlatScene* f = nullptr;
            rc = tray_host_scene_flatten(h, 0, &f);
            if (rc == TRAY_OK) { TraySceneInfo info; tray_host_scene_info(h, &info); rc = tray_host_scene_flatten(h, info.frames > 1 ? info.frames - 1 : 0, &f); }
        }
        if (rc == TRAY_OK) std::printf("ok\n");
        else std::printf("err %d %s\n", rc, tray_last_error());
        if (h) tray_host_scene_free(h);
    }
    return 0;
}
