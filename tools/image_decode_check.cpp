// Standalone harness for the loader's image decoders (csrc/host/image.hpp), meant to be built with the sanitizers:
//   g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all tools/image_decode_check.cpp -o /tmp/image_decode_check
//   /tmp/image_decode_check file...      prints "ok WxH" or "err <message>" per file; any memory error aborts
// (tools/fuzz_images.py --asan drives it over mutated files.)
#include <cstdio>
#include <string>
#include "../tray_rust_amd/csrc/host/image.hpp"

int main(int argc, char** argv) {
    for (int i = 1; i < argc; ++i) {
        trayh::ImageRGBA8 img;
        std::string err;
        if (trayh::load_image(argv[i], img, err)) std::printf("ok %ux%u\n", img.width, img.height);
        else std::printf("err %s\n", err.c_str());
    }
    return 0;
}
