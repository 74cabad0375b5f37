// WebP textures as image 0.18 reads them (scene.rs:317-394 -> image::open -> webp::WebpDecoder -> vp8::VP8Decoder; the crate is a Cargo.lock
// dependency, not vendored by the reference): that version takes the SIMPLE lossy container only ("RIFF" size "WEBP" "VP8 " size, anything else
// -- "VP8L", "VP8X" -- is its "Invalid VP8 signature" error), decodes the key frame's LUMA plane and presents it as a grey image
// (ColorType::Gray(8)): no chroma reconstruction, no loop filter. This file restates that decoder from RFC 6386 (the VP8 data format): frame and
// segment headers (s. 9, 19.2), the boolean entropy decoder (s. 7), key-frame mode parsing (s. 8, 11), DCT token decoding with the default /
// updated coefficient probabilities (s. 13), dequantisation (s. 14.1), the inverse WHT / DCT (s. 14.3, 14.4) and the luma intra predictors with
// the reference decoder's edge conventions (s. 12.2, 12.3: 127 above the first row, 129 left of the first column, above-right of the last
// macroblock of a row = the last pixel above, repeated). Chroma tokens are parsed (they share the partitions) and dropped.
// UNPINNED against the crate itself (no Rust here): checked bit for bit against libwebp's luma plane (WebPDecodeYUV) on files whose header says
// "loop filter level 0", and to the loop filter's reach on the others (tests/test_textures.py). The constant tables are RFC 6386's
// (s. 13.4 / 13.5 coefficient probabilities, s. 11.5 subblock mode probabilities in the mode order DC, TM, VE, HE, RD, VR, LD, VL, HD, HU,
// s. 14.1 quantiser look-ups).
#pragma once
#include <algorithm>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

namespace trayh {
namespace vp8_detail {

static const uint8_t VP8_COEFF_PROBS[1056] = {
    128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128,
    128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128,
    128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128,
    253, 136, 254, 255, 228, 219, 128, 128, 128, 128, 128,
    189, 129, 242, 255, 227, 213, 255, 219, 128, 128, 128,
    106, 126, 227, 252, 214, 209, 255, 255, 128, 128, 128,
    1, 98, 248, 255, 236, 226, 255, 255, 128, 128, 128,
    181, 133, 238, 254, 221, 234, 255, 154, 128, 128, 128,
    78, 134, 202, 247, 198, 180, 255, 219, 128, 128, 128,
    1, 185, 249, 255, 243, 255, 128, 128, 128, 128, 128,
    184, 150, 247, 255, 236, 224, 128, 128, 128, 128, 128,
    77, 110, 216, 255, 236, 230, 128, 128, 128, 128, 128,
    1, 101, 251, 255, 241, 255, 128, 128, 128, 128, 128,
    170, 139, 241, 252, 236, 209, 255, 255, 128, 128, 128,
    37, 116, 196, 243, 228, 255, 255, 255, 128, 128, 128,
    1, 204, 254, 255, 245, 255, 128, 128, 128, 128, 128,
    207, 160, 250, 255, 238, 128, 128, 128, 128, 128, 128,
    102, 103, 231, 255, 211, 171, 128, 128, 128, 128, 128,
    1, 152, 252, 255, 240, 255, 128, 128, 128, 128, 128,
    177, 135, 243, 255, 234, 225, 128, 128, 128, 128, 128,
    80, 129, 211, 255, 194, 224, 128, 128, 128, 128, 128,
    1, 1, 255, 128, 128, 128, 128, 128, 128, 128, 128,
    246, 1, 255, 128, 128, 128, 128, 128, 128, 128, 128,
    255, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128,
    198, 35, 237, 223, 193, 187, 162, 160, 145, 155, 62,
    131, 45, 198, 221, 172, 176, 220, 157, 252, 221, 1,
    68, 47, 146, 208, 149, 167, 221, 162, 255, 223, 128,
    1, 149, 241, 255, 221, 224, 255, 255, 128, 128, 128,
    184, 141, 234, 253, 222, 220, 255, 199, 128, 128, 128,
    81, 99, 181, 242, 176, 190, 249, 202, 255, 255, 128,
    1, 129, 232, 253, 214, 197, 242, 196, 255, 255, 128,
    99, 121, 210, 250, 201, 198, 255, 202, 128, 128, 128,
    23, 91, 163, 242, 170, 187, 247, 210, 255, 255, 128,
    1, 200, 246, 255, 234, 255, 128, 128, 128, 128, 128,
    109, 178, 241, 255, 231, 245, 255, 255, 128, 128, 128,
    44, 130, 201, 253, 205, 192, 255, 255, 128, 128, 128,
    1, 132, 239, 251, 219, 209, 255, 165, 128, 128, 128,
    94, 136, 225, 251, 218, 190, 255, 255, 128, 128, 128,
    22, 100, 174, 245, 186, 161, 255, 199, 128, 128, 128,
    1, 182, 249, 255, 232, 235, 128, 128, 128, 128, 128,
    124, 143, 241, 255, 227, 234, 128, 128, 128, 128, 128,
    35, 77, 181, 251, 193, 211, 255, 205, 128, 128, 128,
    1, 157, 247, 255, 236, 231, 255, 255, 128, 128, 128,
    121, 141, 235, 255, 225, 227, 255, 255, 128, 128, 128,
    45, 99, 188, 251, 195, 217, 255, 224, 128, 128, 128,
    1, 1, 251, 255, 213, 255, 128, 128, 128, 128, 128,
    203, 1, 248, 255, 255, 128, 128, 128, 128, 128, 128,
    137, 1, 177, 255, 224, 255, 128, 128, 128, 128, 128,
    253, 9, 248, 251, 207, 208, 255, 192, 128, 128, 128,
    175, 13, 224, 243, 193, 185, 249, 198, 255, 255, 128,
    73, 17, 171, 221, 161, 179, 236, 167, 255, 234, 128,
    1, 95, 247, 253, 212, 183, 255, 255, 128, 128, 128,
    239, 90, 244, 250, 211, 209, 255, 255, 128, 128, 128,
    155, 77, 195, 248, 188, 195, 255, 255, 128, 128, 128,
    1, 24, 239, 251, 218, 219, 255, 205, 128, 128, 128,
    201, 51, 219, 255, 196, 186, 128, 128, 128, 128, 128,
    69, 46, 190, 239, 201, 218, 255, 228, 128, 128, 128,
    1, 191, 251, 255, 255, 128, 128, 128, 128, 128, 128,
    223, 165, 249, 255, 213, 255, 128, 128, 128, 128, 128,
    141, 124, 248, 255, 255, 128, 128, 128, 128, 128, 128,
    1, 16, 248, 255, 255, 128, 128, 128, 128, 128, 128,
    190, 36, 230, 255, 236, 255, 128, 128, 128, 128, 128,
    149, 1, 255, 128, 128, 128, 128, 128, 128, 128, 128,
    1, 226, 255, 128, 128, 128, 128, 128, 128, 128, 128,
    247, 192, 255, 128, 128, 128, 128, 128, 128, 128, 128,
    240, 128, 255, 128, 128, 128, 128, 128, 128, 128, 128,
    1, 134, 252, 255, 255, 128, 128, 128, 128, 128, 128,
    213, 62, 250, 255, 255, 128, 128, 128, 128, 128, 128,
    55, 93, 255, 128, 128, 128, 128, 128, 128, 128, 128,
    128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128,
    128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128,
    128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128,
    202, 24, 213, 235, 186, 191, 220, 160, 240, 175, 255,
    126, 38, 182, 232, 169, 184, 228, 174, 255, 187, 128,
    61, 46, 138, 219, 151, 178, 240, 170, 255, 216, 128,
    1, 112, 230, 250, 199, 191, 247, 159, 255, 255, 128,
    166, 109, 228, 252, 211, 215, 255, 174, 128, 128, 128,
    39, 77, 162, 232, 172, 180, 245, 178, 255, 255, 128,
    1, 52, 220, 246, 198, 199, 249, 220, 255, 255, 128,
    124, 74, 191, 243, 183, 193, 250, 221, 255, 255, 128,
    24, 71, 130, 219, 154, 170, 243, 182, 255, 255, 128,
    1, 182, 225, 249, 219, 240, 255, 224, 128, 128, 128,
    149, 150, 226, 252, 216, 205, 255, 171, 128, 128, 128,
    28, 108, 170, 242, 183, 194, 254, 223, 255, 255, 128,
    1, 81, 230, 252, 204, 203, 255, 192, 128, 128, 128,
    123, 102, 209, 247, 188, 196, 255, 233, 128, 128, 128,
    20, 95, 153, 243, 164, 173, 255, 203, 128, 128, 128,
    1, 222, 248, 255, 216, 213, 128, 128, 128, 128, 128,
    168, 175, 246, 252, 235, 205, 255, 255, 128, 128, 128,
    47, 116, 215, 255, 211, 212, 255, 255, 128, 128, 128,
    1, 121, 236, 253, 212, 214, 255, 255, 128, 128, 128,
    141, 84, 213, 252, 201, 202, 255, 219, 128, 128, 128,
    42, 80, 160, 240, 162, 185, 255, 205, 128, 128, 128,
    1, 1, 255, 128, 128, 128, 128, 128, 128, 128, 128,
    244, 1, 255, 128, 128, 128, 128, 128, 128, 128, 128,
    238, 1, 255, 128, 128, 128, 128, 128, 128, 128, 128,
};
static const uint8_t VP8_COEFF_UPDATE_PROBS[1056] = {
    255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255,
    255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255,
    255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255,
    176, 246, 255, 255, 255, 255, 255, 255, 255, 255, 255,
    223, 241, 252, 255, 255, 255, 255, 255, 255, 255, 255,
    249, 253, 253, 255, 255, 255, 255, 255, 255, 255, 255,
    255, 244, 252, 255, 255, 255, 255, 255, 255, 255, 255,
    234, 254, 254, 255, 255, 255, 255, 255, 255, 255, 255,
    253, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255,
    255, 246, 254, 255, 255, 255, 255, 255, 255, 255, 255,
    239, 253, 254, 255, 255, 255, 255, 255, 255, 255, 255,
    254, 255, 254, 255, 255, 255, 255, 255, 255, 255, 255,
    255, 248, 254, 255, 255, 255, 255, 255, 255, 255, 255,
    251, 255, 254, 255, 255, 255, 255, 255, 255, 255, 255,
    255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255,
    255, 253, 254, 255, 255, 255, 255, 255, 255, 255, 255,
    251, 254, 254, 255, 255, 255, 255, 255, 255, 255, 255,
    254, 255, 254, 255, 255, 255, 255, 255, 255, 255, 255,
    255, 254, 253, 255, 254, 255, 255, 255, 255, 255, 255,
    250, 255, 254, 255, 254, 255, 255, 255, 255, 255, 255,
    254, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255,
    255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255,
    255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255,
    255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255,
    217, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255,
    225, 252, 241, 253, 255, 255, 254, 255, 255, 255, 255,
    234, 250, 241, 250, 253, 255, 253, 254, 255, 255, 255,
    255, 254, 255, 255, 255, 255, 255, 255, 255, 255, 255,
    223, 254, 254, 255, 255, 255, 255, 255, 255, 255, 255,
    238, 253, 254, 254, 255, 255, 255, 255, 255, 255, 255,
    255, 248, 254, 255, 255, 255, 255, 255, 255, 255, 255,
    249, 254, 255, 255, 255, 255, 255, 255, 255, 255, 255,
    255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255,
    255, 253, 255, 255, 255, 255, 255, 255, 255, 255, 255,
    247, 254, 255, 255, 255, 255, 255, 255, 255, 255, 255,
    255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255,
    255, 253, 254, 255, 255, 255, 255, 255, 255, 255, 255,
    252, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255,
    255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255,
    255, 254, 254, 255, 255, 255, 255, 255, 255, 255, 255,
    253, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255,
    255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255,
    255, 254, 253, 255, 255, 255, 255, 255, 255, 255, 255,
    250, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255,
    254, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255,
    255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255,
    255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255,
    255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255,
    186, 251, 250, 255, 255, 255, 255, 255, 255, 255, 255,
    234, 251, 244, 254, 255, 255, 255, 255, 255, 255, 255,
    251, 251, 243, 253, 254, 255, 254, 255, 255, 255, 255,
    255, 253, 254, 255, 255, 255, 255, 255, 255, 255, 255,
    236, 253, 254, 255, 255, 255, 255, 255, 255, 255, 255,
    251, 253, 253, 254, 254, 255, 255, 255, 255, 255, 255,
    255, 254, 254, 255, 255, 255, 255, 255, 255, 255, 255,
    254, 254, 254, 255, 255, 255, 255, 255, 255, 255, 255,
    255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255,
    255, 254, 255, 255, 255, 255, 255, 255, 255, 255, 255,
    254, 254, 255, 255, 255, 255, 255, 255, 255, 255, 255,
    254, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255,
    255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255,
    254, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255,
    255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255,
    255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255,
    255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255,
    255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255,
    255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255,
    255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255,
    255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255,
    255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255,
    255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255,
    255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255,
    248, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255,
    250, 254, 252, 254, 255, 255, 255, 255, 255, 255, 255,
    248, 254, 249, 253, 255, 255, 255, 255, 255, 255, 255,
    255, 253, 253, 255, 255, 255, 255, 255, 255, 255, 255,
    246, 253, 253, 255, 255, 255, 255, 255, 255, 255, 255,
    252, 254, 251, 254, 254, 255, 255, 255, 255, 255, 255,
    255, 254, 252, 255, 255, 255, 255, 255, 255, 255, 255,
    248, 254, 253, 255, 255, 255, 255, 255, 255, 255, 255,
    253, 255, 254, 254, 255, 255, 255, 255, 255, 255, 255,
    255, 251, 254, 255, 255, 255, 255, 255, 255, 255, 255,
    245, 251, 254, 255, 255, 255, 255, 255, 255, 255, 255,
    253, 253, 254, 255, 255, 255, 255, 255, 255, 255, 255,
    255, 251, 253, 255, 255, 255, 255, 255, 255, 255, 255,
    252, 253, 254, 255, 255, 255, 255, 255, 255, 255, 255,
    255, 254, 255, 255, 255, 255, 255, 255, 255, 255, 255,
    255, 252, 255, 255, 255, 255, 255, 255, 255, 255, 255,
    249, 255, 254, 255, 255, 255, 255, 255, 255, 255, 255,
    255, 255, 254, 255, 255, 255, 255, 255, 255, 255, 255,
    255, 255, 253, 255, 255, 255, 255, 255, 255, 255, 255,
    250, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255,
    255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255,
    255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255,
    254, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255,
    255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255,
};
static const uint8_t VP8_BMODE_PROBS[900] = {
    231, 120, 48, 89, 115, 113, 120, 152, 112,
    152, 179, 64, 126, 170, 118, 46, 70, 95,
    175, 69, 143, 80, 85, 82, 72, 155, 103,
    56, 58, 10, 171, 218, 189, 17, 13, 152,
    114, 26, 17, 163, 44, 195, 21, 10, 173,
    121, 24, 80, 195, 26, 62, 44, 64, 85,
    144, 71, 10, 38, 171, 213, 144, 34, 26,
    170, 46, 55, 19, 136, 160, 33, 206, 71,
    63, 20, 8, 114, 114, 208, 12, 9, 226,
    81, 40, 11, 96, 182, 84, 29, 16, 36,
    134, 183, 89, 137, 98, 101, 106, 165, 148,
    72, 187, 100, 130, 157, 111, 32, 75, 80,
    66, 102, 167, 99, 74, 62, 40, 234, 128,
    41, 53, 9, 178, 241, 141, 26, 8, 107,
    74, 43, 26, 146, 73, 166, 49, 23, 157,
    65, 38, 105, 160, 51, 52, 31, 115, 128,
    104, 79, 12, 27, 217, 255, 87, 17, 7,
    87, 68, 71, 44, 114, 51, 15, 186, 23,
    47, 41, 14, 110, 182, 183, 21, 17, 194,
    66, 45, 25, 102, 197, 189, 23, 18, 22,
    88, 88, 147, 150, 42, 46, 45, 196, 205,
    43, 97, 183, 117, 85, 38, 35, 179, 61,
    39, 53, 200, 87, 26, 21, 43, 232, 171,
    56, 34, 51, 104, 114, 102, 29, 93, 77,
    39, 28, 85, 171, 58, 165, 90, 98, 64,
    34, 22, 116, 206, 23, 34, 43, 166, 73,
    107, 54, 32, 26, 51, 1, 81, 43, 31,
    68, 25, 106, 22, 64, 171, 36, 225, 114,
    34, 19, 21, 102, 132, 188, 16, 76, 124,
    62, 18, 78, 95, 85, 57, 50, 48, 51,
    193, 101, 35, 159, 215, 111, 89, 46, 111,
    60, 148, 31, 172, 219, 228, 21, 18, 111,
    112, 113, 77, 85, 179, 255, 38, 120, 114,
    40, 42, 1, 196, 245, 209, 10, 25, 109,
    88, 43, 29, 140, 166, 213, 37, 43, 154,
    61, 63, 30, 155, 67, 45, 68, 1, 209,
    100, 80, 8, 43, 154, 1, 51, 26, 71,
    142, 78, 78, 16, 255, 128, 34, 197, 171,
    41, 40, 5, 102, 211, 183, 4, 1, 221,
    51, 50, 17, 168, 209, 192, 23, 25, 82,
    138, 31, 36, 171, 27, 166, 38, 44, 229,
    67, 87, 58, 169, 82, 115, 26, 59, 179,
    63, 59, 90, 180, 59, 166, 93, 73, 154,
    40, 40, 21, 116, 143, 209, 34, 39, 175,
    47, 15, 16, 183, 34, 223, 49, 45, 183,
    46, 17, 33, 183, 6, 98, 15, 32, 183,
    57, 46, 22, 24, 128, 1, 54, 17, 37,
    65, 32, 73, 115, 28, 128, 23, 128, 205,
    40, 3, 9, 115, 51, 192, 18, 6, 223,
    87, 37, 9, 115, 59, 77, 64, 21, 47,
    104, 55, 44, 218, 9, 54, 53, 130, 226,
    64, 90, 70, 205, 40, 41, 23, 26, 57,
    54, 57, 112, 184, 5, 41, 38, 166, 213,
    30, 34, 26, 133, 152, 116, 10, 32, 134,
    39, 19, 53, 221, 26, 114, 32, 73, 255,
    31, 9, 65, 234, 2, 15, 1, 118, 73,
    75, 32, 12, 51, 192, 255, 160, 43, 51,
    88, 31, 35, 67, 102, 85, 55, 186, 85,
    56, 21, 23, 111, 59, 205, 45, 37, 192,
    55, 38, 70, 124, 73, 102, 1, 34, 98,
    125, 98, 42, 88, 104, 85, 117, 175, 82,
    95, 84, 53, 89, 128, 100, 113, 101, 45,
    75, 79, 123, 47, 51, 128, 81, 171, 1,
    57, 17, 5, 71, 102, 57, 53, 41, 49,
    38, 33, 13, 121, 57, 73, 26, 1, 85,
    41, 10, 67, 138, 77, 110, 90, 47, 114,
    115, 21, 2, 10, 102, 255, 166, 23, 6,
    101, 29, 16, 10, 85, 128, 101, 196, 26,
    57, 18, 10, 102, 102, 213, 34, 20, 43,
    117, 20, 15, 36, 163, 128, 68, 1, 26,
    102, 61, 71, 37, 34, 53, 31, 243, 192,
    69, 60, 71, 38, 73, 119, 28, 222, 37,
    68, 45, 128, 34, 1, 47, 11, 245, 171,
    62, 17, 19, 70, 146, 85, 55, 62, 70,
    37, 43, 37, 154, 100, 163, 85, 160, 1,
    63, 9, 92, 136, 28, 64, 32, 201, 85,
    75, 15, 9, 9, 64, 255, 184, 119, 16,
    86, 6, 28, 5, 64, 255, 25, 248, 1,
    56, 8, 17, 132, 137, 255, 55, 116, 128,
    58, 15, 20, 82, 135, 57, 26, 121, 40,
    164, 50, 31, 137, 154, 133, 25, 35, 218,
    51, 103, 44, 131, 131, 123, 31, 6, 158,
    86, 40, 64, 135, 148, 224, 45, 183, 128,
    22, 26, 17, 131, 240, 154, 14, 1, 209,
    45, 16, 21, 91, 64, 222, 7, 1, 197,
    56, 21, 39, 155, 60, 138, 23, 102, 213,
    83, 12, 13, 54, 192, 255, 68, 47, 28,
    85, 26, 85, 85, 128, 128, 32, 146, 171,
    18, 11, 7, 63, 144, 171, 4, 4, 246,
    35, 27, 10, 146, 174, 171, 12, 26, 128,
    190, 80, 35, 99, 180, 80, 126, 54, 45,
    85, 126, 47, 87, 176, 51, 41, 20, 32,
    101, 75, 128, 139, 118, 146, 116, 128, 85,
    56, 41, 15, 176, 236, 85, 37, 9, 62,
    71, 30, 17, 119, 118, 255, 17, 18, 138,
    101, 38, 60, 138, 55, 70, 43, 26, 142,
    146, 36, 19, 30, 171, 255, 97, 27, 20,
    138, 45, 61, 62, 219, 1, 81, 188, 64,
    32, 41, 20, 117, 151, 142, 20, 21, 163,
    112, 19, 12, 61, 195, 128, 48, 4, 24,
};
static const uint8_t VP8_DC_Q[128] = {
    4, 5, 6, 7, 8, 9, 10, 10, 11, 12, 13, 14, 15, 16, 17, 17,
    18, 19, 20, 20, 21, 21, 22, 22, 23, 23, 24, 25, 25, 26, 27, 28,
    29, 30, 31, 32, 33, 34, 35, 36, 37, 37, 38, 39, 40, 41, 42, 43,
    44, 45, 46, 46, 47, 48, 49, 50, 51, 52, 53, 54, 55, 56, 57, 58,
    59, 60, 61, 62, 63, 64, 65, 66, 67, 68, 69, 70, 71, 72, 73, 74,
    75, 76, 76, 77, 78, 79, 80, 81, 82, 83, 84, 85, 86, 87, 88, 89,
    91, 93, 95, 96, 98, 100, 101, 102, 104, 106, 108, 110, 112, 114, 116, 118,
    122, 124, 126, 128, 130, 132, 134, 136, 138, 140, 143, 145, 148, 151, 154, 157,
};
static const uint16_t VP8_AC_Q[128] = {
    4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19,
    20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31, 32, 33, 34, 35,
    36, 37, 38, 39, 40, 41, 42, 43, 44, 45, 46, 47, 48, 49, 50, 51,
    52, 53, 54, 55, 56, 57, 58, 60, 62, 64, 66, 68, 70, 72, 74, 76,
    78, 80, 82, 84, 86, 88, 90, 92, 94, 96, 98, 100, 102, 104, 106, 108,
    110, 112, 114, 116, 119, 122, 125, 128, 131, 134, 137, 140, 143, 146, 149, 152,
    155, 158, 161, 164, 167, 170, 173, 177, 181, 185, 189, 193, 197, 201, 205, 209,
    213, 217, 221, 225, 229, 234, 239, 245, 249, 254, 259, 264, 269, 274, 279, 284,
};

static const uint8_t VP8_ZIGZAG[16] = {0, 1, 4, 8, 5, 2, 3, 6, 9, 12, 13, 10, 7, 11, 14, 15};
static const uint8_t VP8_BANDS[17] = {0, 1, 2, 3, 6, 4, 5, 6, 6, 6, 6, 6, 6, 6, 6, 7, 0};
static const uint8_t VP8_CAT3[] = {173, 148, 140, 0}, VP8_CAT4[] = {176, 155, 140, 135, 0}, VP8_CAT5[] = {180, 157, 141, 134, 130, 0},
                     VP8_CAT6[] = {254, 254, 243, 230, 196, 177, 153, 140, 133, 130, 129, 0};
static const uint8_t* const VP8_CAT3456[4] = {VP8_CAT3, VP8_CAT4, VP8_CAT5, VP8_CAT6};
enum { B_DC = 0, B_TM, B_VE, B_HE, B_RD, B_VR, B_LD, B_VL, B_HD, B_HU };   // (the order VP8_BMODE_PROBS is indexed in)
enum { Y_DC = 0, Y_V, Y_H, Y_TM, Y_B };

// RFC 6386 s. 7.3: the boolean decoder (bytes past the end read as zero; `over` counts them)
struct BoolDecoder {
    const uint8_t* p = nullptr; size_t n = 0, pos = 0, over = 0;
    uint32_t value = 0, range = 255; int bit_count = 0;
    void init(const uint8_t* d, size_t len) { p = d; n = len; pos = 0; over = 0; range = 255; bit_count = 0; value = (uint32_t)next() << 8; value |= next(); }
    uint8_t next() { if (pos < n) return p[pos++]; ++over; return 0; }
    int get(int prob) {
        const uint32_t split = 1u + (((range - 1u) * (uint32_t)prob) >> 8), big = split << 8;
        int bit;
        if (value >= big) { bit = 1; range -= split; value -= big; } else { bit = 0; range = split; }
        while (range < 128u) {
            value <<= 1; range <<= 1;
            if (++bit_count == 8) { bit_count = 0; value |= next(); }
        }
        return bit;
    }
    uint32_t literal(int bits) { uint32_t v = 0; while (bits-- > 0) v = (v << 1) | (uint32_t)get(128); return v; }
    int flagged_signed(int bits) { if (!get(128)) return 0; const int v = (int)literal(bits); return get(128) ? -v : v; }   // "flag, magnitude, sign"
};

inline uint8_t clamp255(int v) { return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v); }
inline int avg2(int a, int b) { return (a + b + 1) >> 1; }
inline int avg3(int a, int b, int c) { return (a + 2 * b + c + 2) >> 2; }
inline int mul_sin(int v) { return (int)(((int64_t)v * 35468) >> 16); }    // sqrt(2) * sin(pi / 8), 16-bit fixed point (64-bit product: a forged coefficient may be large)
inline int mul_cos(int v) { return (int)(((int64_t)v * 20091) >> 16); }    // sqrt(2) * cos(pi / 8) - 1

// s. 14.3: inverse Walsh-Hadamard transform of the Y2 block -> the DC of the sixteen luma subblocks
inline void inverse_wht(const int* in, int* out) {
    int t[16];
    for (int i = 0; i < 4; ++i) {
        const int a1 = in[i] + in[12 + i], b1 = in[4 + i] + in[8 + i], c1 = in[4 + i] - in[8 + i], d1 = in[i] - in[12 + i];
        t[i] = a1 + b1; t[4 + i] = c1 + d1; t[8 + i] = a1 - b1; t[12 + i] = d1 - c1;
    }
    for (int i = 0; i < 4; ++i) {
        const int* r = t + 4 * i;
        const int a1 = r[0] + r[3], b1 = r[1] + r[2], c1 = r[1] - r[2], d1 = r[0] - r[3];
        out[4 * i] = (a1 + b1 + 3) >> 3; out[4 * i + 1] = (c1 + d1 + 3) >> 3; out[4 * i + 2] = (a1 - b1 + 3) >> 3; out[4 * i + 3] = (d1 - c1 + 3) >> 3;
    }
}
// s. 14.4: inverse DCT of one subblock, added to the prediction at dst (stride in bytes)
inline void inverse_dct_add(const int* in, uint8_t* dst, int stride) {
    int t[16];
    for (int i = 0; i < 4; ++i) {
        const int a1 = in[i] + in[8 + i], b1 = in[i] - in[8 + i];
        int t1 = mul_sin(in[4 + i]), t2 = in[12 + i] + mul_cos(in[12 + i]);
        const int c1 = t1 - t2;
        t1 = in[4 + i] + mul_cos(in[4 + i]); t2 = mul_sin(in[12 + i]);
        const int d1 = t1 + t2;
        t[i] = a1 + d1; t[12 + i] = a1 - d1; t[4 + i] = b1 + c1; t[8 + i] = b1 - c1;
    }
    for (int i = 0; i < 4; ++i) {
        const int* r = t + 4 * i;
        const int a1 = r[0] + r[2], b1 = r[0] - r[2];
        int t1 = mul_sin(r[1]), t2 = r[3] + mul_cos(r[3]);
        const int c1 = t1 - t2;
        t1 = r[1] + mul_cos(r[1]); t2 = mul_sin(r[3]);
        const int d1 = t1 + t2;
        uint8_t* o = dst + i * stride;
        o[0] = clamp255(o[0] + ((a1 + d1 + 4) >> 3)); o[3] = clamp255(o[3] + ((a1 - d1 + 4) >> 3));
        o[1] = clamp255(o[1] + ((b1 + c1 + 4) >> 3)); o[2] = clamp255(o[2] + ((b1 - c1 + 4) >> 3));
    }
}

// s. 12.3: one 4 x 4 subblock predictor. w points at the subblock's top-left pixel inside the macroblock's work area (stride s): row -1 holds the
// pixels above (eight of them: above and above-right), column -1 the pixels to the left, w[-s - 1] the corner
inline void predict_subblock(int mode, uint8_t* w, int s) {
    const uint8_t* A = w - s;
    const int P = w[-s - 1];
    const int L[4] = {w[-1], w[s - 1], w[2 * s - 1], w[3 * s - 1]};
    const int E[9] = {L[3], L[2], L[1], L[0], P, A[0], A[1], A[2], A[3]};
    uint8_t B[4][4];
    switch (mode) {
    case B_DC: {
        int v = 4;
        for (int i = 0; i < 4; ++i) v += A[i] + L[i];
        std::memset(B, v >> 3, sizeof B);
        break;
    }
    case B_TM:
        for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) B[r][c] = clamp255(L[r] + A[c] - P);
        break;
    case B_VE:
        for (int c = 0; c < 4; ++c) { const uint8_t v = (uint8_t)avg3(c ? A[c - 1] : P, A[c], A[c + 1]); for (int r = 0; r < 4; ++r) B[r][c] = v; }
        break;
    case B_HE: {
        const uint8_t v[4] = {(uint8_t)avg3(P, L[0], L[1]), (uint8_t)avg3(L[0], L[1], L[2]), (uint8_t)avg3(L[1], L[2], L[3]), (uint8_t)avg3(L[2], L[3], L[3])};
        for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) B[r][c] = v[r];
        break;
    }
    case B_LD:
        for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) { const int i = r + c; B[r][c] = (uint8_t)(i == 6 ? avg3(A[6], A[7], A[7]) : avg3(A[i], A[i + 1], A[i + 2])); }
        break;
    case B_RD:
        for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) { const int i = 4 - r + c; B[r][c] = (uint8_t)avg3(E[i - 1], E[i], E[i + 1]); }
        break;
    case B_VR:
        B[3][0] = (uint8_t)avg3(E[1], E[2], E[3]);
        B[2][0] = (uint8_t)avg3(E[2], E[3], E[4]);
        B[3][1] = B[1][0] = (uint8_t)avg3(E[3], E[4], E[5]);
        B[2][1] = B[0][0] = (uint8_t)avg2(E[4], E[5]);
        B[3][2] = B[1][1] = (uint8_t)avg3(E[4], E[5], E[6]);
        B[2][2] = B[0][1] = (uint8_t)avg2(E[5], E[6]);
        B[3][3] = B[1][2] = (uint8_t)avg3(E[5], E[6], E[7]);
        B[2][3] = B[0][2] = (uint8_t)avg2(E[6], E[7]);
        B[1][3] = (uint8_t)avg3(E[6], E[7], E[8]);
        B[0][3] = (uint8_t)avg2(E[7], E[8]);
        break;
    case B_VL:
        B[0][0] = (uint8_t)avg2(A[0], A[1]);
        B[1][0] = (uint8_t)avg3(A[0], A[1], A[2]);
        B[2][0] = B[0][1] = (uint8_t)avg2(A[1], A[2]);
        B[1][1] = B[3][0] = (uint8_t)avg3(A[1], A[2], A[3]);
        B[2][1] = B[0][2] = (uint8_t)avg2(A[2], A[3]);
        B[3][1] = B[1][2] = (uint8_t)avg3(A[2], A[3], A[4]);
        B[2][2] = B[0][3] = (uint8_t)avg2(A[3], A[4]);
        B[3][2] = B[1][3] = (uint8_t)avg3(A[3], A[4], A[5]);
        B[2][3] = (uint8_t)avg3(A[4], A[5], A[6]);   // (the last two leave the pattern)
        B[3][3] = (uint8_t)avg3(A[5], A[6], A[7]);
        break;
    case B_HD:
        B[3][0] = (uint8_t)avg2(E[0], E[1]);
        B[3][1] = (uint8_t)avg3(E[0], E[1], E[2]);
        B[2][0] = B[3][2] = (uint8_t)avg2(E[1], E[2]);
        B[2][1] = B[3][3] = (uint8_t)avg3(E[1], E[2], E[3]);
        B[2][2] = B[1][0] = (uint8_t)avg2(E[2], E[3]);
        B[2][3] = B[1][1] = (uint8_t)avg3(E[2], E[3], E[4]);
        B[1][2] = B[0][0] = (uint8_t)avg2(E[3], E[4]);
        B[1][3] = B[0][1] = (uint8_t)avg3(E[3], E[4], E[5]);
        B[0][2] = (uint8_t)avg3(E[4], E[5], E[6]);
        B[0][3] = (uint8_t)avg3(E[5], E[6], E[7]);
        break;
    default:   // B_HU
        B[0][0] = (uint8_t)avg2(L[0], L[1]);
        B[0][1] = (uint8_t)avg3(L[0], L[1], L[2]);
        B[0][2] = B[1][0] = (uint8_t)avg2(L[1], L[2]);
        B[0][3] = B[1][1] = (uint8_t)avg3(L[1], L[2], L[3]);
        B[1][2] = B[2][0] = (uint8_t)avg2(L[2], L[3]);
        B[1][3] = B[2][1] = (uint8_t)avg3(L[2], L[3], L[3]);
        B[2][2] = B[2][3] = B[3][0] = B[3][1] = B[3][2] = B[3][3] = (uint8_t)L[3];
        break;
    }
    for (int r = 0; r < 4; ++r) std::memcpy(w + r * s, B[r], 4);
}

struct Quant { int y_dc, y_ac, y2_dc, y2_ac; };

// s. 13: the tokens of one block -> dequantised coefficients in raster order; returns whether the block's first token was anything but
// "end of block" (the flag the neighbours' contexts use). probs = this block type's [8 bands][3 contexts][11]
inline bool read_block(BoolDecoder& br, const uint8_t* probs, int ctx, int first, int dc_q, int ac_q, int* out) {
    const uint8_t* p = probs + (VP8_BANDS[first] * 3 + ctx) * 11;
    for (int n = first; n < 16; ++n) {
        if (!br.get(p[0])) return n > first;   // end of block
        while (!br.get(p[1])) {                // zeros (no end of block may follow one)
            if (++n == 16) return true;
            p = probs + (VP8_BANDS[n] * 3 + 0) * 11;
        }
        int v, next_ctx;
        if (!br.get(p[2])) { v = 1; next_ctx = 1; }
        else {
            next_ctx = 2;
            if (!br.get(p[3])) v = !br.get(p[4]) ? 2 : 3 + br.get(p[5]);
            else if (!br.get(p[6])) {
                if (!br.get(p[7])) v = 5 + br.get(159);
                else { v = 7 + 2 * br.get(165); v += br.get(145); }
            } else {
                const int b1 = br.get(p[8]), b0 = br.get(p[9 + b1]), cat = 2 * b1 + b0;
                v = 0;
                for (const uint8_t* t = VP8_CAT3456[cat]; *t; ++t) v += v + br.get(*t);
                v += 3 + (8 << cat);
            }
        }
        if (br.get(128)) v = -v;
        out[VP8_ZIGZAG[n]] = v * (n > 0 ? ac_q : dc_q);
        p = probs + (VP8_BANDS[n + 1] * 3 + next_ctx) * 11;
    }
    return true;
}

}  // namespace vp8_detail

// the luma plane of a simple lossy WebP file (w x h bytes, row 0 on top)
inline bool decode_webp_luma(const std::vector<uint8_t>& f, uint32_t& width, uint32_t& height, std::vector<uint8_t>& luma, std::string& err) {
    using namespace vp8_detail;
    if (f.size() < 20 || std::memcmp(f.data(), "RIFF", 4) || std::memcmp(f.data() + 8, "WEBP", 4)) { err = "not a WebP file"; return false; }
    if (std::memcmp(f.data() + 12, "VP8 ", 4)) { err = "Invalid VP8 signature (image 0.18 reads the simple lossy WebP container only: no VP8L / VP8X)"; return false; }
    const uint8_t* d = f.data() + 20;   // (the chunk's length field is not used: the frame is the rest of the file, as in the crate)
    const size_t n = f.size() - 20;
    if (n < 10) { err = "truncated VP8 frame header"; return false; }
    const uint32_t tag = d[0] | (uint32_t)d[1] << 8 | (uint32_t)d[2] << 16;
    const size_t first_size = tag >> 5;
    if (tag & 1u) { err = "WebP: the VP8 frame is not a key frame"; return false; }
    if (d[3] != 0x9d || d[4] != 0x01 || d[5] != 0x2a) { err = "WebP: bad VP8 start code"; return false; }
    const uint32_t w = (d[6] | (uint32_t)d[7] << 8) & 0x3fffu, h = (d[8] | (uint32_t)d[9] << 8) & 0x3fffu;
    if (w == 0 || h == 0) { err = "WebP: empty frame"; return false; }
    if (first_size == 0 || 10 + first_size > n) { err = "WebP: first partition beyond the end of the file"; return false; }
    const uint32_t mbw = (w + 15) / 16, mbh = (h + 15) / 16;
    if ((uint64_t)mbw * mbh > (uint64_t)first_size * 8) { err = "WebP: more macroblocks than the first partition can describe"; return false; }   // (a macroblock header costs more than a bit)
    if ((uint64_t)w * h > std::max<uint64_t>(1u << 20, (uint64_t)f.size() * 1024u)) { err = "WebP dimensions out of proportion to the file size"; return false; }   // (as the other decoders: no allocation from a forged header)
    BoolDecoder hd;
    hd.init(d + 10, first_size);
    hd.get(128);   // colour space
    hd.get(128);   // clamping type (the reconstruction clamps either way: "no clamping needed" is a promise of the encoder)
    // s. 9.3 segments
    bool seg_enabled = hd.get(128) != 0, seg_update_map = false, seg_absolute = false;
    int seg_quant[4] = {0, 0, 0, 0};
    uint8_t seg_probs[3] = {255, 255, 255};
    if (seg_enabled) {
        seg_update_map = hd.get(128) != 0;
        if (hd.get(128)) {
            seg_absolute = hd.get(128) != 0;
            for (int i = 0; i < 4; ++i) seg_quant[i] = hd.flagged_signed(7);
            for (int i = 0; i < 4; ++i) hd.flagged_signed(6);   // loop filter strengths: unused
        }
        if (seg_update_map) for (int i = 0; i < 3; ++i) seg_probs[i] = hd.get(128) ? (uint8_t)hd.literal(8) : 255;
    }
    // s. 9.6 loop filter parameters: parsed, not applied (image 0.18 shows the unfiltered reconstruction)
    hd.get(128); hd.literal(6); hd.literal(3);
    if (hd.get(128) && hd.get(128)) { for (int i = 0; i < 8; ++i) hd.flagged_signed(6); }
    // s. 9.5 token partitions
    const uint32_t n_parts = 1u << hd.literal(2);
    const size_t sizes_at = 10 + first_size;
    if (sizes_at + 3 * (size_t)(n_parts - 1) > n) { err = "WebP: truncated partition table"; return false; }
    std::vector<BoolDecoder> parts(n_parts);
    {
        size_t at = sizes_at + 3 * (size_t)(n_parts - 1);
        for (uint32_t i = 0; i < n_parts; ++i) {
            size_t len = n - at;
            if (i + 1 < n_parts) {
                const uint8_t* s = d + sizes_at + 3 * i;
                const size_t want = s[0] | (size_t)s[1] << 8 | (size_t)s[2] << 16;
                if (want > len) { err = "WebP: token partition beyond the end of the file"; return false; }
                len = want;
            }
            parts[i].init(d + at, len);
            at += len;
        }
    }
    // s. 9.6 quantiser indices
    const int q_base = (int)hd.literal(7);
    auto delta = [&]() { if (!hd.get(128)) return 0; const int v = (int)hd.literal(4); return hd.get(128) ? -v : v; };
    const int dq_y_dc = delta(), dq_y2_dc = delta(), dq_y2_ac = delta();
    delta(); delta();   // chroma
    auto qi = [](int v) { return v < 0 ? 0 : v > 127 ? 127 : v; };
    Quant quant[4];
    for (int i = 0; i < 4; ++i) {
        int q = q_base;
        if (seg_enabled) q = seg_absolute ? seg_quant[i] : q_base + seg_quant[i];
        q = qi(q);
        quant[i].y_dc = VP8_DC_Q[qi(q + dq_y_dc)];
        quant[i].y_ac = VP8_AC_Q[q];
        quant[i].y2_dc = VP8_DC_Q[qi(q + dq_y2_dc)] * 2;
        quant[i].y2_ac = VP8_AC_Q[qi(q + dq_y2_ac)] * 155 / 100;
        if (quant[i].y2_ac < 8) quant[i].y2_ac = 8;
    }
    hd.get(128);   // refresh_entropy_probs: one frame only
    // s. 13.4 coefficient probability updates
    uint8_t coeff[1056];
    std::memcpy(coeff, VP8_COEFF_PROBS, sizeof coeff);
    for (int i = 0; i < 1056; ++i) if (hd.get(VP8_COEFF_UPDATE_PROBS[i])) coeff[i] = (uint8_t)hd.literal(8);
    const bool use_skip = hd.get(128) != 0;
    const int skip_prob = use_skip ? (int)hd.literal(8) : 0;
    if (hd.over > 2) { err = "WebP: truncated frame header"; return false; }

    const size_t stride = (size_t)mbw * 16;
    std::vector<uint8_t> frame(stride * mbh * 16);
    std::vector<uint8_t> above_modes((size_t)mbw * 4, (uint8_t)B_DC), seg_ids((size_t)mbw * mbh, 0);
    // "any coefficient" flags of the blocks above / to the left: 4 luma + 2 + 2 chroma columns per macroblock, one Y2 flag
    std::vector<uint8_t> above_nz((size_t)mbw * 9, 0);
    constexpr int WS = 1 + 16 + 4;   // work area: corner + 16 columns + 4 above-right; row 0 = the pixels above
    uint8_t ws[17 * WS];
    for (uint32_t mby = 0; mby < mbh; ++mby) {
        BoolDecoder& tk = parts[mby % n_parts];
        uint8_t left_modes[4] = {B_DC, B_DC, B_DC, B_DC};
        uint8_t left_nz[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (uint32_t mbx = 0; mbx < mbw; ++mbx) {
            // ---- s. 19.3 macroblock header
            uint8_t& seg = seg_ids[(size_t)mby * mbw + mbx];
            if (seg_update_map) seg = (uint8_t)(hd.get(seg_probs[0]) ? 2 + hd.get(seg_probs[2]) : hd.get(seg_probs[1]));
            const bool skip = use_skip && hd.get(skip_prob);
            int ymode;
            uint8_t modes[16];
            uint8_t* am = &above_modes[(size_t)mbx * 4];
            if (!hd.get(145)) {
                ymode = Y_B;
                for (int r = 0; r < 4; ++r)
                    for (int c = 0; c < 4; ++c) {
                        const int a = r ? modes[4 * (r - 1) + c] : am[c], l = c ? modes[4 * r + c - 1] : left_modes[r];
                        const uint8_t* p = VP8_BMODE_PROBS + (a * 10 + l) * 9;
                        int m;
                        if (!hd.get(p[0])) m = B_DC;
                        else if (!hd.get(p[1])) m = B_TM;
                        else if (!hd.get(p[2])) m = B_VE;
                        else if (!hd.get(p[3])) m = !hd.get(p[4]) ? B_HE : (!hd.get(p[5]) ? B_RD : B_VR);
                        else if (!hd.get(p[6])) m = B_LD;
                        else if (!hd.get(p[7])) m = B_VL;
                        else m = !hd.get(p[8]) ? B_HD : B_HU;
                        modes[4 * r + c] = (uint8_t)m;
                    }
            } else {
                ymode = hd.get(156) ? (hd.get(128) ? Y_TM : Y_H) : (hd.get(163) ? Y_V : Y_DC);
                static const uint8_t implied[4] = {B_DC, B_VE, B_HE, B_TM};
                std::memset(modes, implied[ymode], 16);
            }
            for (int i = 0; i < 4; ++i) { am[i] = modes[12 + i]; left_modes[i] = modes[4 * i + 3]; }
            if (hd.get(142)) { if (hd.get(114)) hd.get(183); }   // chroma mode: parsed, unused
            // ---- s. 13 residual tokens
            int y_coef[16][16];
            std::memset(y_coef, 0, sizeof y_coef);
            uint8_t* anz = &above_nz[(size_t)mbx * 9];
            const Quant& Q = quant[seg];
            if (!skip) {
                int first = 0;
                const uint8_t* y_probs = coeff + 3 * 264;   // block type 3: luma with its own DC
                if (ymode != Y_B) {
                    int y2[16], dc[16];
                    std::memset(y2, 0, sizeof y2);
                    const bool nz = read_block(tk, coeff + 1 * 264, anz[8] + left_nz[8], 0, Q.y2_dc, Q.y2_ac, y2);
                    anz[8] = left_nz[8] = nz;
                    inverse_wht(y2, dc);
                    for (int b = 0; b < 16; ++b) y_coef[b][0] = dc[b];
                    first = 1;
                    y_probs = coeff;   // block type 0: luma after Y2
                }
                for (int r = 0; r < 4; ++r)
                    for (int c = 0; c < 4; ++c) {
                        const bool nz = read_block(tk, y_probs, anz[c] + left_nz[r], first, Q.y_dc, Q.y_ac, y_coef[4 * r + c]);
                        anz[c] = left_nz[r] = nz;
                    }
                int dropped[16];
                for (int plane = 0; plane < 2; ++plane)   // U then V: parsed for the bit stream's sake (dequantised with 1: the values are not used)
                    for (int r = 0; r < 2; ++r)
                        for (int c = 0; c < 2; ++c) {
                            const int ai = 4 + 2 * plane + c, li = 4 + 2 * plane + r;
                            const bool nz = read_block(tk, coeff + 2 * 264, anz[ai] + left_nz[li], 0, 1, 1, dropped);
                            anz[ai] = left_nz[li] = nz;
                        }
            } else {
                for (int i = 0; i < 8; ++i) anz[i] = left_nz[i] = 0;
                if (ymode != Y_B) anz[8] = left_nz[8] = 0;   // (a macroblock without Y2 leaves the Y2 context alone)
            }
            // ---- s. 12 prediction + residue in the work area
            uint8_t* const mb = &frame[(size_t)mby * 16 * stride + (size_t)mbx * 16];
            if (mby == 0) std::memset(ws + 1, 127, 20);
            else {
                std::memcpy(ws + 1, mb - stride, 16);
                if (mbx + 1 < mbw) std::memcpy(ws + 17, mb - stride + 16, 4);
                else std::memset(ws + 17, mb[-(ptrdiff_t)stride + 15], 4);
            }
            for (int i = 17; i < 21; ++i) ws[4 * WS + i] = ws[8 * WS + i] = ws[12 * WS + i] = ws[i];
            for (int r = 0; r < 16; ++r) ws[(r + 1) * WS] = mbx == 0 ? 129 : mb[(size_t)r * stride - 1];
            ws[0] = mby == 0 ? 127 : mbx == 0 ? 129 : mb[-(ptrdiff_t)stride - 1];
            uint8_t* const px = ws + WS + 1;
            if (ymode == Y_B) {
                for (int r = 0; r < 4; ++r)
                    for (int c = 0; c < 4; ++c) {
                        uint8_t* sb = px + 4 * r * WS + 4 * c;
                        predict_subblock(modes[4 * r + c], sb, WS);
                        inverse_dct_add(y_coef[4 * r + c], sb, WS);
                    }
            } else {
                if (ymode == Y_DC) {
                    int v = 128;
                    if (mbx > 0 || mby > 0) {
                        int sum = 0, shift = 3;
                        if (mby > 0) { for (int i = 0; i < 16; ++i) sum += ws[1 + i]; ++shift; }
                        if (mbx > 0) { for (int i = 0; i < 16; ++i) sum += ws[(i + 1) * WS]; ++shift; }
                        v = (sum + (1 << (shift - 1))) >> shift;
                    }
                    for (int r = 0; r < 16; ++r) std::memset(px + r * WS, v, 16);
                } else if (ymode == Y_V) {
                    for (int r = 0; r < 16; ++r) std::memcpy(px + r * WS, ws + 1, 16);
                } else if (ymode == Y_H) {
                    for (int r = 0; r < 16; ++r) std::memset(px + r * WS, ws[(r + 1) * WS], 16);
                } else {
                    for (int r = 0; r < 16; ++r) for (int c = 0; c < 16; ++c) px[r * WS + c] = clamp255(ws[(r + 1) * WS] + ws[1 + c] - ws[0]);
                }
                for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) inverse_dct_add(y_coef[4 * r + c], px + 4 * r * WS + 4 * c, WS);
            }
            for (int r = 0; r < 16; ++r) std::memcpy(mb + (size_t)r * stride, px + r * WS, 16);
        }
        if (hd.over > 2) { err = "WebP: truncated macroblock headers"; return false; }
    }
    for (auto& p : parts) if (p.over > 2) { err = "WebP: truncated token partition"; return false; }
    width = w; height = h;
    luma.resize((size_t)w * h);
    for (uint32_t y = 0; y < h; ++y) std::memcpy(&luma[(size_t)y * w], &frame[(size_t)y * stride], w);
    return true;
}

}  // namespace trayh
