"""Tiny PNG writer (RGB8) for eyeballing renders; test infrastructure."""
import struct
import zlib


def write_png(path, rgb8, width, height):
    raw = b"".join(b"\x00" + bytes(rgb8[y * width * 3:(y + 1) * width * 3]) for y in range(height))

    def chunk(tag, data):
        c = struct.pack(">I", len(data)) + tag + data
        return c + struct.pack(">I", zlib.crc32(tag + data) & 0xffffffff)

    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", width, height, 8, 2, 0, 0, 0)) +
                chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))
