"""Test infrastructure: the MASTER side of the reference's distributed mode, restated from src/exec/distrib/master.rs so that the
worker of tray_rust_amd/distrib.py can be driven over a real TCP connection without a Rust toolchain:

    partition      master.rs:91-93,218-227   queue.len() / workers blocks each, the remainder to the last worker
    instructions   master.rs:217-237         one Instructions message per worker, as soon as its socket is writable
    frames         master.rs:166-192,240-263 8 bytes of size, the rest, decode, save_results
    save_results   master.rs:124-163         film::Image::add_blocks of every worker's blocks into the frame's image

Its encoder / decoder are written here a second time (struct only), independently of tray_rust_amd.distrib."""
import socket
import struct

import numpy as np


def encode_instructions(scene, frames, block_start, block_count):
    s = scene.encode("utf-8")
    body = struct.pack("<Q", len(s)) + s + struct.pack("<4Q", frames[0], frames[1], block_start, block_count)
    return struct.pack("<Q", 8 + len(body)) + body


def decode_frame(buf):
    size, frame, bw, bh, nb = struct.unpack_from("<5Q", buf, 0)
    assert size == len(buf), (size, len(buf))
    off = 40
    blocks = np.frombuffer(buf, "<u8", 2 * nb, off).reshape(-1, 2); off += 16 * nb
    (npx,) = struct.unpack_from("<Q", buf, off); off += 8
    pixels = np.frombuffer(buf, "<f4", npx, off); off += 4 * npx
    assert off == len(buf) and npx == nb * bw * bh * 4
    return frame, (bw, bh), blocks, pixels


def _read(sock, n):
    b = bytearray()
    while len(b) < n:
        chunk = sock.recv(n - len(b))
        assert chunk, "worker hung up"
        b += chunk
    return bytes(b)


def _connect(host, port, patience=120.0):
    """(the reference's master panics if a worker is not listening yet, master.rs:100-112; the tests start workers and master together)"""
    import time
    t0 = time.time()
    while True:
        try:
            return socket.create_connection((host, port), timeout=600)
        except OSError:
            if time.time() - t0 > patience:
                raise
            time.sleep(0.2)


def add_blocks(image, block_size, blocks, pixels):
    """film/image.rs:36-50, with plain loops"""
    bw, bh = block_size
    stride = bw * bh * 4
    for i, (x0, y0) in enumerate(blocks):
        px = pixels[stride * i:stride * (i + 1)]
        for by in range(bh):
            for bx in range(bw):
                for c in range(4):
                    image[int(y0) + by, int(x0) + bx, c] += px[by * bw * 4 + bx * 4 + c]


def run_master(workers, scene_file, frames, img_dim, n_blocks):
    """workers: [(host, port)]. Returns {frame: H x W x 4 image} once every worker has reported every frame."""
    per, rem = n_blocks // len(workers), n_blocks % len(workers)
    conns = []
    for i, (host, port) in enumerate(workers):
        c = _connect(host, port)
        count = per + rem if i == len(workers) - 1 else per
        c.sendall(encode_instructions(scene_file, frames, i * per, count))
        conns.append(c)
    images, reports = {}, {}
    for c in conns:   # (the reference multiplexes with mio; the result does not depend on the order the frames arrive in)
        for _ in range(frames[0], frames[1] + 1):
            head = _read(c, 8)
            (size,) = struct.unpack("<Q", head)
            frame, bs, blocks, pixels = decode_frame(head + _read(c, size - 8))
            img = images.setdefault(frame, np.zeros((img_dim[1], img_dim[0], 4), np.float32))
            add_blocks(img, bs, blocks, pixels)
            reports[frame] = reports.get(frame, 0) + 1
        c.close()
    assert all(v == len(workers) for v in reports.values()) and len(reports) == frames[1] - frames[0] + 1
    return images
