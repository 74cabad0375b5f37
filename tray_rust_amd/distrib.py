"""Wire-compatible `--worker`: a process that speaks the reference's master protocol, so that an unmodified
`tray_rust scene.json --master host...` can farm slices of the Morton tile queue to an MI355X box.

Mirrors src/exec/distrib/{mod,worker}.rs and the worker_node loop of src/main.rs:148-166:

    Instructions, Frame            exec/distrib/mod.rs:52-100   the two messages, each prefixed by its own encoded size
    PORT                           exec/distrib/worker.rs:17    63234
    Worker.listen_for_master(...)  exec/distrib/worker.rs:37-47,59-90
    Worker.send_results()          exec/distrib/worker.rs:49-56 (RenderTarget::get_rendered_blocks, render_target.rs:215-241)
    worker_node(exec)              main.rs:148-166              render frame, send, clear, next frame; exit after the last one

Wire format: the reference serialises with bincode 0.9.2 (Cargo.lock) through `serialize(&msg, Infinite)`. That crate is not under
/root/reference; its published encoding is restated here: little-endian, fixed width, no field names; usize as u64; a String as
u64 length + UTF-8 bytes; a tuple as its fields; a Vec as u64 length + elements; f32 as its 4 IEEE bytes. Nothing in the
reference holds serialised bytes to pin this against and no Rust toolchain exists here to run its master: the byte layout is
**unpinned** (tests/test_distrib.py holds hand-assembled vectors and a restated master, tests/_ref_master.py).

The scene path in the instructions must be valid on this machine (the reference assumes a shared file system,
exec/distrib/mod.rs:29-30). Rendering goes through the `exec` object the caller passes (tray_rust_amd.Hip: libtrayhip.so on the
GPU); there is no CPU fallback."""
import socket
import struct
import time

import numpy as np

from . import Config, Scene

PORT = 63234   # exec/distrib/worker.rs:17


class WireError(Exception):
    pass


class _Reader:
    def __init__(self, data):
        self.data, self.pos = bytes(data), 0

    def take(self, n):
        if n < 0 or self.pos + n > len(self.data):
            raise WireError(f"message truncated: {n} bytes wanted at offset {self.pos} of {len(self.data)}")
        b = self.data[self.pos:self.pos + n]
        self.pos += n
        return b

    def u64(self):
        return struct.unpack("<Q", self.take(8))[0]

    def done(self):
        if self.pos != len(self.data):
            raise WireError(f"{len(self.data) - self.pos} bytes left over")


class Instructions:
    """exec/distrib/mod.rs:52-74: which scene, which frames (inclusive) and which slice (start, count) of the 8x8-tile Morton queue"""

    def __init__(self, scene, frames, block_start, block_count):
        self.scene, self.frames = str(scene), (int(frames[0]), int(frames[1]))
        self.block_start, self.block_count = int(block_start), int(block_count)
        self.encoded_size = len(self._body()) + 8   # serialized_size(&instr), the header included (mod.rs:69-71)

    def _body(self):
        s = self.scene.encode("utf-8")
        return struct.pack("<Q", len(s)) + s + struct.pack("<QQQQ", self.frames[0], self.frames[1], self.block_start, self.block_count)

    def encode(self):
        return struct.pack("<Q", self.encoded_size) + self._body()

    @staticmethod
    def decode(data):
        r = _Reader(data)
        size = r.u64()
        n = r.u64()
        try:
            scene = r.take(n).decode("utf-8")
        except UnicodeDecodeError as e:
            raise WireError(f"scene path is not UTF-8: {e}")
        frames = (r.u64(), r.u64())
        ins = Instructions(scene, frames, r.u64(), r.u64())
        r.done()
        if size != ins.encoded_size:
            raise WireError(f"encoded_size {size} does not match the message ({ins.encoded_size} bytes)")
        return ins

    def __repr__(self):   # #[derive(Debug)]: what the reference's worker prints (worker.rs:84)
        return (f'Instructions {{ encoded_size: {self.encoded_size}, scene: "{self.scene}", frames: ({self.frames[0]}, {self.frames[1]}), '
                f"block_start: {self.block_start}, block_count: {self.block_count} }}")


class Frame:
    """exec/distrib/mod.rs:79-100: one finished frame of one worker: the positions (in pixels) of the blocks it touched and their RGBW"""

    def __init__(self, frame, block_size, blocks, pixels):
        self.frame, self.block_size = int(frame), (int(block_size[0]), int(block_size[1]))
        self.blocks = np.ascontiguousarray(np.asarray(blocks, dtype="<u8").reshape(-1, 2))
        self.pixels = np.ascontiguousarray(np.asarray(pixels, dtype="<f4").reshape(-1))
        self.encoded_size = 8 + 8 + 16 + 8 + 16 * len(self.blocks) + 8 + 4 * len(self.pixels)

    def encode(self):
        return b"".join([struct.pack("<QQQQ", self.encoded_size, self.frame, self.block_size[0], self.block_size[1]),
                         struct.pack("<Q", len(self.blocks)), self.blocks.tobytes(),
                         struct.pack("<Q", len(self.pixels)), self.pixels.tobytes()])

    @staticmethod
    def decode(data):
        r = _Reader(data)
        size, frame, bw, bh = r.u64(), r.u64(), r.u64(), r.u64()
        nb = r.u64()
        blocks = np.frombuffer(r.take(16 * nb), dtype="<u8").reshape(-1, 2)
        npx = r.u64()
        pixels = np.frombuffer(r.take(4 * npx), dtype="<f4")
        r.done()
        f = Frame(frame, (bw, bh), blocks, pixels)
        if size != f.encoded_size:
            raise WireError(f"encoded_size {size} does not match the message ({f.encoded_size} bytes)")
        return f


MAX_INSTRUCTIONS_BYTES = 1 << 16   # Instructions: five u64 + a scene path; anything longer is not a master speaking


def read_message(stream, max_size=MAX_INSTRUCTIONS_BYTES):
    """one length-prefixed message off a socket: 8 bytes of size, then the rest (worker.rs:63-83, master.rs:166-192). The size a peer
    may announce is capped by what the caller expects (the worker only ever reads Instructions: a few hundred bytes; a reader of Frames
    passes 8 + 40 + blocks * (16 + 64) for its film) -- whoever connects to the listening port must not make the worker buffer gigabytes."""
    def read_exact(n):
        buf = bytearray()
        while len(buf) < n:
            chunk = stream.recv(min(n - len(buf), 1 << 20))
            if not chunk:
                raise WireError(f"connection closed after {len(buf)} of {n} bytes")
            buf += chunk
        return bytes(buf)
    head = read_exact(8)
    size = struct.unpack("<Q", head)[0]
    if size < 8 or size > max_size:
        raise WireError(f"implausible message size {size} (at most {max_size} bytes expected)")
    return head + read_exact(size - 8)


class Worker:
    """exec/distrib/worker.rs:23-57"""

    def __init__(self, instructions, render_target, scene, config, master):
        self.instructions, self.render_target, self.scene, self.config, self.master = instructions, render_target, scene, config, master

    @staticmethod
    def listen_for_master(num_threads=1, port=PORT, host="0.0.0.0", ready=None):
        """Blocks until the master connects and has sent its instructions, then loads the scene it names (worker.rs:37-47).
        `port` / `host` / `ready` (an Event set once the socket listens) exist for tests; the reference's port is fixed."""
        listener = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
        listener.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        try:
            listener.bind((host, port))
        except OSError as e:
            raise RuntimeError(f"Worker failed to get port: {e}")
        listener.listen(1)
        print(f"Worker listening for master on {listener.getsockname()[1]}", flush=True)
        if ready is not None:
            ready.set()
        try:
            master, _ = listener.accept()
        finally:
            listener.close()
        instructions = Instructions.decode(read_message(master))
        print(f"Received instructions: {instructions!r}", flush=True)
        scene, rt, spp, frame_info = Scene.load_file(instructions.scene)
        frame_info.start, frame_info.end = instructions.frames
        config = Config("/tmp", instructions.scene, spp, num_threads, frame_info, (instructions.block_start, instructions.block_count))
        return Worker(instructions, rt, scene, config, master)

    def send_results(self):
        """worker.rs:49-56"""
        block_size, blocks, pixels = self.render_target.get_rendered_blocks()
        frame = Frame(self.config.current_frame, block_size, blocks, pixels)
        try:
            self.master.sendall(frame.encode())
        except OSError as e:
            raise RuntimeError(f"Failed to send frame to {self.master}: {e}")

    def close(self):
        try:
            self.master.shutdown(socket.SHUT_RDWR)
        except OSError:
            pass
        self.master.close()


def worker_node(exec_, num_threads=1, port=PORT, host="0.0.0.0", ready=None):
    """main.rs:148-166 with `exec_` in the place of exec::MultiThreaded: wait for the master, render the frames it asks for, send each
    one back as soon as it is finished, exit after the last."""
    worker = Worker.listen_for_master(num_threads, port, host, ready)
    scene_start = time.time()
    try:
        for i in range(worker.config.frame_info.start, worker.config.frame_info.end + 1):
            worker.config.current_frame = i
            exec_.render(worker.scene, worker.render_target, worker.config)
            worker.send_results()
            worker.render_target.clear()
            print("--------------------", flush=True)
    finally:
        worker.close()
    print(f"Rendering entire sequence took {time.time() - scene_start:.4f}s", flush=True)
    return worker
