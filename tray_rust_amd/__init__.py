"""tray_rust_amd — MI355X-native tile worker for tray_rust scenes.

Host-side mirror of the reference's interface for the hot path (names, argument meaning and error
behaviour follow /root/reference):

    Scene.load_file(path) -> (scene, rt, spp, frame_info)      src/scene.rs:101-145
    FrameInfo, Config                                           src/film/mod.rs:19-36, src/exec/mod.rs:17-38
    Hip().render(scene, rt, config)                             trait Exec, src/exec/mod.rs:41-49
    RenderTarget.get_renderf32 / get_render / clear             src/film/render_target.rs:168-266
    RenderTarget.get_rendered_blocks / add_blocks               src/film/render_target.rs:215-241, src/film/image.rs:36-50
    distrib.worker_node(Hip()) / `python -m tray_rust_amd --worker`   src/exec/distrib/worker.rs, src/main.rs:148-166
    BlockQueue(img, dim, select_blocks)                         src/sampler/block_queue.rs:11-66
    sampler.LowDiscrepancy / Uniform / Adaptive                 src/sampler/{ld,uniform,adaptive}.rs (Hip(sampler=...))

Everything below the Python layer is libtrayhip.so (include/trayhip.h): a C++ scene loader and
hand-written HIP kernels for gfx950. The reference panics on invalid input; here the same
preconditions raise TrayError. There is no CPU fallback: without the library / a GPU the calls fail.
"""
import ctypes as C
import os

import numpy as np

from . import _lib
from . import sampler
from ._lib import TrayError, lib, check

__all__ = ["Scene", "FrameInfo", "Config", "RenderTarget", "Hip", "BlockQueue", "TrayError", "round_spp", "sampler"]


def round_spp(spp):
    """LowDiscrepancy::new rounds spp up to a power of two (src/sampler/ld.rs:22-25)."""
    r = int(lib().tray_round_spp(int(spp)))
    if r != spp:
        print(f"Warning: LowDiscrepancy sampler requires power of two samples per pixel, rounding up to {r}")
    return r


class FrameInfo:
    """src/film/mod.rs:19-36"""

    def __init__(self, frames, time, start, end):
        self.frames, self.time, self.start, self.end = int(frames), float(time), int(start), int(end)

    def __repr__(self):
        return f"FrameInfo(frames={self.frames}, time={self.time}, start={self.start}, end={self.end})"


class Config:
    """src/exec/mod.rs:17-38; select_blocks = (start, count) into the Morton tile queue, count 0 = all."""

    def __init__(self, out_path, scene_file, spp, num_threads, frame_info, select_blocks=(0, 0)):
        self.out_path, self.scene_file, self.spp, self.num_threads = out_path, scene_file, int(spp), int(num_threads)
        self.frame_info, self.current_frame, self.select_blocks = frame_info, frame_info.start, tuple(select_blocks)


class BlockQueue:
    """src/sampler/block_queue.rs:28-48: 8x8 tiles in Morton order, optional (start, count) sub-range."""

    def __init__(self, img, dim=(8, 8), select_blocks=(0, 0)):
        if tuple(dim) != (8, 8):
            raise TrayError(_lib.TRAY_E_UNSUPPORTED, "only (8, 8) blocks are supported (exec/multithreaded.rs:32)")
        n = C.c_uint32(0)
        check(lib().tray_block_queue(img[0], img[1], select_blocks[0], select_blocks[1], None, 0, C.byref(n)))
        buf = (C.c_uint32 * (2 * n.value))()
        check(lib().tray_block_queue(img[0], img[1], select_blocks[0], select_blocks[1], buf, n.value, C.byref(n)))
        self.blocks = [(buf[2 * i], buf[2 * i + 1]) for i in range(n.value)]
        self.dimensions = (8, 8)
        if not self.blocks:
            print("Warning: This block queue is empty!")

    def block_dim(self):
        return self.dimensions

    def __len__(self):
        return len(self.blocks)

    def __iter__(self):
        return iter(self.blocks)


class RenderTarget:
    """Accumulated RGBW f32 film in the layout of RenderTarget::get_renderf32
    (src/film/render_target.rs:243-266). The GPU accumulates on the device; this host object merges
    results by addition like film::Image::add_pixels (src/film/image.rs:21-34)."""

    def __init__(self, width, height):
        self.width, self.height = int(width), int(height)
        self.pixels = np.zeros(self.width * self.height * 4, dtype=np.float32)

    def dimensions(self):
        return (self.width, self.height)

    def clear(self):
        self.pixels[:] = 0.0

    def add_pixels(self, rgbw):
        self.pixels += np.asarray(rgbw, dtype=np.float32).reshape(-1)

    def get_renderf32(self):
        return self.pixels.copy()

    lock_size = (2, 2)   # the reference's lock blocks, hard-coded by its loader (scene.rs:224)

    def get_rendered_blocks(self):
        """render_target.rs:215-241: (block size, positions in pixels of the 2x2 blocks whose FOUR pixels all have a non-zero
        weight, their RGBW one block after the other) -- what a distributed worker sends to the master. Blocks in row order,
        pixels row-major inside a block. (A block with an untouched pixel is left out whole, as in the reference.)"""
        bw, bh = self.lock_size
        xb, yb = self.width // bw, self.height // bh
        img = self.pixels.reshape(self.height, self.width, 4)[:yb * bh, :xb * bw]
        tiles = img.reshape(yb, bh, xb, bw, 4).transpose(0, 2, 1, 3, 4)          # [by][bx][y][x][rgbw]
        full = (tiles[..., 3] != 0.0).all(axis=(2, 3))
        by, bx = np.nonzero(full)
        blocks = np.stack([bx * bw, by * bh], axis=1).astype(np.uint64)
        return (bw, bh), blocks, np.ascontiguousarray(tiles[by, bx]).reshape(-1)

    def add_blocks(self, block_size, blocks, pixels):
        """film::Image::add_blocks (src/film/image.rs:36-50): what the master does with a worker's blocks"""
        bw, bh = int(block_size[0]), int(block_size[1])
        img = self.pixels.reshape(self.height, self.width, 4)
        px = np.asarray(pixels, dtype=np.float32).reshape(-1, bh, bw, 4)
        for k, (x, y) in enumerate(np.asarray(blocks, dtype=np.int64).reshape(-1, 2)):
            img[y:y + bh, x:x + bw] += px[k]

    def get_render(self):
        """sRGB8, 3 bytes per pixel (render_target.rs:185-210)."""
        out = np.zeros(self.width * self.height * 3, dtype=np.uint8)
        check(lib().tray_resolve_srgb8(self.pixels.ctypes.data, self.width, self.height, out.ctypes.data))
        return out


class Scene:
    """Loaded scene (host side) + its device copy for the current frame."""

    def __init__(self, handle):
        self._h = handle
        self._dev = None
        self._dev_frame = None
        self._dev_device = None
        info = _lib.TraySceneInfo()
        check(lib().tray_host_scene_info(self._h, C.byref(info)))
        self.info = info

    @staticmethod
    def load_file(path):
        """Scene::load_file -> (scene, render_target, spp, frame_info)"""
        h = C.c_void_p()
        check(lib().tray_scene_load_file(os.fsencode(path), C.byref(h)))
        return Scene._finish(h)

    @staticmethod
    def load_string(text, base_dir=""):
        h = C.c_void_p()
        check(lib().tray_scene_load_string(text.encode("utf-8"), os.fsencode(base_dir), C.byref(h)))
        return Scene._finish(h)

    @staticmethod
    def _finish(h):
        s = Scene(h)
        i = s.info
        return s, RenderTarget(i.width, i.height), int(i.spp), FrameInfo(i.frames, i.scene_time, i.start_frame, i.end_frame)

    def flatten(self, frame=0):
        """Scene::update_frame + lowering to the TrayFlatScene POD. The view borrows from this scene -- it is valid until the next
        flatten() / close() -- so the returned pointer keeps the scene alive (`T.Scene.load_file(p)[0].flatten(0)` must not dangle)."""
        p = C.POINTER(_lib.TrayFlatScene)()
        check(lib().tray_host_scene_flatten(self._h, int(frame), C.byref(p)))
        p._scene = self
        return p

    def device_scene(self, frame=0, device=None):
        if self._dev is not None and device is not None and device != self._dev_device:
            self.release_device()
        if self._dev is not None and self._dev_frame != frame:
            # Scene::update_frame (scene.rs:152-176): the device copy moves to the new frame, meshes / tables / pools stay where they are
            try:
                check(lib().tray_scene_update_frame(self._dev, self.flatten(frame)))
            except TrayError:
                self.release_device()   # a failed update leaves a handle that can only be destroyed
                raise
            self._dev_frame = frame
        if self._dev is None:
            flat = self.flatten(frame)   # host work first: a scene that cannot be flattened fails here, with or without a GPU
            if device is not None:
                check(lib().tray_init(int(device)))
            d = C.c_void_p()
            check(lib().tray_scene_create(flat, C.byref(d)))
            self._dev, self._dev_frame, self._dev_device = d, frame, device
        return self._dev

    def release_device(self):
        if self._dev is not None:
            lib().tray_scene_destroy(self._dev)
            self._dev = None

    def close(self):
        self.release_device()
        if self._h is not None:
            lib().tray_host_scene_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Hip:
    """Execution backend in the place of exec::MultiThreaded (src/exec/multithreaded.rs:20-70): renders
    config.select_blocks of frame config.current_frame on one MI355X and adds the result into rt."""

    def __init__(self, device=0, seed=1, sampler=None):
        """sampler: callable (block_dim, spp) -> sampler.LowDiscrepancy / Uniform / Adaptive -- what thread_work constructs per worker
        (exec/multithreaded.rs:74); None = LowDiscrepancy::new(block_dim, spp)."""
        self.device, self.seed = int(device), int(seed)
        self.sampler = sampler
        check(lib().tray_init(self.device))
        self.last_timing = None
        self._multi, self._multi_key, self._multi_frame, self._multi_scene = None, None, None, None

    def _select_sampler(self, dev, spp, multi=False):
        """tells the device scene which Sampler the next render stands for; returns the spp the render call takes"""
        if self.sampler is None:
            smp_kind, lo, hi, spp = sampler.LOW_DISCREPANCY, 1, 1, round_spp(spp)
        else:
            smp = self.sampler((8, 8), spp)     # BlockQueue::block_dim (main.rs:101: (8, 8))
            smp_kind, lo, hi, spp = smp.KIND, smp.min_spp, smp.max_spp_, smp.spp
        check((lib().tray_multi_set_sampler if multi else lib().tray_scene_set_sampler)(dev, smp_kind, lo, hi))
        return spp

    def render(self, scene, rt, config):
        dev = scene.device_scene(config.current_frame, self.device)
        spp = self._select_sampler(dev, config.spp)
        start, count = config.select_blocks
        check(lib().tray_render_tiles(dev, int(start), int(count), spp, self.seed, rt.pixels.ctypes.data))
        t = _lib.TrayKernelTiming()
        if lib().tray_last_timing(dev, C.byref(t)) == _lib.TRAY_OK:
            self.last_timing = t
            print(f"Frame {config.current_frame}: rendering took {t.render_ms * 1e-3:.4f}s")

    def render_device(self, scene, frame, select_blocks, spp, rgbw_ptr, stream=None):
        """Asynchronous variant: accumulate into a device RGBW buffer (e.g. torch tensor .data_ptr())."""
        dev = scene.device_scene(frame, self.device)
        spp = self._select_sampler(dev, spp)
        check(lib().tray_render_tiles_device(dev, int(select_blocks[0]), int(select_blocks[1]), int(spp), self.seed,
                                             C.c_void_p(int(rgbw_ptr)), C.c_void_p(int(stream)) if stream else None))
        return dev

    def render_shard_device(self, scene, frame, shard, n_shards, spp, rgbw_ptr, chunk_tiles=16, stream=None):
        """One rank's share of a frame (round-robin chunks of the Morton queue); merge = sum over ranks."""
        dev = scene.device_scene(frame, self.device)
        spp = self._select_sampler(dev, spp)
        check(lib().tray_render_shard_device(dev, int(shard), int(n_shards), int(chunk_tiles), int(spp), self.seed,
                                             C.c_void_p(int(rgbw_ptr)), C.c_void_p(int(stream)) if stream else None))
        return dev

    def render_multi(self, scene, rt, config, devices):
        """One frame on several GPUs of this process: tiles sharded round-robin over `devices`, the per-device films summed onto
        the first one by RCCL inside the library (tray_render_frame_multi; the master's Image::add_blocks merge,
        exec/distrib/master.rs:124-163, film/image.rs:36-50), the result added into rt. Returns (per-device timings, reduce ms).
        The per-device scenes and the communicators are kept between calls: another frame of the same scene on the same devices
        is a tray_multi_update_frame (scene.rs:152-176), not a new ncclCommInitAll. close_multi() releases them."""
        key = tuple(int(d) for d in devices)
        # the cached device copies belong to ONE scene object, held by a strong reference and compared by identity: an id() alone
        # could be reused by another Scene allocated at the same address after this one was collected, and that scene's frame would
        # then be grafted onto the old one's meshes and tables by tray_multi_update_frame
        if self._multi is not None and (self._multi_scene is not scene or self._multi_key != key):
            self.close_multi()
        flat = scene.flatten(config.current_frame)
        if self._multi is None:
            ids = (C.c_int * len(devices))(*[int(d) for d in devices])
            m = C.c_void_p()
            check(lib().tray_multi_create(flat, len(devices), ids, C.byref(m)))
            self._multi, self._multi_key, self._multi_frame, self._multi_scene = m, key, config.current_frame, scene
        elif self._multi_frame != config.current_frame:
            try:
                check(lib().tray_multi_update_frame(self._multi, flat))
            except TrayError:
                self.close_multi()
                raise
            self._multi_frame = config.current_frame
        spp = self._select_sampler(self._multi, config.spp, multi=True)
        check(lib().tray_render_frame_multi(self._multi, spp, self.seed, rt.pixels.ctypes.data))
        per = (_lib.TrayKernelTiming * len(devices))()
        ms = C.c_float()
        check(lib().tray_multi_timing(self._multi, per, C.byref(ms)))
        return list(per), float(ms.value)

    def close_multi(self):
        if self._multi is not None:
            lib().tray_multi_destroy(self._multi)
            self._multi = None
        self._multi_scene = None

    def __del__(self):
        try:
            self.close_multi()
        except Exception:
            pass

    def schedule(self, scene):
        """the schedule the last render call of `scene` on this device ran with (tray_last_schedule): pool slots, views, slices, bytes"""
        info = _lib.TrayScheduleInfo()
        check(lib().tray_last_schedule(scene.device_scene(scene._dev_frame, self.device), C.byref(info)))
        return {name: int(getattr(info, name)) for name, _ in info._fields_}

    def set_wavefront(self, scene, pool_slots=0, views=0, slices=0):
        """tray_scene_set_wavefront on the scene's device copy (0 = the library's own rule)"""
        check(lib().tray_scene_set_wavefront(scene.device_scene(scene._dev_frame, self.device), int(pool_slots), int(views), int(slices)))

    def set_transform_table(self, scene, mode=-1):
        """tray_scene_set_transform_table: 1 = the frame's table of transforms by shutter-time index, 0 = per-path evaluation, -1 = by sample count"""
        check(lib().tray_scene_set_transform_table(scene.device_scene(scene._dev_frame, self.device), int(mode)))

    def timing(self, scene):
        t = _lib.TrayKernelTiming()
        check(lib().tray_last_timing(scene.device_scene(scene._dev_frame, self.device), C.byref(t)))
        self.last_timing = t
        return t
