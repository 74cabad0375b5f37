"""Multi-GPU sharding of one frame: one process per GPU, tiles of the Morton queue dealt round-robin
in chunks, per-rank RGBW films merged by addition on rank 0.

Mirrors the reference's distributed mode (src/exec/distrib/master.rs:91-93,124-163,218-227: each
worker renders a slice of the block queue, the master sums the returned RGBW blocks,
src/film/image.rs:21-50) with the TCP transport replaced by one RCCL sum-reduce over xGMI
(`torch.distributed` backend "nccl" is RCCL on ROCm; "gloo" works for CPU tests)."""
import ctypes as C

from ._lib import check, lib

DEFAULT_CHUNK_TILES = 16


def shard_tiles(n_tiles, rank, world, chunk_tiles=DEFAULT_CHUNK_TILES):
    """Morton-queue indices rank `rank` of `world` renders (same mapping as tray_render_shard_device)."""
    n = C.c_uint32()
    check(lib().tray_shard_tiles(n_tiles, rank, world, chunk_tiles, None, 0, C.byref(n)))
    buf = (C.c_uint32 * max(n.value, 1))()
    check(lib().tray_shard_tiles(n_tiles, rank, world, chunk_tiles, buf, n.value, C.byref(n)))
    return list(buf[:n.value])


def merge_film(film, dst=0):
    """Sum the per-rank RGBW films into rank `dst` (film::Image::add_pixels semantics). No-op for one rank."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.reduce(film, dst=dst, op=dist.ReduceOp.SUM)
    return film


def render_frame_sharded(render_shard, film, rank, world, dst=0):
    """render_shard(rank, world, film) accumulates this rank's tiles into `film` (a torch tensor);
    afterwards rank `dst` holds the whole frame."""
    render_shard(rank, world, film)
    return merge_film(film, dst)


def shard_frames(start_frame, end_frame, rank, world):
    """Frames [start_frame, end_frame] dealt round-robin over the ranks (BASELINE.json configs[4]: an animation shards by
    frame first; frames are independent, src/main.rs:91-106, so no collective is needed -- each rank writes its own
    frames). With fewer frames than ranks, shard the tiles of each frame instead (render_frame_sharded)."""
    return list(range(int(start_frame) + int(rank), int(end_frame) + 1, int(world)))
