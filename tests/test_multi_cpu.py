"""world_size-2 tests of the multi-GPU path on CPU (gloo): shard partition + sum-merge reproduce the single-process frame.
The per-rank renderer is (a) the CPU oracle standing in for the HIP tile worker, and (b) the DEVICE code of the tile worker
itself in the host emulation (tests/_emu.py), launched with the shard arguments of tray_render_shard_device -- so the index map
inside k_path_tiles (work item -> entry of the Morton queue) is what the two ranks execute, through the same
multi.render_frame_sharded closure bench.py drives."""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, scene_path, out_path, renderer="oracle"):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    import tray_rust_amd as T
    from tray_rust_amd import multi
    import _oracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    scene, rt, spp, fi = T.Scene.load_file(scene_path)
    flat = scene.flatten(0)
    n_tiles = (rt.width // 8) * (rt.height // 8)

    def render_shard(r, w, film):
        acc = np.zeros((rt.height, rt.width, 4), np.float32)
        if renderer == "emu":   # k_path_tiles itself, given (shard, n_shards, chunk_tiles) like tray_render_shard_device gives them to the GPU
            import _emu as E
            tiles = np.array(T.BlockQueue((rt.width, rt.height), (8, 8)).blocks, np.uint32).reshape(-1, 2)
            acc, _ = E.render_tiles(flat, tiles, spp, 11, blocks=2, shard=(r, w, 3))
        elif renderer == "emu_adaptive":   # sampler::Adaptive(2, 16): k_sampler_pass / k_sampler_decide over the rank's tiles (every number keyed by pixel, round, index)
            import _emu as E
            tiles = np.array(T.BlockQueue((rt.width, rt.height), (8, 8)).blocks, np.uint32).reshape(-1, 2)
            mine = tiles[multi.shard_tiles(n_tiles, r, w, chunk_tiles=3)]
            acc, _ = E.render_sampler(flat, mine, O.SAMPLER_ADAPTIVE, 2, 16, seed=11)
        else:
            for t in multi.shard_tiles(n_tiles, r, w, chunk_tiles=3):
                img, _ = O.render_tiles(flat, spp, seed=11, tile_start=t, tile_count=1, threads=1)
                acc += img
        film += torch.from_numpy(acc.reshape(-1))

    film = torch.zeros(rt.width * rt.height * 4, dtype=torch.float32)
    multi.render_frame_sharded(render_shard, film, rank, world, dst=0)
    if rank == 0:
        np.save(out_path, film.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("renderer", ["oracle", "emu", "emu_adaptive"])
def test_two_rank_sharded_frame_equals_single_process(renderer, tmp_path, built):
    import torch.multiprocessing as mp
    import tray_rust_amd as T
    from tray_rust_amd import scenes
    import _oracle as O
    scenes.write_assets(str(tmp_path))
    scene_path = os.path.join(str(tmp_path), "s.json")
    json.dump(scenes.cornell_box(48, 32, 8), open(scene_path, "w"))
    out_path = os.path.join(str(tmp_path), "merged.npy")
    port = 29500 + os.getpid() % 2000
    if renderer != "oracle":
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import _emu as E
        E.emu()   # build the emulation library once, before two processes race for it
    mp.spawn(_worker, args=(2, port, scene_path, out_path, renderer), nprocs=2, join=True)
    merged = np.load(out_path).reshape(32, 48, 4)
    scene, *_ = T.Scene.load_file(scene_path)
    if renderer == "emu_adaptive":   # the same Sampler in ONE process: the oracle's thread_work over the whole block queue
        whole, st, counts = O.render_tiles_sampler(scene.flatten(0), O.SAMPLER_ADAPTIVE, 2, 16, seed=11)
        assert counts.max() > 2
        np.testing.assert_allclose(merged, whole, rtol=5e-5, atol=5e-5)
        return
    whole, _ = O.render_tiles(scene.flatten(0), 8, seed=11)
    # (the emulated tile kernel bins the film by rows: same sums grouped differently, a few ulps of the largest pixel value)
    diff = float(np.abs(merged - whole).max())
    assert diff <= (3e-6 if renderer == "oracle" else 1e-5 * float(whole.max())), (diff, float(whole.max()))
    if renderer == "emu":   # and against the same device code run as ONE shard: only the order of the partial sums differs
        import _emu as E
        tiles = np.array(T.BlockQueue((48, 32), (8, 8)).blocks, np.uint32).reshape(-1, 2)
        one, _ = E.render_tiles(scene.flatten(0), tiles, 8, 11, blocks=2)
        assert float(np.abs(merged - one).max()) <= 2e-6 * float(one.max())
    assert (merged[..., 3] > 0).all()


def test_shard_lists_are_disjoint_and_balanced(built):
    from tray_rust_amd import multi
    n = 32400
    shards = [multi.shard_tiles(n, r, 8) for r in range(8)]
    assert sorted(sum(shards, [])) == list(range(n))
    sizes = [len(s) for s in shards]
    assert max(sizes) - min(sizes) <= 16


def test_frames_shard_round_robin():
    from tray_rust_amd import multi
    world = 8
    parts = [multi.shard_frames(0, 127, r, world) for r in range(world)]
    assert sorted(sum(parts, [])) == list(range(128)) and all(len(p) == 16 for p in parts)
    assert multi.shard_frames(5, 6, 3, 8) == [] and multi.shard_frames(5, 6, 1, 8) == [6]


def _frame_worker(rank, world, port, scene_path, out_dir, first, last):
    """BASELINE.json configs[4] on two ranks: the frames of a sequence dealt round-robin (multi.shard_frames), every rank renders WHOLE
    frames with the tile worker's device code (host emulation) and writes them itself -- frames are independent (main.rs:91-106), there is
    no collective on the data path; the process group only provides the ranks and the final barrier."""
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    import tray_rust_amd as T
    from tray_rust_amd import multi
    import _emu as E
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    scene, rt, spp, fi = T.Scene.load_file(scene_path)
    tiles = np.array(T.BlockQueue((rt.width, rt.height), (8, 8)).blocks, np.uint32).reshape(-1, 2)
    mine = multi.shard_frames(first, last, rank, world)
    for fr in mine:
        img, _ = E.render_tiles(scene.flatten(fr), tiles, spp, 11, blocks=2)
        np.save(os.path.join(out_dir, f"frame{fr}_rank{rank}.npy"), img)
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_render_the_frames_of_a_sequence(tmp_path, built):
    """frames f .. f+2 of the moving test scene over two ranks: rank 0 gets f and f+2, rank 1 gets f+1; every frame equals the one a single
    process renders with the same device code, bit for bit (nothing is merged), and no frame is rendered twice or left out"""
    import glob
    import torch.multiprocessing as mp
    import tray_rust_amd as T
    from tray_rust_amd import scenes
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _emu as E
    d = str(tmp_path)
    scenes.write_moving_box(d, width=32, height=24, samples=4)
    scene_path = os.path.join(d, "moving_box.json")
    E.emu()
    first, last = 2, 4
    port = 31500 + os.getpid() % 2000
    mp.spawn(_frame_worker, args=(2, port, scene_path, d, first, last), nprocs=2, join=True)
    files = sorted(os.path.basename(f) for f in glob.glob(os.path.join(d, "frame*_rank*.npy")))
    assert files == ["frame2_rank0.npy", "frame3_rank1.npy", "frame4_rank0.npy"]
    scene, rt, spp, fi = T.Scene.load_file(scene_path)
    tiles = np.array(T.BlockQueue((rt.width, rt.height), (8, 8)).blocks, np.uint32).reshape(-1, 2)
    frames = []
    for fr, name in zip(range(first, last + 1), files):
        flat = scene.flatten(fr)
        assert flat.contents.animated
        one, _ = E.render_tiles(flat, tiles, spp, 11, blocks=2)
        got = np.load(os.path.join(d, name))
        assert got.tobytes() == one.tobytes(), f"frame {fr}"
        assert (got[..., 3] > 0).all()
        frames.append(got)
    assert np.abs(frames[0] - frames[1]).max() > 1e-3      # the box moves: consecutive frames differ


# ---- the same two-rank flow with the REAL per-rank renderer (tray_render_shard_device) and RCCL: needs two GPUs ----
def _gpu_worker(rank, world, port, scene_path, out_path):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import tray_rust_amd as T
    from tray_rust_amd import multi
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    scene, rt, spp, fi = T.Scene.load_file(scene_path)
    hip = T.Hip(device=rank, seed=11)
    film = torch.zeros(rt.width * rt.height * 4, dtype=torch.float32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    multi.render_frame_sharded(lambda r, w, f: hip.render_shard_device(scene, 0, r, w, T.round_spp(spp), f.data_ptr(), chunk_tiles=3, stream=stream),
                               film, rank, world, dst=0)
    torch.cuda.synchronize()
    if rank == 0:
        np.save(out_path, film.cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_two_gpu_ranks_shard_and_rccl_reduce(tmp_path):
    """bench.py's N > 1 path on hardware: one process per GPU, tray_render_shard_device per rank, ONE RCCL sum-reduce to rank 0.
    Runs where the box has two GPUs (the driver's 8-GPU node); skipped on the 1-GPU boxes."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    import torch.multiprocessing as mp
    import tray_rust_amd as T
    from tray_rust_amd import scenes
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _oracle as O
    scenes.write_assets(str(tmp_path))
    scene_path = os.path.join(str(tmp_path), "s.json")
    json.dump(scenes.cornell_box(160, 96, 16), open(scene_path, "w"))
    out_path = os.path.join(str(tmp_path), "merged.npy")
    mp.spawn(_gpu_worker, args=(2, 29500 + os.getpid() % 2000, scene_path, out_path), nprocs=2, join=True)
    merged = np.load(out_path).reshape(96, 160, 4)
    scene, *_ = T.Scene.load_file(scene_path)
    whole, _ = O.render_tiles(scene.flatten(0), 16, seed=11)
    rgb = lambda i: i[..., :3] / np.maximum(i[..., 3:], 1e-20)
    assert (merged[..., 3] > 0).all() and float(np.sqrt(np.mean((rgb(merged) - rgb(whole)) ** 2))) < 1e-4
