#!/usr/bin/env python
"""bench.py — Msamples/s of the tray_rust tile worker on MI355X (BASELINE.json metric).

A step = one pass of the hot path over one frame of BASELINE.json configs[1]:
scenes/cornell_box.json at 1920x1080, 1024 spp (2.123e9 camera samples), inputs resident in HBM.
With N GPUs the frame's tiles are sharded round-robin over the ranks (one process per GPU) and the
per-rank RGBW buffers are merged with one RCCL sum-reduce to rank 0 inside the timed region
(strong scaling of one frame, the reference's distributed mode: exec/distrib/master.rs:91-93).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0 (contract in the task statement), with `roofline` (HBM-bound
accounting of SURVEY 8d: 368 B per path vertex + 16 B per pixel) and `cpu_baseline` (the C++
oracle in faithful-transform mode on all host cores, bounded tile sample, N=1 only).
"""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WIDTH, HEIGHT, SPP = 1920, 1080, 1024
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8 TB/s HBM3E
BYTES_PER_VERTEX = 368.0   # SURVEY 8(d) wavefront accounting
BYTES_PER_PIXEL = 16.0
WORKLOAD_CONFIG = {"cornell_box": 1, "smallpt": 2, "dragon": 3, "tr15_like": 4}


def cpu_baseline(flat, spp, target_seconds=8.0):
    """Oracle ('port' of the reference incl. the per-intersection transform rebuild of geometry/receiver.rs:30) on all host
    cores over every k-th tile of the Morton queue. The per-sample cost of the path tracer does not depend on the sample
    count, so the sample runs at min(spp, 64) spp over MANY tiles (>= 16 per thread): with one expensive tile per thread
    the slowest tile would set the time."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _oracle as O
    cores = os.cpu_count() or 1
    n_tiles = (WIDTH // 8) * (HEIGHT // 8)
    cpu_spp = min(int(spp), 64)
    # probe: 4 tiles per thread at 16 spp
    stride = max(1, n_tiles // (4 * cores))
    _, st = O.render_tiles(flat, 16, seed=1, stride=stride, threads=cores, flags=O.FAITHFUL_XF)
    rate = st.samples / max(st.seconds, 1e-9)
    want_tiles = int(rate * target_seconds / (64 * cpu_spp))
    want_tiles = max(16 * cores, min(n_tiles, want_tiles))
    stride = max(1, n_tiles // want_tiles)
    _, st = O.render_tiles(flat, cpu_spp, seed=1, stride=stride, threads=cores, flags=O.FAITHFUL_XF)
    tiles = (n_tiles + stride - 1) // stride
    return {
        "value": round(st.samples / st.seconds / 1e6, 4), "unit": "Msamples/s", "cores": cores, "kind": "port",
        "sample": f"every {stride}th 8x8 tile of the Morton queue ({tiles} tiles, {st.samples} samples at {cpu_spp} spp) in {st.seconds:.1f}s; "
                  "C++ oracle, faithful per-intersection transforms, -O3 -ffp-contract=off",
        "vertices_per_sample": round(st.vertices / max(st.samples, 1), 4),
    }


def device_code_hash():
    """md5 of the gfx950 code objects of the library this process runs (tools/device_code_hash.sh), or None"""
    import subprocess
    try:
        return subprocess.run([os.path.join(ROOT, "tools", "device_code_hash.sh")], capture_output=True, text=True, check=True, timeout=60).stdout.strip() or None
    except Exception:
        return None


# VALU peak, MEASURED (profiles/r04_ubench_valu.txt, tools/ubench_valu.hip): under a pure VALU load the shader clock holds 2.24 - 2.40 GHz
# (s_memtime against the 100 MHz s_memrealtime) and a SIMD retires one plain wave64 instruction (v_mul_f32 / v_add_f32 / v_mov_b32 /
# v_xor_b32) every 3.8 cycles at >= 4 waves per SIMD (4.05 at 2; v_fma_f32 3.05; the guide's nominal figure is 2) -- 64 lanes each.
# Packed f32 (v_pk_mul_f32 / v_pk_add_f32: two f32 per lane) costs 4.5 - 4.9 cycles at >= 4 waves, 5.9 at 2: 0.6 - 0.73 of the plain price
# per f32. The peak below is the plain rate; a kernel made of packed instructions could exceed it by up to 1.6x.
VALU_CLOCK_GHZ = 2.33
VALU_CYCLES_PER_WAVE_INSTRUCTION = 3.8
VALU_PEAK_TLANE = 256 * 4 * 64 * VALU_CLOCK_GHZ * 1e9 / VALU_CYCLES_PER_WAVE_INSTRUCTION / 1e12   # 40.2e12 lane-instructions/s
VALU_PEAK_NOMINAL_TLANE = 256 * 4 * 64 * 2.4e9 / 2.0 / 1e12   # 78.6: MI355X_MICROARCH.md's SIMD-32 figure (a wave64 instruction every 2 cycles at 2.4 GHz)
VALU_PEAK_NOTE = ("peak = MEASURED issue rate of plain VALU instructions: 256 CUs x 4 SIMDs x 64 lanes per 3.8 cycles at 2.33 GHz under load "
                  "(profiles/r04_ubench_valu.txt; the nominal 2 cycles per wave64 instruction at 2.4 GHz would be 78.6)")


def pmc_views(workload, samples, kernel_seconds):
    """roofline.traffic (HBM-side bytes per launch), the VALU view and the compute view of a workload's kernels from the committed
    counter passes (profiles/pmc_latest.json, tools/pmc_workloads.py). The counters belong to ONE build: they are used only when
    the md5 of this library's device code equals the one recorded with them; a different build prints nulls and says why.
    The passes profile a cut-down sample count; HBM-side bytes and instruction counts scale with the sample count of the launch.
    valu.achieved = VALU wave-instructions x 64 x lane utilisation (= active lane-instructions, from SQ_INSTS_VALU and
    SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU) per sample x this launch's samples / this launch's HIP-event seconds."""
    path = os.path.join(ROOT, "profiles", "pmc_latest.json")
    have = device_code_hash()
    static = None   # static registers of the tile kernel: only those of THIS device code (tools/static_registers.sh records the hash)
    try:
        static = json.load(open(os.path.join(ROOT, "profiles", "static_registers_latest.json")))
        if not have or static.get("device_code_hash") != have:
            static = {"note": f"profiles/static_registers_latest.json belongs to device code {static.get('device_code_hash')}, this library is {have}: "
                              "re-run tools/static_registers.sh"}
    except Exception:
        pass
    try:
        p = json.load(open(path))
    except Exception:
        return None, None, {"source": None, "note": "no profiles/pmc_latest.json", "static": static}
    if not have or have != p.get("device_code_hash"):
        return None, None, {"source": "profiles/pmc_latest.json", "static": static,
                            "note": f"counters belong to device code {p.get('device_code_hash')}, this library is {have}: re-run tools/pmc_workloads.py"}
    d = (p.get("workloads") or {}).get(workload)
    if not d:
        return None, None, {"source": "profiles/pmc_latest.json", "note": f"no counters for workload {workload}", "static": static}
    per_sample = d.get("hbm_bytes_per_sample")
    traffic = int(per_sample * samples) if per_sample else None
    valu = None
    if d.get("valu_instructions_per_sample") and d.get("valu_lane_utilisation") and kernel_seconds > 0:
        lane_instr = d["valu_instructions_per_sample"] * 64.0 * d["valu_lane_utilisation"] * samples
        achieved = lane_instr / kernel_seconds / 1e12
        valu = {"achieved": round(achieved, 3), "peak": round(VALU_PEAK_TLANE, 2), "unit": "T lane-instructions/s", "frac": round(achieved / VALU_PEAK_TLANE, 4),
                "peak_nominal": round(VALU_PEAK_NOMINAL_TLANE, 2), "frac_nominal": round(achieved / VALU_PEAK_NOMINAL_TLANE, 4),
                "wave_instructions_per_sample": round(d["valu_instructions_per_sample"], 1), "lane_utilisation": round(d["valu_lane_utilisation"], 4),
                "note": VALU_PEAK_NOTE + (f"; the SQ counters show the VALU pipe {min(1.0, d.get('valu_busy', 0.0)):.2f} time-busy" if d.get("valu_busy") else "")}
    compute = {"source": "profiles/pmc_latest.json (rocprofv3 --pmc passes around a %d-spp launch of this device code, tools/pmc_workloads.py)" % d.get("spp", 0),
               "device_code_hash": have, "kernel": d.get("kernel"),
               # (SQ_ACTIVE_INST_VALU x SQ_WAVES / (SIMDs x SQ_WAVE_CYCLES): an estimate -- counters of separate passes, waves that do not all live the whole launch --
               # which can come out a few per cent above 1 for an issue-bound kernel; reported as measured, the note above clamps it)
               "valu_busy": d.get("valu_busy"), "valu_lane_util": d.get("valu_lane_utilisation"),
               "waves_per_simd": d.get("waves_per_simd"), "waiting_share_of_wave_cycles": d.get("waiting_share_of_wave_cycles"),
               "dominant_kernel_share_of_time": d.get("dominant_kernel_share_of_time"),
               "hbm_bytes_basis": d.get("hbm_bytes_basis"), "static": static if workload == "cornell_box" else None,
               "schedule_of_the_counter_launch": d.get("schedule")}
    return traffic, valu, compute


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--spp", type=int, default=0, help="debug only: the reported config is 1024 spp")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-workloads", action="store_true", help="only the reported line (configs[1]), without the configs[2..4] lines attached to it")
    ap.add_argument("--workload", choices=["cornell_box", "smallpt", "dragon", "tr15_like"], default="cornell_box",
                    help="cornell_box = BASELINE.json configs[1] (the reported line); smallpt = configs[2] at 4096 spp; "
                         "dragon = configs[3] stand-in (871 200 triangles + MERL) at 2048 spp; tr15_like = configs[4] stand-in "
                         "(59 instances, 3.1 M triangles, moving camera / objects / lights), one frame at 512 spp")
    ap.add_argument("--frame", type=int, default=63, help="first frame of a moving workload (tr15_like). BASELINE.json configs[4] is frames 0..127 (main.rs:91-106); of the stand-in's "
                    "59 instances 8 move within a frame up to frame 63 and 11 from 64 on (rounds 2-4 benched frames 330-331, where 2 move)")
    ap.add_argument("--frames", type=int, default=1, help="tr15_like only: a step renders the SEQUENCE of frames [frame, frame + frames) -- the device scene moves from "
                    "frame to frame with tray_scene_update_frame (Scene::update_frame, scene.rs:152-176) inside the timed region; with N GPUs the frames are "
                    "dealt round-robin over the ranks (multi.shard_frames, BASELINE.json configs[4]), each rank renders whole frames, no collective")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    import tray_rust_amd as T
    from tray_rust_amd import multi, scenes

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and args.gpus > 1:
        if "WORLD_SIZE" in os.environ or "RANK" in os.environ:
            raise SystemExit(f"--gpus {args.gpus} but the launcher started WORLD_SIZE={world}")
        # started plainly (`python bench.py --gpus N`): become the launcher of N ranks, one per GPU, on this node
        import socket
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        os.execvp(sys.executable, [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
                                   "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:])
    distributed = world > 1
    torch.cuda.set_device(local_rank)
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    stream = torch.cuda.current_stream().cuda_stream
    hip = T.Hip(device=local_rank, seed=1)
    film = torch.zeros(WIDTH * HEIGHT * 4, dtype=torch.float32, device="cuda")
    default_spp = {"cornell_box": SPP, "smallpt": 4096, "dragon": 2048, "tr15_like": 512}

    def fence():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    def run_workload(name, want_spp, steps, warmup, warmup_spp=None, frames=1):
        """K timed steps of one workload (a step = one frame, or a sequence of `frames` frames of the moving workload), bracketed as the
        contract says; returns the measurements of this rank (and the scene)"""
        tmp = tempfile.mkdtemp(prefix=f"traybench{rank}_")
        frame = args.frame if name == "tr15_like" else 0
        if name == "dragon":
            scenes.write_dragon_assets(tmp, film=(WIDTH, HEIGHT, want_spp))
        elif name == "tr15_like":
            scenes.write_tr15_like_assets(tmp, film=(WIDTH, HEIGHT, want_spp))
        else:
            scenes.write_assets(tmp, cornell=(WIDTH, HEIGHT, want_spp), small=(WIDTH, HEIGHT, want_spp))
        scene, rt, spp, frame_info = T.Scene.load_file(os.path.join(tmp, name + ".json"))
        spp = T.round_spp(spp)

        seq = name == "tr15_like" and frames > 1

        def step(step_spp):
            if seq:   # this rank's frames of the sequence, one after the other through the same device scene
                for fr in multi.shard_frames(frame, frame + frames - 1, rank, world):
                    film.zero_()
                    hip.render_device(scene, fr, (0, 0), step_spp, film.data_ptr(), stream=stream)
                    seq_counts.append(hip.timing(scene))   # (synchronises: the next frame's update waits for the device anyway)
                return
            film.zero_()
            if distributed:
                multi.render_frame_sharded(   # tiles round-robin over ranks, then film::Image::add_pixels as one RCCL sum-reduce
                    lambda r, w, f: hip.render_shard_device(scene, frame, r, w, step_spp, f.data_ptr(), chunk_tiles=multi.DEFAULT_CHUNK_TILES, stream=stream),
                    film, rank, world, dst=0)
            else:
                hip.render_device(scene, frame, (0, 0), step_spp, film.data_ptr(), stream=stream)

        seq_counts = []
        for _ in range(warmup):
            step(warmup_spp or spp)
        fence()
        kernel_ms, samples, vertices, launches = [], 0, 0, 0
        t0 = time.perf_counter()
        step_ms = []   # wall time of every step (tray_last_timing below waits for the step's launches, so each figure is a finished step)
        for _ in range(steps):
            t_step = time.perf_counter()
            seq_counts.clear()
            step(spp)
            if seq:   # sums over this rank's frames of the sequence
                kernel_ms.append(sum(t.render_ms for t in seq_counts))
                samples, vertices = sum(int(t.samples) for t in seq_counts), sum(int(t.vertices) for t in seq_counts)
                launches = sum(int(t.launches) for t in seq_counts)
            else:
                tim = hip.timing(scene)   # HIP events on the launch stream around the schedule's kernels
                kernel_ms.append(tim.render_ms)
                samples, vertices, launches = tim.samples, tim.vertices, tim.launches
            step_ms.append((time.perf_counter() - t_step) * 1e3)
        fence()
        elapsed = time.perf_counter() - t0
        if distributed:
            tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            elapsed = float(tmax.item())
            cnt = torch.tensor([samples, vertices], dtype=torch.float64, device="cuda")
            dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
            total_samples, total_vertices = float(cnt[0].item()), float(cnt[1].item())
        else:
            total_samples, total_vertices = float(samples), float(vertices)
        try:
            sched = hip.schedule(scene)   # pool slots / views / slices of the last launch (tray_last_schedule)
        except Exception:
            sched = None
        return {"name": name, "scene": scene, "frame": frame, "spp": spp, "steps": steps, "elapsed": elapsed, "kernel_ms": sum(kernel_ms) / len(kernel_ms), "schedule": sched,
                "n_frames": frames if seq else 1, "step_ms": step_ms,
                "samples": samples, "vertices": vertices, "launches": launches, "total_samples": total_samples, "total_vertices": total_vertices}

    def line_of(m):
        """metric, roofline (SURVEY 8d's HBM accounting + the VALU view from the hash-matched counters) of one measured workload"""
        name, spp = m["name"], m["spp"]
        frame_samples = WIDTH * HEIGHT * spp * m["n_frames"]   # samples of one step (a frame, or the sequence of frames)
        assert abs(m["total_samples"] - frame_samples) < 0.5, (m["total_samples"], frame_samples)
        ms_per_step = m["elapsed"] * 1e3 / m["steps"]
        value = frame_samples / (ms_per_step * 1e-3) / 1e6
        vbar = m["total_vertices"] / m["total_samples"]
        k_ms = m["kernel_ms"]
        algo_bytes = m["samples"] * BYTES_PER_VERTEX * (m["vertices"] / max(m["samples"], 1)) + (WIDTH * HEIGHT * BYTES_PER_PIXEL) * m["n_frames"] / world
        achieved = algo_bytes / (k_ms * 1e-3) / 1e9
        traffic, valu, compute = pmc_views(name, m["samples"], k_ms * 1e-3)
        fs = m["scene"].flatten(m["frame"]).contents
        wave = m["launches"] > 1
        frac = achieved / HBM_PEAK_GBS
        schedule = ("wavefront stage kernels over the HBM path pool (compacted ray queues, persistent dynamic-fetch traversal)" if wave
                    else "tile megakernel k_path_tiles (wave-synchronous vertex stepping)")
        roofline = {"bound": "valu" if (valu and valu["frac"] > frac) else "hbm",
                    "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(frac, 5), "traffic": traffic,
                    "valu": valu, "compute": compute,
                    "kernel": "the stage kernels of a round -- advance, regen, trace A, sort, shade per material kind, trace B -- (HIP events around the whole schedule)" if wave else "k_path_tiles",
                    "kernel_ms": round(k_ms, 3), "algorithmic_bytes_per_launch": int(algo_bytes),
                    "schedule": m.get("schedule"),
                    "counters_from_this_schedule": (None if not (compute and compute.get("schedule_of_the_counter_launch") and m.get("schedule")) else
                                                    all(compute["schedule_of_the_counter_launch"].get(k) == m["schedule"].get(k) for k in ("pool_slots", "views", "slices", "launched_wavefront", "transform_table"))),
                    "note": "achieved / peak / frac are SURVEY 8(d)'s accounting of the WAVEFRONT formulation (368 B per path vertex + 16 B per pixel) over the HIP-event "
                            "time of the launch; `bound` names the resource the launch is closer to, `valu` prices the VALU lane-issue rate from the counters. " +
                            ("Traversal of the 3.1 M-triangle scene is memory-latency bound." if wave
                             else "The tile megakernel keeps path state in registers and the film in LDS: its HBM traffic (`traffic`) is a small "
                                  "multiple of the 33 MB film, the scene is cache resident, and the kernel is bound by VALU issue at partial lane utilisation.")}
        config = {"workload": f"{name} 1920x1080 {spp}spp, path tracer min_depth {fs.min_depth} max_depth {fs.max_depth}"
                              f"{(', frames %d..%d, device scene updated from frame to frame inside the timed region' % (m['frame'], m['frame'] + m['n_frames'] - 1)) if m['n_frames'] > 1 else (', frame %d' % m['frame'] if name == 'tr15_like' else '')}"
                              f" (BASELINE.json configs[{WORKLOAD_CONFIG[name]}])",
                  "schedule": schedule, "samples_per_step": frame_samples,
                  "parallelism": (f"frames round-robin over {world} GPU(s), no collective" if m["n_frames"] > 1 else f"tiles round-robin over {world} GPU(s), RCCL sum-reduce")
                  if distributed else "1 GPU", "seed": 1,
                  "vertices_per_sample": round(vbar, 4)}
        return value, ms_per_step, config, roofline

    main_m = run_workload(args.workload, args.spp or default_spp[args.workload], args.steps, args.warmup, frames=args.frames)
    if rank == 0:
        value, ms_per_step, config, roofline = line_of(main_m)
        out = {
            "metric": "Msamples/s (whole node) at 1920x1080; achieved HBM GB/s vs peak",
            "value": round(value, 3), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic", "config": config, "roofline": roofline,
        }
    # BASELINE.json configs[2..4] beside the reported line (N = 1 only, like the CPU leg): one warm-up launch at a cut-down sample
    # count, then full-size frames -- so that the driver's own run sees every workload, not only cornell_box
    if world == 1 and args.workload == "cornell_box" and not args.spp and not args.no_other_workloads:
        others = []
        # (configs[4] is a SEQUENCE: the tr15 stand-in runs two consecutive frames per step, the device scene moved from one to the next with
        # tray_scene_update_frame inside the timed region; `frame_kernel_value` is the rate of the frames' kernels alone, HIP events)
        for name, steps, frames in (("smallpt", 2, 1), ("dragon", 2, 1), ("tr15_like", 2, 2)):   # (round 6: the sequence is stepped twice and both steps are printed)
            m = run_workload(name, default_spp[name], steps, 1, warmup_spp=16, frames=frames)
            v, ms, cfg, rf = line_of(m)
            entry = {"workload": cfg["workload"], "schedule": cfg["schedule"], "value": round(v, 3), "unit": "Msamples/s", "steps": steps,
                     "warmup": "1 launch at 16 spp", "ms_per_step": round(ms, 3), "each_step_ms": [round(x, 1) for x in m["step_ms"]],
                     "each_step_value": [round(WIDTH * HEIGHT * m["spp"] * m["n_frames"] / (x * 1e-3) / 1e6, 1) for x in m["step_ms"]],
                     "vertices_per_sample": cfg["vertices_per_sample"],
                     "roofline": {k: rf[k] for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "valu", "kernel", "kernel_ms", "schedule", "counters_from_this_schedule")}}
            if frames > 1:
                entry["frames_per_step"] = frames
                entry["frame_kernel_value"] = round(m["samples"] / (m["kernel_ms"] * 1e-3) / 1e6, 3)
            others.append(entry)
        out["workloads"] = others
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(main_m["scene"].flatten(main_m["frame"]), main_m["spp"])
        print(json.dumps(out), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
