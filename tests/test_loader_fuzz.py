"""The loader behind the C ABI must answer malformed input with an error code: nothing may abort, crash or hang across the
boundary (the reference panics instead: scene.rs:104-136). A short deterministic mutation run of tools/fuzz_loader.py and
tools/fuzz_assets.py (scene JSON; OBJ and MERL files) in subprocesses, so that a crash is reported instead of killing pytest."""
import os
import re
import subprocess
import sys

import pytest

from tray_rust_amd import scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_tool(args):
    p = subprocess.run([sys.executable] + args, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    return p.stdout


@pytest.mark.parametrize("which", ["cornell_box", "moving_box", "waving_flag"])
def test_mutated_scene_files_never_crash_the_loader(which, tmp_path, built):
    d = str(tmp_path)
    scenes.write_assets(d, cornell=(64, 64, 4), small=(64, 64, 4))
    scenes.write_moving_box(d, width=64, height=64, samples=4)
    scenes.write_waving_flag(d, grid=6, n_keys=3, width=64, height=64, samples=4)      # (the animated_mesh entry and its keyframe files)
    out = run_tool([os.path.join(ROOT, "tools", "fuzz_loader.py"), "11", "80", os.path.join(d, which + ".json")])
    assert "CRASH" not in out and "pyexc" not in out, out
    counts = dict((k, int(v)) for k, v in re.findall(r"'(\w+)': (\d+)", out))
    assert counts.get("ok", 0) + counts.get("err", 0) == 80 and counts.get("err", 0) > 20 and counts.get("ok", 0) > 5, out


def test_mutated_obj_and_merl_files_never_crash_the_loader(built):
    out = run_tool([os.path.join(ROOT, "tools", "fuzz_assets.py"), "3", "24"])
    assert "rc 0 done 24 of 24" in out and "pyexc" not in out, out


def test_mutated_image_files_never_crash_the_loader(built):
    """PNG, baseline / progressive JPEG, GIF, BMP, TGA textures with flipped, inserted, deleted and truncated bytes: a picture or an error"""
    pytest.importorskip("PIL.Image")
    out = run_tool([os.path.join(ROOT, "tools", "fuzz_images.py"), "5", "64"])
    assert "rc 0 done 64 of 64" in out and "pyexc" not in out and "CRASH" not in out, out


def test_image_decoders_under_the_sanitizers(tmp_path):
    """csrc/host/image.hpp alone, built with AddressSanitizer and UndefinedBehaviorSanitizer, over 1500 mutated files (round 4: a corrupt
    JPEG's coefficients overflowed the 32-bit IDCT; the transform is 64 bits wide since)"""
    pytest.importorskip("PIL.Image")
    exe = str(tmp_path / "image_decode_check")
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                        os.path.join(ROOT, "tools", "image_decode_check.cpp"), "-o", exe], capture_output=True, text=True)
    if r.returncode != 0:
        pytest.skip("no sanitizer runtime in this toolchain: " + r.stderr[-200:])
    out = run_tool([os.path.join(ROOT, "tools", "fuzz_images_asan.py"), "21", "1500", exe])
    assert "failing batches: 0" in out, out


def test_loader_under_the_sanitizers(tmp_path):
    """csrc/host/scene.cpp + capi_host.cpp (JSON, OBJ, MERL, animated_mesh keyframes, textures, BVH builds, flattening) built with
    AddressSanitizer and UndefinedBehaviorSanitizer (tools/loader_check.cpp; no HIP involved) over mutated scene and asset files of the five
    test scenes. (Round 4 ran 6 600 of them clean; this is the short version.)"""
    exe = str(tmp_path / "loader_check")
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-I" + os.path.join(ROOT, "include"),
                        os.path.join(ROOT, "tools", "loader_check.cpp"), os.path.join(ROOT, "tray_rust_amd", "csrc", "host", "scene.cpp"),
                        os.path.join(ROOT, "tray_rust_amd", "csrc", "host", "capi_host.cpp"), "-pthread", "-o", exe], capture_output=True, text=True)
    if r.returncode != 0:
        pytest.skip("no sanitizer runtime in this toolchain: " + r.stderr[-200:])
    out = run_tool([os.path.join(ROOT, "tools", "fuzz_loader_asan.py"), "31", "120", exe])
    assert "failing batches: 0" in out and "'ok'" in out, out
