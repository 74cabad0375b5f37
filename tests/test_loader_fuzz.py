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


@pytest.mark.parametrize("which", ["cornell_box", "moving_box"])
def test_mutated_scene_files_never_crash_the_loader(which, tmp_path, built):
    d = str(tmp_path)
    scenes.write_assets(d, cornell=(64, 64, 4), small=(64, 64, 4))
    scenes.write_moving_box(d, width=64, height=64, samples=4)
    out = run_tool([os.path.join(ROOT, "tools", "fuzz_loader.py"), "11", "80", os.path.join(d, which + ".json")])
    assert "CRASH" not in out and "pyexc" not in out, out
    counts = dict((k, int(v)) for k, v in re.findall(r"'(\w+)': (\d+)", out))
    assert counts.get("ok", 0) + counts.get("err", 0) == 80 and counts.get("err", 0) > 20 and counts.get("ok", 0) > 5, out


def test_mutated_obj_and_merl_files_never_crash_the_loader(built):
    out = run_tool([os.path.join(ROOT, "tools", "fuzz_assets.py"), "3", "24"])
    assert "rc 0 done 24 of 24" in out and "pyexc" not in out, out
