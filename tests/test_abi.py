"""The C-ABI library loads and exports every symbol include/trayhip.h declares (no GPU needed)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    text = open(os.path.join(ROOT, "include", "trayhip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(tray_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported(built):
    from tray_rust_amd import _lib as L
    names = header_functions()
    assert len(names) >= 20
    handle = C.CDLL(L.LIB_PATH)
    for n in names:
        assert hasattr(handle, n), f"{n} declared in include/trayhip.h but not exported"
    assert set(names) == set(L.SYMBOLS), set(names) ^ set(L.SYMBOLS)


def test_struct_sizes_match_header(built):
    from tray_rust_amd import _lib as L
    assert C.sizeof(L.TrayBvhNode) == 32
    assert C.sizeof(L.TrayTriVerts) == 48
    assert C.sizeof(L.TrayTriAttrs) == 64
    assert C.sizeof(L.TrayInstance) == 4 * 4 + 16 + 16 + 64 + 64 + 32
    assert C.sizeof(L.TrayMaterial) == 80
    assert C.sizeof(L.TrayRay) == 36
    assert C.sizeof(L.TrayHit) == 4 * 3 + 12 * 3 + 8 + 24


def test_ctypes_layouts_match_the_compiled_library(built):
    import tray_rust_amd as T
    from tray_rust_amd import _lib as L
    names = [n for n in dir(L) if n.startswith("Tray") and isinstance(getattr(L, n), type) and issubclass(getattr(L, n), C.Structure)]
    assert len(names) >= 15
    for n in names:
        want = T.lib().tray_abi_sizeof(n.encode())
        assert want == C.sizeof(getattr(L, n)), (n, want, C.sizeof(getattr(L, n)))


def test_version_and_error_channel(built):
    import tray_rust_amd as T
    assert b"trayhip" in T.lib().tray_version()
    with pytest.raises(T.TrayError) as e:
        T.Scene.load_file("/nonexistent/scene.json")
    assert e.value.code == T._lib.TRAY_E_IO and "Failed to open scene file" in e.value.message


def test_wavefront_entry_points_check_their_arguments(built):
    import tray_rust_amd as T
    from tray_rust_amd import _lib as L
    lib = T.lib()
    info = L.TrayScheduleInfo()
    assert lib.tray_scene_set_wavefront(None, 0, 0, 0) == L.TRAY_E_INVALID and b"null" in lib.tray_last_error()
    assert lib.tray_multi_set_wavefront(None, 0, 0, 0) == L.TRAY_E_INVALID
    assert lib.tray_last_schedule(None, C.byref(info)) == L.TRAY_E_INVALID
    assert C.sizeof(L.TrayScheduleInfo) == 8 * 4 + 3 * 8 + 2 * 4 + 8
    assert lib.tray_scene_set_transform_table(None, 1) == L.TRAY_E_INVALID and lib.tray_multi_set_transform_table(None, 0) == L.TRAY_E_INVALID


def test_device_calls_fail_loudly_without_gpu(assets):
    """No silent CPU fallback: creating a device scene without a GPU is an error, never a no-op."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import tray_rust_amd as T
    scene, rt, spp, fi = T.Scene.load_file(os.path.join(assets, "cornell_box.json"))
    with pytest.raises(T.TrayError):
        T.Hip(0).render(scene, rt, T.Config(".", "x", spp, 1, fi))


def test_product_does_not_depend_on_the_oracle(built):
    """oracle/ is test infrastructure: the shipped library must not link it and the package must not import it."""
    import re
    import subprocess
    from tray_rust_amd import _lib as L
    needed = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "-d", L.LIB_PATH], capture_output=True, text=True).stdout
    libs = re.findall(r"\(NEEDED\)\s+Shared library: \[(.*?)\]", needed)
    assert libs and not any("oracle" in l for l in libs), libs
    pkg = os.path.dirname(os.path.abspath(L.__file__))
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hpp", ".h", ".cpp", ".hip")):
                text = open(os.path.join(root, f), errors="replace").read()
                assert "liboracle" not in text and "oracle/" not in text.replace("the oracle/", ""), os.path.join(root, f)


def _build_c_example(tmp_path):
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(str(tmp_path), "trayhip_render")
    libdir = os.path.join(root, "tray_rust_amd")
    subprocess.run(["gcc", "-O2", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(root, "include"), os.path.join(root, "examples", "trayhip_render.c"),
                    "-L" + libdir, "-ltrayhip", "-Wl,-rpath," + libdir, "-o", exe], check=True)
    return exe


def test_plain_c_host_compiles_and_fails_loudly_without_a_gpu(assets, tmp_path):
    """include/trayhip.h is a C header; examples/trayhip_render.c drives the whole boundary from C. Without a GPU the render must
    stop with the library's error, not fall back to anything."""
    import subprocess
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu test")
    exe = _build_c_example(tmp_path)
    r = subprocess.run([exe, os.path.join(assets, "cornell_box.json"), os.path.join(str(tmp_path), "o.ppm")], capture_output=True, text=True)
    assert r.returncode == 1 and "tray_init failed" in r.stderr
    assert not os.path.exists(os.path.join(str(tmp_path), "o.ppm"))


@pytest.mark.gpu
def test_plain_c_host_renders(tmp_path):
    import subprocess
    import numpy as np
    from tray_rust_amd import scenes
    exe = _build_c_example(tmp_path)
    scenes.write_assets(str(tmp_path), cornell=(64, 48, 16))
    out = os.path.join(str(tmp_path), "o.ppm")
    r = subprocess.run([exe, os.path.join(str(tmp_path), "cornell_box.json"), out], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "rendering took" in r.stdout
    data = open(out, "rb").read()
    assert data.startswith(b"P6\n64 48\n255\n")
    px = np.frombuffer(data[len(b"P6\n64 48\n255\n"):], np.uint8)
    assert px.size == 64 * 48 * 3 and px.max() > 100
    # ... and with sampler::Adaptive::new(dim, 4, 32) in LowDiscrepancy's place (tray_scene_set_sampler from plain C)
    r = subprocess.run([exe, os.path.join(str(tmp_path), "cornell_box.json"), out, "0", "1", "adaptive:4:32"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    px2 = np.frombuffer(open(out, "rb").read()[len(b"P6\n64 48\n255\n"):], np.uint8)
    assert px2.size == px.size and np.abs(px2.astype(int) - px.astype(int)).mean() < 12
    r = subprocess.run([exe, os.path.join(str(tmp_path), "cornell_box.json"), out, "0", "1", "stratified"], capture_output=True, text=True)
    assert r.returncode == 2 and "unknown sampler" in r.stderr
