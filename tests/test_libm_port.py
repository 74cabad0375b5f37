"""tray_rust_amd/csrc/hip/dev_libm.h restates glibc's acosf / sinf / cosf -- what the reference's f32::acos / sin / cos in Quaternion::slerp
(quaternion.rs:101-113) resolve to on Linux, and what the oracle calls -- so that moving instances get the reference's bits. The checker
tools/libm_port_check.cpp compiles THAT header for the host and compares it with the system libm; exhaustively it takes a minute
(profiles/r04_libm_port_check.txt: 0 of 2 130 706 434 / 1 078 774 990 arguments differ), here every 1021st float."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_device_libm_equals_the_system_libm(tmp_path):
    exe = str(tmp_path / "libm_port_check")
    subprocess.run(["g++", "-O2", "-fno-builtin", "-ffp-contract=off", os.path.join(ROOT, "tools", "libm_port_check.cpp"), "-o", exe, "-lm"], check=True)
    out = subprocess.run([exe, "1021"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout
    assert "acosf: 0 of" in out.stdout and "sinf: 0, cosf: 0 of" in out.stdout, out.stdout
