"""tray_rust_amd/csrc/hip/dev_libm.h restates glibc's acosf / sinf / cosf / atanf / atan2f / expf / logf -- what the reference's f32::acos / sin /
cos / atan2 / exp / ln resolve to on Linux (Quaternion::slerp quaternion.rs:101-113, bxdf/merl.rs:63-75, microfacet/beckmann.rs:33-48,
mc.rs:49-50, sphere.rs:71), and what the oracle calls -- so that a camera sample's radiance is the reference arithmetic's bit for bit. The
checker tools/libm_port_check.cpp compiles THAT header for the host and compares it with the system libm; exhaustively it takes a minute
and a half (profiles/r05_libm_port_check.txt: 0 of 2^32 bit patterns differ for atanf / expf / logf, 0 of 2.2e9 for sinf / cosf on |x| < 119,
0 of 2.1e9 for acosf on [-1, 1], 0 of 1.6e9 pairs for atan2f), here every 1021st float."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_device_libm_equals_the_system_libm(tmp_path):
    exe = str(tmp_path / "libm_port_check")
    subprocess.run(["g++", "-O2", "-fno-builtin", "-ffp-contract=off", os.path.join(ROOT, "tools", "libm_port_check.cpp"), "-o", exe, "-lm", "-pthread"], check=True)
    out = subprocess.run([exe, "1021"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout
    assert "acosf: 0 of" in out.stdout and "sinf: 0, cosf: 0 of" in out.stdout, out.stdout
    assert "atanf: 0, expf: 0, logf: 0 of" in out.stdout and "(shared form, |x| < 119): 0 of" in out.stdout and "atan2f: 0 of" in out.stdout, out.stdout
