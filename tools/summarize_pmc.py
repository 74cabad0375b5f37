"""Condense the rocprofv3 passes of tools/pmc_tile.sh into profiles-ready files:
  <tag>_pmc_<kernel>.csv   counters of the dominant kernel (summed over XCDs / SEs) + derived figures
  pmc_latest.json          what bench.py reads for roofline.traffic and the compute view -- carries the md5 of the device code
                           (tools/device_code_hash.sh) the counters were measured on; bench.py ignores it for any other build
usage: summarize_pmc.py <dir> <tag> <workload> <spp>"""
import csv, glob, json, os, re, subprocess, sys

out_dir, tag = sys.argv[1].rstrip("/"), sys.argv[2]
workload = sys.argv[3] if len(sys.argv) > 3 else "cornell_box"
spp = int(sys.argv[4]) if len(sys.argv) > 4 else 64
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dest = os.path.join(os.path.dirname(out_dir), f"summary_{tag}")
os.makedirs(dest, exist_ok=True)
N_SIMD = 256 * 4


def read(prefix):
    counters, kernel_ns = {}, {}
    for path in glob.glob(os.path.join(out_dir, prefix + "*", "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            k = row["Kernel_Name"].split("(")[0]
            counters.setdefault(k, {}).setdefault(row["Counter_Name"], 0.0)
            counters[k][row["Counter_Name"]] += float(row["Counter_Value"])
    for path in glob.glob(os.path.join(out_dir, prefix + "*", "**", "*kernel_trace.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            k = row["Kernel_Name"].split("(")[0]
            kernel_ns.setdefault(k, []).append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    return counters, kernel_ns


counters, kernel_ns = read("pmc_")
if not counters:
    sys.exit("no counter_collection.csv under " + out_dir)
dominant = max(counters, key=lambda k: counters[k].get("SQ_WAVE_CYCLES", 0.0) + counters[k].get("FETCH_SIZE", 0.0))
# the runner launches the kernel more than once per process (tools/mini_ab.py: 2 repetitions): per-LAUNCH values
n_passes = max(1, len(glob.glob(os.path.join(out_dir, "pmc_*", "**", "*kernel_trace.csv"), recursive=True)))
launches_per_pass = max(1, round(len(kernel_ns.get(dominant, [0])) / n_passes))
c = {k: v / launches_per_pass for k, v in counters[dominant].items()}
d = {}
if c.get("SQ_ACTIVE_INST_VALU"):
    d["valu_lane_utilisation"] = c.get("SQ_THREAD_CYCLES_VALU", 0) / (64 * c["SQ_ACTIVE_INST_VALU"])
    d["cycles_per_valu_instruction"] = 4 * c["SQ_ACTIVE_INST_VALU"] / max(c.get("SQ_INSTS_VALU", 1), 1)
if c.get("SQ_WAVE_CYCLES") and c.get("SQ_WAVES"):
    # persistent kernel: every wave lives for the whole launch, so WAVE_CYCLES / WAVES is the launch in (quad-)cycles and the
    # VALU pipe of a SIMD is busy for ACTIVE_INST_VALU / n_SIMD of them (SQ_* cycle counters count in the same unit)
    d["waves_per_simd"] = c["SQ_WAVES"] / N_SIMD
    d["valu_busy"] = c.get("SQ_ACTIVE_INST_VALU", 0) * c["SQ_WAVES"] / (N_SIMD * c["SQ_WAVE_CYCLES"])
    d["waiting_share_of_wave_cycles"] = c.get("SQ_WAIT_ANY", 0) / c["SQ_WAVE_CYCLES"]
    d["issue_stall_share_of_wave_cycles"] = c.get("SQ_WAIT_INST_ANY", 0) / c["SQ_WAVE_CYCLES"]
    d["valu_instructions_per_wave"] = c.get("SQ_INSTS_VALU", 0) / c["SQ_WAVES"]
if dominant in kernel_ns:
    d["kernel_ms_under_pmc_mean"] = sum(kernel_ns[dominant]) / len(kernel_ns[dominant]) / 1e6

# calibration of FETCH_SIZE / WRITE_SIZE (KB) on the scratch pattern with a known byte count
calib = {}
cc, _ = read("calib_")
for k, v in cc.items():
    if "k_scratch" in k:
        calib.update(v)
log = ""
for p in glob.glob(os.path.join(out_dir, "calib_*.log")):
    log = open(p).read()
m = re.search(r"([\d.e+]+) bytes stored, ([\d.e+]+) bytes loaded.*the warm-up launch adds 1/(\d+)", log)
if m and calib:
    known = float(m.group(1)) * (1.0 + 1.0 / int(m.group(3)))   # both launches are in the counter sums
    if "FETCH_SIZE" in calib:
        d["calib_fetch_reported_over_known"] = calib["FETCH_SIZE"] * 1024 / known
    if "WRITE_SIZE" in calib:
        d["calib_write_reported_over_known"] = calib["WRITE_SIZE"] * 1024 / known
    d["calib_known_bytes_each_way"] = known
samples = 1920 * 1080 * spp
if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
    fr, wr = c["FETCH_SIZE"] * 1024, c["WRITE_SIZE"] * 1024
    d["fetch_bytes_reported"], d["write_bytes_reported"] = fr, wr
    kf, kw = d.get("calib_fetch_reported_over_known"), d.get("calib_write_reported_over_known")
    if kf and kw and kf > 0.05 and kw > 0.05:
        d["hbm_bytes_per_launch"] = fr / kf + wr / kw
        d["hbm_bytes_basis"] = "FETCH_SIZE / WRITE_SIZE divided by the reported / known ratios of tools/scratch_calib (one-dword-per-lane scratch traffic)"
    else:
        d["hbm_bytes_per_launch"] = 2 * fr + wr
        d["hbm_bytes_basis"] = "2 x FETCH_SIZE + WRITE_SIZE (MI355X_MICROARCH.md: gfx950 tallies 128-B reads as 64 B; WRITE_SIZE uncalibrated)"
    d["hbm_bytes_per_sample"] = d["hbm_bytes_per_launch"] / samples

lines = [f"# rocprofv3 --kernel-trace --pmc <set> -- python tools/mini_ab.py run <dir> pmc {workload}:{spp}   ({workload} 1920x1080, {spp} spp, one launch)",
         "# separate passes per counter set; values summed over all XCDs / SEs; kernel: " + dominant, "counter,value"]
lines.insert(2, f"# values per launch (the runner launched the kernel {launches_per_pass}x per pass)")
lines += [f"{k},{v:.6g}" for k, v in sorted(c.items())]
lines += [f"# derived: {k} = {v if isinstance(v, str) else format(v, '.6g')}" for k, v in d.items()]
name = re.sub(r"[^A-Za-z0-9_]+", "_", dominant.split("::")[-1]).strip("_")
open(os.path.join(dest, f"{tag}_pmc_{name}.csv"), "w").write("\n".join(lines) + "\n")
try:
    dev_hash = subprocess.run([os.path.join(ROOT, "tools", "device_code_hash.sh")], capture_output=True, text=True, check=True).stdout.strip()
except Exception:
    dev_hash = None
latest = {"kernel": dominant, "workload": workload, "spp": spp, "samples_per_launch": samples, "device_code_hash": dev_hash,
          "counters": c, "derived": d,
          "k_path_tiles_hbm_bytes_per_launch_at_profiled_spp": d.get("hbm_bytes_per_launch"),
          "note": "traffic scales with the sample count: bench.py multiplies hbm_bytes_per_sample by the samples of its launch"}
json.dump(latest, open(os.path.join(dest, "pmc_latest.json"), "w"), indent=1)
print("\n".join(lines[-len(d) - 2:]))
print("summaries in", dest, os.listdir(dest))
