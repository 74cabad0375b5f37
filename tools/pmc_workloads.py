"""Counter passes of ALL bench workloads for the build in tray_rust_amd/libtrayhip.so (run on the GPU box):
    python tools/pmc_workloads.py <tag> [workload:spp ...]      default: cornell_box:64 smallpt:64 dragon:32 tr15_like:128 at FULL scene size
(round 5: tr15_like at 128 spp, frame 64 -- the sample count from which the host rules cut tiles into 4 slices, fill a 33 M-slot pool and run ONE view, i.e. the
schedule of the 512-spp bench launch; 16 spp, rounds 2-4, ran whole tiles in 8 M slots on two views. The schedule of every measured launch is recorded.)
One rocprofv3 --pmc pass per counter set (utilisation set, FETCH_SIZE, WRITE_SIZE) around a torch-free launch (tools/mini_ab.py), plus the
FETCH_SIZE / WRITE_SIZE calibration on a scratch pattern of known size (tools/scratch_calib). Writes
    gpurun_out/summary_<tag>/pmc_latest.json           {"device_code_hash", "workloads": {name: derived figures}}  -> copy to profiles/
    gpurun_out/summary_<tag>/<tag>_pmc_<workload>.csv  raw per-launch counters per kernel + derived figures
bench.py uses pmc_latest.json only for the device code whose md5 it carries."""
import csv, glob, json, os, re, shutil, subprocess, sys

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
tag = sys.argv[1]
items = sys.argv[2:] or ["cornell_box:64", "smallpt:64", "dragon:32", "tr15_like:256"]   # (round 6: 256 spp is the sample count from which launch_wavefront cuts tiles into 16 slices and fills the 132.7 M-slot pool, as at the bench's 512)
dest = os.path.join(ROOT, "gpurun_out", f"summary_{tag}")
os.makedirs(dest, exist_ok=True)
D = "/tmp/pmc_full"   # (its own directory: /tmp/mini_full is also what tools/ab.sh calls prepare, possibly with another MINI_TR15_DETAIL, and a GPU box can come back with its /tmp as it was)
env0 = dict(os.environ, TMPDIR="/tmp", MINI_DRAGON_GRID=os.environ.get("MINI_DRAGON_GRID", "660"), MINI_TR15_DETAIL=os.environ.get("MINI_TR15_DETAIL", "1.0"))
if not os.path.exists(os.path.join(D, "cornell_box.json")):
    subprocess.run(["python", os.path.join(ROOT, "tools", "mini_ab.py"), "prepare", D], env=env0, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
N_SIMD = 256 * 4
SETS = {"util": "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU",
        "fetch": "FETCH_SIZE", "write": "WRITE_SIZE"}


def prof(name, counters, cmd, timeout=600):
    d = f"/tmp/pmcw_{name}"
    shutil.rmtree(d, ignore_errors=True)
    subprocess.run(["timeout", str(timeout), "rocprofv3", "--kernel-trace", "--pmc"] + counters.split() + ["--output-format", "csv", "-d", d, "--"] + cmd,
                   cwd="/tmp", env=env0, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    per_kernel, times = {}, {}
    for path in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            k = row["Kernel_Name"].split("(")[0]
            times.setdefault(k, []).append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            k = row["Kernel_Name"].split("(")[0]
            per_kernel.setdefault(k, {}).setdefault(row["Counter_Name"], 0.0)
            per_kernel[k][row["Counter_Name"]] += float(row["Counter_Value"])
    return per_kernel, times


# ---- calibration of FETCH_SIZE / WRITE_SIZE on a known scratch pattern
calib = {}
exe = os.path.join(ROOT, "tools", "scratch_calib")
if not os.path.exists(exe):
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", exe + ".hip", "-o", exe], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
if os.path.exists(exe):
    log = subprocess.run([exe, "64"], capture_output=True, text=True).stdout
    m = re.search(r"([\d.e+]+) bytes stored, ([\d.e+]+) bytes loaded.*the warm-up launch adds 1/(\d+)", log)
    if m:
        known = float(m.group(1)) * (1.0 + 1.0 / int(m.group(3)))
        for key, cname in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
            pk, _ = prof("calib_" + key, cname, [exe, "64"], 120)
            tot = sum(v.get(cname, 0.0) for k, v in pk.items() if "k_scratch" in k)
            if tot:
                calib[cname] = tot * 1024 / known
try:
    dev_hash = subprocess.run([os.path.join(ROOT, "tools", "device_code_hash.sh")], capture_output=True, text=True, check=True).stdout.strip()
except Exception:
    dev_hash = None
latest = {"device_code_hash": dev_hash, "calibration_reported_over_known": calib,
          "note": "per-launch counters of cut-down launches; traffic and instruction counts scale with the sample count (bench.py multiplies the per-sample figures)",
          "workloads": {}}
for item in items:
    wl, spp = item.split(":")
    spp = int(spp)
    samples = 1920 * 1080 * spp
    cmd = ["python", os.path.join(ROOT, "tools", "mini_ab.py"), "run", D, "pmc", item]
    merged, times = {}, {}
    for key, counters in SETS.items():
        pk, tm = prof(f"{wl}_{key}", counters, cmd)
        if key == "util":
            times = tm
        for k, v in pk.items():
            merged.setdefault(k, {}).update(v)
    if not merged:
        print(f"{wl}: no counters collected"); continue
    # mini_ab launches every frame twice per process: per-LAUNCH(-of-the-frame) values
    reps = 2.0
    kernels = {k: {c: v / reps for c, v in cs.items()} for k, cs in merged.items() if k.startswith("void k_") or "k_path_tiles" in k or "k_wf_" in k}
    ktime = {k: sum(times.get(k, [0])) / reps / 1e6 for k in kernels}
    total_ms = sum(ktime.values()) or 1.0
    dominant = max(ktime, key=ktime.get)
    tot = {}
    for cs in kernels.values():
        for c, v in cs.items():
            tot[c] = tot.get(c, 0.0) + v
    d = {"spp": spp, "samples_per_launch": samples, "kernel": dominant, "dominant_kernel_share_of_time": round(ktime[dominant] / total_ms, 4),
         "kernel_ms_under_pmc": round(total_ms, 3)}
    if tot.get("SQ_ACTIVE_INST_VALU"):
        d["valu_instructions_per_sample"] = tot["SQ_INSTS_VALU"] / samples
        d["valu_lane_utilisation"] = round(tot["SQ_THREAD_CYCLES_VALU"] / (64 * tot["SQ_ACTIVE_INST_VALU"]), 4)
        dk = kernels[dominant]
        if dk.get("SQ_ACTIVE_INST_VALU"):
            d["dominant_kernel_lane_utilisation"] = round(dk["SQ_THREAD_CYCLES_VALU"] / (64 * dk["SQ_ACTIVE_INST_VALU"]), 4)
            d["dominant_kernel_waiting_share_of_wave_cycles"] = round(dk.get("SQ_WAIT_ANY", 0) / max(dk.get("SQ_WAVE_CYCLES", 1), 1), 4)
        if len(kernels) == 1 and tot.get("SQ_WAVES") and tot.get("SQ_WAVE_CYCLES"):   # persistent tile kernel: every wave lives for the whole launch
            d["waves_per_simd"] = tot["SQ_WAVES"] / N_SIMD
            d["valu_busy"] = round(tot["SQ_ACTIVE_INST_VALU"] * tot["SQ_WAVES"] / (N_SIMD * tot["SQ_WAVE_CYCLES"]), 4)
        d["waiting_share_of_wave_cycles"] = round(tot.get("SQ_WAIT_ANY", 0) / max(tot.get("SQ_WAVE_CYCLES", 1), 1), 4)
    if "FETCH_SIZE" in tot and "WRITE_SIZE" in tot:
        fr, wr = tot["FETCH_SIZE"] * 1024, tot["WRITE_SIZE"] * 1024
        kf, kw = calib.get("FETCH_SIZE"), calib.get("WRITE_SIZE")
        if kf and kw and kf > 0.05 and kw > 0.05:
            d["hbm_bytes_per_launch"] = fr / kf + wr / kw
            d["hbm_bytes_basis"] = "FETCH_SIZE / WRITE_SIZE divided by the reported / known ratios of tools/scratch_calib (one-dword-per-lane scratch traffic)"
        else:
            d["hbm_bytes_per_launch"] = 2 * fr + wr
            d["hbm_bytes_basis"] = "2 x FETCH_SIZE + WRITE_SIZE (MI355X_MICROARCH.md: gfx950 tallies 128-B reads as 64 B; WRITE_SIZE uncalibrated)"
        d["hbm_bytes_per_sample"] = d["hbm_bytes_per_launch"] / samples
    try:   # slots / views / slices / frame of the measured launch (tools/mini_ab.py writes it after its render calls)
        d["schedule"] = json.load(open(os.path.join(D, f"schedule_{wl}.json")))
    except Exception:
        d["schedule"] = None
    latest["workloads"][wl] = d
    lines = [f"# rocprofv3 --kernel-trace --pmc <set> -- python tools/mini_ab.py run <dir> pmc {item}   ({wl} 1920x1080 at full scene size, {spp} spp); one pass per counter set",
             f"# device code {dev_hash}; values per frame launch, summed over XCDs / SEs", "kernel,ms,counter,value"]
    for k in sorted(kernels, key=lambda k: -ktime[k]):
        for c, v in sorted(kernels[k].items()):
            lines.append(f"\"{k}\",{ktime[k]:.3f},{c},{v:.6g}")
    lines += [f"# derived: {k} = {v}" for k, v in d.items()]
    open(os.path.join(dest, f"{tag}_pmc_{wl}.csv"), "w").write("\n".join(lines) + "\n")
    print(wl, json.dumps(d), flush=True)
json.dump(latest, open(os.path.join(dest, "pmc_latest.json"), "w"), indent=1)
print("summaries in", dest, os.listdir(dest))
