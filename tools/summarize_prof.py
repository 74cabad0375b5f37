"""Condense the rocprofv3 output of tools/profile_round.sh into the two small files committed under profiles/:
<tag>_kernel_stats.csv (verbatim kernel_stats of the bench step) and <tag>_pmc_<kernel>.csv (counters summed over
XCDs/SEs for the dominant kernel, plus derived figures). usage: summarize_prof.py <dir> <tag> <workload>"""
import csv, glob, json, os, sys

out_dir, tag = sys.argv[1], sys.argv[2]
workload = sys.argv[3] if len(sys.argv) > 3 else "cornell_box"
dest = os.path.join(os.path.dirname(out_dir.rstrip("/")), f"summary_{tag}")
os.makedirs(dest, exist_ok=True)

stats = glob.glob(os.path.join(out_dir, "stats", "**", "*kernel_stats.csv"), recursive=True)
if stats:
    open(os.path.join(dest, f"{tag}_kernel_stats.csv"), "w").write(open(stats[0]).read())

counters, kernel_ns, dominant = {}, {}, None
for path in glob.glob(os.path.join(out_dir, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(path)):
        k = row["Kernel_Name"].split("(")[0]
        counters.setdefault(k, {}).setdefault(row["Counter_Name"], 0.0)
        counters[k][row["Counter_Name"]] += float(row["Counter_Value"])
for path in glob.glob(os.path.join(out_dir, "pmc_*", "**", "*kernel_trace.csv"), recursive=True):
    for row in csv.DictReader(open(path)):
        k = row["Kernel_Name"].split("(")[0]
        kernel_ns.setdefault(k, []).append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
if counters:
    per_kernel = []
    for k, c in sorted(counters.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0.0)):
        if c.get("SQ_ACTIVE_INST_VALU"):
            per_kernel.append(f"# kernel {k}: wave_cycles {c.get('SQ_WAVE_CYCLES', 0):.4g} valu_insts {c.get('SQ_INSTS_VALU', 0):.4g} "
                              f"lane_util {c.get('SQ_THREAD_CYCLES_VALU', 0) / (64 * c['SQ_ACTIVE_INST_VALU']):.3f} "
                              f"wait_share {c.get('SQ_WAIT_ANY', 0) / max(c.get('SQ_WAVE_CYCLES', 1), 1):.3f} "
                              f"total_ms {sum(kernel_ns.get(k, [0])) / 1e6 / max(1, len(glob.glob(os.path.join(out_dir, 'pmc_*')) ) // 2 or 1):.1f}")
    open(os.path.join(dest, f"{tag}_per_kernel.txt"), "w").write("\n".join(per_kernel) + "\n")
    dominant = max(counters, key=lambda k: counters[k].get("SQ_WAVE_CYCLES", 0.0) + counters[k].get("FETCH_SIZE", 0.0))
    c = counters[dominant]
    lines = [f"# rocprofv3 --kernel-trace --pmc <set> -- python tools/bench_small.py 64 1 {workload}  ({workload} 1920x1080, 64 spp, one launch)",
             "# separate passes per counter set; values summed over all XCDs/SEs; kernel: " + dominant, "counter,value"]
    lines += [f"{k},{v:.6g}" for k, v in sorted(c.items())]
    d = {}
    if "SQ_THREAD_CYCLES_VALU" in c and c.get("SQ_ACTIVE_INST_VALU"):
        d["valu_lane_utilisation"] = c["SQ_THREAD_CYCLES_VALU"] / (64 * c["SQ_ACTIVE_INST_VALU"])
    if c.get("SQ_WAVE_CYCLES"):
        d["waiting_share_of_wave_cycles"] = c.get("SQ_WAIT_ANY", 0) / c["SQ_WAVE_CYCLES"]
        d["valu_busy_share_of_wave_cycles"] = c.get("SQ_ACTIVE_INST_VALU", 0) / c["SQ_WAVE_CYCLES"]
    # MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE are in KB; gfx950 tallies 128-B read requests as 64 B -> fetch x2 (upper bound)
    if "FETCH_SIZE" in c:
        d["fetch_bytes_reported"] = c["FETCH_SIZE"] * 1024
        d["fetch_bytes_corrected"] = 2 * c["FETCH_SIZE"] * 1024
    if "WRITE_SIZE" in c:
        d["write_bytes"] = c["WRITE_SIZE"] * 1024
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        d["hbm_bytes_per_64spp_launch"] = d["fetch_bytes_corrected"] + d["write_bytes"]
        d["hbm_bytes_per_sample"] = d["hbm_bytes_per_64spp_launch"] / (1920 * 1080 * 64)
    if dominant in kernel_ns:
        d["kernel_ms_under_pmc_mean"] = sum(kernel_ns[dominant]) / len(kernel_ns[dominant]) / 1e6
    lines += [f"# derived: {k} = {v:.6g}" for k, v in d.items()]
    open(os.path.join(dest, f"{tag}_pmc_{dominant.split('::')[-1]}.csv"), "w").write("\n".join(lines) + "\n")
    json.dump({"kernel": dominant, "workload": workload, "counters": c, "derived": d}, open(os.path.join(dest, f"{tag}_pmc.json"), "w"), indent=1)
print("summaries in", dest, os.listdir(dest))
