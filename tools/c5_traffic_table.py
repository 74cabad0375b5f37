"""Per-kernel HBM-side traffic of a counter run of tools/pmc_workloads.py:  python tools/c5_traffic_table.py gpurun_out/summary_<tag> [workload]
Reads <tag>_pmc_<workload>.csv (per-launch FETCH_SIZE / WRITE_SIZE and the SQ counters per kernel) and pmc_latest.json (calibration of the
two byte counters on a scratch pattern of known size, schedule of the measured launch) and prints, per kernel: time, GB fetched / written,
bytes per camera sample, GB/s, VALU lane utilisation, waiting share of the wave cycles, VALU wave-instructions per sample."""
import collections, csv, io, json, os, sys

d = sys.argv[1]
wl = sys.argv[2] if len(sys.argv) > 2 else "tr15_like"
p = json.load(open(os.path.join(d, "pmc_latest.json")))
w = p["workloads"][wl]
path = [f for f in os.listdir(d) if f.endswith(f"_pmc_{wl}.csv")][0]
rows = [l for l in open(os.path.join(d, path)) if not l.startswith("#")]
ks = collections.OrderedDict()
for row in csv.DictReader(io.StringIO("".join(rows))):
    ks.setdefault(row["kernel"], {"ms": float(row["ms"])})[row["counter"]] = float(row["value"])
kf, kw = p["calibration_reported_over_known"]["FETCH_SIZE"], p["calibration_reported_over_known"]["WRITE_SIZE"]
S = w["samples_per_launch"]
print(f"# {wl} 1920x1080 {w['spp']} spp, device code {p['device_code_hash']}; schedule of the measured launch: {json.dumps(w.get('schedule'))}")
print(f"# counters per frame launch (rocprofv3 --pmc, one pass per counter set; FETCH_SIZE / WRITE_SIZE in KB, divided by the reported / known ratios {kf:.4f} / {kw:.4f} of tools/scratch_calib)")
print(f"{'kernel':34s} {'ms':>8s} {'fetched GB':>11s} {'written GB':>11s} {'B / sample':>11s} {'GB/s':>7s} {'lanes':>6s} {'waiting':>8s} {'VALU wave-instr / sample':>25s}")
tot = tot_ms = 0.0
for k, v in ks.items():
    f = v.get("FETCH_SIZE", 0) * 1024 / kf; wr = v.get("WRITE_SIZE", 0) * 1024 / kw
    tot += f + wr; tot_ms += v["ms"]
    lanes = v.get("SQ_THREAD_CYCLES_VALU", 0) / (64 * v["SQ_ACTIVE_INST_VALU"]) if v.get("SQ_ACTIVE_INST_VALU") else 0
    wait = v.get("SQ_WAIT_ANY", 0) / max(v.get("SQ_WAVE_CYCLES", 1), 1)
    name = k.replace("void tr::", "")
    print(f"{name[:34]:34s} {v['ms']:8.1f} {f / 1e9:11.2f} {wr / 1e9:11.2f} {(f + wr) / S:11.1f} {(f + wr) / v['ms'] / 1e6 if v['ms'] else 0:7.0f} {lanes:6.3f} {wait:8.3f} {v.get('SQ_INSTS_VALU', 0) / S:25.1f}")
print(f"{'all kernels':34s} {tot_ms:8.1f} {'':11s} {'':11s} {tot / S:11.1f} {tot / tot_ms / 1e6:7.0f}")
