"""Counter A/B of library builds on one workload (one rocprofv3 --pmc pass per build and counter set, torch-free runner):
    python tools/pmc_ab.py <workload:spp> <lib.so>...          -> gpurun_out/pmc_ab.txt
per build: VALU wave-instructions per camera sample, lane utilisation, VALU busy, waiting share, SALU / SMEM / LDS instructions per sample."""
import csv, glob, os, subprocess, sys, shutil

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
wl = sys.argv[1]
name, spp = wl.split(":")
samples = 1920 * 1080 * int(spp)
SETS = ["SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU",
        "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_FLAT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA"]
if os.environ.get("PMC_SETS"): SETS = SETS[:int(os.environ["PMC_SETS"])]
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
out = open(os.path.join(ROOT, "gpurun_out", "pmc_ab.txt"), "a")
subprocess.run(["python", os.path.join(ROOT, "tools", "mini_ab.py"), "prepare", "/tmp/mini_ab"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
for lib in sys.argv[2:]:
    c, ns = {}, []
    for i, s in enumerate(SETS):
        d = f"/tmp/pmc_ab_{os.path.basename(lib)}_{i}"
        shutil.rmtree(d, ignore_errors=True)
        env = dict(os.environ, TRAYHIP_LIB=os.path.join(ROOT, "tray_rust_amd", lib), TMPDIR="/tmp")
        subprocess.run(["timeout", "120", "rocprofv3", "--kernel-trace", "--pmc"] + s.split() + ["--output-format", "csv", "-d", d, "--",
                        "python", os.path.join(ROOT, "tools", "mini_ab.py"), "run", "/tmp/mini_ab", "pmc", wl], cwd="/tmp", env=env,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        n_launch = 0
        for path in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
            for row in csv.DictReader(open(path)):
                if "k_path_tiles" in row["Kernel_Name"]:
                    n_launch += 1
                    if i == 0: ns.append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
        for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(path)):
                if "k_path_tiles" in row["Kernel_Name"]:
                    c[row["Counter_Name"]] = c.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"]) / max(n_launch, 1)
    if not c.get("SQ_INSTS_VALU"):
        line = f"{lib}: no counters"
    else:
        nsimd = 1024
        line = (f"{lib:28s} {wl}: VALU wave-instr/sample {c['SQ_INSTS_VALU'] / samples:7.1f}  lanes {c['SQ_THREAD_CYCLES_VALU'] / (64 * c['SQ_ACTIVE_INST_VALU']):.3f}  "
                f"VALU busy {c['SQ_ACTIVE_INST_VALU'] * c['SQ_WAVES'] / (nsimd * c['SQ_WAVE_CYCLES']):.3f}  waiting {c['SQ_WAIT_ANY'] / c['SQ_WAVE_CYCLES']:.3f}  "
                f"issue-stall {c['SQ_WAIT_INST_ANY'] / c['SQ_WAVE_CYCLES']:.3f}  cyc/VALU {4 * c['SQ_ACTIVE_INST_VALU'] / c['SQ_INSTS_VALU']:.2f}  "
                f"SALU/sample {c.get('SQ_INSTS_SALU', 0) / samples:6.1f}  SMEM {c.get('SQ_INSTS_SMEM', 0) / samples:5.1f}  LDS {c.get('SQ_INSTS_LDS', 0) / samples:5.1f}  "
                f"VMEM {(c.get('SQ_INSTS_VMEM_RD', 0) + c.get('SQ_INSTS_VMEM_WR', 0)) / samples:5.2f}  ms(pmc) {sum(ns) / max(len(ns), 1) / 1e6:.1f}")
    print(line, flush=True); out.write(line + "\n"); out.flush()
