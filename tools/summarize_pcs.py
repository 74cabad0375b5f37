"""Condense a rocprofv3 PC-sampling CSV (host_trap or stochastic) of one tile-kernel launch into a per-source-line table:
    samples, share, mean active lanes (popcount of Exec_Mask), and for stochastic samples the share that issued an instruction.
usage: python tools/summarize_pcs.py <dir with *pc_sampling*.csv> <out prefix> [kernel substring]
The library must be built with -gline-tables-only (tools/variant.sh g -gline-tables-only) so that Instruction_Comment carries file:line."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

src, out = sys.argv[1], sys.argv[2]
files = [f for f in glob.glob(os.path.join(src, "**", "*.csv"), recursive=True) if "pc_sampling" in os.path.basename(f) and "stats" not in os.path.basename(f)]
if not files:
    print("no pc sampling csv under", src); sys.exit(1)
csv.field_size_limit(1 << 30)
by_line = defaultdict(lambda: [0, 0, 0])   # samples, lanes, issued
by_file = defaultdict(lambda: [0, 0, 0])
by_inst = defaultdict(lambda: [0, 0, 0])
by_stall = defaultdict(int)
total = 0
header = None
for f in files:
    with open(f, newline="") as fh:
        rd = csv.DictReader(fh)
        header = rd.fieldnames
        for row in rd:
            comment = row.get("Instruction_Comment") or ""
            inst = (row.get("Instruction") or "").split(" ")[0]
            try:
                lanes = bin(int(row.get("Exec_Mask") or "0")).count("1")
            except ValueError:
                lanes = bin(int(row.get("Exec_Mask"), 16)).count("1")
            issued = 1 if (row.get("Wave_Issued_Instruction") or "1") in ("1", "true", "True") else 0
            m = re.search(r"([A-Za-z0-9_./-]+\.(?:h|hip|hpp|cpp)):(\d+)", comment)
            key = (os.path.basename(m.group(1)), int(m.group(2))) if m else ("?", 0)
            for table, k in ((by_line, key), (by_file, key[0]), (by_inst, inst)):
                e = table[k]; e[0] += 1; e[1] += lanes; e[2] += issued
            if "Stall_Reason" in row: by_stall[row["Stall_Reason"]] += 1
            total += 1
with open(out + "_lines.txt", "w") as o:
    o.write(f"# {total} samples from {len(files)} file(s); columns: {header}\n")
    o.write("# share  samples  lanes/64  issued  file:line\n")
    for k, (n, l, i) in sorted(by_line.items(), key=lambda kv: -kv[1][0]):
        if n * 5000 < total: continue
        o.write(f"{100.0 * n / total:6.2f}% {n:8d}  {l / n / 64:5.2f}  {i / n:5.2f}  {k[0]}:{k[1]}\n")
with open(out + "_summary.txt", "w") as o:
    o.write(f"# {total} samples; mean active lanes {sum(v[1] for v in by_file.values()) / max(total, 1) / 64:.3f}\n# by file\n")
    for k, (n, l, i) in sorted(by_file.items(), key=lambda kv: -kv[1][0]):
        o.write(f"{100.0 * n / total:6.2f}% {n:8d}  lanes {l / n / 64:5.2f}  issued {i / n:5.2f}  {k}\n")
    o.write("# by opcode\n")
    for k, (n, l, i) in sorted(by_inst.items(), key=lambda kv: -kv[1][0])[:60]:
        o.write(f"{100.0 * n / total:6.2f}% {n:8d}  lanes {l / n / 64:5.2f}  issued {i / n:5.2f}  {k}\n")
    if by_stall:
        o.write("# by stall reason\n")
        for k, n in sorted(by_stall.items(), key=lambda kv: -kv[1]):
            o.write(f"{100.0 * n / total:6.2f}% {n:8d}  {k}\n")
print(open(out + "_summary.txt").read())
