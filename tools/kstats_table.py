"""Per-kernel table of a rocprofv3 --kernel-trace --stats run: python tools/kstats_table.py <dir> -> name, calls, total ms, share, avg us"""
import csv, glob, os, re, sys
f = [p for p in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_stats.csv"), recursive=True)]
rows = list(csv.DictReader(open(f[0])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:24]:
    name = re.sub(r"\(.*", "", r["Name"]).replace("void tr::", "").replace("void ", "")
    print(f"{name:34s} calls {int(r['Calls']):6d}  total {float(r['TotalDurationNs']) / 1e6:9.1f} ms  {100 * float(r['TotalDurationNs']) / tot:5.1f} %  avg {float(r['AverageNs']) / 1e3:9.1f} us")
