"""Print a rocprofv3 kernel_stats.csv as name / calls / total ms / average us: python scripts/kstats.py <dir-or-file> [rows]."""
import csv, os, sys
path = sys.argv[1]
if os.path.isdir(path):
    path = next(os.path.join(d, f) for d, _, fs in os.walk(path) for f in fs if f.endswith("kernel_stats.csv"))
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 16
for i, r in enumerate(csv.reader(open(path))):
    if i == 0 or i > rows:
        continue
    print(r[0][:70].ljust(72), r[1].rjust(7), "%9.3f ms" % (float(r[2]) / 1e6), "%9.2f us" % (float(r[3]) / 1e3))
