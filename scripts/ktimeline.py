"""Timeline of the last solve in a rocprofv3 kernel trace: python scripts/ktimeline.py <dir> [n_last_kernels]
Prints start offset, duration and the gap to the previous kernel's end (same process), so that launch gaps / tails show."""
import csv, os, sys
path = sys.argv[1]
f = next(os.path.join(d, x) for d, _, fs in os.walk(path) for x in fs if x.endswith("kernel_trace.csv"))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-n:]
t0 = int(rows[0]["Start_Timestamp"]); prev_end = t0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].replace("void sadvio::", "")[:34]
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.1f}  gap {(s - prev_end) / 1e3:6.1f}  {name}  grid {r.get('Grid_Size_X', r.get('Grid_Size', ''))} wg {r.get('Workgroup_Size_X', r.get('Workgroup_Size', ''))}")
    prev_end = max(prev_end, e)
