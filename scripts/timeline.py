"""Print the kernel timeline of the last steps of a rocprofv3 --kernel-trace CSV.

usage: python scripts/timeline.py <kernel_trace.csv> [n_last_rows]
"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-n:]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    s = (int(r["Start_Timestamp"]) - t0) / 1e3
    e = (int(r["End_Timestamp"]) - t0) / 1e3
    name = r["Kernel_Name"].split("(")[0].split("::")[-1][:28]
    print(f"{s:10.1f} {e:10.1f} {e - s:8.1f} us  q={r.get('Queue_Id','?'):>3} {name}")
