"""Fill and drain of the stage pipeline in a short timed region: every kernel of the LAST n calls of a kernel trace
(bench.py --steps n), start / end relative to the first of them.
usage: region_timeline.py <kernel_trace.csv> [n=20]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
by = collections.defaultdict(list)
for r in rows:
    name = r["Kernel_Name"].split("(")[0].split("::")[-1].split("<")[0]
    by[name].append((int(r["Start_Timestamp"]) / 1e3, int(r["End_Timestamp"]) / 1e3))
names = [k for k in ("fir_sign_kernel", "pll3_kernel", "pll_kernel", "hdlc_events_kernel", "hdlc_crc_kernel") if len(by.get(k, [])) >= n]
for k in names: by[k].sort()
t0 = by[names[0]][-n][0]
end = max(by[k][-1][1] for k in names)
print(f"region: {n} calls, first FIR start -> last kernel end {end - t0:.1f} us = {(end - t0) / n:.1f} us per call")
print("call  " + "  ".join(f"{k[:12]:>25s}" for k in names))
for i in range(n):
    print(f"{i:4d}  " + "  ".join(f"{by[k][-n + i][0] - t0:9.1f} -{by[k][-n + i][1] - t0:9.1f} ({by[k][-n + i][1] - by[k][-n + i][0]:5.0f})" for k in names))
