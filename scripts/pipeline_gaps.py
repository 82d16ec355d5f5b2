"""Steady-state structure of the stage pipeline from a rocprofv3 kernel trace: per kernel the mean duration, the
start-to-start period and the idle gap on its queue, over the calls in the middle of the timed loop.
usage: pipeline_gaps.py <kernel_trace.csv> [first_call last_call]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
lo, hi = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (20, 40)
by = collections.defaultdict(list)
for r in rows:
    name = r["Kernel_Name"].split("(")[0].split("::")[-1].split("<")[0]
    by[name].append((int(r["Start_Timestamp"]) / 1e3, int(r["End_Timestamp"]) / 1e3))
for name in ("fir_sign_kernel", "pll3_kernel", "pll_kernel", "hdlc_events_kernel", "hdlc_crc_kernel"):
    v = by.get(name, [])[lo:hi]
    if len(v) < 3: continue
    dur = sum(e - s for s, e in v) / len(v)
    per = (v[-1][0] - v[0][0]) / (len(v) - 1)
    gap = sum(v[i + 1][0] - v[i][1] for i in range(len(v) - 1)) / (len(v) - 1)
    print(f"{name:22s} calls {lo}..{hi}: duration {dur:7.1f} us  period {per:7.1f} us  gap to the next launch of it {gap:7.1f} us")
f = by["fir_sign_kernel"][lo:hi]; p = (by.get("pll3_kernel") or by.get("pll_kernel"))[lo:hi]; d = by["hdlc_events_kernel"][lo:hi]; c = by["hdlc_crc_kernel"][lo:hi]
print("call: FIR start | FIR end -> PLL start | PLL end -> K2b start | K2b end -> K3 start | K3 end   (us, relative to the call's FIR start)")
for i in range(min(8, len(f))):
    t = f[i][0]
    print(f"  {f[i][0]-f[0][0]:8.1f} | {f[i][1]-t:6.1f} -> {p[i][0]-t:6.1f} | {p[i][1]-t:6.1f} -> {d[i][0]-t:6.1f} | {d[i][1]-t:6.1f} -> {c[i][0]-t:6.1f} | {c[i][1]-t:6.1f}")
