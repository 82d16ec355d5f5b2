"""Reads a rocprofv3 --kernel-trace CSV of scripts/time_stream_nmea.py and prints per-kernel periods,
durations and a window of the timeline."""
import csv, collections, sys
rows = list(csv.DictReader(open(sys.argv[1])))
KEYS = ('fir_sign', 'pll_kernel', 'hdlc_events', 'hdlc_crc', 'nmea_write', 'chunk_scan', 'nmea_meta', 'wrapped_scan',
        'init_lookback', 'text_copy', 'slot_info', 'copyBuffer', 'fillBuffer')
ks = collections.defaultdict(list)
for r in rows:
    for key in KEYS:
        if key in r['Kernel_Name']:
            ks[key].append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r.get('Queue_Id')))
t0 = min(v[0][0] for v in ks.values())
for k, v in ks.items():
    v.sort()
    st = [x[0] for x in v]
    per = [(st[i + 1] - st[i]) / 1e3 for i in range(len(st) - 1)]
    print(k, 'q', set(x[2] for x in v), 'n', len(v), 'start-to-start', [round(p) for p in per[30:38]], 'dur', [round((x[1] - x[0]) / 1e3) for x in v[30:38]])
ev = sorted((x[0], x[1], k, x[2]) for k, v in ks.items() for x in v)
lo = ks['fir_sign'][40][0]
for a, b, k, q in ev:
    if lo <= a < lo + float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else lo + 3e6:
        print(f"{(a - t0) / 1e3:10.0f} {(b - t0) / 1e3:10.0f} q{q} {k}")
