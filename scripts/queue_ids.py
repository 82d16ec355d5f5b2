"""Which queue did each kernel of each batch object run on?  (reads a rocprofv3 kernel trace of
MODE=seq scripts/alloc_experiment.py)"""
import csv, sys
from collections import defaultdict, OrderedDict
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# split into batches by the hdlc_reset kernel (one per create/reset)
batch = -1
seen = defaultdict(lambda: OrderedDict())
for r in rows:
    n = r["Kernel_Name"].split("(")[0].split("::")[-1]
    if "hdlc_reset" in n:
        batch += 1
    if batch >= 0 and any(k in n for k in ("fir_sign", "pll_kernel", "hdlc_deframe", "hdlc_crc_kernel")):
        seen[batch].setdefault(n[:20], set()).add(r["Queue_Id"])
for b, d in seen.items():
    print("batch", b, {k: sorted(v) for k, v in d.items()})
