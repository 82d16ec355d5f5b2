#!/bin/bash
# per-kernel counters of the C5 chain's FIR (PMC pass: kernel trace only beside it)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/c5pmc
NCH=${NCH:-16384} rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU -d $R/gpurun_out/c5pmc -o c5 --output-format csv -- python $R/scripts/time_c5_taps.py ${1:-0} > $R/gpurun_out/c5pmc.log 2>&1
tail -2 $R/gpurun_out/c5pmc.log
python - <<'P'
import csv, glob, os, collections
root=os.environ['GRAFT_REPO_ROOT']
f=glob.glob(root+'/gpurun_out/c5pmc/**/*counter_collection.csv', recursive=True)
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for p in f:
    for r in csv.DictReader(open(p)):
        k=r['Kernel_Name'][:60]
        acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k,d in acc.items():
    if 'fir' not in k: continue
    print(k, {c: round(sum(v)/len(v)) for c,v in d.items()}, 'launches', len(next(iter(d.values()))))
P
