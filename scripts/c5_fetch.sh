#!/bin/bash
# FETCH_SIZE per launch of C5's FIR kernels for a given option string of scripts/time_c5_taps.py (calibration of the counter
# on this access pattern: the prologue's share of a segment changes with fir_T, the rest does not)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for a in "$@"; do
  rm -rf $R/gpurun_out/c5f
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/c5f -o c5 --output-format csv -- python $R/scripts/time_c5_taps.py $a > /dev/null 2>&1
  python - "$a" <<'P'
import csv, glob, os, sys, collections
root=os.environ['GRAFT_REPO_ROOT']
acc=collections.defaultdict(list)
for p in glob.glob(root+'/gpurun_out/c5f/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(p)):
        if 'fir_sign' in r['Kernel_Name']: acc[r['Kernel_Name'][:48]].append(float(r['Counter_Value']))
print(sys.argv[1], {k: round(sum(v)/len(v)*1024*2/1e9, 3) for k,v in acc.items()}, 'GB per launch (FETCH_SIZE KiB x 2); input 6.291 GB')
P
done
