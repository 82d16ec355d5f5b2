mkdir -p gpurun_out/r3/pmc3
R=$PWD
cd /tmp && export TMPDIR=/tmp
for pk in 1 0; do
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "GRBM_GUI_ACTIVE"; do
  tag=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/gpurun_out/r3/pmc3/pk${pk}_$tag -o pmc -- python $R/scripts/fir_only_loop.py 1 0 6 768 $pk > $R/gpurun_out/r3/pmc3/pk${pk}_$tag.log 2>&1
done; done
cd $R
python - <<'PY'
import csv, glob, collections, os
for d in sorted(glob.glob("gpurun_out/r3/pmc3/pk*_*")):
    if not os.path.isdir(d): continue
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: [0.0, 0])
        for r in csv.DictReader(open(f)):
            if "fir_sign" not in r["Kernel_Name"]: continue
            a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
        print(d.split("/")[-1], {k: round(v[0] / max(v[1], 1)) for k, v in acc.items()})
PY
find gpurun_out/r3/pmc3 -name "*.csv" -delete
