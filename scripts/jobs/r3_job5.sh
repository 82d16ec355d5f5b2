mkdir -p gpurun_out/r3/pmc
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $R/gpurun_out/r3/counters_list.txt 2>&1
for spec in "1 0" "2 1"; do set -- $spec
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_LEVEL_WAVES" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_IFETCH SQ_INSTS_VALU" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  tag=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/gpurun_out/r3/pmc/c$1_$tag -o pmc -- python $R/scripts/fir_only_loop.py $1 $2 8 > $R/gpurun_out/r3/pmc/c$1_$tag.log 2>&1
done; done
cd $R
python - <<'PY'
import csv, glob, collections, os
for d in sorted(glob.glob("gpurun_out/r3/pmc/c*_*")):
    if not os.path.isdir(d): continue
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: [0.0, 0])
        for r in csv.DictReader(open(f)):
            if "fir_sign" not in r["Kernel_Name"]: continue
            a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
        print(d, {k: round(v[0] / max(v[1], 1)) for k, v in acc.items()})
PY
find gpurun_out/r3/pmc -name "*.csv" -size +1M -delete
