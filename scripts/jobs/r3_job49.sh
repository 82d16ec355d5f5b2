cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_pll_forms.txt
echo "scripts/time_pll4.py: the PLL workgroup's forms (pll_variant 3 / 32 / 4 / 51 / 52 / 6), 256 channels alone (NCH=256: ms/step ~ the PLL stage) and C3 inside the pipeline (ms/step = the period; per-stage in-pipeline durations)" > $O
echo "== 256 channels" >> $O
NCH=256 REPS=1 LPWS=16 PVS=3,32,4,51,52,6 timeout 300 python scripts/time_pll4.py 2>&1 | grep "^lag" >> $O
echo "== C3 (16384 channels)" >> $O
REPS=2 LPWS=16 PVS=3,32,4,51,52,6 timeout 400 python scripts/time_pll4.py 2>&1 | grep "^lag" >> $O
cat $O
