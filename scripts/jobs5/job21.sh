# round 5, job 21: priority of the PLL stage's recurrence wave / helpers (the FIR shares their SIMDs)
mkdir -p gpurun_out/r5
run() { timeout 600 python scripts/time_pll_forms.py 0:0x1f 0:0x03 0:0x1f 2>&1 | grep -v amdgpu.ids; }
{
echo "== R 3, helpers 3 (as built)"; run
for cfg in "-DPLLH3_R_PRIO=0 -DPLLH3_H_PRIO=0" "-DPLLH3_R_PRIO=1 -DPLLH3_H_PRIO=0" "-DPLLH3_R_PRIO=3 -DPLLH3_H_PRIO=0"; do
  rm -f gnuais_amd/csrc/build/pll_h3.o
  make -s -C gnuais_amd/csrc EXTRA="$cfg" 2>&1 | grep -iE " error"
  echo "== $cfg"; run
done
} > gpurun_out/r5/job21_pll_prio.txt 2>&1
cat gpurun_out/r5/job21_pll_prio.txt
