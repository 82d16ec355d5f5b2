# round 5, job 16: wave priorities of the deframer and of the PLL stage's helpers inside the C3 pipeline
mkdir -p gpurun_out/r5
run() { timeout 600 python scripts/time_pll_forms.py 0:0x1f 0:0x1f 2>&1 | grep -v amdgpu.ids; }
{
echo "== as built (deframer 3, helpers 3)"; run
for cfg in "-DEV_PRIO=1" "-DEV_PRIO=0" "-DPLLH3_HELPER_PRIO=1" "-DPLLH3_HELPER_PRIO=0" "-DEV_PRIO=1 -DPLLH3_HELPER_PRIO=1"; do
  rm -f gnuais_amd/csrc/build/hdlc_events.o gnuais_amd/csrc/build/pll_h3.o
  make -s -C gnuais_amd/csrc EXTRA="$cfg" 2>&1 | grep -iE " error"
  echo "== $cfg"; run
done
} > gpurun_out/r5/job16_prio.txt 2>&1
cat gpurun_out/r5/job16_prio.txt
