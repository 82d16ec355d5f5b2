# round 5, job 20: the FIR's unrolled body against the instruction cache the stages share (63 KB of code at four words per turn)
mkdir -p gpurun_out/r5
run() { timeout 600 python scripts/time_pll_forms.py 0:0x01 0:0x1f 0:0x1f 0:0x03 2>&1 | grep -v amdgpu.ids; }
{
echo "== four words per loop turn (as built)"; run
for u in 2 1; do
  rm -f gnuais_amd/csrc/build/fir_scalar.o
  make -s -C gnuais_amd/csrc EXTRA="-DFIR_DIRECT_UNROLL=$u" 2>&1 | grep -iE " error"
  echo "== FIR_DIRECT_UNROLL=$u"; run
done
} > gpurun_out/r5/job20_fir_unroll.txt 2>&1
cat gpurun_out/r5/job20_fir_unroll.txt
