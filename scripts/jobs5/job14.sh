# round 5, job 14: the exact FIR's register claim (exact chain), and the sign-exact FIR's segment length
mkdir -p gpurun_out/r5
run() { timeout 600 python scripts/time_pll_forms.py "$@" 2>&1 | grep -v amdgpu.ids; }
{
echo "== exact FIR as built (85 registers, five waves per SIMD)"
run 0:0x1f:fir_variant=0 0:0x01:fir_variant=0
for claim in v103 v127; do
  rm -f gnuais_amd/csrc/build/fir_scalar.o
  make -s -C gnuais_amd/csrc EXTRA="-DFIR_EXACT_CLAIM='\"$claim\"'" 2>&1 | grep -iE " error"
  echo "== exact FIR claiming $claim"
  run 0:0x1f:fir_variant=0 0:0x01:fir_variant=0
done
echo "== K1s segment length"
run 0:0x1f 0:0x1f:fir_T=1024 0:0x1f:fir_T=2048 0:0x01:fir_T=512 0:0x01:fir_T=1024 0:0x01:fir_T=2048 0:0x1f:fir_T=256 0:0x1f
} > gpurun_out/r5/job14_exact_claim_and_T.txt 2>&1
cat gpurun_out/r5/job14_exact_claim_and_T.txt
