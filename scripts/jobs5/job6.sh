# round 5, job 6: pll_h3 with three and two helpers x deframer width
mkdir -p gpurun_out/r5
for nh in 3 2; do
  rm -f gnuais_amd/csrc/build/pll_h3.o
  make -s -C gnuais_amd/csrc EXTRA="-DPLLH3_HELPERS=$nh" 2>&1 | grep -iE "error"
  echo "== helpers $nh"
  ( timeout 300 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "ragged_chunks or noise_only" 2>&1 | tail -2 )
  timeout 900 python scripts/time_pll_forms.py 8:0x02 8:0x1f:hdlc_lpw=64 8:0x1f:hdlc_lpw=32 8:0x1f:hdlc_lpw=16 8:0x1f:hdlc_lpw=64 3:0x1f 2>&1 | grep -v amdgpu.ids
done > gpurun_out/r5/job6_helpers.txt 2>&1
cat gpurun_out/r5/job6_helpers.txt
