cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_ubench_mfma_slicer.txt
echo "scripts/ubench/mfma_slicer.hip, -DPFD=2 -mllvm -amdgpu-mfma-vgpr-form, C3 shape (16384 x 48000), T 1536; STAGE: 0 loads only, 1 + int8 digits (xor, v_perm), 2 + 30 MFMAs, 3 + combine and flag words, 4 + word assembly and stores (the whole slicer, signs checked)" > $O
for st in 0 1 2 3 4; do echo "== STAGE $st" >> $O; timeout 120 scripts/ubench/mfma_slicer_s$st.bin 16384 48000 1536 | grep "ms per\|signs" | tail -2 >> $O; done
cat $O
