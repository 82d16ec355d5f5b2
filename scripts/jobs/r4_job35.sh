# round 4, job 35: the round's closing run on the final defaults (FL2, twelve taps, K3 on the deframer's stream): GPU suite, smoke,
# profiles, the node line, then a fuzz soak in every mode
mkdir -p gpurun_out/r4
( timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 ) > gpurun_out/r4/job35_pytest.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r4/job35_smoke.txt 2>&1
bash scripts/collect_profiles.sh r04 > gpurun_out/r4/job35_collect.log 2>&1
timeout 600 python bench.py --gpus 2 --devices 0,0 --steps 20 --warmup 5 > gpurun_out/r4/job35_bench_node.json 2> gpurun_out/r4/job35_bench_node.err
f=gpurun_out/r4/job35_fuzz.txt
rm -f $f
( timeout 700 python scripts/fuzz_parity.py 600 700000 2>&1 | tail -1 ) >> $f
( TABLE=192k timeout 500 python scripts/fuzz_parity.py 400 710000 2>&1 | tail -1 ) >> $f
( DEFRAMER=1 timeout 400 python scripts/fuzz_parity.py 300 720000 2>&1 | tail -1 ) >> $f
( PIPE=1 timeout 400 python scripts/fuzz_parity.py 300 730000 2>&1 | tail -1 ) >> $f
( GNUAIS_FIR_FLAG2=0 timeout 300 python scripts/fuzz_parity.py 200 740000 2>&1 | tail -1 ) >> $f
( GNUAIS_K3_SAME=0 GNUAIS_NBUF=4 timeout 300 python scripts/fuzz_parity.py 200 750000 2>&1 | tail -1 ) >> $f
cat gpurun_out/r4/job35_pytest.txt gpurun_out/r4/job35_smoke.txt $f; tail -3 gpurun_out/r4/job35_collect.log | cut -c1-400
