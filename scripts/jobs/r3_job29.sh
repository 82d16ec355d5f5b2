mkdir -p gpurun_out/r3
( timeout 700 python scripts/fuzz_parity.py 600 3000 2>&1 | tail -1
  TABLE=192k timeout 700 python scripts/fuzz_parity.py 600 4000 2>&1 | tail -1
  PIPE=1 timeout 400 python scripts/fuzz_parity.py 300 5000 2>&1 | tail -1
  DEFRAMER=1 timeout 400 python scripts/fuzz_parity.py 300 6000 2>&1 | tail -1
  GNUAIS_FIR_PK=1 timeout 400 python scripts/fuzz_parity.py 300 7000 2>&1 | tail -1
  GNUAIS_FIR_CPL=2 GNUAIS_FIR_FORM=133 timeout 400 python scripts/fuzz_parity.py 300 8000 2>&1 | tail -1
) > gpurun_out/r3/fuzz_soak.txt 2>&1
cat gpurun_out/r3/fuzz_soak.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 10 --warmup 3 --channels 4096 --len 24000 --no-cpu 2>&1 | tail -1 | cut -c1-300
