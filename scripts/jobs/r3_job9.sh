mkdir -p gpurun_out/r3
( python scripts/time_chain_matrix.py; GNUAIS_K2B_LAG=2 python scripts/time_chain_matrix.py ) > gpurun_out/r3/chain_matrix.txt 2>&1
cat gpurun_out/r3/chain_matrix.txt
