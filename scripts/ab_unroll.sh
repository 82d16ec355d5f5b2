#!/bin/bash
# A/B on one GPU box: words unrolled per loop turn in K1s's direct form (code size vs tail moves)
# Round 1: unroll 3: 0.798 0.793 0.794 0.796; unroll 2: 0.801 0.795 0.796 0.795; unroll 1: 0.806 0.806 0.808 0.807 ms per call -> 3 stays.
cd $GRAFT_REPO_ROOT
run() { for i in 1 2; do timeout 300 python bench.py --no-cpu 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],4), round(d['kernel_ms']['fir_slice'],4), round(d['kernel_ms_isolated']['fir_slice'],4))"; done; }
build() { rm -f gnuais_amd/csrc/build/fir_scalar.o; make -s -C gnuais_amd/csrc EXTRA="$1" 2>&1 | grep -i error; }
for u in 3 2 1 3 2 1; do build "-DFIR_DIRECT_UNROLL=$u"; run "unroll=$u"; done
build ""
