#!/bin/bash
# A/B on one GPU box: K1s central taps in transposed form (accumulator ring, 18 ops per sample) vs direct
# form with symmetric pre-adds (17 ops), two bench runs each, twice.
cd $GRAFT_REPO_ROOT
run() { for i in 1 2; do timeout 300 python bench.py --no-cpu 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],4), round(d['kernel_ms']['fir_slice'],4), round(d['kernel_ms_isolated']['fir_slice'],4))"; done; }
build() { rm -f gnuais_amd/csrc/build/fir_scalar.o gnuais_amd/csrc/build/gnuais_capi.o; make -s -C gnuais_amd/csrc EXTRA="$1" 2>&1 | grep -i error; }
for rep in 1 2; do
  build "-DK1S_DIRECT_12=0"; run transposed
  build "-DK1S_DIRECT_12=1"; run direct
done
build ""
