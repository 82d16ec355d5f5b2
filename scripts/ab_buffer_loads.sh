#!/bin/bash
# A/B on one GPU box: K1s with plain global loads vs buffer loads (descriptor + scalar row offset), two
# bench runs each, twice.  Round 1: plain 0.825 / 0.815 / 0.827 / 0.833, buffer 0.810 / 0.808 / 0.804 / 0.804 ms per call.
cd $GRAFT_REPO_ROOT
run() { for i in 1 2; do timeout 300 python bench.py --no-cpu 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],4), round(d['kernel_ms']['fir_slice'],4), round(d['kernel_ms_isolated']['fir_slice'],4))"; done; }
build() { rm -f gnuais_amd/csrc/build/fir_scalar.o; make -s -C gnuais_amd/csrc EXTRA="$1" 2>&1 | grep -i error; }
for rep in 1 2; do
  build "-DFIR_BUFFER_LOADS=${A:-0}"; run "loads=${A:-0}"
  build "-DFIR_BUFFER_LOADS=${B:-1}"; run "loads=${B:-1}"
done
build ""
