#!/bin/bash
# A/B on one GPU box: VGPRs the 12-tap K1s claims (waves per SIMD it leaves to the other stages)
cd $GRAFT_REPO_ROOT
run() { for i in 1 2; do timeout 300 python bench.py --no-cpu --no-others 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],4), {k:round(v,3) for k,v in d['kernel_ms'].items()}, 'iso fir', round(d['kernel_ms_isolated']['fir_slice'],4), 'e2e', round(d['end_to_end']['ms_per_step'],4))"; done; }
build() { rm -f gnuais_amd/csrc/build/fir_scalar.o; make -s -C gnuais_amd/csrc EXTRA="$1" 2>&1 | grep -i error; }
for c in v87 v71 v79 v95 v87 v71; do build "-DFIR_SIGN_CLAIM=\\\"$c\\\""; run $c; done
build ""
