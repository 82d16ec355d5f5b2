#!/bin/bash
# A/B on one GPU box: sign words per loop turn of the 12-tap K1s (code size of its loop: the instruction
# cache is shared with the PLL stage's, the deframer's and K3's code)
cd $GRAFT_REPO_ROOT
run() { for i in 1 2; do timeout 300 python bench.py --no-cpu --no-others 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],4), {k:round(v,3) for k,v in d['kernel_ms'].items()}, 'iso fir', round(d['kernel_ms_isolated']['fir_slice'],4), 'e2e', round(d['end_to_end']['ms_per_step'],4))"; done; }
build() { rm -f gnuais_amd/csrc/build/fir_scalar.o gnuais_amd/csrc/build/gnuais_capi.o; make -s -C gnuais_amd/csrc EXTRA="$1" 2>&1 | grep -i error; }
python -m pytest tests/test_hip_parity.py -m gpu -x -q 2>&1 | tail -1
for u in 4 2 1 4 2 1; do build "-DFIR_DIRECT_UNROLL=$u"; run unroll$u; done
build ""
