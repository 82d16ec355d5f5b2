#!/bin/bash
# A/B on one GPU box: block slots between the PLL kernel's scanner, recurrence and togglers.
cd $GRAFT_REPO_ROOT
run() { for i in 1 2; do timeout 300 python bench.py --no-cpu 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); o=d['other_configs']; print('$1', round(d['ms_per_step'],4), {k:round(v,3) for k,v in d['kernel_ms'].items()}, 'iso pll', round(d['kernel_ms_isolated']['pll'],4), 'e2e', round(d['end_to_end']['ms_per_step'],4), 'C2', round(o['C2']['ms_per_step'],4), 'C5', round(o['C5']['ms_per_step'],3))"; done; }
build() { rm -f gnuais_amd/csrc/build/pll_nrzi.o; make -s -C gnuais_amd/csrc EXTRA="$1" 2>&1 | grep -i error; }
python -m pytest tests/test_hip_parity.py -m gpu -x -q 2>&1 | tail -2
for s in 6 4 5; do build "-DPLL_SLOTS_N=$s"; run slots$s; done
build ""
