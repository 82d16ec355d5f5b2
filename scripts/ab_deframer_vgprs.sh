#!/bin/bash
# A/B on one GPU box: VGPR cap of the event deframer (116 free; 96 = five waves per SIMD, 26 spills; 80, 40 spills)
cd $GRAFT_REPO_ROOT
run() { for i in 1 2; do timeout 300 python bench.py --no-cpu --no-others 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],4), {k:round(v,3) for k,v in d['kernel_ms'].items()}, 'iso', round(d['kernel_ms_isolated']['hdlc_deframe'],4))"; done; }
build() { rm -f gnuais_amd/csrc/build/hdlc_events.o; make -s -C gnuais_amd/csrc EXTRA="$1" 2>&1 | grep -i error; }
for v in 0 5 6 0 5; do build "-DEV_WAVES_PER_EU=$v"; run occ$v; done
build ""
