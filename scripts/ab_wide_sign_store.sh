#!/bin/bash
# A/B on one GPU box: K1s sign words stored one by one (three words per loop turn, T = 576) vs four words
# of a lane as one 16-byte store (four words per turn, T = 512 / 640 / 768).
cd $GRAFT_REPO_ROOT
run() { for i in 1 2; do GNUAIS_FIR_T=$2 timeout 300 python bench.py --no-cpu --no-others 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 T=$2', round(d['ms_per_step'],4), round(d['kernel_ms']['fir_slice'],4), round(d['kernel_ms_isolated']['fir_slice'],4))"; done; }
build() { rm -f gnuais_amd/csrc/build/fir_scalar.o gnuais_amd/csrc/build/gnuais_capi.o; make -s -C gnuais_amd/csrc EXTRA="$1" 2>&1 | grep -i error; }
python -m pytest tests/test_hip_parity.py -m gpu -x -q 2>&1 | tail -2
for rep in 1 2; do
  build "-DFIR_DIRECT_UNROLL=3"; run narrow 512
  build "-DFIR_DIRECT_UNROLL=4"; run wide 512; run wide 640; run wide 768
done
build ""
