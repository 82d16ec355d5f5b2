#!/bin/bash
# A/B on one GPU box: the 48-tap K1s instantiation with plain loads vs typed buffer loads (vector row offset)
# Round 1: plain 6.73 / 6.82 ms FIR per 16384 x 192000 call, typed 6.56 / 6.63.
cd $GRAFT_REPO_ROOT
build() { rm -f gnuais_amd/csrc/build/fir_scalar.o; make -s -C gnuais_amd/csrc EXTRA="$1" 2>&1 | grep -i error; }
for v in 0 1 0 1; do build "-DFIR_TYPED_LOADS_48=$v"; echo "typed48=$v"; FAST=1 timeout 300 python scripts/time_c5.py 2>&1 | tail -2 | cut -c1-150; done
build ""
