#!/bin/bash
# A/B on one GPU box: the three-wave PLL kernel (recurrence toggles its own bits, one scanner) against the
# six-wave one (two scanners, recurrence, two togglers, writer), alone and beside the other stages.
cd $GRAFT_REPO_ROOT
cat > /tmp/masks.py <<'PY'
import sys, os, time
sys.path.insert(0, os.getcwd())
import torch
from gnuais_amd import ReceiverBatch, synth, tile_channels
n_ch, total = 16384, 48000
base, _ = synth.make_base_streams(64, total)
x = tile_channels(torch.from_numpy(base).cuda(), n_ch)
res = []
for mask in (0x01, 0x02, 0x03, 0x0b, 0x1f):
    b = ReceiverBatch(n_ch, max_len=total)
    b.autotune(x)
    b.set_option("stage_mask", mask)
    for i in range(20): b.run(x, sync=False); b.discard_frames()
    b.sync(); torch.cuda.synchronize(); t = time.perf_counter()
    for i in range(150): b.run(x, sync=False); b.discard_frames()
    b.sync(); torch.cuda.synchronize()
    res.append((hex(mask), round((time.perf_counter() - t) / 150 * 1e3, 4)))
    del b
print(sys.argv[1], res, flush=True)
PY
build() { rm -f gnuais_amd/csrc/build/pll_nrzi.o; make -s -C gnuais_amd/csrc 2>&1 | grep -i error; }
cp gnuais_amd/csrc/pll_nrzi.hip /tmp/pll_new.hip
for rep in 1 2; do
  cp scripts/variants/pll_nrzi_3wave.hip.txt gnuais_amd/csrc/pll_nrzi.hip; build; python /tmp/masks.py three-wave 2>&1 | grep wave
  cp /tmp/pll_new.hip gnuais_amd/csrc/pll_nrzi.hip; build; python /tmp/masks.py six-wave 2>&1 | grep wave
done
