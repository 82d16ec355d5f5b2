"""C3 pipelined step for the PLL workgroup forms x deframer widths (autotuned stream placement each)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gnuais_amd import ReceiverBatch, synth, tile_channels
n_ch, total = 16384, 48000
base, _ = synth.make_base_streams(64, total)
x = tile_channels(torch.from_numpy(base).cuda(), n_ch)
for rep in range(2):
    for pv in (3, 6):
        for lpw in (16, 32, 64):
            b = ReceiverBatch(n_ch, max_len=total)
            b.set_option("pll_variant", pv); b.set_option("hdlc_lpw", lpw)
            b.autotune(x)
            b.set_option("pll_variant", pv); b.set_option("hdlc_lpw", lpw)
            for i in range(20): b.run(x, sync=False); b.discard_frames()
            b.sync(); torch.cuda.synchronize(); t = time.perf_counter()
            for i in range(150): b.run(x, sync=False); b.discard_frames()
            b.sync(); torch.cuda.synchronize()
            print("pll", pv, "lpw", lpw, round((time.perf_counter() - t) / 150 * 1e3, 4), flush=True)
            del b
