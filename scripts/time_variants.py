"""Exploration: per-kernel timing of the chain at BASELINE C3 size for the K1
variants / segment lengths (HIP events inside the library)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gnuais_amd import ReceiverBatch, synth, tile_channels

n_ch = int(os.environ.get("NCH", 16384)); total = int(os.environ.get("LEN", 48000))
base, placed = synth.make_base_streams(64, total)
x = tile_channels(torch.from_numpy(base).cuda(), n_ch)
torch.cuda.synchronize()
b = ReceiverBatch(n_ch, max_len=total)
b.set_timing(True)
for variant in (0, 2):
    for T in (512, 1024):
        b.set_option("fir_variant", variant); b.set_option("fir_T", T)
        res = []
        for it in range(4):
            b.run(x); res.append(b.last_timing()); b.drain_frames()
        r = res[-1]
        print(f"variant={variant} T={T}: fir {r['fir_slice']:.3f} ms  pll {r['pll']:.3f} ms  hdlc {r['hdlc_deframe']:.3f}+{r['hdlc_crc']:.3f} ms  total {r['total']:.3f} ms  "
              f"-> fir {n_ch*total/r['fir_slice']/1e9:.3f} Tsample/s, chain {n_ch*total/r['total']/1e9:.3f} Tsample/s", flush=True)
for lpw in (1, 2, 4, 8, 16, 32, 64):
    b.set_option("hdlc_lpw", lpw)
    for it in range(3):
        b.run(x); r = b.last_timing(); b.drain_frames()
    print(f"hdlc_lpw={lpw}: hdlc {r['hdlc_deframe']:.3f}+{r['hdlc_crc']:.3f} ms total {r['total']:.3f} ms", flush=True)
print("received total", b.total_received())
