"""C3 steady state (200 steps) per PLL form x stages running: does a shorter PLL launch shorten the period now that the
FIR issues a quarter fewer instructions?  usage: time_pll_forms.py [variant:mask ...]  (mask: 1 FIR, 2 PLL, 8 deframer, 16 K3)"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gnuais_amd import ReceiverBatch, synth, tile_channels
n_ch, total = int(os.environ.get("NCH", 16384)), 48000
base, _ = synth.make_base_streams(256, total)
x = tile_channels(torch.from_numpy(base).cuda(), n_ch)
stream = torch.cuda.current_stream().cuda_stream


def measure(variant, mask, extra):
    b = ReceiverBatch(n_ch, max_len=total)
    b.set_option("pll_variant", variant)
    for k, v in extra.items():
        b.set_option(k, v)
    b.autotune(x, stream)

    def step():
        b.run(x, stream=stream, sync=False)
        b.discard_frames(stream)
    for _ in range(8):
        step()
    b.sync()
    torch.cuda.synchronize()
    b.set_option("stage_mask", mask)
    for _ in range(10):
        step()
    b.sync()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(20):
        step()
    torch.cuda.synchronize()
    short = (time.perf_counter() - t) / 20 * 1e3
    b.set_timing(True)
    b.set_option("timing_stride", 4)
    t = time.perf_counter()
    for _ in range(200):
        step()
    b.sync()
    torch.cuda.synchronize()
    steady = (time.perf_counter() - t) / 200 * 1e3
    live = b.mean_timing()
    del b
    return short, steady, {k: round(float(live[k]), 3) for k in ("fir_slice", "pll", "hdlc_deframe", "hdlc_crc")}


for a in sys.argv[1:] or ["8:31"]:
    parts = a.split(":")
    variant, mask = int(parts[0]), int(parts[1], 0)
    extra = dict(kv.split("=") for kv in parts[2:])
    extra = {k: int(v) for k, v in extra.items()}
    short, steady, k = measure(variant, mask, extra)
    print(f"pll_variant {variant} stage_mask {mask:#04x} {extra}: 20 steps {short:.3f}  steady {steady:.3f} ms/step  {k}", flush=True)
