"""C5 (BASELINE configs: 16384 channels at 192 kHz = 144 taps / pllinc 3276, 192000 samples per call):
isolated per-kernel times of the chain with the sign-exact slicer (48 central taps) and with the
generic direct-form FIR, then the pipelined steady state.
    LEN=192000 python scripts/time_c5.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                                                  # noqa: E402
from gnuais_amd import ReceiverBatch, params, synth, tile_channels          # noqa: E402

n_ch, total = 16384, int(os.environ.get("LEN", 192000))
base, _ = synth.make_base_streams(64, total, sps=20)
x = tile_channels(torch.from_numpy(base).cuda(), n_ch)
for variant in (3, 0) if not os.environ.get("FAST") else (3,):
    b = ReceiverBatch(n_ch, taps=params.taps_192k(), pllinc=params.PLLINC_192K, max_len=total)
    b.set_option("fir_variant", variant)
    b.set_option("pipeline", 0)
    b.set_timing(True)
    ts = []
    for _ in range(3):
        b.run(x, sync=True)
        b.discard_frames()
        ts.append(b.last_timing())
    t = min(ts, key=lambda d: d["fir_slice"])
    print("fir_variant", variant, {k: round(v, 3) for k, v in t.items()}, "received", b.total_received(),
          flush=True)
    del b

b = ReceiverBatch(n_ch, taps=params.taps_192k(), pllinc=params.PLLINC_192K, max_len=total)
stream = torch.cuda.current_stream().cuda_stream
ms = b.autotune(x, stream)
for _ in range(3):
    b.run(x, stream=stream, sync=False)
    b.discard_frames(stream)
torch.cuda.synchronize()
steps = 20
t0 = time.perf_counter()
for _ in range(steps):
    b.run(x, stream=stream, sync=False)
    b.discard_frames(stream)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print(f"pipelined: {dt * 1e3:.3f} ms per call of {n_ch} x {total} samples = "
      f"{n_ch * total / dt / 1e9:.1f} Gsamples/s ({n_ch * total / dt / 192000:.3g} x real-time channels), "
      f"autotune {ms}")
