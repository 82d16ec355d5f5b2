"""C5 (BASELINE configs: 16384 channels, 192 kHz table = 144 taps / pllinc 3276): FIR time of the
sign-exact slicer (48 central taps) vs the generic direct-form kernel; isolated, per call."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gnuais_amd import ReceiverBatch, synth, tile_channels, params
n_ch, total = 16384, int(os.environ.get("LEN", 48000))
base, _ = synth.make_base_streams(64, total, sps=20) if "sps" in synth.make_base_streams.__code__.co_varnames else synth.make_base_streams(64, total)
x = tile_channels(torch.from_numpy(base).cuda(), n_ch)
for variant in (3, 0):
    b = ReceiverBatch(n_ch, taps=params.taps_192k(), pllinc=params.PLLINC_192K, max_len=total)
    b.set_option("fir_variant", variant)
    b.set_option("pipeline", 0)
    b.set_timing(True)
    ts = []
    for _ in range(3):
        b.run(x, sync=True); b.discard_frames()
        ts.append(b.last_timing())
    t = min(ts, key=lambda d: d["fir_slice"])
    print("fir_variant", variant, {k: round(v, 3) for k, v in t.items()}, "received", b.total_received(), flush=True)
    del b
