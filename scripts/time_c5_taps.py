"""C5 (16384 channels x 192 000 samples, the 144-tap table): the packed slicer with 40 against 48 central taps --
FIR alone and the whole chain in steady state.  usage: time_c5_taps.py [pk_taps ...]   (0 = 40 where allowed, 48)"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gnuais_amd import ReceiverBatch, params, synth, tile_channels
n_ch, total = int(os.environ.get("NCH", 16384)), 192000
base, _ = synth.make_base_streams(256, total, sps=20, seed=72)
x = tile_channels(torch.from_numpy(base).cuda(), n_ch)
stream = torch.cuda.current_stream().cuda_stream
for a in sys.argv[1:] or ["0", "48"]:
    parts = a.split(":")
    b = ReceiverBatch(n_ch, taps=params.taps_192k(), pllinc=params.PLLINC_192K, max_len=total)
    b.set_option("fir_pk_taps", int(parts[0]))
    for kv in parts[1:]:
        k, v = kv.split("=")
        b.set_option(k, int(v))
    b.autotune(x, stream)
    res = []
    for mask in (0x01, 0x1f):
        b.set_option("stage_mask", mask)
        for _ in range(4):
            b.run(x, stream=stream, sync=False)
            b.discard_frames(stream)
        b.sync()
        torch.cuda.synchronize()
        t = time.perf_counter()
        n = 16
        for _ in range(n):
            b.run(x, stream=stream, sync=False)
            b.discard_frames(stream)
        b.sync()
        torch.cuda.synchronize()
        res.append((time.perf_counter() - t) / n * 1e3)
    print(f"fir_pk_taps {a}: central taps {int(b.info('sign_central_taps'))}, eps seen/ahead {b.info('sign_eps_seen'):.4f}/{b.info('sign_eps_ahead'):.5f}: "
          f"FIR alone {res[0]:.3f} ms ({n_ch * total * 2 / res[0] / 1e9:.2f} TB/s), chain {res[1]:.3f} ms per call", flush=True)
    del b
