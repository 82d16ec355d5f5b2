"""Row f1 on the device vs on the host, one C3 call's worth of frames (~3e5) and four calls' worth:
  host path   gnuais_batch_drain_frames() + gnuais_nmea_from_frames()  (D2H of the records, host threads)
  device path gnuais_batch_drain_nmea()                                (sort + scans + write kernel, D2H of the text)
and the device kernels alone (no D2H), from a second call on warm buffers."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                                              # noqa: E402
from gnuais_amd import ReceiverBatch, nmea_from_frames, synth, tile_channels   # noqa: E402

n_ch, total = 16384, 48000
base, _ = synth.make_base_streams(256, total)
x = tile_channels(torch.from_numpy(base).cuda(), n_ch)
for calls in (1, 4):
    res = {}
    for rep in range(3):
        a = ReceiverBatch(n_ch, max_len=total, frame_capacity=calls * n_ch * 24)
        b = ReceiverBatch(n_ch, max_len=total, frame_capacity=calls * n_ch * 24)
        for _ in range(calls):
            a.run(x)
            b.run(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        frames = a.drain_frames()
        t1 = time.perf_counter()
        seq = np.zeros(n_ch, dtype=np.uint8)
        want = nmea_from_frames(frames, seq)
        t2 = time.perf_counter()
        seq2 = np.zeros(n_ch, dtype=np.uint8)
        got, ns, nf = b.drain_nmea(seq2)
        t3 = time.perf_counter()
        assert got == want and np.array_equal(seq, seq2) and nf == len(frames)
        res = {"frames": nf, "drain_frames_ms": (t1 - t0) * 1e3, "host_format_ms": (t2 - t1) * 1e3,
               "drain_nmea_ms": (t3 - t2) * 1e3, "text_MB": len(got) / 1e6}
        del a, b
    host = res["drain_frames_ms"] + res["host_format_ms"]
    print(f"{calls} call(s): {res['frames']} frames, {res['text_MB']:.1f} MB of text | host path "
          f"{res['drain_frames_ms']:.1f} + {res['host_format_ms']:.1f} ms = {res['frames'] / host / 1e3:.1f} M frames/s | "
          f"device path {res['drain_nmea_ms']:.1f} ms = {res['frames'] / res['drain_nmea_ms'] / 1e3:.1f} M frames/s")
