"""What a fresh process gets: the C3 region (20 steps after 200 warm-up calls) with the library's default stage -> stream
assignment, then what gnuais_batch_autotune() picks (pool indices in creation order) and the region again."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gnuais_amd import ReceiverBatch, params, synth, tile_channels
c5 = os.environ.get("C5", "0") == "1"
n_ch, total = int(os.environ.get("NCH", 16384)), int(os.environ.get("LEN", 192000 if c5 else 48000))
base, _ = synth.make_base_streams(min(256, n_ch), total, sps=20 if c5 else 5)
x = tile_channels(torch.from_numpy(base).cuda(), n_ch)
stream = torch.cuda.current_stream().cuda_stream
b = ReceiverBatch(n_ch, max_len=total, **(dict(taps=params.taps_192k(), pllinc=params.PLLINC_192K) if c5 else {}))
b.set_option("stage_mask", int(os.environ.get("MASK", "0x1f"), 0))


def region():
    for _ in range(200):
        b.run(x, stream=stream, sync=False)
        b.discard_frames(stream)
    b.sync()
    torch.cuda.synchronize()
    out = []
    for _ in range(3):
        t = time.perf_counter()
        for _ in range(20):
            b.run(x, stream=stream, sync=False)
            b.discard_frames(stream)
        b.sync()
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t) / 20 * 1e3)
    return sorted(out)[1]


def assignment():
    return [int(b.info("stream_of_stage_%d" % q)) for q in range(4)]


print("default", assignment(), "%.3f ms" % region(), flush=True)
b.autotune(x, stream)
print("tuned  ", assignment(), "%.3f ms" % region(), flush=True)
