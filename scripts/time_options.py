"""C3 (or NCH / LEN) per option set: the driver's region (20 steps after a sync) and the steady state (200 steps), with the
stage durations inside the loop.  usage: time_options.py "k=v,k=v[,noauto]" ...   ("" = the defaults)"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gnuais_amd import ReceiverBatch, synth, tile_channels, params
n_ch, total = int(os.environ.get("NCH", 16384)), int(os.environ.get("LEN", 48000))
c5 = os.environ.get("C5", "0") == "1"
if c5:
    total = int(os.environ.get("LEN", 192000))
    base, _ = synth.make_base_streams(256, total, sps=20)
else:
    base, _ = synth.make_base_streams(256, total)
x = tile_channels(torch.from_numpy(base).cuda(), n_ch)
stream = torch.cuda.current_stream().cuda_stream


def measure(extra, auto):
    kw = dict(taps=params.taps_192k(), pllinc=params.PLLINC_192K) if c5 else {}
    b = ReceiverBatch(n_ch, max_len=total, **kw)
    for k, v in extra.items():
        b.set_option(k, v)
    if auto:
        b.autotune(x, stream)

    def step():
        b.run(x, stream=stream, sync=False)
        b.discard_frames(stream)
    for _ in range(8):
        step()
    b.sync()
    torch.cuda.synchronize()
    shorts = []
    for _ in range(5):
        t = time.perf_counter()
        for _ in range(20):
            step()
        b.sync()
        torch.cuda.synchronize()
        shorts.append((time.perf_counter() - t) / 20 * 1e3)
    b.set_timing(True)
    b.set_option("timing_stride", 4)
    t = time.perf_counter()
    for _ in range(200):
        step()
    b.sync()
    torch.cuda.synchronize()
    steady = (time.perf_counter() - t) / 200 * 1e3
    live = b.mean_timing()
    rx = int(b.counters()["receivedframes"].sum())
    del b
    return shorts, steady, {k: round(float(live[k]), 3) for k in ("fir_slice", "pll", "hdlc_deframe", "hdlc_crc")}, rx


for a in sys.argv[1:] or [""]:
    parts = [p for p in a.split(",") if p]
    auto = "noauto" not in parts
    extra = {k: int(v, 0) for k, v in (p.split("=") for p in parts if "=" in p)}
    shorts, steady, k, rx = measure(extra, auto)
    print(f"{extra}{'' if auto else ' uncalibrated'}: 20 steps {sorted(shorts)[len(shorts) // 2]:.3f} (min {min(shorts):.3f} max {max(shorts):.3f})  "
          f"steady {steady:.3f} ms/step  {k}  rx {rx}", flush=True)
