"""C3 chain period with the PLL workgroup's three forms (three, four, six waves).
usage: [GNUAIS_K2B_LAG=2] time_pll4.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gnuais_amd import ReceiverBatch, synth, tile_channels
n_ch, total = int(os.environ.get("NCH", 16384)), 48000
base, _ = synth.make_base_streams(256, total)
x = tile_channels(torch.from_numpy(base).cuda(), n_ch)
stream = torch.cuda.current_stream().cuda_stream
def measure(opts, steps=100, tune=True):
    b = ReceiverBatch(n_ch, max_len=total)
    for k, v in opts.items(): b.set_option(k, v)
    if tune: b.autotune(x, stream)
    for _ in range(10):
        b.run(x, stream=stream, sync=False); b.discard_frames(stream)
    torch.cuda.synchronize()
    b.set_timing(True); b.set_option("timing_stride", 4)
    t = time.perf_counter()
    for _ in range(steps):
        b.run(x, stream=stream, sync=False); b.discard_frames(stream)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / steps * 1e3
    live = b.mean_timing()
    del b
    return dt, {k: round(float(live[k]), 3) for k in ("fir_slice", "pll", "hdlc_deframe", "hdlc_crc")}
lag = os.environ.get("GNUAIS_K2B_LAG", "1")
for rep in range(int(os.environ.get("REPS", 2))):
    for pv in [int(v) for v in os.environ.get("PVS", "3,4,6").split(",")]:
        for lpw in [int(v) for v in os.environ.get("LPWS", "16,32").split(",")]:
            dt, k = measure(dict(pll_variant=pv, hdlc_lpw=lpw))
            print(f"lag {lag} pll {pv} lpw {lpw}: {dt:.3f} ms/step  {k}", flush=True)
