"""C3 pipeline with stages switched off (stage_mask: 1 FIR, 2 PLL, 8 deframer, 16 K3): which stages set the period?
The later stages re-read what the last full call left in the hand-off buffers, so their work is the real one.
usage: time_masks.py [mask,nbuf ...]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gnuais_amd import ReceiverBatch, synth, tile_channels
n_ch, total = 16384, 48000
base, _ = synth.make_base_streams(256, total)
x = tile_channels(torch.from_numpy(base).cuda(), n_ch)
stream = torch.cuda.current_stream().cuda_stream

def measure(mask, nbuf):
    b = ReceiverBatch(n_ch, max_len=total)
    b.set_option("nbuf", nbuf)
    b.autotune(x, stream)
    def step():
        b.run(x, stream=stream, sync=False); b.discard_frames(stream)
    for _ in range(2 * nbuf): step()          # every hand-off set holds a real call's data
    b.sync(); torch.cuda.synchronize()
    b.set_option("stage_mask", mask)
    for _ in range(10): step()
    b.sync(); torch.cuda.synchronize()
    b.set_timing(True); b.set_option("timing_stride", 4)
    t = time.perf_counter()
    for _ in range(200): step()
    b.sync(); torch.cuda.synchronize()
    steady = (time.perf_counter() - t) / 200 * 1e3
    live = b.mean_timing()
    del b
    return steady, {k: round(float(live[k]), 3) for k in ("fir_slice", "pll", "hdlc_deframe", "hdlc_crc")}

for a in sys.argv[1:] or ["31,3"]:
    mask, nbuf = (int(v, 0) for v in a.split(","))
    steady, k = measure(mask, nbuf)
    print(f"stage_mask {mask:#04x} nbuf {nbuf}: steady {steady:.3f} ms/step  {k}", flush=True)
