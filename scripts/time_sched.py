"""C3 chain under the host-side scheduling knobs of round 4: hand-off depth (nbuf), the cold-start hold, FIR launches on
two streams, candidate-ring lag.  Per setting: the driver's 20-step region (5 warm-up steps, sync, 20 steps, sync; median
of REPS) and a 200-step steady state, with the per-stage in-pipeline durations.
usage: time_sched.py [setting ...]   a setting is nbuf,hold,firstreams,lag[,pll[,lpw]]  e.g. 4,-1,1,1  6,0,2,2"""
import sys, os, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gnuais_amd import ReceiverBatch, synth, tile_channels
n_ch, total = 16384, 48000
base, _ = synth.make_base_streams(256, total)
x = tile_channels(torch.from_numpy(base).cuda(), n_ch)
stream = torch.cuda.current_stream().cuda_stream
REPS = int(os.environ.get("REPS", "7"))

def measure(nbuf, hold, firs, lag, pll=0, lpw=0):
    os.environ["GNUAIS_K2B_LAG"] = str(lag)
    b = ReceiverBatch(n_ch, max_len=total)
    b.set_option("nbuf", nbuf); b.set_option("cold_hold_us", hold); b.set_option("fir_streams", firs)
    if pll: b.set_option("pll_variant", pll)
    if lpw: b.set_option("hdlc_lpw", lpw)
    if os.environ.get("FIR_LDS"): b.set_option("fir_lds", int(os.environ["FIR_LDS"]))    # LDS claimed per FIR wave (caps its occupancy)
    b.autotune(x, stream)
    def step():
        b.run(x, stream=stream, sync=False); b.discard_frames(stream)
    region = []
    for _ in range(REPS):
        for _ in range(5): step()
        b.sync(); torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(20): step()
        b.sync(); torch.cuda.synchronize()
        region.append((time.perf_counter() - t) / 20 * 1e3)
    b.set_timing(True); b.set_option("timing_stride", 4)
    for _ in range(10): step()
    b.sync(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(200): step()
    b.sync(); torch.cuda.synchronize()
    steady = (time.perf_counter() - t) / 200 * 1e3
    live = b.mean_timing()
    rx = int(b.total_received()) if hasattr(b, "total_received") else -1
    del b
    return region, steady, {k: round(float(live[k]), 3) for k in ("fir_slice", "pll", "hdlc_deframe", "hdlc_crc")}, rx

settings = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]] or [(4, -1, 1, 1)]
for st in settings:
    region, steady, k, rx = measure(*st)
    print(f"nbuf {st[0]} hold {st[1]} firstreams {st[2]} lag {st[3]}" + (f" pll {st[4]}" if len(st) > 4 else "") + (f" lpw {st[5]}" if len(st) > 5 else "") +
          f": 20-step {statistics.median(region):.3f} (min {min(region):.3f} max {max(region):.3f})  steady {steady:.3f} ms/step  {k}  rx {rx}", flush=True)
