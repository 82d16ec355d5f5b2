"""C3 chain with every step's batch handed to the device as S time slices (S calls of L/S rows each, same total work per
step): does a shallower pipeline in TIME (a slice's latency is 1/S of a call's) shorten the driver's 20-step region?
Per setting: 20-step region (5 warm-up steps, sync, 20 steps, sync; median of REPS) and a 200-step steady state.
usage: time_slices.py [S,nbuf ...]   e.g. 1,3 2,3 2,4 4,4 4,6"""
import sys, os, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gnuais_amd import ReceiverBatch, synth, tile_channels
n_ch, total = 16384, 48000
base, _ = synth.make_base_streams(256, total)
x = tile_channels(torch.from_numpy(base).cuda(), n_ch)
stream = torch.cuda.current_stream().cuda_stream
REPS = int(os.environ.get("REPS", "7"))

def measure(S, nbuf):
    b = ReceiverBatch(n_ch, max_len=total)
    b.set_option("nbuf", nbuf)
    b.autotune(x, stream)
    rows = (total + S - 1) // S
    parts = [x[i:min(i + rows, total)] for i in range(0, total, rows)]
    def step():
        for p in parts:
            b.run(p, stream=stream, sync=False)
        b.discard_frames(stream)
    region = []
    for _ in range(REPS):
        for _ in range(5): step()
        b.sync(); torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(20): step()
        b.sync(); torch.cuda.synchronize()
        region.append((time.perf_counter() - t) / 20 * 1e3)
    for _ in range(10): step()
    b.sync(); torch.cuda.synchronize()
    rx0 = int(b.total_received())
    t = time.perf_counter()
    for _ in range(200): step()
    b.sync(); torch.cuda.synchronize()
    steady = (time.perf_counter() - t) / 200 * 1e3
    rx = int(b.total_received()) - rx0
    del b
    return region, steady, rx

settings = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]] or [(1, 3), (2, 3)]
for st in settings:
    region, steady, rx = measure(*st)
    print(f"slices {st[0]} nbuf {st[1]}: 20-step {statistics.median(region):.3f} (min {min(region):.3f} max {max(region):.3f})"
          f"  steady {steady:.3f} ms/step  frames in 200 steps {rx}", flush=True)
