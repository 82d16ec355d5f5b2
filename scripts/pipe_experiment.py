"""Wall-clock ms/step of the pipelined chain at C3 size under option sweeps.

env: SWEEP="name=v1,v2;name2=..." (cartesian), TIMING=0/1, STEPS
"""
import sys, os, time, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gnuais_amd import ReceiverBatch, synth, tile_channels
n_ch, total = 16384, 48000
base, _ = synth.make_base_streams(64, total)
x = tile_channels(torch.from_numpy(base).cuda(), n_ch)
steps = int(os.environ.get("STEPS", 40))
sweep = [kv.split("=") for kv in os.environ.get("SWEEP", "fir_wpb=1").split(";") if kv]
names = [k for k, _ in sweep]
for combo in itertools.product(*[v.split(",") for _, v in sweep]):
    for timing in [int(t) for t in os.environ.get("TIMING", "1").split(",")]:
        b = ReceiverBatch(n_ch, max_len=total)
        for k, v in zip(names, combo):
            b.set_option(k, int(v))
        b.set_timing(bool(timing))
        for _ in range(4):
            b.run(x, sync=False); b.discard_frames()
        b.sync(); torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(steps):
            b.run(x, sync=False); b.discard_frames()
        t_enq = time.perf_counter() - t
        b.sync(); torch.cuda.synchronize()
        dt = time.perf_counter() - t
        km = b.mean_timing() if timing else None
        print(dict(zip(names, combo)), "timing", timing, "ms/step %.3f" % (dt / steps * 1e3),
              "enqueue ms/step %.3f" % (t_enq / steps * 1e3),
              {k: round(v, 2) for k, v in km.items()} if km else "", flush=True)
        del b
