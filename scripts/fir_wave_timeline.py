"""Occupancy of K1s over the life of one launch: every wave's start / end (100 MHz wall clock, option fir_stamps).
usage: fir_wave_timeline.py cpl form [T]"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gnuais_amd import ReceiverBatch, synth, tile_channels
from gnuais_amd.lib import load
cpl, form = int(sys.argv[1]), int(sys.argv[2], 0)
T = int(sys.argv[3]) if len(sys.argv) > 3 else 512
extra = {a.split("=")[0]: int(a.split("=")[1]) for a in sys.argv[4:]}
n_ch, total = 16384, 48000
base, _ = synth.make_base_streams(256, total)
x = tile_channels(torch.from_numpy(base).cuda(), n_ch)
b = ReceiverBatch(n_ch, max_len=total)
for k, v in dict(fir_cpl=cpl, fir_form=form, fir_T=T, stage_mask=1, fir_stamps=1, **extra).items(): b.set_option(k, v)
for _ in range(6): b.run(x, sync=False)
b.sync()
n_waves = (n_ch // (64 * cpl)) * ((total + 127) // 128 + 2)
st = np.zeros((n_waves, 2), dtype=np.uint64)
fn = load().gnuais_debug_fir_stamps; fn.restype = C.c_int; fn.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
assert fn(b.handle if hasattr(b, "handle") else b._h, st.ctypes.data, n_waves) == 0
st = st[st[:, 1] != 0]
n_waves = len(st)
s = (st[:, 0] - st[:, 0].min()).astype(np.float64) / 100.0      # us
e = (st[:, 1] - st[:, 0].min()).astype(np.float64) / 100.0
life = e - s
print(f"cpl {cpl} form {form:#x} T {T} {extra}: {n_waves} waves, launch {e.max():.1f} us; lifetime mean {life.mean():.1f} median {np.median(life):.1f} "
      f"p10 {np.percentile(life,10):.1f} p90 {np.percentile(life,90):.1f} max {life.max():.1f} us; wave-time / launch = {life.sum()/e.max():.0f} waves resident on average "
      f"({life.sum()/e.max()/1024:.2f} per SIMD)")
edges = np.linspace(0, e.max(), 21)
for a, z in zip(edges[:-1], edges[1:]):
    mid = (a + z) / 2
    res = np.sum((s <= mid) & (e > mid))
    started = np.sum((s >= a) & (s < z))
    lm = life[(s >= a) & (s < z)]
    print(f"  t = {mid:6.1f} us: resident {res:5d} ({res/1024:.2f}/SIMD)  started in bin {started:5d}  their mean lifetime {lm.mean() if len(lm) else 0:6.1f} us")
