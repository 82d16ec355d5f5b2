"""pll_h3.hip's recurrence wave and helper 0: where their clock ticks go (measurement build: EXTRA=-DPLLH3_BUDGET after
removing build/pll_h3.o)."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from gnuais_amd import ReceiverBatch, synth, tile_channels, lib

n_ch, total = 16384, 48000
base, _ = synth.make_base_streams(256, total)
x = tile_channels(torch.from_numpy(base).cuda(), n_ch)
stream = torch.cuda.current_stream().cuda_stream
L = lib.load()
fn = L.gnuais_debug_pllh3_budget
fn.restype, fn.argtypes = C.c_int, [C.c_void_p, C.c_int]
n_wg = n_ch // 64


def show(tag, ms):
    bud = np.zeros((n_wg, 16), dtype=np.uint64)
    assert fn(bud.ctypes.data, n_wg) == 0
    m = bud.astype(np.float64).mean(0)
    print(f"--- {tag}: PLL launch {ms:.3f} ms")
    print(f"  recurrence: total {m[0]:.0f} ticks (max {bud[:, 0].max()}), waiting for a scan {m[1]:.0f} ({m[1] / m[0]:.3f}), rows "
          f"{m[3]:.0f} ({m[3] / m[0]:.3f}); {m[4]:.0f} rows of four in {m[5]:.0f} blocks: {m[3] / m[4] / 4:.1f} ticks per step")
    print(f"  core clock while the recurrence wave ran: {m[0] / max(m[6], 1) * 0.1:.3f} GHz (clock ticks / 100 MHz ticks)")
    print(f"  helper 0: total {m[8]:.0f}, scanning {m[9]:.0f} ({m[9] / m[8]:.3f}), waiting for a free slot {m[10]:.0f} "
          f"({m[10] / m[8]:.3f}), packs {m[12]:.0f} ({m[12] / m[8]:.3f}); per own block: scan {m[9] / (m[5] / 3):.0f}")


b = ReceiverBatch(n_ch, max_len=total)
b.set_option("pll_variant", 8)
b.autotune(x, stream)
for _ in range(4):
    b.run(x, stream=stream, sync=True)
    b.discard_frames(stream)
b.set_timing(True)
b.set_option("pipeline", 0)
acc = []
for _ in range(3):
    b.run(x, stream=stream, sync=True)
    acc.append(b.last_timing()["pll"])
    b.discard_frames(stream)
torch.cuda.synchronize()
show("one call at a time", float(np.mean(acc)))
b.set_option("pipeline", 1)
b.set_option("timing_stride", 2)
for _ in range(80):
    b.run(x, stream=stream, sync=False)
    b.discard_frames(stream)
# NOT synchronised: the host is at most `nbuf` calls ahead, so the counters read now (a copy on the null stream, which
# the library's non-blocking streams do not wait for) are those of a launch in the middle of the running loop
snap = np.zeros((n_wg, 16), dtype=np.uint64)
assert fn(snap.ctypes.data, n_wg) == 0
torch.cuda.synchronize()
live = float(b.mean_timing()["pll"])


def show_snap(tag, ms, bud):
    m = bud.astype(np.float64).mean(0)
    print(f"--- {tag}: PLL launch {ms:.3f} ms (mean of the loop)")
    print(f"  recurrence: total {m[0]:.0f} ticks, waiting for a scan {m[1]:.0f} ({m[1] / m[0]:.3f}), rows {m[3]:.0f} ({m[3] / m[0]:.3f}): "
          f"{m[3] / m[4] / 4:.1f} ticks per step")
    print(f"  core clock while the recurrence wave ran: {m[0] / max(m[6], 1) * 0.1:.3f} GHz; its span {m[6] / 100:.1f} us")
    print(f"  helper 0: total {m[8]:.0f}, scanning {m[9]:.0f} ({m[9] / m[8]:.3f}), waiting for a free slot {m[10]:.0f} ({m[10] / m[8]:.3f}), "
          f"packs {m[12]:.0f} ({m[12] / m[8]:.3f}); per own block: scan {m[9] / (m[5] / 3):.0f}")


show_snap("inside the pipelined loop (a launch in the middle of it, read while the loop runs)", live, snap)
