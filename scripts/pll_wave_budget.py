"""Where the PLL launch's waves spend their clock ticks (measurement build: make -C gnuais_amd/csrc EXTRA=-DPLL3_BUDGET
after removing build/pll_nrzi3.o).  One C3 call with the three-wave form, stages one at a time (nothing else on the
chip), then the same inside the pipelined loop; per-workgroup counters read back through gnuais_debug_pll_budget().
usage: pll_wave_budget.py > profiles/r05_pll_wave_budget.txt"""
import ctypes as C
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from gnuais_amd import ReceiverBatch, synth, tile_channels, lib

n_ch, total = 16384, 48000
base, _ = synth.make_base_streams(256, total)
x = tile_channels(torch.from_numpy(base).cuda(), n_ch)
stream = torch.cuda.current_stream().cuda_stream
L = lib.load()
fn = L.gnuais_debug_pll_budget
fn.restype, fn.argtypes = C.c_int, [C.c_void_p, C.c_int]
n_wg = n_ch // 64


def budget():
    out = np.zeros((n_wg, 16), dtype=np.uint64)
    assert fn(out.ctypes.data, n_wg) == 0
    return out.astype(np.float64)


def show(tag, ms, bud):
    m = bud.mean(0)
    tick_ns = ms * 1e6 / max(bud[:, 0].max(), 1.0)            # the slowest workgroup's recurrence spans the launch
    print(f"--- {tag}: PLL launch {ms:.3f} ms; mean over {n_wg} workgroups (min .. max); one tick ~ {tick_ns:.3f} ns "
          f"if the slowest recurrence wave spans the launch")
    names = ["recurrence: total ticks", "  waiting for the scanner", "  waiting for the writer", "  in the rows",
             "rows of four walked", "blocks", "transitions of all 64 lanes", "largest lane total",
             "scanner: total ticks", "  waiting for a free slot", "  expanding", "writer: total ticks"]
    for i, n in enumerate(names):
        print(f"  {n:32s} {m[i]:12.0f}  ({bud[:, i].min():.0f} .. {bud[:, i].max():.0f})")
    rows, tot, mx = m[4], m[6], m[7]
    print(f"  ticks per row of four            {m[3] / rows:12.1f}   = {m[3] / rows / 4:.1f} per transition step")
    print(f"  rows x 4 (what the wave pays, re-synchronised every 256 samples)  {rows * 4:.0f}")
    print(f"  largest lane total (what it would pay running ahead freely)       {mx:.0f}   tax {rows * 4 / mx:.3f}")
    print(f"  mean lane total                                                    {tot / 64:.0f}")
    print(f"  shares of the recurrence wave: rows {m[3] / m[0]:.3f}, waiting for the scanner {m[1] / m[0]:.3f}, "
          f"for the writer {m[2] / m[0]:.3f}, other {1 - (m[1] + m[2] + m[3]) / m[0]:.3f}")
    print(f"  shares of the scanner wave: expanding {m[10] / m[8]:.3f}, waiting for a slot {m[9] / m[8]:.3f}")


b = ReceiverBatch(n_ch, max_len=total)
b.set_option("pll_variant", 3)
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
show("one call at a time, stages one after the other", float(np.mean(acc)), budget())
b.set_option("pipeline", 1)
b.set_option("timing_stride", 2)
for _ in range(40):
    b.run(x, stream=stream, sync=False)
    b.discard_frames(stream)
torch.cuda.synchronize()
live = b.mean_timing()
show("inside the pipelined loop (last launch's counters)", float(live["pll"]), budget())
