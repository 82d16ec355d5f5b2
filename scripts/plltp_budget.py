"""pll_tp.hip at C2 (256 channels x 48 000 samples): where a launch's time goes.  Measurement build: EXTRA=-DPLLTP_BUDGET
after removing build/pll_tp.o.  Stamps are the constant 100 MHz clock, shared by all workgroups."""
import ctypes as C
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from gnuais_amd import ReceiverBatch, synth, lib, tile_channels

n_ch, total = int(os.environ.get("NCH", 256)), 48000
base, _ = synth.make_base_streams(256, total)
x = tile_channels(torch.from_numpy(base).cuda(), n_ch)
stream = torch.cuda.current_stream().cuda_stream
L = lib.load()
fn = L.gnuais_debug_plltp_budget
fn.restype, fn.argtypes = C.c_int, [C.c_void_p, C.c_int]


def show(tag, ms):
    bud = np.zeros((n_ch, 16), dtype=np.uint64)
    assert fn(bud.ctypes.data, n_ch) == 0
    t = bud[:, :7].astype(np.int64)
    t0 = t[:, 0].min()
    us = lambda v: v / 100.0
    print(f"--- {tag}: PLL launch {ms * 1e3:.1f} us by its events")
    print(f"  first workgroup in to last workgroup out: {us(t[:, 6].max() - t0):.1f} us; entries spread over {us(t[:, 0].max() - t0):.1f} us, "
          f"exits over {us(t[:, 6].max() - t[:, 6].min()):.1f} us")
    d = np.diff(t, axis=1)
    names = ["zero packs + counts + scan", "chunks (pass 1 + walk)", "pass 3 (bits)", "packs formed", "parity (one lane)", "packs written"]
    print(f"  a workgroup: {us((t[:, 6] - t[:, 0]).mean()):.1f} us (max {us((t[:, 6] - t[:, 0]).max()):.1f})")
    for i, n in enumerate(names):
        print(f"    {n:30s} {us(d[:, i].mean()):7.1f} us  (max {us(d[:, i].max()):.1f})")
    wg = (t[:, 6] - t[:, 0]).astype(np.float64)
    fall, ntr = bud[:, 10].astype(np.float64), bud[:, 11].astype(np.float64)
    order = np.argsort(wg)
    print(f"    chunks (one more than planned each time the walk left the window): mean {fall.mean():.1f}, max {fall.max():.0f}; transitions per call mean {ntr.mean():.0f}, max {ntr.max():.0f}")
    for i in list(order[:3]) + list(order[-6:]):
        print(f"      channel {i:3d}: {us(wg[i]):6.1f} us, {fall[i]:3.0f} chunks, {ntr[i]:6.0f} transitions, pass 1 {us(float(bud[i, 8])):.1f}, walk {us(float(bud[i, 9])):.1f}")
    print(f"    of the chunks: pass 1 {us(bud[:, 8].astype(np.float64).mean()):.1f} us, walk {us(bud[:, 9].astype(np.float64).mean()):.1f} us")


print("GNUAIS_TP_CHUNK", os.environ.get("GNUAIS_TP_CHUNK", "default"))
for mask in (0x03,):
    b = ReceiverBatch(n_ch, max_len=total)
    b.set_option("stage_mask", 0x03)
    b.autotune(x, stream)
    for _ in range(4):
        b.run(x, stream=stream, sync=True)
    b.set_option("stage_mask", mask)
    b.set_timing(True)
    b.set_option("pipeline", 0)
    acc = []
    for _ in range(5):
        b.run(x, stream=stream, sync=True)
        acc.append(b.last_timing()["pll"])
    torch.cuda.synchronize()
    show(f"stage_mask {mask:#x}, one call at a time", float(np.mean(acc)))
    b.set_option("pipeline", 1)
    b.set_option("timing_stride", 2)
    for _ in range(20):
        b.run(x, stream=stream, sync=False)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(200):
        b.run(x, stream=stream, sync=False)
    snap_ms = float(b.mean_timing()["pll"])
    torch.cuda.synchronize()
    per = (time.perf_counter() - t) / 200
    show(f"stage_mask {mask:#x}, pipelined loop: {per * 1e6:.1f} us per call", float(b.mean_timing()["pll"]))
    del b
