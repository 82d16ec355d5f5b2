"""The time-parallel PLL (pll_tp.hip, pll_variant 7) at C2's shape: the stage's duration and the phases of workgroup 0
(pre-pass + scan | first chunk: pass 1, walk | later chunks | pass 3 | packs out), beside the lane-per-channel forms.
usage: time_pll_tp.py [channels=256]"""
import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gnuais_amd import ReceiverBatch, synth, tile_channels
n_ch, total = int(sys.argv[1]) if len(sys.argv) > 1 else 256, 48000
base, _ = synth.make_base_streams(min(256, n_ch), total)
x = tile_channels(torch.from_numpy(base).cuda(), n_ch)
stream = torch.cuda.current_stream().cuda_stream
for pv in (7, 6, 3):
    b = ReceiverBatch(n_ch, max_len=total)
    b.set_option("pll_variant", pv)
    b.set_option("stage_mask", 0x03)
    for _ in range(5):
        b.run(x, stream=stream, sync=False)
    b.sync()
    b.set_timing(True)
    ts = []
    for _ in range(10):
        b.run(x, stream=stream, sync=True)
        ts.append(b.last_timing()["pll"])
    line = f"{n_ch} channels, pll_variant {pv}: PLL stage alone {min(ts) * 1e3:.1f} us (median {sorted(ts)[5] * 1e3:.1f})"
    t0 = time.perf_counter()
    for _ in range(100):
        b.run(x, stream=stream, sync=False)
    b.sync()
    line += f"; FIR + PLL pipelined {(time.perf_counter() - t0) / 100 * 1e3:.4f} ms per call"
    if pv == 7:
        st = (C.c_ulonglong * 8)()
        f = b._lib.gnuais_debug_pll_tp_stamps
        f.argtypes = [C.c_void_p, C.c_void_p]
        if f(b._h, st) == 0:
            s = [v / 100.0 for v in st]
            line += (f"\n   workgroup 0 (us): pre-pass+scan {s[1] - s[0]:.1f} | chunk 0: pass 1 {s[2] - s[1]:.1f}, walk {s[3] - s[2]:.1f}"
                     f" | last chunk: pass 1 end at {s[4] - s[0]:.1f}, walk end at {s[5] - s[0]:.1f} | pass 3 {s[6] - s[5]:.1f}"
                     f" | packs out {s[7] - s[6]:.1f} | total {s[7] - s[0]:.1f}")
    print(line, flush=True)
    del b
