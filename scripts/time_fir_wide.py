"""K1s with 1 / 2 / 4 channels per lane (fir_sign_wide.hip): the FIR alone back to back, and the whole C3 chain,
for chosen forms (packed fp32, prefetch mode, rows per group) and segment lengths.
usage: time_fir_wide.py [all|fir] [cpl:form:T ...]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gnuais_amd import ReceiverBatch, synth, tile_channels

n_ch, total = 16384, 48000
base, _ = synth.make_base_streams(256, total)
x = tile_channels(torch.from_numpy(base).cuda(), n_ch)

def measure(cpl, form, T, mask, steps=60, extra=None):
    b = ReceiverBatch(n_ch, max_len=total)
    b.set_option("fir_cpl", cpl)
    b.set_option("fir_form", form)
    b.set_option("fir_T", T)
    for k, v in (extra or {}).items():
        b.set_option(k, v)
    if mask != 0x1f:
        b.set_option("stage_mask", mask)
    for _ in range(8):
        b.run(x, sync=False)
        if mask == 0x1f:
            b.discard_frames()
    b.sync(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(steps):
        b.run(x, sync=False)
        if mask == 0x1f:
            b.discard_frames()
    b.sync(); torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / steps
    del b
    return dt * 1e3

which = sys.argv[1] if len(sys.argv) > 1 else "all"
cases = [tuple(int(v, 0) for v in a.split(":")) for a in sys.argv[2:]]
if not cases:
    cases = [(1, 0, 512)]
    for T in (256, 512):
        for form in (0x01, 0x04, 0x05, 0x84, 0x85):
            cases.append((2, form, T))
    cases += [(4, 0x85, 256), (4, 0x85, 512)]
for cpl, form, T in cases:
    fir = measure(cpl, form, T, 1)
    line = f"cpl {cpl} form {form:#04x} T {T}: FIR alone {fir:.3f} ms ({n_ch*total*2/fir/1e9:.2f} TB/s)"
    if which == "all":
        chain = measure(cpl, form, T, 0x1f, steps=100)
        line += f"   chain {chain:.3f} ms/step"
    print(line, flush=True)
