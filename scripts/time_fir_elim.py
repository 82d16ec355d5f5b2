"""What bounds K1s?  The FIR alone, back to back, with parts of the kernel switched off (library built with
EXTRA=-DWIDE_DEBUG_FORMS; results of those runs are wrong by construction) and with the occupancy capped through
an LDS claim."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gnuais_amd import ReceiverBatch, synth, tile_channels

n_ch, total = 16384, 48000
base, _ = synth.make_base_streams(256, total)
x = tile_channels(torch.from_numpy(base).cuda(), n_ch)

def measure(opts, steps=60):
    b = ReceiverBatch(n_ch, max_len=total)
    for k, v in opts.items():
        b.set_option(k, v)
    b.set_option("stage_mask", 1)
    for _ in range(8):
        b.run(x, sync=False)
    b.sync(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(steps):
        b.run(x, sync=False)
    b.sync(); torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / steps
    del b
    return dt * 1e3

names = {0: "full kernel", 1: "no exact path", 2: "no sign stores", 4: "no central sum", 8: "no peak", 16: "no epilogue",
         5: "no exact path, no central sum", 27: "only loads + central sum", 31: "only loads"}
print("cpl 1 (fir_slice.hip):", f"{measure(dict(fir_cpl=1)):.3f} ms")
for dbg in (0, 1, 2, 4, 8, 16, 5, 27, 31):
    ms = measure(dict(fir_cpl=2, fir_form=1, fir_dbg=dbg))
    print(f"cpl 2 pk G16  dbg {dbg:2d} ({names[dbg]:32s}): {ms:.3f} ms  {n_ch*total*2/ms/1e9:.2f} TB/s", flush=True)
for waves in (2, 3, 4):
    lds = (160 * 1024 // (4 * waves)) // 256 * 256 - 256
    for cpl, form in ((1, 0), (2, 1), (2, 3)):
        ms = measure(dict(fir_cpl=cpl, fir_form=form, fir_lds=lds))
        print(f"cpl {cpl} form {form}  <= {waves} waves/SIMD: {ms:.3f} ms", flush=True)
# single launches after idle (the chip at its full clock)
b = ReceiverBatch(n_ch, max_len=total); b.set_option("stage_mask", 1); b.set_option("fir_cpl", 2); b.set_option("fir_form", 1)
b.run(x, sync=True)
for _ in range(5):
    time.sleep(0.3)
    torch.cuda.synchronize(); t = time.perf_counter(); b.run(x, sync=False); b.sync(); torch.cuda.synchronize()
    print(f"single launch after idle: {(time.perf_counter()-t)*1e3:.3f} ms")
