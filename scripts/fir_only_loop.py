"""The FIR stage alone, back to back (for rocprofv3 passes): fir_only_loop.py cpl form [steps] [T] [fir_pk]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gnuais_amd import ReceiverBatch, synth, tile_channels
cpl, form = int(sys.argv[1]), int(sys.argv[2], 0)
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
n_ch, total = 16384, 48000
base, _ = synth.make_base_streams(256, total)
x = tile_channels(torch.from_numpy(base).cuda(), n_ch)
b = ReceiverBatch(n_ch, max_len=total)
b.set_option("fir_cpl", cpl); b.set_option("fir_form", form); b.set_option("stage_mask", 1)
if len(sys.argv) > 4: b.set_option("fir_T", int(sys.argv[4]))
if len(sys.argv) > 5: b.set_option("fir_pk", int(sys.argv[5]))
for _ in range(3): b.run(x, sync=False)
b.sync(); torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(steps): b.run(x, sync=False)
b.sync(); torch.cuda.synchronize()
print(f"cpl {cpl} form {form:#x}: {(time.perf_counter()-t)/steps*1e3:.3f} ms/launch")
