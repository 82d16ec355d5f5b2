"""C5's FIR stage alone, back to back (for rocprofv3 passes): fir_only_loop_c5.py fir_pk [steps]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gnuais_amd import ReceiverBatch, params, synth, tile_channels
pk = int(sys.argv[1]); steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
n_ch, total = 16384, 192000
base, _ = synth.make_base_streams(64, total, sps=20)
x = tile_channels(torch.from_numpy(base).cuda(), n_ch)
b = ReceiverBatch(n_ch, taps=params.taps_192k(), pllinc=params.PLLINC_192K, max_len=total)
b.set_option("fir_pk", pk); b.set_option("stage_mask", 1)
for _ in range(2): b.run(x, sync=False)
b.sync(); torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(steps): b.run(x, sync=False)
b.sync(); torch.cuda.synchronize()
print(f"C5 fir_pk {pk}: {(time.perf_counter()-t)/steps*1e3:.3f} ms/launch")
