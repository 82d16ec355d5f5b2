"""C5's FIR (48 central taps of the 144-tap table): the per-segment pre-pass (fir_inloop 0) against the running
window maximum in the loop (1): FIR alone back to back, and the whole chain pipelined."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gnuais_amd import ReceiverBatch, params, synth, tile_channels
n_ch, total = 16384, 192000
base, _ = synth.make_base_streams(64, total, sps=20)
x = tile_channels(torch.from_numpy(base).cuda(), n_ch)
stream = torch.cuda.current_stream().cuda_stream
for inloop in (0, 1, 0, 1):
    b = ReceiverBatch(n_ch, taps=params.taps_192k(), pllinc=params.PLLINC_192K, max_len=total)
    b.set_option("fir_inloop", inloop)
    b.set_option("stage_mask", 1)
    for _ in range(3): b.run(x, stream=stream, sync=False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): b.run(x, stream=stream, sync=False)
    torch.cuda.synchronize(); fir = (time.perf_counter() - t0) / 10 * 1e3
    del b
    b = ReceiverBatch(n_ch, taps=params.taps_192k(), pllinc=params.PLLINC_192K, max_len=total)
    b.set_option("fir_inloop", inloop)
    b.autotune(x, stream)
    for _ in range(3):
        b.run(x, stream=stream, sync=False); b.discard_frames(stream)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        b.run(x, stream=stream, sync=False); b.discard_frames(stream)
    torch.cuda.synchronize(); chain = (time.perf_counter() - t0) / 20 * 1e3
    print(f"fir_inloop {inloop}: FIR alone {fir:.3f} ms ({n_ch*total*2/fir/1e9:.2f} TB/s)  chain {chain:.3f} ms/call  received {b.total_received()}  eps seen/ahead {b.info('sign_eps_seen'):.4f}/{b.info('sign_eps_ahead'):.5f} (whole {b.info('sign_eps'):.4f})", flush=True)
    del b
