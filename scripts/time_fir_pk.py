"""K1s on register pairs (fir_sign_pk.hip, option fir_pk) against the scalar kernels: FIR alone back to back and the
pipelined chain, C3 (12 central taps) and C5 (48)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gnuais_amd import ReceiverBatch, params, synth, tile_channels
stream = torch.cuda.current_stream().cuda_stream
def run(name, n_ch, total, sps, kw, Ts):
    base, _ = synth.make_base_streams(64, total, sps=sps)
    x = tile_channels(torch.from_numpy(base).cuda(), n_ch)
    for pk, T in [(0, 512)] + [(1, t) for t in Ts] + [(0, 512)]:
        def mk():
            b = ReceiverBatch(n_ch, max_len=total, **kw)
            b.set_option("fir_pk", pk); b.set_option("fir_T", T)
            return b
        b = mk(); b.set_option("stage_mask", 1)
        for _ in range(4): b.run(x, stream=stream, sync=False)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = 40 if total < 100000 else 10
        for _ in range(n): b.run(x, stream=stream, sync=False)
        torch.cuda.synchronize(); fir = (time.perf_counter() - t0) / n * 1e3
        del b
        b = mk(); b.autotune(x, stream)
        for _ in range(5):
            b.run(x, stream=stream, sync=False); b.discard_frames(stream)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = 100 if total < 100000 else 20
        for _ in range(n):
            b.run(x, stream=stream, sync=False); b.discard_frames(stream)
        torch.cuda.synchronize(); chain = (time.perf_counter() - t0) / n * 1e3
        print(f"{name} fir_pk {pk} T {T}: FIR alone {fir:.3f} ms ({n_ch*total*2/fir/1e9:.2f} TB/s)  chain {chain:.3f} ms/call  received {b.total_received()}", flush=True)
        del b
which = sys.argv[1] if len(sys.argv) > 1 else "both"
if which in ("both", "C3"): run("C3", 16384, 48000, 5, {}, (384, 768))
if which in ("both", "C5"): run("C5", 16384, 192000, 20, dict(taps=params.taps_192k(), pllinc=params.PLLINC_192K), (768,))
