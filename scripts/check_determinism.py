"""Is a pipelined run of the chain deterministic?  The same calls three times over (fresh batch each time), frames drained
every few calls and hashed; C5=1: the 192 kHz parameter set at 16384 channels.  Exits non-zero when two runs differ."""
import hashlib
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gnuais_amd import ReceiverBatch, params, synth, tile_channels
c5 = os.environ.get("C5", "0") == "1"
n_ch = int(os.environ.get("NCH", 16384))
total = int(os.environ.get("LEN", 192000 if c5 else 48000))
base, _ = synth.make_base_streams(256, total, sps=20 if c5 else 5)
x = tile_channels(torch.from_numpy(base).cuda(), n_ch)
stream = torch.cuda.current_stream().cuda_stream
kw = dict(taps=params.taps_192k(), pllinc=params.PLLINC_192K) if c5 else {}
seen = []
for rep in range(int(os.environ.get("REPS", 3))):
    b = ReceiverBatch(n_ch, max_len=total, frame_capacity=n_ch * 48 * 6, **kw)
    h = hashlib.sha256()
    for i in range(int(os.environ.get("CALLS", 40))):
        b.run(x, stream=stream, sync=os.environ.get('CALL_SYNC') == '1')
        if i % 5 == 4:
            f = b.drain_frames()
            h.update(f.tobytes())
    c = b.counters()
    h.update(c.tobytes())
    p = b.pll_state()
    h.update(p.tobytes())
    seen.append(h.hexdigest()[:16])
    print(rep, seen[-1], int(c["receivedframes"].sum()), flush=True)
    del b
sys.exit(0 if len(set(seen)) == 1 else 1)
