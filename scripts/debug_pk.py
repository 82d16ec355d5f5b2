import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gnuais_amd import ReceiverBatch, params, synth
rng = np.random.default_rng(5)
for name, kw, n_ch, total, T in (("48k", {}, 70, 5000, 512), ("192k", dict(taps=params.taps_192k(), pllinc=params.PLLINC_192K), 70, 9000, 512)):
    x = (rng.normal(0, 3000, (total, n_ch))).astype(np.int16)
    xd = torch.from_numpy(x).cuda()
    for rep in range(3):
        a = ReceiverBatch(n_ch, max_len=total, **kw); b = ReceiverBatch(n_ch, max_len=total, **kw)
        b.set_option("fir_pk", 1)
        a.run(xd); b.run(xd)
        sa, sb = a.last_signs(total), b.last_signs(total)
        d = np.argwhere(sa != sb)
        print(name, "rep", rep, "differing (channel, sample):", len(d), d[:12].tolist(), "samples mod 768:", sorted(set((d[:, 1] % 768).tolist()))[:20], "channels:", sorted(set(d[:, 0].tolist()))[:20], flush=True)
        # second call on the carried state
        a.run(xd[:1000]); b.run(xd[:1000])
        sa, sb = a.last_signs(1000), b.last_signs(1000)
        d = np.argwhere(sa != sb)
        print("   second call:", len(d), d[:8].tolist(), flush=True)
