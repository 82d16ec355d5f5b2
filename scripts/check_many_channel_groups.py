"""pll_h3 with more channel groups than CUs (two / three workgroups per CU): 33 000 and 50 000 channels against the oracle
on the channels at both ends and in the middle."""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from gnuais_amd import synth, ReceiverBatch, tile_channels
from oracle_lib import Oracle
for n_ch, total in ((33000, 9000), (50001, 5000)):
    base, _ = synth.make_base_streams(256, total, seed=81)
    xb = tile_channels(torch.from_numpy(base).cuda(), n_ch)
    b = ReceiverBatch(n_ch, max_len=total)
    b.run(xb[:total - 777]); b.run(xb[total - 777:])
    f = b.drain_frames()
    pick = np.r_[0:64, n_ch // 2:n_ch // 2 + 64, n_ch - 64:n_ch]
    xs = xb[:, torch.from_numpy(pick).cuda()].cpu().numpy()
    o = Oracle(len(pick)); o.run(xs[:total - 777]); o.run(xs[total - 777:])
    sel = f[np.isin(f["channel"], pick)]
    remap = {int(c): i for i, c in enumerate(pick)}
    sel["channel"] = [remap[int(c)] for c in sel["channel"]]
    p = b.pll_state()
    ok_pll = [(int(p["pll"][c]), int(p["prev"][c]), int(p["lastbit"][c])) for c in pick] == [o.pll(i) for i in range(len(pick))]
    print(n_ch, total, "frames", len(f), "subset equal:", sel.tobytes() == o.frames().tobytes(), "pll equal:", ok_pll, flush=True)
