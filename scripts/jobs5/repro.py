import sys, numpy as np, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from gnuais_amd import synth, ReceiverBatch, tile_channels
from oracle_lib import Oracle
def dev(x): return torch.from_numpy(np.ascontiguousarray(x)).cuda()
for n_ch,total,call in ((16384,15360,6000),(16384,15360,15360),(16384,48000,6000),(2048,15360,6000),(8192,15360,6000)):
    base,_=synth.make_base_streams(256,total,seed=76)
    xb=tile_channels(dev(base),n_ch)
    for opts in ({}, {"pll_variant":8,"hdlc_lpw":16}, {"hdlc_lpw":64}):
        b=ReceiverBatch(n_ch,max_len=total)
        for k,v in opts.items(): b.set_option(k,v)
        b.run(xb[:call]); f=b.drain_frames()
        xs=xb[:call,:64].cpu().numpy(); o=Oracle(64); o.run(xs)
        g=f[f["channel"]<64]
        print(n_ch,total,call,opts,"frames",len(f),"first64 equal oracle:",g.tobytes()==o.frames().tobytes(), len(o.frames()), flush=True)
