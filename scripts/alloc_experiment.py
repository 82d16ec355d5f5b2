"""Does the state of the allocator matter?  ms/step of consecutive batch objects in one process."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gnuais_amd import ReceiverBatch, synth, tile_channels
n_ch, total = 16384, 48000
base, _ = synth.make_base_streams(64, total)
x = tile_channels(torch.from_numpy(base).cuda(), n_ch)
def bench(b, steps=60):
    for _ in range(4):
        b.run(x, sync=False); b.discard_frames()
    b.sync(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(steps):
        b.run(x, sync=False); b.discard_frames()
    b.sync(); torch.cuda.synchronize()
    return (time.perf_counter() - t) / steps * 1e3
pre = int(os.environ.get("PRE_STREAMS", 0))      # streams the application created (and used) before us
keep = []
for i in range(pre):
    st = torch.cuda.Stream(priority=-1 if i % 2 else 0)
    with torch.cuda.stream(st):
        keep.append(torch.zeros(16, device="cuda") + 1)
torch.cuda.synchronize()
mode = os.environ.get("MODE", "seq")
if mode == "autotune":
    b = ReceiverBatch(n_ch, max_len=total)
    print("before autotune: ms/step %.3f" % bench(b), flush=True)
    t = time.perf_counter(); best = b.autotune(x); dt = time.perf_counter() - t
    print("autotune took %.2f s, best seen %.3f ms" % (dt, best), flush=True)
    for i in range(2):
        print("after autotune, round", i, "ms/step %.3f" % bench(b), flush=True)
elif mode == "seq":
    for i in range(4):
        b = ReceiverBatch(n_ch, max_len=total)
        print("batch", i, "ms/step %.3f" % bench(b), flush=True)
        del b
elif mode == "same":
    b = ReceiverBatch(n_ch, max_len=total)
    for i in range(4):
        print("same batch, round", i, "ms/step %.3f" % bench(b), flush=True)
elif mode == "prealloc":
    b0 = ReceiverBatch(n_ch, max_len=total); del b0
    b = ReceiverBatch(n_ch, max_len=total)
    for i in range(3):
        print("after create/destroy/create, round", i, "ms/step %.3f" % bench(b), flush=True)
elif mode == "two_alive":
    a = ReceiverBatch(n_ch, max_len=total)
    b = ReceiverBatch(n_ch, max_len=total)
    print("second batch while first alive: ms/step %.3f" % bench(b), flush=True)
    print("first batch: ms/step %.3f" % bench(a), flush=True)
elif mode == "streams_first":
    ss = [torch.cuda.Stream(priority=-1) for _ in range(4)]
    b = ReceiverBatch(n_ch, max_len=total)
    print("after creating 4 unrelated priority streams: ms/step %.3f" % bench(b), flush=True)
elif mode == "mem_first":
    junk = [torch.empty(256 << 20, dtype=torch.uint8, device="cuda") for _ in range(6)]
    del junk; torch.cuda.empty_cache()
    b = ReceiverBatch(n_ch, max_len=total)
    print("after allocating+freeing 1.5 GB through torch: ms/step %.3f" % bench(b), flush=True)
