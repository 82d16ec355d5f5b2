"""gnuais_batch_drain_messages() on one C3 call's frames: sentences + stdout lines formatted on the device,
against drain_frames + the host formatter (gnuais_messages_from_frames, threaded)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gnuais_amd import ReceiverBatch, synth, tile_channels, messages_from_frames
n_ch, total = 16384, 48000
base, _ = synth.make_base_streams(64, total)
x = tile_channels(torch.from_numpy(base).cuda(), n_ch)
b = ReceiverBatch(n_ch, max_len=total)
seq = np.zeros(n_ch, dtype=np.uint8)
for it in range(3):
    b.run(x)
    t = time.perf_counter()
    nm, tx, ns, nl, nf = b.drain_messages(seq)
    dt = time.perf_counter() - t
    print(f"device: {nf} frames, {ns} sentences, {nl} lines, {len(nm)/1e6:.1f} + {len(tx)/1e6:.1f} MB in {dt*1e3:.1f} ms = {nf/dt/1e6:.1f} M frames/s", flush=True)
seq2 = np.zeros(n_ch, dtype=np.uint8)
b.run(x)
t = time.perf_counter()
fr = b.drain_frames()
t1 = time.perf_counter()
nm2, tx2 = messages_from_frames(fr, seq2)
dt = time.perf_counter() - t
print(f"host: drain {1e3*(t1-t):.1f} ms + format {1e3*(time.perf_counter()-t1):.1f} ms = {len(fr)/dt/1e6:.1f} M frames/s; same text: {tx2 == tx and len(nm2) == len(nm)}")
# the C call alone, into buffers that exist already (what a C caller sees)
import ctypes as C
lib = b._lib
n = 400000
nmb = np.zeros(164 * n, dtype=np.uint8); txb = np.zeros(512 * n, dtype=np.uint8)
for it in range(3):
    b.run(x)
    nl, tl, ns, nlines, nf = C.c_size_t(0), C.c_size_t(0), C.c_int(0), C.c_int(0), C.c_int(0)
    t = time.perf_counter()
    rc = lib.gnuais_batch_drain_messages(b._h, seq.ctypes.data, None, nmb.ctypes.data, nmb.size, C.byref(nl), C.byref(ns),
                                         txb.ctypes.data, txb.size, C.byref(tl), C.byref(nlines), C.byref(nf))
    dt = time.perf_counter() - t
    print(f"C call: rc {rc}, {nf.value} frames in {dt*1e3:.1f} ms = {nf.value/dt/1e6:.1f} M frames/s", flush=True)
# row f3: the batch's vessel table folded on the device (before the drain), the C call alone
from gnuais_amd import VESSEL_DTYPE
tab = np.zeros(n, dtype=VESSEL_DTYPE)
for it in range(3):
    b.run(x)
    nv = C.c_int(0)
    t = time.perf_counter()
    rc = lib.gnuais_batch_fold_vessels(b._h, tab.ctypes.data, len(tab), C.byref(nv))
    dt = time.perf_counter() - t
    pend = b.pending_frames()
    print(f"fold_vessels: rc {rc}, {pend} frames -> {nv.value} vessels in {dt*1e3:.2f} ms = {pend/dt/1e6:.1f} M frames/s", flush=True)
    b.discard_frames()
