"""Host post-stage throughput: frames -> NMEA sentences (+ stdout text), serial vs threaded."""
import sys, os, time, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cases
from gnuais_amd import lib as L
lib = L.load()
fr, n = cases.nmea_frames(n_random=2000, n_channels=4096)
big = np.tile(fr, 150)
grouped = np.ascontiguousarray(big[np.argsort(big["channel"], kind="stable")])
big = np.ascontiguousarray(big)
nm = np.empty(170 * len(big), dtype=np.uint8); tx = np.empty(400 * len(big), dtype=np.uint8)
nm[:] = 0; tx[:] = 0
print("host threads available:", os.cpu_count())
for name, arr, text in (("serial, NMEA only", big, False), ("threaded, NMEA only", grouped, False),
                        ("serial, NMEA + text", big, True), ("threaded, NMEA + text", grouped, True)):
    seq = np.zeros(n, dtype=np.uint8); a, b = C.c_size_t(0), C.c_size_t(0)
    t = time.perf_counter()
    rc = lib.gnuais_messages_from_frames(arr.ctypes.data, len(arr), seq.ctypes.data, None, n, nm.ctypes.data, nm.size,
                                         C.byref(a), None, tx.ctypes.data if text else None, tx.size if text else 0,
                                         C.byref(b) if text else None, None)
    dt = time.perf_counter() - t
    print(f"{name:24s} rc {rc} {len(arr)} frames {dt*1e3:7.1f} ms  {len(arr)/dt/1e6:6.2f} M frames/s", flush=True)
