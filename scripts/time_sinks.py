"""Row f3 measurement: host post-stage throughput, messages/s, of
  (a) the reference's per-message path -- protodec_getdata() feeding serial_write(), printf +
      fflush(stdout) and the position cache one message at a time (oracle/_ref, its own code), and
  (b) the batched adapter (gnuais_amd/csrc/sinks_batch.c) feeding the SAME sink functions once per
      batch / vessel,
on identical frame records, serial port = /dev/null, stdout = /dev/null.
    python scripts/time_sinks.py [n_frames] [batch]"""
import ctypes as C
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases                                                    # noqa: E402
from oracle_lib import REF_SO, reference                        # noqa: E402
from test_sinks import Serial, Sinks                            # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 400000
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
    from gnuais_amd import lib
    ref = reference()
    n_ch = 64
    base, _ = cases.vessel_frames(seed=5, n_channels=n_ch, n=20000, n_mmsi=5000)
    fr = np.ascontiguousarray(np.tile(base, (n + len(base) - 1) // len(base))[:n])
    fr = fr[np.argsort(fr["channel"], kind="stable")]           # as drained: grouped by channel
    ref.reset()
    ref.add_receivers(n_ch)
    ref.lib.ref_cache_enable()
    ref.lib.ref_getdata_many.restype = C.c_double
    ref.lib.ref_getdata_many.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
    devnull = os.open("/dev/null", os.O_WRONLY)
    saved = os.dup(1)
    sys.stdout.flush()
    os.dup2(devnull, 1)                                          # the reference prints on fd 1
    t_ref = ref.lib.ref_getdata_many(fr.ctypes.data, len(fr), devnull, 1)
    t_ref_quiet = ref.lib.ref_getdata_many(fr.ctypes.data, len(fr), devnull, 0)
    os.dup2(saved, 1)

    with tempfile.TemporaryDirectory() as d:
        so = os.path.join(d, "libsinks.so")
        subprocess.check_call(["gcc", "-O2", "-std=gnu11", "-shared", "-fPIC", "-I", os.path.join(ROOT, "include"),
                               os.path.join(ROOT, "gnuais_amd", "csrc", "sinks_batch.c"), "-o", so])
        C.CDLL(REF_SO, mode=C.RTLD_GLOBAL)
        C.CDLL(lib.LIB_PATH, mode=C.RTLD_GLOBAL)
        L = C.CDLL(so)
        L.gnuais_sinks_deliver.argtypes = [C.POINTER(Sinks), C.c_void_p, C.c_int]
        libc = C.CDLL(None)
        libc.fdopen.restype = C.c_void_p
        libc.fdopen.argtypes = [C.c_int, C.c_char_p]
        ser = Serial(devnull)
        seq = np.zeros(n_ch, dtype=np.uint8)
        res = {}
        for label, text in (("all sinks", True), ("no stdout text", False)):
            s = Sinks()
            s.serial = C.pointer(ser)
            s.text_out = libc.fdopen(os.dup(devnull), b"w") if text else None
            s.use_cache, s.seqnr, s.n_channels = 1, seq.ctypes.data, n_ch
            t0 = time.perf_counter()
            for i in range(0, len(fr), batch):
                part = fr[i:i + batch]
                assert L.gnuais_sinks_deliver(C.byref(s), part.ctypes.data, len(part)) == 0
            res[label] = (time.perf_counter() - t0, s.serial_calls, s.cache_calls, s.flushes, s.vessels)
    print(f"{len(fr)} frames, {n_ch} channels, batches of {batch}, host cores {os.cpu_count()}")
    print(f"reference per-message path, all sinks      : {len(fr) / t_ref / 1e6:8.3f} M msgs/s")
    print(f"reference per-message path, no stdout text : {len(fr) / t_ref_quiet / 1e6:8.3f} M msgs/s")
    for label, (t, sc, cc, fl, nv) in res.items():
        print(f"batched adapter, {label:<26}: {len(fr) / t / 1e6:8.3f} M msgs/s   "
              f"({sc} serial_write, {cc} cache calls for {nv} vessel entries, {fl} fflush)")


if __name__ == "__main__":
    main()
