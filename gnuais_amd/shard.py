"""Channel sharding over the GPUs of one node.

Channels are independent receivers (reference src/ais.c:141-147 creates one
struct receiver per channel, nothing is shared), so the path shards with NO
data-path collective: GPU g owns the contiguous block [g*N/G, (g+1)*N/G) of the
interleaved channel axis and runs its own batch.  torch.distributed is used only
for the benchmark's barrier and for reducing per-rank timings and counters.
"""
from __future__ import annotations


def shard_range(n_channels: int, world: int, rank: int):
    """[lo, hi) of rank's contiguous channel block; blocks tile [0, n_channels)."""
    assert 0 <= rank < world and n_channels >= 0
    return n_channels * rank // world, n_channels * (rank + 1) // world


def reduce_bench(dist, device, seconds: float, msgs: float, samples: float):
    """max over ranks of the elapsed time, sums of messages and samples."""
    import torch
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    s = torch.tensor([msgs, samples], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(s, op=dist.ReduceOp.SUM)
    return float(t.item()), float(s[0].item()), float(s[1].item())


class ReceiverNode:
    """ctypes face of the node object (include/gnuais_hip.h: gnuais_node_*; gnuais_amd/csrc/node.hip): N channels in
    contiguous blocks over `devices`, one batch and one host thread per device inside the library, results merged in
    the reference's order with global channel numbers.  The product path of BASELINE's C4."""

    def __init__(self, n_channels: int, devices=None, taps=None, pllinc: int = 0, max_len: int = 48000,
                 frame_capacity: int = 0):
        import ctypes as C
        import numpy as np
        from .lib import check, load
        self._C, self._np, self._check = C, np, check
        self._lib = load()
        self._h = C.c_void_p()
        dv = None if devices is None else np.ascontiguousarray(devices, dtype=np.int32)
        tp = None if taps is None else np.ascontiguousarray(taps, dtype=np.float32)
        rc = self._lib.gnuais_node_create(C.byref(self._h), None if dv is None else dv.ctypes.data,
                                          0 if dv is None else len(dv), n_channels,
                                          None if tp is None else tp.ctypes.data, 0 if tp is None else len(tp),
                                          pllinc, max_len, frame_capacity)
        self._raise(rc)
        self.n_channels, self.max_len = n_channels, max_len
        self.shards = []
        for i in range(self._lib.gnuais_node_n_devices(self._h)):
            d, f, n = C.c_int(), C.c_int(), C.c_int()
            self._lib.gnuais_node_shard(self._h, i, C.byref(d), C.byref(f), C.byref(n), None)
            self.shards.append((d.value, f.value, n.value))

    def _raise(self, rc, allow=()):
        if rc != 0 and rc not in allow:
            from .lib import GnuaisError
            raise GnuaisError(rc, self._lib.gnuais_node_last_error().decode())
        return rc

    def close(self):
        if getattr(self, "_h", None):
            self._lib.gnuais_node_destroy(self._h)
            self._h = None

    __del__ = close

    def run_host(self, samples):
        x = self._np.ascontiguousarray(samples, dtype=self._np.int16)
        assert x.ndim == 2 and x.shape[1] == self.n_channels
        self._raise(self._lib.gnuais_node_run_host(self._h, x.ctypes.data, int(x.shape[0])))

    def run(self, slabs, streams=None):
        """slabs: one CUDA/HIP int16 tensor [len][n_i] per shard, each on its shard's device."""
        C = self._C
        assert len(slabs) == len(self.shards)
        ln = int(slabs[0].shape[0])
        for t, (d, f, n) in zip(slabs, self.shards):
            assert t.is_cuda and t.is_contiguous() and t.shape == (ln, n) and t.device.index == d
        ptrs = (C.c_void_p * len(slabs))(*[t.data_ptr() for t in slabs])
        st = None if streams is None else (C.c_void_p * len(slabs))(*streams)
        self._raise(self._lib.gnuais_node_run(self._h, ptrs, ln, st))

    def autotune(self, slabs, streams=None) -> float:
        C = self._C
        ptrs = (C.c_void_p * len(slabs))(*[t.data_ptr() for t in slabs])
        st = None if streams is None else (C.c_void_p * len(slabs))(*streams)
        ms = C.c_float(0)
        self._raise(self._lib.gnuais_node_autotune(self._h, ptrs, int(slabs[0].shape[0]), st, C.byref(ms)))
        return ms.value

    def sync(self):
        self._raise(self._lib.gnuais_node_sync(self._h))

    def set_option(self, name: str, value: int):
        self._raise(self._lib.gnuais_node_set_option(self._h, name.encode(), int(value)))

    def reset(self):
        self._raise(self._lib.gnuais_node_reset(self._h))

    def pending_frames(self) -> int:
        n = self._C.c_int()
        self._raise(self._lib.gnuais_node_pending_frames(self._h, self._C.byref(n)))
        return n.value

    def stream_nmea(self):
        """gnuais_node_stream_nmea(): call after every run(); -> (sentences of every shard in shard order = the node's
        text for the call `stream_depth` calls ago, sentences, frames); frames == -1 while the pipelines fill.
        If a shard fails, the GnuaisError raised carries what the others delivered as `.partial` (bytes)."""
        C = self._C
        n = len(self.shards)
        texts, lens = (C.c_void_p * n)(), (C.c_size_t * n)()
        ns, nf = C.c_int(0), C.c_int(0)
        rc = self._lib.gnuais_node_stream_nmea(self._h, texts, lens, C.byref(ns), C.byref(nf))
        out = b"".join(C.string_at(texts[i], lens[i]) for i in range(n) if texts[i] and lens[i])
        try:
            self._raise(rc)
        except Exception as e:
            e.partial = out
            raise
        return out, ns.value, nf.value

    def mark(self):
        """start a per-shard measurement (gnuais_node_mark)"""
        self._raise(self._lib.gnuais_node_mark(self._h))

    def warnings(self):
        """what gnuais_node_create() could not do without failing (unpinned host threads), one text per shard"""
        return [l for l in self._lib.gnuais_node_warnings(self._h).decode().splitlines() if l]

    def shard_stats(self):
        """after sync(): per shard, where its host thread runs and how long it was busy since mark()"""
        C = self._C

        class Stat(C.Structure):
            _fields_ = [("device", C.c_int32), ("first_channel", C.c_int32), ("n_channels", C.c_int32),
                        ("numa_node", C.c_int32), ("pinned_cpus", C.c_int32), ("calls", C.c_longlong),
                        ("submit_ms", C.c_double), ("busy_ms", C.c_double), ("pci", C.c_char * 32)]
        out = []
        for i in range(len(self.shards)):
            st = Stat()
            self._raise(self._lib.gnuais_node_shard_stats(self._h, i, C.byref(st)))
            out.append(dict(device=st.device, first_channel=st.first_channel, n_channels=st.n_channels,
                            pci=st.pci.decode(), numa_node=st.numa_node, pinned_cpus=st.pinned_cpus, calls=st.calls,
                            submit_ms=st.submit_ms, busy_ms=st.busy_ms))
        return out

    def discard_frames(self):
        self._raise(self._lib.gnuais_node_discard_frames(self._h))

    def drain_frames(self):
        from .lib import FRAME_DTYPE
        out = self._np.zeros(max(self.pending_frames(), 1), dtype=FRAME_DTYPE)
        got = self._C.c_int()
        self._raise(self._lib.gnuais_node_drain_frames(self._h, out.ctypes.data, len(out), self._C.byref(got)))
        return out[: got.value].copy()

    def counters(self):
        from .lib import COUNTERS_DTYPE
        out = self._np.zeros(self.n_channels, dtype=COUNTERS_DTYPE)
        self._raise(self._lib.gnuais_node_counters(self._h, out.ctypes.data))
        return out

    def total_received(self) -> int:
        t = self._C.c_longlong()
        self._raise(self._lib.gnuais_node_total_received(self._h, self._C.byref(t)))
        return t.value

    def maxval(self):
        out = self._np.zeros(self.n_channels, dtype=self._np.int16)
        self._raise(self._lib.gnuais_node_maxval(self._h, out.ctypes.data))
        return out

    def pll_state(self):
        from .lib import PLL_DTYPE
        out = self._np.zeros(self.n_channels, dtype=PLL_DTYPE)
        self._raise(self._lib.gnuais_node_pll_state(self._h, out.ctypes.data))
        return out
