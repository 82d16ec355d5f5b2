"""ctypes binding of the C ABI (include/gnuais_hip.h -> gnuais_amd/libgnuais_hip.so).

There is no fallback: if the HIP library has not been built, or no HIP device
is usable, importing or calling raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libgnuais_hip.so")

OK, E_ARG, E_HIP, E_OVERFLOW, E_STATE = 0, -1, -2, -3, -4

FRAME_DTYPE = np.dtype([("channel", "<u4"), ("end_bit", "<u4"), ("payload", "u1", (53,)),
                        ("flags", "u1"), ("nbits", "<u2")])
COUNTERS_DTYPE = np.dtype([("receivedframes", "<i4"), ("lostframes", "<i4"), ("lostframes2", "<i4")])
PLL_DTYPE = np.dtype([("pll", "<u4"), ("prev", "<i4"), ("lastbit", "<i4")])
FSM_DTYPE = np.dtype([("state", "<i4"), ("nstartsign", "<i4"), ("antallpreamble", "<i4"),
                      ("antallenner", "<i4"), ("bitstuff", "<i4"), ("last", "<i4"),
                      ("bufferpos", "<i4")])
assert FRAME_DTYPE.itemsize == 64

# every symbol include/gnuais_hip.h declares: (restype, argtypes)
_P, _I, _U = C.c_void_p, C.c_int, C.c_uint
SYMBOLS = {
    "gnuais_batch_create": (_I, [C.POINTER(_P), _I, _I, _P, _I, _U, _I, _I]),
    "gnuais_batch_destroy": (None, [_P]),
    "gnuais_batch_reset": (_I, [_P]),
    "gnuais_batch_run": (_I, [_P, _P, _I, _P]),
    "gnuais_batch_run_host": (_I, [_P, _P, _I]),
    "gnuais_batch_run_host_async": (_I, [_P, _P, _I]),
    "gnuais_wav_open": (_I, [C.POINTER(_P), C.c_char_p, _I]),
    "gnuais_wav_channels": (_I, [_P]),
    "gnuais_wav_rate": (_I, [_P]),
    "gnuais_wav_read": (C.c_long, [_P, _P, C.c_long]),
    "gnuais_wav_close": (None, [_P]),
    "gnuais_batch_sync": (_I, [_P]),
    "gnuais_batch_filter": (_I, [_P, _P, _I, _P, _P]),
    "gnuais_batch_filter_host": (_I, [_P, _P, _I, _P]),
    "gnuais_batch_decode_bits": (_I, [_P, _P, _I, _P]),
    "gnuais_batch_last_bits": (_I, [_P, _P, _I, _P]),
    "gnuais_batch_last_signs": (_I, [_P, _P, _I]),
    "gnuais_batch_info": (_I, [_P, C.c_char_p, C.POINTER(C.c_double)]),
    "gnuais_batch_drain_frames": (_I, [_P, _P, _I, C.POINTER(_I)]),
    "gnuais_batch_pending_frames": (_I, [_P, C.POINTER(_I)]),
    "gnuais_batch_discard_frames": (_I, [_P, _P]),
    "gnuais_batch_counters": (_I, [_P, _P]),
    "gnuais_batch_total_received": (_I, [_P, C.POINTER(C.c_longlong)]),
    "gnuais_batch_maxval": (_I, [_P, _P]),
    "gnuais_batch_pll_state": (_I, [_P, _P]),
    "gnuais_batch_fsm_state": (_I, [_P, _P]),
    "gnuais_batch_protodec_reset": (_I, [_P]),
    "gnuais_batch_frame_bits": (_I, [_P, _I, _P, _I, C.POINTER(_I)]),
    "gnuais_batch_history": (_I, [_P, _P]),
    "gnuais_batch_n_channels": (_I, [_P]),
    "gnuais_batch_n_taps": (_I, [_P]),
    "gnuais_default_taps": (_I, [_P]),
    "gnuais_crc16_batch": (_I, [_I, _P, _I, _P, _I, _P]),
    "gnuais_crc16_bits": (_I, [_I, _P, _I, _P, _P, _I]),
    "gnuais_nmea_from_frames": (_I, [_P, _I, _P, _I, _P, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(_I)]),
    "gnuais_messages_from_frames": (_I, [_P, _I, _P, _P, _I, _P, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(_I),
                                         _P, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(_I)]),
    "gnuais_batch_drain_nmea": (_I, [_P, _P, _P, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(_I), C.POINTER(_I)]),
    "gnuais_batch_drain_messages": (_I, [_P, _P, _P, _P, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(_I), _P, C.c_size_t,
                                         C.POINTER(C.c_size_t), C.POINTER(_I), C.POINTER(_I)]),
    "gnuais_batch_fold_vessels": (_I, [_P, _P, _I, C.POINTER(_I)]),
    "gnuais_batch_vessel_table_enable": (_I, [_P, _I]),
    "gnuais_batch_vessel_table_update": (_I, [_P]),
    "gnuais_batch_vessel_table": (_I, [_P, _P, _I, C.POINTER(_I)]),
    "gnuais_batch_vessel_table_clear": (_I, [_P]),
    "gnuais_batch_stream_nmea": (_I, [_P, C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), C.POINTER(_I), C.POINTER(_I)]),
    "gnuais_batch_drain_frames_nmea": (_I, [_P, _P, _I, C.POINTER(_I), _P, _P, C.c_size_t, C.POINTER(C.c_size_t),
                                            C.POINTER(_I)]),
    "gnuais_range_from_frames": (_I, [_P, _I, _I, C.c_float, C.c_float, _P]),
    "gnuais_vessels_from_frames": (_I, [_P, _I, _P, _I, C.POINTER(_I)]),
    "gnuais_sql_plan_from_frames": (_I, [_P, _I, _P, _I, C.POINTER(_I)]),
    "gnuais_sql_calls_from_frames": (_I, [_P, _I, _I, _P, _I, C.POINTER(_I)]),
    "gnuais_tile_channels": (_I, [_P, _I, _I, _P, _I, _P]),
    "gnuais_batch_set_timing": (_I, [_P, _I]),
    "gnuais_batch_last_timing": (_I, [_P, _P]),
    "gnuais_batch_mean_timing": (_I, [_P, _P, C.POINTER(_I)]),
    "gnuais_batch_autotune": (_I, [_P, _P, _I, _P, C.POINTER(C.c_float)]),
    "gnuais_batch_autotune_delivery": (_I, [_P, _P, _I, _P, C.POINTER(C.c_float)]),
    "gnuais_batch_set_option": (_I, [_P, C.c_char_p, _I]),
    "gnuais_node_create": (_I, [C.POINTER(_P), _P, _I, _I, _P, _I, _U, _I, _I]),
    "gnuais_node_destroy": (None, [_P]),
    "gnuais_node_reset": (_I, [_P]),
    "gnuais_node_n_devices": (_I, [_P]),
    "gnuais_node_n_channels": (_I, [_P]),
    "gnuais_node_shard": (_I, [_P, _I, C.POINTER(_I), C.POINTER(_I), C.POINTER(_I), C.POINTER(_P)]),
    "gnuais_node_run_host": (_I, [_P, _P, _I]),
    "gnuais_node_run": (_I, [_P, _P, _I, _P]),
    "gnuais_node_sync": (_I, [_P]),
    "gnuais_node_pending_frames": (_I, [_P, C.POINTER(_I)]),
    "gnuais_node_drain_frames": (_I, [_P, _P, _I, C.POINTER(_I)]),
    "gnuais_node_stream_nmea": (_I, [_P, _P, _P, C.POINTER(_I), C.POINTER(_I)]),
    "gnuais_node_discard_frames": (_I, [_P]),
    "gnuais_node_counters": (_I, [_P, _P]),
    "gnuais_node_total_received": (_I, [_P, C.POINTER(C.c_longlong)]),
    "gnuais_node_maxval": (_I, [_P, _P]),
    "gnuais_node_pll_state": (_I, [_P, _P]),
    "gnuais_node_set_option": (_I, [_P, C.c_char_p, _I]),
    "gnuais_node_autotune": (_I, [_P, _P, _I, _P, C.POINTER(C.c_float)]),
    "gnuais_node_last_error": (C.c_char_p, []),
    "gnuais_node_warnings": (C.c_char_p, [_P]),
    "gnuais_node_mark": (_I, [_P]),
    "gnuais_node_shard_stats": (_I, [_P, _I, _P]),
    "gnuais_last_error": (C.c_char_p, []),
    "gnuais_version": (C.c_char_p, []),
}


class GnuaisError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"gnuais_hip error {code}: {msg}")
        self.code = code


_LIB = None


def load() -> C.CDLL:
    """dlopen the in-tree HIP library and bind every declared symbol."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `make -C gnuais_amd/csrc` "
                "(or __graft_entry__.build()); there is no CPU fallback")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(lib, name)          # AttributeError if the export is missing
            fn.restype = res
            fn.argtypes = args
        _LIB = lib
    return _LIB


def check(rc: int, allow=()) -> int:
    if rc != OK and rc not in allow:
        raise GnuaisError(rc, load().gnuais_last_error().decode())
    return rc


def default_taps() -> np.ndarray:
    t = np.zeros(36, dtype=np.float32)
    n = load().gnuais_default_taps(t.ctypes.data)
    assert n == 36
    return t
