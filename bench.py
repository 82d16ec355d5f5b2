#!/usr/bin/env python3
"""bench.py -- throughput of the batch AIS receive chain on MI355X.

Metric (BASELINE.json): Msamples/s demodulated (+ valid-CRC AIS msgs/s) over an
N-channel batch.  One step = one pass of the chain over one batch of synthetic
input resident in HBM.  Workloads (BASELINE.json configs):

  C3 (default, the headline): 16384 synthetic GMSK channels x 48000 samples (1 s at
      48 kHz) per GPU, full chain FIR -> slicer/PLL/NRZI -> HDLC deframe + CRC-16.
      (C3 rather than configs[1]: the metric counts valid-CRC messages, which only
      the full chain produces; configs[1] stops before the deframer.)
  C2: 256 channels x 48000 samples, filter.c + receiver.c only (no protodec).
  C5: 16384 channels x 192000 samples at 192 kHz (144 taps, pllinc 0x10000/20).

`--config` picks the one the headline line is about; a default single-GPU run also
measures the other two briefly and reports them under "other_configs".

Channels shard embarrassingly over GPUs (weak scaling, no data-path collective):
`--gpus N` started plainly runs ONE process that drives every device through the library's
node object (gnuais_node_*: a batch and a host thread per device, merged results);
`--gpus N --procs` starts one worker process per device instead -- N independent batches,
nothing shared but a start/stop barrier; launched by torch.distributed.run, the ranks it
created are used and RCCL provides the barrier and the max-over-ranks time.

Rank 0 prints ONE compact JSON line (< 4 KB: the contract's keys, `roofline`, `cpu_baseline`, the other
configs' ms / frac / bound) as the LAST line of stdout; every other measurement (per-kernel VALU tables, PMC
traffic detail, stage masks, the texts that say what each figure is) goes to bench_detail.json.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
T_START = time.perf_counter()

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s measured)
KERNEL_LEG_CALLS = 120         # the leg behind the timed region that `kernel_ms` comes from: events on every 2nd of
KERNEL_LEG_STRIDE = 2          # these calls = 60 samples per kernel (the library's event ring holds 64 calls);
                               # --no-kernel-leg shortens it to 12 calls (the profiler's child runs)

CONFIGS = {
    # stage_mask: bit 0 FIR/slicer, bit 1 PLL + NRZI, bit 3 HDLC deframer, bit 4 CRC + delivery
    "C2": dict(channels=256, len=48000, rate=48000, sps=5, wide=False, stage_mask=0x03,
               what="C2: 256 ch x 48000 samples @48 kHz, filter.c + receiver.c only (no protodec)"),
    "C3": dict(channels=16384, len=48000, rate=48000, sps=5, wide=False, stage_mask=0x1f,
               what="C3: 16384 ch x 48000 samples @48 kHz per GPU, full chain incl. HDLC/CRC-16"),
    "C5": dict(channels=16384, len=192000, rate=192000, sps=20, wide=True, stage_mask=0x1f,
               what="C5: 16384 ch x 192000 samples @192 kHz (144 taps, pllinc 0x10000/20) per GPU, "
                    "full chain incl. HDLC/CRC-16"),
}


def effective_cpus():
    """Host CPUs this process may actually use: affinity mask, capped by the cgroup CPU quota
    (the GPU box's container shows 256 logical CPUs and a quota of 16)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except Exception:
            pass
    return n


def cpu_baseline(x_host, n_sample_ch, total, x_wide=None, post_stage=False):
    """The reference's own code (oracle/_ref, kind 'reference') -- or the C
    restatement (kind 'port') when the prebuilt reference is absent -- timed on a
    bounded sample of the same workload on this host, single thread."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    xs = np.ascontiguousarray(x_host[:, :n_sample_ch])
    sample = f"{n_sample_ch} of the bench's channels x {total} samples, 1020-sample chunks"
    if oracle_lib.have_reference():
        ref = oracle_lib.reference()
        ref.add_receivers(n_sample_ch)
        t = time.perf_counter()
        got = ref.lib.ref_bench_run(xs.ctypes.data, total, 1020)
        dt = time.perf_counter() - t
        kind = "reference"
    else:
        o = oracle_lib.Oracle(n_sample_ch)
        t = time.perf_counter()
        o.run(xs)
        dt = time.perf_counter() - t
        got = int(o.counters()[:, 0].sum())
        kind = "port"
    res = {"value": n_sample_ch * total / dt / 1e6, "unit": "Msamples/s", "cores": 1, "cpu": cpu_model(),
           "kind": kind, "sample": sample, "seconds": round(dt, 2), "msgs": int(got)}
    # context: the same work on every host core (the reference itself is single-threaded by
    # design; this is the C restatement with channels partitioned over pthreads)
    if x_wide is not None:
        cores = effective_cpus()
        o = oracle_lib.Oracle(x_wide.shape[1])
        o.run(x_wide[:1020], threads=cores)                       # thread start-up, page faults
        o = oracle_lib.Oracle(x_wide.shape[1])
        t = time.perf_counter()
        o.run(x_wide, threads=cores)
        dt = time.perf_counter() - t
        res["all_cores"] = {"value": x_wide.shape[1] * total / dt / 1e6, "unit": "Msamples/s", "cores": cores,
                            "kind": "port", "sample": f"{x_wide.shape[1]} of the bench's channels x {total} samples, "
                            "interleaved input as the reference reads it, channels partitioned over threads",
                            "seconds": round(dt, 2)}
        # and the CPU's best case: the same channels de-interleaved first (planar, unit stride) --
        # not how the reference reads its input
        xp = np.ascontiguousarray(x_wide.T)
        o = oracle_lib.Oracle(x_wide.shape[1])
        o.run_planar(xp[:, :1020], cores)
        o = oracle_lib.Oracle(x_wide.shape[1])
        t = time.perf_counter()
        o.run_planar(xp, cores)
        dt = time.perf_counter() - t
        res["all_cores_planar"] = {"value": x_wide.shape[1] * total / dt / 1e6, "unit": "Msamples/s", "cores": cores,
                                   "kind": "port", "sample": f"{x_wide.shape[1]} of the bench's channels x {total} "
                                   "samples, de-interleaved beforehand (planar), channels partitioned over threads",
                                   "seconds": round(dt, 2), "msgs": int(o.counters()[:, 0].sum())}
    if post_stage:
        # row f3's host post-stage ("measure host post-stage msgs/s"): the reference's per-message path
        # (protodec_getdata -> serial_write, printf + fflush, cache) against the batched adapter (sinks_batch.c) in front of
        # the SAME unchanged sink functions, identical frame records, /dev/null behind both -- scripts/time_sinks.py, bounded
        import subprocess
        try:
            out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "time_sinks.py"), "200000", "100000"],
                                 capture_output=True, timeout=180, cwd=ROOT).stdout.decode()
            ps = {}
            for line in out.splitlines():
                if "M msgs/s" not in line:
                    continue
                label, rest = line.split(":", 1)
                ps[" ".join(label.split())] = float(rest.split()[0]) * 1e6
            if ps:
                res["post_stage"] = {"msgs_per_s": ps, "unit": "msgs/s", "cores": 1,
                                     "what": "host side behind the frame records, 200 000 frames: the reference's own "
                                             "per-message path vs the batched adapter feeding the same sink functions"}
        except Exception as e:                          # noqa: BLE001
            res["post_stage"] = {"error": str(e)}
    return res


class LocalSync:
    """Start/stop barrier and result exchange of the one-process-per-GPU workers bench.py starts
    itself: a multiprocessing barrier and a queue.  Nothing else is shared between the ranks."""

    def __init__(self, rank, world, barrier, queue):
        self.rank, self.world, self._barrier, self._queue = rank, world, barrier, queue

    def barrier(self):
        if self._barrier is not None:
            self._barrier.wait()

    def reduce(self, seconds, msgs, samples, per_rank):
        """-> (max seconds, sum msgs, sum samples, list of per-rank dicts) on rank 0."""
        if self.world == 1:
            return seconds, msgs, samples, [per_rank]
        if self.rank != 0:
            self._queue.put((self.rank, seconds, msgs, samples, per_rank))
            return None
        got = [(0, seconds, msgs, samples, per_rank)] + [self._queue.get() for _ in range(self.world - 1)]
        got.sort(key=lambda g: g[0])
        return (max(g[1] for g in got), sum(g[2] for g in got), sum(g[3] for g in got), [g[4] for g in got])


class DistSync:
    """The same over torch.distributed (RCCL) when torch.distributed.run created the ranks."""

    def __init__(self, dist, device, rank, world):
        self.dist, self.device, self.rank, self.world = dist, device, rank, world

    def barrier(self):
        self.dist.barrier()

    def reduce(self, seconds, msgs, samples, per_rank):
        from gnuais_amd.shard import reduce_bench
        t, m, s = reduce_bench(self.dist, self.device, seconds, msgs, samples)
        ranks = [None] * self.world
        self.dist.all_gather_object(ranks, per_rank)
        return (t, m, s, ranks) if self.rank == 0 else None


def measure(cfg, args, local, rank, sync, steps, warmup, isolated=True, post=False, keep_input=False, kernel_leg=True,
            options=None, uncalibrated=False):
    """One workload on this rank's GPU.  Returns a dict of raw measurements."""
    import torch
    from gnuais_amd import ReceiverBatch, params, synth, tile_channels

    device = torch.device("cuda", local)
    n_ch, total = cfg["channels"], cfg["len"]
    # synthetic input: base streams on the host once, tiled on the device
    base, _ = synth.make_base_streams(min(args.base, n_ch), total, seed=synth.SEED + rank, sps=cfg["sps"])
    x = tile_channels(torch.from_numpy(base).to(device), n_ch)
    torch.cuda.synchronize()
    kw = dict(taps=params.taps_192k(), pllinc=params.PLLINC_192K) if cfg["wide"] else {}
    b = ReceiverBatch(n_ch, max_len=total, device=local, **kw)
    stream = torch.cuda.current_stream(device).cuda_stream

    def step():
        b.run(x, stream=stream, sync=False)
        b.discard_frames(stream)

    # one-time calibration (untimed, before the warm-up): which internal stream serves which stage.
    # The hardware queue a stream gets depends on what the process created before and decides a good
    # part of the pipeline's speed; the library measures it on this input.
    b.set_option("stage_mask", cfg["stage_mask"])    # before the calibration: its calls run this workload's stages only
    uncal = None
    if uncalibrated:
        # what a plain gnuais_batch_run() caller gets: the same region with the library's default stage -> stream assignment
        for _ in range(max(warmup, 200)):       # as warm as the calibration's ~1400 calls leave the chip for the headline region
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        uncal = (time.perf_counter() - t0) / steps * 1e3
    b.autotune(x, stream)
    for k_, v_ in (options or {}).items():
        b.set_option(k_, v_)
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()

    rx0 = b.total_received()

    # timed region: exactly K steps, asynchronous, nothing else on the streams -- the per-kernel HIP events the library
    # can record cost stream time themselves (ten records a call: the 20-step figure was 4 % higher with them on every
    # 4th call), so they are NOT taken here but in the leg that continues this loop right behind the region (below)
    b.set_timing(False)
    sync.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt_own = time.perf_counter() - t0
    sync.barrier()
    dt = time.perf_counter() - t0
    rx1 = b.total_received()
    iso = None
    if isolated:        # one call at a time, nothing overlapping: context only (BEHIND the region: its syncs leave the chip idle)
        b.set_timing(True)
        acc = {k: [] for k in b.KERNELS + ("total",)}
        for _ in range(3):
            step()
            t = b.last_timing()
            for k in acc:
                acc[k].append(t[k])
        torch.cuda.synchronize()
        b.set_timing(False)
        iso = {k: float(np.mean(v)) for k, v in acc.items()}
        iso_total = iso.pop("total")
    out = {"n_ch": n_ch, "len": total, "dt": dt, "dt_own": dt_own, "steps": steps, "msgs": float(rx1 - rx0),
           "kernel_ms_isolated": iso, "uncalibrated_ms_per_step": uncal,
           "call_latency_isolated_ms": iso_total if isolated else None}
    # `kernel_ms`: the same loop again, long enough for a stable mean -- a 20-step region sampled on every 4th call gives
    # five samples per kernel, and which of two stages of nearly equal length "dominates" then flips from run to run
    leg = KERNEL_LEG_CALLS if kernel_leg else 12
    if leg:
        b.set_timing(True)
        b.set_option("timing_stride", KERNEL_LEG_STRIDE)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(leg):
            step()
        torch.cuda.synchronize()
        t_leg = time.perf_counter() - t0
        live2 = b.mean_timing()
        b.set_timing(False)
        b.set_option("timing_stride", 1)
        out["kernel_ms"] = {k: float(live2[k]) for k in b.KERNELS}
        out["call_latency_in_loop_ms"] = float(live2["total"])      # a call's first kernel start -> its last kernel's end
        out["kernel_ms_calls"] = int(live2["calls"])
        out["kernel_ms_leg"] = {"calls": leg, "events_on_every": KERNEL_LEG_STRIDE, "ms_per_step": t_leg / leg * 1e3,
                                "what": "the timed loop continued for %d calls with the library's per-kernel HIP events on "
                                        "every %d-th call (each kernel on the stream it is launched on); kernel_ms is the mean "
                                        "over the sampled calls" % (leg, KERNEL_LEG_STRIDE)}

    if post and steps < 100:
        # beside a short timed region (the driver's 20 steps carry one fill and one drain of the stage pipeline:
        # a call is about 1.5 ms from its first kernel to its last): the same loop over 200 steps
        n_ss = 200
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n_ss):
            step()
        torch.cuda.synchronize()
        t_ss = time.perf_counter() - t0
        out["steady_state"] = {"steps": n_ss, "ms_per_step": t_ss / n_ss * 1e3,
                               "Msamples_per_s": n_ch * total * n_ss / t_ss / 1e6,
                               "what": "the timed loop again over 200 steps (not the headline: context for the "
                                       "fill + drain share of a short region)"}

    if post and cfg["stage_mask"] == 0x1f and steps < 100:
        # which stages set the period: the same loop with stages switched off (stage_mask: 1 FIR, 2 PLL, 8 deframer, 16 K3;
        # the later stages re-read what the last full call left in the hand-off buffers, so their work is the real one) --
        # instruction counts cannot tell a chain that is issue-bound from one that waits on two serial recurrences
        legs = {}
        for name, mask in (("pll_alone", 0x02), ("deframer_and_k3", 0x18), ("without_fir", 0x1e), ("fir_alone", 0x01),
                           ("fir_and_pll", 0x03), ("all", 0x1f)):
            b.set_option("stage_mask", mask)
            for _ in range(8):
                step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(60):
                step()
            torch.cuda.synchronize()
            legs[name] = (time.perf_counter() - t0) / 60 * 1e3
        b.reset()                   # the masked legs left the receivers' state between stages inconsistent
        for _ in range(4):
            step()
        torch.cuda.synchronize()
        out["stage_masks"] = dict(legs, steps=60, what="ms per step of the pipelined loop with only the named stages running "
                                  "(gnuais_batch_set_option stage_mask; results are discarded): the PLL stage's launches are "
                                  "serial (they carry the receivers' state), so pll_alone is a floor under every arrangement")

    if post and args.e2e and cfg["stage_mask"] == 0x1f:
        # the rest of row f1 from the device: sentences AND the stdout line of every accepted frame
        # (gnuais_batch_drain_messages), one step's frames, the C call alone into buffers that exist already
        import ctypes as C
        b.sync()
        b.discard_frames(stream)
        cap = n_ch * 48
        # pinned host buffers, as a consumer at this rate would use (pageable memory halves the copy's speed)
        nmb = torch.zeros(164 * cap, dtype=torch.uint8, pin_memory=True).numpy()
        txb = torch.zeros(512 * cap, dtype=torch.uint8, pin_memory=True).numpy()
        seq = np.zeros(n_ch, dtype=np.uint8)
        best = None
        for _ in range(3):
            b.run(x, stream=stream, sync=True)
            nl, tl, ns, nln, nf = C.c_size_t(0), C.c_size_t(0), C.c_int(0), C.c_int(0), C.c_int(0)
            t0 = time.perf_counter()
            rc = b._lib.gnuais_batch_drain_messages(b._h, seq.ctypes.data, None, nmb.ctypes.data, nmb.size, C.byref(nl),
                                                    C.byref(ns), txb.ctypes.data, txb.size, C.byref(tl), C.byref(nln),
                                                    C.byref(nf))
            dt = time.perf_counter() - t0
            if rc == 0 and (best is None or dt < best[0]):
                best = (dt, nf.value, nln.value, nl.value, tl.value)
        if best:
            out["message_lines"] = {"what": "gnuais_batch_drain_messages: NMEA sentences + the stdout line of "
                                            "protodec_getdata() for one step's frames, formatted on the device, both "
                                            "texts in (pinned) host memory", "frames": best[1], "lines": best[2], "ms": best[0] * 1e3,
                                    "frames_per_s": best[1] / best[0], "nmea_bytes": best[3], "text_bytes": best[4]}
        del nmb, txb
        # end to end: every step's frames leave the device as NMEA text (formatted on the device, row
        # f1) and arrive in pinned host memory -- gnuais_batch_stream_nmea(), one call per step, the
        # formatter and the copy of step i overlapping the chain of steps i+1..i+3
        b.sync()
        b.on_overflow = "keep"                     # an overflowed slot still delivers what fitted (counted, reported)
        b.autotune_delivery(x, stream)              # the copy stream's place, measured like the stages' (resets)
        n_e2e = 100                                 # its own step count (reported): 20 would mostly time fill and drain
        for _ in range(12):                         # the first use allocates rings, text and scratch buffers
            b.run(x, stream=stream, sync=False)
            b.stream_nmea(copy=False)
        torch.cuda.synchronize()
        depth = b.stream_depth

        def delivery_loop():
            torch.cuda.synchronize()
            frames = text = sent = 0
            t0 = time.perf_counter()
            for i in range(n_e2e + depth):          # `depth` more calls deliver what is in flight
                if i < n_e2e:
                    b.run(x, stream=stream, sync=False)
                tx, ns, nf = b.stream_nmea(copy=False)
                if i >= depth:
                    frames += nf
                    sent += ns
                    text += len(tx)
            torch.cuda.synchronize()
            return time.perf_counter() - t0, frames, sent, text

        t_e2e, frames, sent, text = delivery_loop()
        out["end_to_end"] = {"what": "run + gnuais_batch_stream_nmea every step: chain, NMEA sentences formatted on the "
                                     "device in the reference's order, text into pinned host memory, handed out "
                                     "%d steps later" % depth,
                             "steps": n_e2e, "ms_per_step": t_e2e / n_e2e * 1e3,
                             "delivered_msgs_per_s": frames / t_e2e, "sentences": sent,
                             "text_bytes_per_step": text / n_e2e, "ring_overflows": b.stream_overflows,
                             "Msamples_per_s": n_ch * total * n_e2e / t_e2e / 1e6}
        # the same loop with the position cache carried on the device (row f3): every step's frames are folded into
        # the table behind their formatter
        b.vessel_table_enable(1 << 16)
        for _ in range(4):
            b.run(x, stream=stream, sync=False)
            b.stream_nmea(copy=False)
        t_tab, frames_t, _, _ = delivery_loop()
        out["end_to_end"]["with_vessel_table"] = {
            "what": "the same loop with gnuais_batch_vessel_table_enable(): sentences to the host + every frame folded "
                    "into the position cache kept on the device", "ms_per_step": t_tab / n_e2e * 1e3,
            "delivered_msgs_per_s": frames_t / t_tab, "vessels": int(len(b.vessel_table()))}
    if keep_input:
        out["x_cpu"] = x[:, : args.cpu_channels].cpu().numpy()
        out["x_wide"] = np.ascontiguousarray(x.cpu().numpy())
    del b, x
    torch.cuda.empty_cache()
    return out


MEASURED_HBM_GBS = 6290.0      # MI355X_MICROARCH.md: float4 copy, 79 % of the 8 TB/s spec


def roofline_of(m, ms_per_step=None, traffic=None, n_taps=36):
    """`roofline` of the bench line.  The unit of work is one input sample = 2 algorithmic bytes, a call
    is N x L of them, and every kernel of the chain works on the same call; `achieved` divides the call's
    algorithmic bytes by the mean duration of the kernel that takes LONGEST inside the pipelined loop (HIP events on
    that kernel's own stream) -- the stage that sets the pipeline's period.  `chain` is the same bytes over the
    driver-visible ms_per_step, `fir` the kernel that actually moves 98 % of them.  `bound` says what binds that
    kernel: north_star asks for the HBM fraction (achieved / peak / frac are always that), but SURVEY 8d / BASELINE.md 3
    also ask which bound binds -- "valu" when the kernel's VALU issue floor (PMC) exceeds the time HBM needs for its
    bytes and fills most of the launch, "hbm" when that time does, "latency" when neither comes near the launch's
    duration (the per-channel recurrences: one wave per 16-64 channels, serial through the call)."""
    alg = m["n_ch"] * m["len"] * 2.0
    km = {k: v for k, v in m["kernel_ms"].items() if v and v > 0}
    order = sorted(km, key=km.get, reverse=True)
    dom = order[0]                  # the longest, whichever it is; the FIR's own figures are under "fir" either way
    ach = alg / (km[dom] * 1e-3) / 1e9
    r = {"bound": "hbm" if (traffic and not traffic.get("error")) else None, "kernel": dom, "kernel_ms": km[dom], "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
         "frac": ach / HBM_PEAK_GBS, "traffic": None, "algorithmic_bytes_per_launch": alg,
         "algorithmic_flops_per_launch": m["n_ch"] * m["len"] * 2.0 * n_taps,
         "kernel_ms_samples": m.get("kernel_ms_calls"),
         "runner_up": ({"kernel": order[1], "kernel_ms": km[order[1]]} if len(order) > 1 else None),
         "why_this_kernel": "largest mean duration of the chain's kernels in the pipelined loop (all of them process "
                            "the same N x L samples per launch)"}
    if "fir_slice" in km:
        f = alg / (km["fir_slice"] * 1e-3) / 1e9
        r["fir"] = {"kernel": "fir_slice", "kernel_ms": km["fir_slice"], "achieved": f, "frac": f / HBM_PEAK_GBS,
                    "what": "the FIR/slicer launch: reads every sample once = 98 % of the chain's algorithmic bytes"}
        if m.get("kernel_ms_isolated"):
            r["fir"]["achieved_isolated"] = alg / (m["kernel_ms_isolated"]["fir_slice"] * 1e-3) / 1e9
    if ms_per_step:
        c = alg / (ms_per_step * 1e-3) / 1e9
        r["chain"] = {"achieved": c, "frac": c / HBM_PEAK_GBS, "frac_of_measured_peak": c / MEASURED_HBM_GBS,
                      "measured_peak": MEASURED_HBM_GBS, "ms_per_step": ms_per_step,
                      "what": "N*L*2 bytes / ms_per_step of this line (whole chain, as the driver clocks it)"}
    if traffic and not traffic.get("error"):
        # `traffic`: HBM bytes of the kernel `achieved` is quoted on, per launch; the whole chain's beside it
        r["traffic"] = traffic.get("bytes_per_launch", {}).get(dom, traffic.get("chain_bytes_per_call"))
        r["traffic_chain"] = traffic.get("chain_bytes_per_call")
        r["traffic_detail"] = {k: v for k, v in traffic.items() if k != "valu"}
        valu = traffic.get("valu") or {}
        wave_samples = m["n_ch"] * m["len"] / 64.0
        per = {}
        for k, v in valu.items():
            hbm_ms = (traffic["bytes_per_launch"].get(k, 0.0)) / (HBM_PEAK_GBS * 1e9) * 1e3
            in_pipe = km.get(k)
            if v["issue_floor_ms"] >= hbm_ms and in_pipe and v["issue_floor_ms"] >= 0.5 * v["launch_ms_in_this_pass"]:
                bound = "valu"
            elif in_pipe and hbm_ms >= 0.5 * v["launch_ms_in_this_pass"]:
                bound = "hbm"
            else:
                bound = "latency"
            per[k] = dict(v, insts_per_sample=(v["insts_per_launch"] / wave_samples) if v.get("insts_per_launch") else None,
                          hbm_ms_for_its_traffic=hbm_ms, bound=bound)
        if per:
            r["valu"] = dict(per.get(dom, {}), kernel=dom) if dom in per else None
            r["valu_by_kernel"] = per
            chain_floor = sum(v["issue_floor_ms"] for v in per.values())
            r["valu_chain"] = {"issue_floor_ms": chain_floor, "hbm_ms_for_algorithmic_bytes": alg / (HBM_PEAK_GBS * 1e9) * 1e3,
                               "bound": ("valu" if ms_per_step and chain_floor >= 0.8 * ms_per_step else
                                         ("hbm" if ms_per_step and alg / (HBM_PEAK_GBS * 1e9) * 1e3 >= 0.8 * ms_per_step else "latency")),
                               "what": "sum of the chain's kernels' VALU issue floors (they share the chip's 1024 SIMDs in "
                                       "the pipelined loop) against the time 8 TB/s needs for the call's algorithmic bytes. "
                                       "The floor prices every wave-instruction at the 4 cycles SQ_ACTIVE_INST_VALU counts it "
                                       "as; micro-benchmarked issue costs on this chip are 2.4 (add, mul) to 4.7 (fma with a "
                                       "scalar operand, packed) cycles, so read it as an upper estimate of how full the VALU "
                                       "is: between ~0.65 x and 1 x this figure"}
            sm = m.get("stage_masks")
            if sm and ms_per_step:
                # measured, not priced: if the PLL stage alone already needs most of the period, the chain waits on its
                # recurrence, however full the VALU is
                r["valu_chain"]["stage_masks"] = {k: v for k, v in sm.items() if k not in ("what", "steps")}
                if sm["pll_alone"] >= 0.8 * sm["all"]:
                    r["valu_chain"]["bound"] = "latency"
                    r["valu_chain"]["bound_why"] = ("the PLL stage alone (serial launches of a per-channel recurrence) needs %.3f ms "
                                                    "per step, %.0f %% of the %.3f the whole chain needs in the same loop; without "
                                                    "the FIR %.3f, FIR alone %.3f" % (sm["pll_alone"], 100 * sm["pll_alone"] / sm["all"],
                                                                                     sm["all"], sm["without_fir"], sm["fir_alone"]))
            r["bound"] = per[dom]["bound"] if dom in per else r["bound"]
            if r["bound"] == "latency":
                r["latency_note"] = ("a per-channel recurrence: one wave per 16-64 channels walks the call serially "
                                     "(receiver.c:113-134 / protodec.c:988-1122); the PLL's row costs ~46 clock ticks per "
                                     "transition on its wave (profiles/r03_ubench_pll_step6.txt), ~14 000 steps per call "
                                     "for the noisiest lane of a wave")
            r["bound_why"] = ("%s: VALU issue floor %.3f ms, HBM time for its own traffic %.3f ms, launch alone %.3f ms, "
                              "in the pipeline %.3f ms; the chain as a whole: VALU floor %.3f ms against %.3f ms of HBM time"
                              % (dom, per[dom]["issue_floor_ms"], per[dom]["hbm_ms_for_its_traffic"],
                                 per[dom]["launch_ms_in_this_pass"], km[dom], chain_floor,
                                 alg / (HBM_PEAK_GBS * 1e9) * 1e3)) if dom in per else None
    elif traffic:
        r["traffic_detail"] = traffic
    return r


N_SIMD = 1024                  # 256 CUs x 4 SIMDs (MI355X_MICROARCH.md)
SPEC_CLOCK_HZ = 2.4e9          # max engine clock; the PMC pass reports the clock it saw when GRBM_GUI_ACTIVE is readable


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def pmc_traffic(args, config, device=0):
    """PMC counters per launch, collected as MI355X_MICROARCH.md's HBM / rocprofv3 section prescribes: separate
    rocprofv3 passes with --kernel-trace only, each over a 6-step child run of this same script.
      * HBM bytes: --pmc FETCH_SIZE and --pmc WRITE_SIZE (KiB; FETCH_SIZE doubled: gfx950 tallies 128-byte requests at 64);
      * VALU issue: --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAVES (+ GRBM_GUI_ACTIVE for the clock):
        wave-instructions per launch, the quad-cycles the VALU was issuing, and from them the kernel's issue floor
        = SQ_ACTIVE_INST_VALU x 4 cycles / (1024 SIMDs x clock) and the share of the launch it fills."""
    import csv
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return {"error": "rocprofv3 not found"}
    child = [sys.executable, os.path.join(ROOT, "bench.py"), "--config", config, "--no-cpu", "--no-others", "--no-e2e",
             "--no-traffic", "--no-kernel-leg", "--steps", "6", "--warmup", "1", "--base", str(args.base)]
    short = (("fir_sign", "fir_slice"), ("fir_slice_generic", "fir_slice"), ("fir_slice", "fir_slice"), ("pll_h3_kernel", "pll"), ("pll_tp_kernel", "pll"),
             ("hdlc_events", "hdlc_deframe"), ("hdlc_deframe", "hdlc_deframe"), ("hdlc_crc", "hdlc_crc"))
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    # the child is a one-GPU run on THIS rank's device (a multi-GPU job: rank 0's; every GPU runs the same workload)
    vis = [v for v in os.environ.get("HIP_VISIBLE_DEVICES", "").split(",") if v.strip()]
    env["HIP_VISIBLE_DEVICES"] = vis[device] if device < len(vis) else str(device)
    res, raw = {}, {}
    tmp = tempfile.mkdtemp(prefix="gnuais_pmc_", dir="/tmp")

    def one_pass(tag, counters):
        """-> ({kernel: {counter: mean per launch}}, {kernel: mean duration of a launch in this pass, ms}) or None"""
        d = os.path.join(tmp, tag)
        cmd = [exe, "--kernel-trace", "--pmc"] + counters + ["--output-format", "csv", "-d", d, "-o", "pmc", "--"] + child
        try:
            subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                           timeout=420, check=False)
        except subprocess.TimeoutExpired:
            return None
        acc, dur = {}, {}
        for dp, _, fs in os.walk(d):
            for f in fs:
                if not f.endswith("counter_collection.csv"):
                    continue
                with open(os.path.join(dp, f)) as fh:
                    for row in csv.DictReader(fh):
                        name = next((s_ for pat, s_ in short if pat in row["Kernel_Name"]), None)
                        if not name:
                            continue
                        # a stage may be more than one kernel per call (C5's FIR: the call's first segment and the rest):
                        # the mean per launch of each distinct kernel, summed over the kernels of the stage
                        full = row["Kernel_Name"]
                        acc.setdefault(name, {}).setdefault(full, {}).setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
                        try:
                            dur.setdefault(name, {}).setdefault(full, {})[row["Dispatch_Id"]] = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6
                        except (KeyError, ValueError):
                            pass
        if not acc:
            return None
        out_c, out_d = {}, {}
        for k, per in acc.items():
            for cs in per.values():
                for c, v in cs.items():
                    out_c.setdefault(k, {})[c] = out_c.get(k, {}).get(c, 0.0) + sum(v) / len(v)
        for k, per in dur.items():
            out_d[k] = sum(sum(v.values()) / len(v) for v in per.values() if v)
        return out_c, out_d

    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            got = one_pass(counter, [counter])
            if not got:
                return {"error": f"no {counter} rows in the rocprofv3 output"}
            raw[counter] = {k: v.get(counter, 0.0) for k, v in got[0].items()}
        for k in raw["FETCH_SIZE"]:
            res[k] = (2.0 * raw["FETCH_SIZE"][k] + raw["WRITE_SIZE"].get(k, 0.0)) * 1024.0
        valu = None
        sq = ["SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", "SQ_WAVES"]
        got = one_pass("SQ", sq + ["SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE"]) or one_pass("SQ1", sq + ["GRBM_GUI_ACTIVE"]) or one_pass("SQ2", sq)
        if got:
            valu = {}
            for k, c in got[0].items():
                ms = got[1].get(k)
                if not ms or "SQ_ACTIVE_INST_VALU" not in c:
                    continue
                clock = SPEC_CLOCK_HZ
                if c.get("GRBM_GUI_ACTIVE"):
                    clock = c["GRBM_GUI_ACTIVE"] / 8.0 / (ms * 1e-3)         # the counter sums the 8 XCDs
                # what an instruction costs its SIMD: 4 cycles is what SQ_ACTIVE_INST_VALU counts and what VOP3 / packed
                # forms take (1.9 ns at four waves per SIMD); the 12-tap direct-form stream of fir_sign_kernel is VOP2
                # adds, muls and fmacs and was measured at 1.02 ns per instruction = 2.45 cycles
                # (profiles/r05_ubench_valu_op_rates.txt: direct12_now_8out)
                inst_cycles = 2.45 if (k == "fir_slice" and config in ("C2", "C3")) else 4.0
                floor_ms = c["SQ_ACTIVE_INST_VALU"] * inst_cycles / (N_SIMD * clock) * 1e3
                # a kernel that also runs matrix products (C5's slicer): the pipe's busy cycles (32 per product, summed over
                # the SIMDs) on top -- its two waves per SIMD run in phase, products and vector work do not overlap
                # (profiles/r05_c5_matrix_pipe.txt)
                mfma_ms = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (N_SIMD * clock) * 1e3
                floor_ms += mfma_ms
                valu[k] = {"insts_per_launch": c.get("SQ_INSTS_VALU"), "active_quad_cycles": c["SQ_ACTIVE_INST_VALU"],
                           "issue_floor_ms_at_4_cycles": c["SQ_ACTIVE_INST_VALU"] * 4.0 / (N_SIMD * clock) * 1e3 + mfma_ms,
                           "waves": c.get("SQ_WAVES"), "launch_ms_in_this_pass": ms, "clock_ghz": clock / 1e9,
                           "clock_from": "GRBM_GUI_ACTIVE / 8 XCDs / duration" if c.get("GRBM_GUI_ACTIVE") else "spec",
                           "issue_floor_ms": floor_ms, "mfma_busy_ms": mfma_ms, "busy_frac": floor_ms / ms, "cycles_per_instruction_priced": inst_cycles,
                           "valu_share_of_wave_cycles": (c["SQ_ACTIVE_INST_VALU"] / c["SQ_WAVE_CYCLES"]) if c.get("SQ_WAVE_CYCLES") else None}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return {"bytes_per_launch": res, "chain_bytes_per_call": sum(res.values()), "raw_kib_per_launch": raw, "valu": valu,
            "how": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE | SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES "
                   "SQ_WAVES GRBM_GUI_ACTIVE (three separate passes) -- python bench.py --config " + config +
                   " --no-cpu --no-others --no-e2e --no-traffic --no-kernel-leg --steps 6 --warmup 1, launched by this run; "
                   "sizes in KiB, FETCH_SIZE x 2 (gfx950 correction of MI355X_MICROARCH.md); issue_floor_ms = "
                   "SQ_ACTIVE_INST_VALU x 4 cycles / (1024 SIMDs x clock); launches run one at a time under the profiler"}


def float_path(args, local, shapes=(("C2", 256, 48000), ("C3", 16384, 48000))):
    """north_star's pre-slicer floats: gnuais_batch_filter() (the exact K1, bit-identical to filter_run_buf()) on the
    BASELINE shapes, output [L][N] float32 written to HBM: ms per call, bytes = 2 read + 4 written per sample."""
    import torch
    from gnuais_amd import ReceiverBatch, synth, tile_channels
    import ctypes as C
    device = torch.device("cuda", local)
    out = {}
    for name, n_ch, total in shapes:
        base, _ = synth.make_base_streams(min(args.base, n_ch), total, seed=synth.SEED)
        x = tile_channels(torch.from_numpy(base).to(device), n_ch)
        y = torch.empty((total, n_ch), dtype=torch.float32, device=device)
        b = ReceiverBatch(n_ch, max_len=total, device=local)
        stream = torch.cuda.current_stream(device).cuda_stream
        def call():
            rc = b._lib.gnuais_batch_filter(b._h, x.data_ptr(), total, y.data_ptr(), C.c_void_p(stream))
            assert rc == 0
        for _ in range(3):
            call()
        torch.cuda.synchronize()
        n = 20
        t0 = time.perf_counter()
        for _ in range(n):
            call()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
        moved = n_ch * total * 6.0
        out[name] = {"channels": n_ch, "samples_per_channel": total, "ms": ms, "Msamples_per_s": n_ch * total / ms / 1e3,
                     "GB_per_s": moved / ms / 1e6, "frac_of_hbm_peak": moved / ms / 1e6 / HBM_PEAK_GBS,
                     "bytes": "2 read + 4 written per sample"}
        del b, x, y
        torch.cuda.empty_cache()
    out["what"] = ("gnuais_batch_filter(): the exact FIR (strict tap order, separate mul/add), floats bit-identical to "
                   "filter_run_buf() (tests/test_hip_parity.py, tests/test_hip_fullsize.py), 20 calls back to back")
    return out



COMPACT_LIMIT = 4096           # the driver keeps an 8 KB tail of stdout and parses its LAST line: stay far below


def _r(v, nd=4):
    """Round floats (recursively) so that the compact line spends its bytes on digits that mean something."""
    if isinstance(v, float):
        if v != v or v in (float("inf"), float("-inf")):
            return None
        return round(v, nd) if abs(v) < 1e4 else round(v, 1)
    if isinstance(v, dict):
        return {k: _r(x, nd) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_r(x, nd) for x in v]
    return v


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d and d[k] is not None}


def compact_line(out, detail_path=None):
    """The ONE line the driver parses: the contract's keys + `roofline` + `cpu_baseline`, nothing that is prose.
    Everything else of `out` (per-kernel VALU tables, traffic detail, stage masks, the e2e legs, every `what`)
    goes to the detail file.  Guaranteed shorter than COMPACT_LIMIT bytes: optional blocks are dropped, last first,
    until it is (tests/test_bench_line.py holds it to that on canned measurements)."""
    c = _pick(out, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                    "vs_baseline", "dtype", "data", "valid_crc_msgs_per_s", "x_realtime_channels", "uncalibrated_ms_per_step",
                    "error"))
    c.setdefault("vs_baseline", None)
    cfg = out.get("config") or {}
    c["config"] = _pick(cfg, ("workload", "channels_per_gpu", "samples_per_channel", "parallelism"))
    r = out.get("roofline") or {}
    rr = _pick(r, ("bound", "kernel", "kernel_ms", "achieved", "peak", "unit", "frac", "traffic", "traffic_chain",
                   "algorithmic_bytes_per_launch"))
    rr.setdefault("traffic", None)
    if isinstance(r.get("runner_up"), dict):
        rr["runner_up"] = _pick(r["runner_up"], ("kernel", "kernel_ms"))
    if isinstance(r.get("chain"), dict):
        rr["chain"] = _pick(r["chain"], ("ms_per_step", "achieved", "frac"))
    if isinstance(r.get("fir"), dict):
        rr["fir"] = _pick(r["fir"], ("kernel_ms", "frac"))
    if isinstance(r.get("valu"), dict):
        rr["valu"] = _pick(r["valu"], ("insts_per_sample", "busy_frac", "issue_floor_ms"))
    if isinstance(r.get("valu_chain"), dict):
        rr["valu_chain"] = _pick(r["valu_chain"], ("issue_floor_ms", "bound"))
    c["roofline"] = rr
    cb = out.get("cpu_baseline")
    if isinstance(cb, dict):
        cc = _pick(cb, ("value", "unit", "cores", "cpu", "kind", "sample", "seconds"))
        for k in ("all_cores", "all_cores_planar"):
            if isinstance(cb.get(k), dict):
                cc[k] = _pick(cb[k], ("value", "cores"))
        if isinstance(cb.get("post_stage"), dict) and isinstance(cb["post_stage"].get("msgs_per_s"), dict):
            ps = cb["post_stage"]["msgs_per_s"]
            cc["post_stage"] = {"reference": ps.get("reference per-message path, all sinks"),
                                "batched": ps.get("batched adapter, all sinks")}
        c["cpu_baseline"] = cc
    optional = []                   # (key, value), most dispensable LAST
    if isinstance(out.get("kernel_ms"), dict):
        optional.append(("kernel_ms", out["kernel_ms"]))
    if isinstance(out.get("steady_state"), dict):
        optional.append(("steady_state", _pick(out["steady_state"], ("steps", "ms_per_step"))))
    if isinstance(out.get("call_latency_ms"), dict):
        optional.append(("call_latency_ms", _pick(out["call_latency_ms"], ("isolated", "in_loop"))))
    if isinstance(out.get("stage_masks"), dict):
        optional.append(("stage_masks", {k: v for k, v in out["stage_masks"].items() if k not in ("what", "steps")}))
    oc = out.get("other_configs")
    if isinstance(oc, dict):
        o2 = {}
        for name, o in oc.items():
            e = _pick(o, ("ms_per_step", "value"))
            orf = o.get("roofline") or {}
            if isinstance(orf.get("chain"), dict) and "frac" in orf["chain"]:
                e["frac"] = orf["chain"]["frac"]
            e["bound"] = orf.get("bound")
            e["traffic"] = orf.get("traffic")
            e["kernel"] = orf.get("kernel")
            o2[name] = e
        optional.append(("other_configs", o2))
    if isinstance(out.get("exact_chain"), dict):
        optional.append(("exact_chain", _pick(out["exact_chain"], ("ms_per_step", "frac_of_hbm_peak",
                                                                   "same_msgs_as_the_default_chain"))))
    if isinstance(out.get("kernel_ms_isolated"), dict):
        optional.append(("kernel_ms_isolated", out["kernel_ms_isolated"]))
    if isinstance(out.get("per_gpu"), list):
        optional.append(("per_gpu", [_pick(g, ("rank", "shard", "device", "channels", "ms_per_step"))
                                     for g in out["per_gpu"]]))
    if isinstance(out.get("end_to_end"), dict):
        optional.append(("end_to_end", _pick(out["end_to_end"], ("ms_per_step", "delivered_msgs_per_s"))))
    if out.get("bench_seconds") is not None:
        optional.append(("bench_seconds", out["bench_seconds"]))
    for k, v in optional:
        c[k] = v
    if detail_path:
        c["detail"] = detail_path
    c = _r(c)
    for k in ("value", "ms_per_step"):          # the driver recomputes one from the other: keep them as measured
        if isinstance(out.get(k), float):
            c[k] = float("%.9g" % out[k])
    # the texts are the only unbounded fields: cut them first, then drop optional blocks, most dispensable first
    for holder, key in ((c["config"], "workload"), (c["config"], "parallelism"), (c.get("cpu_baseline", {}), "sample"),
                        (c.get("cpu_baseline", {}), "cpu"), (c, "metric"), (c, "dtype")):
        if isinstance(holder.get(key), str) and len(holder[key]) > 160:
            holder[key] = holder[key][:157] + "..."
    line = json.dumps(c, separators=(",", ":"))
    while len(line) >= COMPACT_LIMIT and optional:
        k, _ = optional.pop()
        c.pop(k, None)
        line = json.dumps(c, separators=(",", ":"))
    assert len(line) < COMPACT_LIMIT, len(line)
    return line


def emit(out, args):
    """Everything measured -> the detail file (bench_detail.json beside this script, and under gpurun_out/ when that
    exists so that it comes back from the GPU box); the compact line -> stdout, LAST."""
    detail = getattr(args, "detail", None) or os.path.join(ROOT, "bench_detail.json")
    written = None
    for path in [detail] + ([os.path.join(ROOT, "gpurun_out", "bench_detail.json")]
                            if os.path.isdir(os.path.join(ROOT, "gpurun_out")) else []):
        try:
            with open(path, "w") as fh:
                json.dump(out, fh, indent=1)
            written = written or os.path.relpath(path, ROOT)
        except OSError:
            pass
    if getattr(args, "print_detail", False):
        print(json.dumps(out), flush=True)             # an EARLIER stdout line, on request only
    sys.stderr.flush()
    print(compact_line(out, written), flush=True)


def rank_main(rank, local, world, args, sync):
    import torch
    torch.cuda.set_device(local)
    cfg = CONFIGS[args.config]
    if args.channels:
        cfg = dict(cfg, channels=args.channels)
    if args.len:
        cfg = dict(cfg, len=args.len)
    want_cpu = rank == 0 and world == 1 and args.cpu and args.config == "C3"
    # the FIRST collective brings RCCL up (communicator, buffers, kernels: hundreds of ms with the device idle); it must not be
    # the barrier that opens the timed region -- behind it the first steps ran at idle clocks (0.59 instead of 0.50 ms per step)
    sync.barrier()
    m = measure(cfg, args, local, rank, sync, args.steps, args.warmup, post=(rank == 0), keep_input=want_cpu,
                kernel_leg=args.kernel_leg, uncalibrated=(world == 1))
    x_cpu, x_wide = m.pop("x_cpu", None), m.pop("x_wide", None)
    per_rank = {"rank": rank, "device": local, "ms_per_step": m["dt_own"] / m["steps"] * 1e3}
    # every rank left the opening barrier together and clocked its own K steps up to its own synchronize; the job's time is
    # the MAX of those over the ranks (the reduction) -- not a clock that also contains the closing barrier's own latency
    # (an RCCL barrier is 0.1 ms per step of a 20-step region)
    red = sync.reduce(m["dt_own"], m["msgs"], float(m["n_ch"]) * m["len"] * m["steps"], per_rank)
    if rank != 0:
        return
    dt, msgs, samples, ranks = red
    value = samples / dt / 1e6
    out = {
        "metric": "Msamples/s demodulated (N-channel batch, " + ("full chain" if cfg["stage_mask"] == 0x1f
                                                                 else "FIR + receiver") + ")",
        "value": value, "unit": "Msamples/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32 FIR on int16 samples; u32 PLL/HDLC/CRC", "data": "synthetic",
        "config": {"workload": cfg["what"], "channels_per_gpu": m["n_ch"], "samples_per_channel": m["len"],
                   "parallelism": f"channels sharded over {world} GPU(s), one process per GPU, no collectives"},
        "valid_crc_msgs_per_s": msgs / dt,
        "x_realtime_channels": value / (cfg["rate"] / 1e6),
        "kernel_ms": m["kernel_ms"], "kernel_ms_isolated": m["kernel_ms_isolated"],
        "kernel_ms_calls": m["kernel_ms_calls"], "kernel_ms_leg": m.get("kernel_ms_leg"),
        "per_gpu": ranks,
        "end_to_end": m.get("end_to_end"),
        "message_lines": m.get("message_lines"),
        "roofline": roofline_of(m, dt / args.steps * 1e3,
                                pmc_traffic(args, args.config, local) if args.traffic else None,
                                n_taps=144 if cfg["wide"] else 36),
        "float_path": float_path(args, local) if (world == 1 and args.e2e and args.config == "C3") else None,
        "steady_state": m.get("steady_state"),
        "stage_masks": m.get("stage_masks"),
        "uncalibrated_ms_per_step": m.get("uncalibrated_ms_per_step"),
        "call_latency_ms": {"isolated": m.get("call_latency_isolated_ms"), "in_loop": m.get("call_latency_in_loop_ms"),
                            "what": "one call, first kernel's start to last kernel's end (HIP events): alone on the chip / "
                                    "inside the pipelined loop, where three calls are in flight"},
        "uncalibrated_what": "the same K steps BEFORE gnuais_batch_autotune() (after 200 warm-up calls): the library's default "
                             "stage -> stream assignment, what a plain gnuais_batch_run() caller gets",
        "note": "the receive path slices the sign of the filter output and never stores the pre-slicer "
                "floats; gnuais_batch_filter() produces them (bit-exact, tests/test_hip_parity.py)",
    }
    if world == 1 and args.others and args.config == "C3" and not args.channels and not args.len:
        # the same chain with the EXACT FIR in it (fir_slice_kernel: the reference's ordered 36-tap fp32 sum formed for
        # every sample, then `out > 0`) instead of the sign-certified slicer the receive path runs by default: same
        # bits, frames and counters (checked here on the message count), the floats' cost made visible
        # (the same sequence of calls as the headline measurement, so that the message counts can be compared)
        e = measure(cfg, args, local, rank, sync, args.steps, args.warmup, isolated=True, kernel_leg=False,
                    options={"fir_variant": 0})
        ems = e["dt"] / e["steps"] * 1e3
        out["exact_chain"] = {
            "what": "full chain with fir_variant = 0: K1 forms the reference's exact ordered fp32 sum for every sample "
                    "(fir_slice_kernel) instead of certifying its sign (K1s); everything downstream unchanged",
            "steps": e["steps"], "ms_per_step": ems, "Msamples_per_s": e["n_ch"] * e["len"] * e["steps"] / e["dt"] / 1e6,
            "frac_of_hbm_peak": e["n_ch"] * e["len"] * 2.0 / (ems * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "kernel_ms": e["kernel_ms"],
            "valid_crc_msgs": e["msgs"], "same_msgs_as_the_default_chain": e["msgs"] == m["msgs"]}
        others = {}
        for name in ("C2", "C5"):
            o = measure(CONFIGS[name], args, local, rank, sync, *((60, 10) if name == "C2" else (20, 4)),
                        kernel_leg=args.kernel_leg)
            v = o["n_ch"] * o["len"] * o["steps"] / o["dt"] / 1e6
            others[name] = {"workload": CONFIGS[name]["what"], "value": v, "unit": "Msamples/s",
                            "ms_per_step": o["dt"] / o["steps"] * 1e3, "steps": o["steps"],
                            "valid_crc_msgs_per_s": o["msgs"] / o["dt"],
                            "x_realtime_channels": v / (CONFIGS[name]["rate"] / 1e6),
                            "kernel_ms": o["kernel_ms"], "kernel_ms_isolated": o["kernel_ms_isolated"],
                            "dominant_kernel": max(o["kernel_ms"], key=o["kernel_ms"].get),
                            "roofline": roofline_of(o, o["dt"] / o["steps"] * 1e3,
                                                    pmc_traffic(args, name) if (args.traffic and args.traffic_others) else None,
                                                    n_taps=144 if CONFIGS[name]["wide"] else 36)}
        out["other_configs"] = others
    if x_cpu is not None:
        out["cpu_baseline"] = cpu_baseline(x_cpu, args.cpu_channels, m["len"], x_wide, post_stage=args.e2e)
    out["bench_seconds"] = time.perf_counter() - T_START
    emit(out, args)


def node_main(world, args):
    """--gpus N started plainly: ONE process, the library's node object (gnuais_node_*, gnuais_amd/csrc/node.hip) --
    a batch and a host thread per device inside the library, 16384 channels per device (weak scaling), every step one
    gnuais_node_run() over device-resident slabs; nothing is exchanged between devices."""
    import torch
    from gnuais_amd import ReceiverNode, params, synth, tile_channels
    cfg = CONFIGS[args.config]
    if args.channels:
        cfg = dict(cfg, channels=args.channels)
    if args.len:
        cfg = dict(cfg, len=args.len)
    devs = [int(d) for d in args.devices.split(",")] if args.devices else list(range(world))
    devs = [devs[g % len(devs)] for g in range(world)]
    per, total = cfg["channels"], cfg["len"]
    kw = dict(taps=params.taps_192k(), pllinc=params.PLLINC_192K) if cfg["wide"] else {}
    node = ReceiverNode(per * world, devices=devs, max_len=total, **kw)
    slabs = []
    for g, (d, first, n) in enumerate(node.shards):
        base, _ = synth.make_base_streams(min(args.base, n), total, seed=synth.SEED + g, sps=cfg["sps"])
        with torch.cuda.device(d):
            slabs.append(tile_channels(torch.from_numpy(base).to(f"cuda:{d}"), n))
    for d in set(devs):
        torch.cuda.synchronize(d)
    for w in node.warnings():
        sys.stderr.write("bench.py: " + w + "\n")
    node.set_option("stage_mask", cfg["stage_mask"])
    node.autotune(slabs)

    def step():
        node.run(slabs)
        node.discard_frames()
    for _ in range(args.warmup):
        step()
    node.sync()
    rx0 = node.total_received()
    for d in set(devs):
        torch.cuda.synchronize(d)
    node.mark()                     # per-shard bookkeeping inside the library: a slow device must show by itself
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    node.sync()
    for d in set(devs):
        torch.cuda.synchronize(d)
    dt = time.perf_counter() - t0
    stats = node.shard_stats()
    msgs = node.total_received() - rx0
    samples = float(per) * world * total * args.steps
    value = samples / dt / 1e6
    alg = per * world * total * 2.0
    c = alg / (dt / args.steps) / 1e9
    out = {"metric": "Msamples/s demodulated (N-channel batch, " + ("full chain" if cfg["stage_mask"] == 0x1f else "FIR + receiver") + ")",
           "value": value, "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "f32 FIR on int16 samples; u32 PLL/HDLC/CRC", "data": "synthetic",
           "config": {"workload": cfg["what"], "channels_per_gpu": per, "samples_per_channel": total,
                      "parallelism": f"channels in {world} contiguous blocks over device(s) {devs}, one process, one host "
                                     "thread + batch per device (gnuais_node_*), no collectives"},
           "valid_crc_msgs_per_s": msgs / dt, "x_realtime_channels": value / (cfg["rate"] / 1e6),
           "per_gpu": [{"shard": g, "device": st["device"], "pci": st["pci"], "numa_node": st["numa_node"],
                        "host_thread_pinned_to_cpus": st["pinned_cpus"], "first_channel": st["first_channel"],
                        "channels": st["n_channels"], "calls": st["calls"],
                        "ms_per_step": st["busy_ms"] / max(1, st["calls"]),
                        "host_submit_ms_per_step": st["submit_ms"] / max(1, st["calls"])}
                       for g, st in enumerate(stats)],
           "per_gpu_what": "per shard, measured inside the library (gnuais_node_mark / gnuais_node_shard_stats): "
                           "ms_per_step = from the shard's first submission of the timed region to the end of its own sync, "
                           "over its calls; host_submit = time its host thread spent inside the run calls",
           "roofline": {"bound": "hbm", "kernel": "chain (all devices)", "achieved": c, "peak": HBM_PEAK_GBS * len(set(devs)),
                        "unit": "GB/s", "frac": c / (HBM_PEAK_GBS * len(set(devs))), "traffic": None,
                        "algorithmic_bytes_per_launch": alg,
                        "what": "N*L*2 bytes of all shards / ms_per_step against 8 TB/s per distinct device; per-kernel "
                                "figures come from the single-GPU line"}}
    if args.traffic and not args.channels and not args.len:
        # HBM bytes of one device's workload (PMC child passes on the first device) x the shards: every shard runs it
        t = pmc_traffic(args, args.config, devs[0])
        if t and not t.get("error"):
            out["roofline"]["traffic"] = t["chain_bytes_per_call"] * world
            out["roofline"]["traffic_what"] = "PMC bytes per call of ONE shard's workload (a one-GPU child run on device %d) x %d shards" % (devs[0], world)
    if args.cpu:
        # the reference's own code on this host beside it (a bounded sample of shard 0's input; one core)
        x0 = slabs[0][:, : min(args.cpu_channels, slabs[0].shape[1])].cpu().numpy()
        out["cpu_baseline"] = cpu_baseline(x0, x0.shape[1], total)
    node.close()
    out["bench_seconds"] = time.perf_counter() - T_START
    emit(out, args)


def preflight(devs, cfg):
    """Before any allocation: is every device this run names there, and has it room for its shards?  -> None or a text
    that names the device and both figures.  (Per shard: the input slab N x L x 2 bytes + the batch's hand-off sets,
    candidate ring and frame ring -- gnuais_batch_create checks its own share again, exactly.)"""
    try:
        import torch
    except Exception as e:                              # noqa: BLE001
        return "torch is not importable: %s" % e
    if not torch.cuda.is_available():
        return "no HIP device is visible (torch.cuda.is_available() is False); the chain has no CPU path"
    have = torch.cuda.device_count()
    bad = sorted(set(d for d in devs if d < 0 or d >= have))
    if bad:
        return "device index %s requested, %d device(s) visible (HIP_VISIBLE_DEVICES=%s)" % (
            bad, have, os.environ.get("HIP_VISIBLE_DEVICES", "unset"))
    per_shard = cfg["channels"] * cfg["len"] * 2.0 * 1.6 + cfg["channels"] * 90e3
    for d in sorted(set(devs)):
        try:
            free_b, total_b = torch.cuda.mem_get_info(d)
        except Exception as e:                          # noqa: BLE001
            return "device %d does not answer (mem_get_info: %s)" % (d, e)
        need = per_shard * devs.count(d)
        if need > free_b:
            return "device %d: %d shard(s) of %d channels x %d samples need about %.1f GB, %.1f GB free of %.1f" % (
                d, devs.count(d), cfg["channels"], cfg["len"], need / 1e9, free_b / 1e9, total_b / 1e9)
    return None


def fail_line(args, world, what, per_rank=None):
    """A run that cannot start still ends in ONE parseable line (value null, `error` says why) and a non-zero status."""
    cfg = CONFIGS[args.config]
    out = {"metric": "Msamples/s demodulated (N-channel batch, full chain)", "value": None, "unit": "Msamples/s",
           "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": None, "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "f32 FIR on int16 samples; u32 PLL/HDLC/CRC", "data": "synthetic",
           "config": {"workload": cfg["what"]}, "error": what}
    if per_rank:
        out["per_gpu"] = per_rank
    sys.stderr.write("bench.py: " + what + "\n")
    sys.stderr.flush()
    print(json.dumps(out, separators=(",", ":")), flush=True)
    sys.exit(3)


def _cfg_of(args):
    cfg = CONFIGS[args.config]
    if args.channels:
        cfg = dict(cfg, channels=args.channels)
    if args.len:
        cfg = dict(cfg, len=args.len)
    return cfg


def _worker(rank, world, args, barrier, queue):
    devs = [int(d) for d in args.devices.split(",")] if args.devices else list(range(world))
    rank_main(rank, devs[rank % len(devs)], world, args, LocalSync(rank, world, barrier, queue))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="C3")
    ap.add_argument("--channels", type=int, default=0, help="override: channels per GPU")
    ap.add_argument("--len", type=int, default=0, help="override: samples per channel per step")
    ap.add_argument("--base", type=int, default=256, help="distinct base streams")
    ap.add_argument("--cpu-channels", type=int, default=8192,
                    help="channels of the batch the single-core CPU baseline runs (about 13 s of CPU work)")
    ap.add_argument("--devices", default="", help="comma list: device of each worker (default 0..N-1); "
                    "repeating a device runs several workers on it")
    ap.add_argument("--procs", action="store_true",
                    help="--gpus N as N worker processes (one per device) instead of the in-process node object")
    ap.add_argument("--no-cpu", dest="cpu", action="store_false")
    ap.add_argument("--no-kernel-leg", dest="kernel_leg", action="store_false",
                    help="skip the 120-call leg behind the timed region that kernel_ms is averaged over")
    ap.add_argument("--no-traffic", dest="traffic", action="store_false",
                    help="skip the rocprofv3 --pmc child runs that fill roofline.traffic")
    ap.add_argument("--e2e", dest="e2e", action="store_true", default=True,
                    help="(the default) the message-layer legs too: message_lines, end_to_end (run + streamed NMEA text into "
                         "pinned host memory every step), float_path; the sinks' host post-stage rides with the CPU baseline")
    ap.add_argument("--no-e2e", dest="e2e", action="store_false", help="skip them (the profiler's child runs do)")
    ap.add_argument("--no-traffic-others", dest="traffic_others", action="store_false",
                    help="skip the rocprofv3 --pmc child runs for C2 and C5 (six passes; other_configs.*.traffic stays null)")
    ap.add_argument("--traffic-others", dest="traffic_others", action="store_true", help="(the default; kept for old command lines)")
    ap.set_defaults(traffic_others=True)
    ap.add_argument("--full", action="store_true", help="everything: --e2e as well")
    ap.add_argument("--detail", default="", help="where the full measurements go (default: bench_detail.json here)")
    ap.add_argument("--print-detail", action="store_true",
                    help="also print the full measurements as an EARLIER stdout line (the compact line stays last)")
    ap.add_argument("--no-others", dest="others", action="store_false",
                    help="skip the brief C2 / C5 measurements of a default single-GPU run")
    args = ap.parse_args()
    if args.full:
        args.e2e = args.traffic_others = True

    if "RANK" in os.environ and "WORLD_SIZE" in os.environ:
        # launched by torch.distributed.run: its ranks, RCCL for the barrier and the reductions
        import torch
        import torch.distributed as dist
        world, rank = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"])
        local = int(os.environ.get("LOCAL_RANK", "0"))
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        # every rank looks at its own device first and all of them learn what the others found BEFORE anything is
        # allocated or timed: a rank that cannot run says so in the final line instead of leaving the others in a barrier
        mine = preflight([local], _cfg_of(args))
        if mine is None:
            torch.cuda.set_device(local)
        # through a plain TCP store next to the launcher's (no process group yet: RCCL cannot be initialised on a rank without
        # a device, and a second group behind a destroyed one does not come up under torch.distributed.run)
        import datetime
        found = [{"rank": rank, "device": local, "error": mine}]
        try:
            store = dist.TCPStore(os.environ["MASTER_ADDR"], int(os.environ["MASTER_PORT"]) + 1, world, rank == 0,
                                  timeout=datetime.timedelta(seconds=120))
            store.set("pre%d" % rank, json.dumps(found[0]))
            found = [json.loads(store.get("pre%d" % r)) for r in range(world)]
            store.set("ack%d" % rank, "1")
            if rank == 0:
                for r in range(world):                  # the store lives in rank 0: stay until everybody has read it
                    store.get("ack%d" % r)
        except Exception as e:                          # noqa: BLE001   (a busy port must not stop a healthy node: go on with what this rank knows)
            sys.stderr.write("bench.py: rank %d: preflight exchange failed (%s); continuing on this rank's own check\n" % (rank, e))
            found = [{"rank": rank, "device": local, "error": mine}]
        if any(f.get("error") for f in found):
            if rank == 0 or (mine is not None and len(found) == 1):
                fail_line(args, world, "; ".join("rank %d (device %d): %s" % (f["rank"], f["device"], f["error"])
                                                for f in found if f.get("error")), found)
            sys.exit(3)
        dist.init_process_group("nccl", rank=rank, world_size=world)
        rank_main(rank, local, world, args, DistSync(dist, torch.device("cuda", local), rank, world))
        dist.destroy_process_group()
        return

    world = max(1, args.gpus)
    devs = [int(d) for d in args.devices.split(",")] if args.devices else list(range(world))
    devs = [devs[g % len(devs)] for g in range(world)]
    bad = preflight(devs, _cfg_of(args))
    if bad:
        fail_line(args, world, bad)
    if world == 1:
        rank_main(0, devs[0] if args.devices else 0, 1, args, LocalSync(0, 1, None, None))
        return
    if not args.procs:
        try:
            node_main(world, args)              # one process: the library's node object drives every device
        except Exception as e:                  # noqa: BLE001   (GnuaisError carries the shard's own text: "device 3 (channels ..): ...")
            fail_line(args, world, "%s: %s" % (type(e).__name__, e))
        return
    # one worker process per device (receivers share nothing, src/ais.c:141-147):
    # independent batches and launches, a barrier around the timed region, no process group
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    barrier, queue = ctx.Barrier(world), ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, args, barrier, queue)) for r in range(1, world)]
    for p in procs:
        p.start()
    try:
        _worker(0, world, args, barrier, queue)
    finally:
        for p in procs:
            p.join(timeout=600)
    if any(p.exitcode not in (0, None) for p in procs):
        sys.exit(1)


if __name__ == "__main__":
    main()
