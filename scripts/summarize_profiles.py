"""Turn what scripts/collect_profiles.sh left under gpurun_out/<tag> into the tracked summaries
under profiles/ (per round: r02_*).

  python scripts/summarize_profiles.py [gpurun_out/r02] [r02] [output dir, default profiles/]

PMC units / corrections as MI355X_MICROARCH.md's HBM section prescribes: FETCH_SIZE and
WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half the bytes of a coalesced stream, so
it is doubled (cross-check: 2 x FETCH of the FIR = the 1.573 GB of samples + the segment halo).
"""
import csv, json, os, shutil, sys
from collections import defaultdict

src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r02"
tag = sys.argv[2] if len(sys.argv) > 2 else "r02"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = sys.argv[3] if len(sys.argv) > 3 else os.path.join(root, "profiles")      # the GPU box writes next to the raw files
os.makedirs(out, exist_ok=True)

SHORT = [("fir_sign_pk", "fir_slice"), ("fir_sign_mfma", "fir_slice"), ("fir_sign_kernel", "fir_slice"), ("fir_slice_kernel", "fir_slice"), ("pll_h3_kernel", "pll"), ("pll_tp_kernel", "pll"),
         ("hdlc_events_kernel", "hdlc_deframe"), ("hdlc_deframe_kernel", "hdlc_deframe"),
         ("hdlc_crc_kernel", "hdlc_crc")]


def short(name):
    for pat, s in SHORT:
        if pat in name:
            return s
    return None


def per_launch(path, n_ch=16384):
    """mean counter value per launch, bench-sized launches only (grid of the C3 workload)"""
    acc = defaultdict(lambda: defaultdict(lambda: defaultdict(list)))
    with open(path) as f:
        for r in csv.DictReader(f):
            s = short(r["Kernel_Name"])
            if s is None:
                continue
            acc[s][r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    # the three isolated + warm-up launches are the same size as the timed ones: plain mean per distinct kernel; a stage
    # that is two kernels per call (C5's FIR: the call's first segment, the rest) is their sum
    out = {}
    for k, per in acc.items():
        out[k] = defaultdict(float)
        for d in per.values():
            for c, v in d.items():
                out[k][c] += sum(v) / len(v)
        out[k] = dict(out[k])
    return out


def find(rel):
    """rocprofv3 puts its files under <dir>/<host>/...: take the first match"""
    base = os.path.join(src, os.path.dirname(rel))
    for dp, _, fs in os.walk(base):
        for f in fs:
            if f.endswith(os.path.basename(rel)):
                return os.path.join(dp, f)
    return os.path.join(src, rel)


for name, dst in (("stats/bench_kernel_stats.csv", f"{tag}_bench_kernel_stats_pipelined.csv"),
                  ("stats_nopipe/bench_kernel_stats.csv", f"{tag}_bench_kernel_stats_sequential.csv"),
                  ("bench.json", f"{tag}_bench.json"),
                  ("bench_under_rocprof.json", f"{tag}_bench_under_rocprof.json"),
                  ("bench_nopipe.json", f"{tag}_bench_sequential.json"),
                  ("c5_stats/bench_kernel_stats.csv", f"{tag}_c5_kernel_stats.csv"),
                  ("c5_bench_under_rocprof.json", f"{tag}_c5_bench_under_rocprof.json"),
                  ("c2_stats/bench_kernel_stats.csv", f"{tag}_c2_kernel_stats.csv"),
                  ("c2_bench_under_rocprof.json", f"{tag}_c2_bench_under_rocprof.json")):
    p = os.path.join(src, name)
    if not os.path.exists(p) and name.endswith(".csv"):
        p = find(name)
    if os.path.exists(p):
        shutil.copy(p, os.path.join(out, dst))

def traffic_of(prefix, cmd):
    fetch = per_launch(find(prefix + "pmc_fetch/pmc_counter_collection.csv"))
    write = per_launch(find(prefix + "pmc_write/pmc_counter_collection.csv"))
    t = {"command": cmd,
         "units": "FETCH_SIZE/WRITE_SIZE are reported in KiB; FETCH_SIZE is doubled (gfx950 reports half the "
                  "bytes of a coalesced stream, MI355X_MICROARCH.md HBM section)",
         "raw_kib_per_launch": {}, "bytes_per_launch": {}}
    for k in fetch:
        f = fetch[k].get("FETCH_SIZE", 0.0)
        w = write.get(k, {}).get("WRITE_SIZE", 0.0)
        t["raw_kib_per_launch"][k] = {"FETCH_SIZE": f, "WRITE_SIZE": w}
        t["bytes_per_launch"][k] = (2.0 * f + w) * 1024.0
    t["chain_bytes_per_call"] = sum(t["bytes_per_launch"].values())
    return t


c5 = traffic_of("c5_", "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE (separate passes) -- python bench.py "
                "--config C5 --no-cpu --steps 4 --warmup 1")
c5["algorithmic_bytes_per_call"] = 16384 * 192000 * 2.0
with open(os.path.join(out, f"{tag}_c5_pmc_traffic.json"), "w") as f:
    json.dump(c5, f, indent=1)

fetch = per_launch(find("pmc_fetch/pmc_counter_collection.csv"))
write = per_launch(find("pmc_write/pmc_counter_collection.csv"))
traffic = {"command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE (separate passes) -- "
                      "python bench.py --no-cpu --no-others --steps 6 --warmup 1",
           "units": "FETCH_SIZE/WRITE_SIZE are reported in KiB; FETCH_SIZE is doubled (gfx950 reports half the "
                    "bytes of a coalesced stream, MI355X_MICROARCH.md HBM section)",
           "raw_kib_per_launch": {}, "bytes_per_launch": {}}
for k in fetch:
    f = fetch[k].get("FETCH_SIZE", 0.0)
    w = write.get(k, {}).get("WRITE_SIZE", 0.0)
    traffic["raw_kib_per_launch"][k] = {"FETCH_SIZE": f, "WRITE_SIZE": w}
    traffic["bytes_per_launch"][k] = (2.0 * f + w) * 1024.0
traffic["chain_bytes_per_call"] = sum(traffic["bytes_per_launch"].values())
traffic["algorithmic_bytes_per_call"] = 16384 * 48000 * 2.0
with open(os.path.join(out, f"{tag}_pmc_traffic.json"), "w") as f:
    json.dump(traffic, f, indent=1)

sq_path = find("pmc_sq/pmc_counter_collection.csv")
if os.path.exists(sq_path):
    sq = per_launch(sq_path)
    for k, d in sq.items():
        if d.get("SQ_WAVE_CYCLES"):
            d["valu_issue_share_of_wave_cycles"] = d.get("SQ_ACTIVE_INST_VALU", 0.0) / d["SQ_WAVE_CYCLES"]
    with open(os.path.join(out, f"{tag}_pmc_sq_counters.json"), "w") as f:
        json.dump({"command": "rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD "
                              "SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU -- python bench.py "
                              "--no-cpu --steps 6 --warmup 1", "per_launch_mean": sq}, f, indent=1)
print(json.dumps(traffic["bytes_per_launch"], indent=1))
