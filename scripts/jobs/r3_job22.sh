mkdir -p gpurun_out/r3
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
TABLE=192k timeout 300 python scripts/fuzz_parity.py 60 2>&1 | tail -1
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r3/bench_b.json 2> gpurun_out/r3/bench_b.err; tail -2 gpurun_out/r3/bench_b.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r3/bench_b.json"))
print("C3", d["ms_per_step"], d["kernel_ms"], d["steady_state"]["ms_per_step"], d["roofline"]["kernel"], round(d["roofline"]["frac"],3), round(d["roofline"]["chain"]["frac"],3))
for n, o in (d.get("other_configs") or {}).items():
    print(n, o["ms_per_step"], o["kernel_ms"], o["roofline"]["kernel"], round(o["roofline"]["frac"], 3))
PY
