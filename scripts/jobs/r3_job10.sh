mkdir -p gpurun_out/r3
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r3/bench_a.json 2> gpurun_out/r3/bench_a.err; tail -3 gpurun_out/r3/bench_a.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r3/bench_a.json"))
for k in ("value", "ms_per_step", "kernel_ms", "steady_state", "float_path", "end_to_end", "message_lines"):
    print(k, json.dumps(d.get(k))[:600])
print("roofline", json.dumps({k: v for k, v in d["roofline"].items() if k != "traffic_detail"})[:1500])
print("traffic_detail", json.dumps(d["roofline"].get("traffic_detail"))[:800])
for n, o in (d.get("other_configs") or {}).items():
    print(n, o["ms_per_step"], o["kernel_ms"], o["roofline"]["kernel"], round(o["roofline"]["frac"], 3))
print("cpu", json.dumps(d.get("cpu_baseline"))[:400])
PY
