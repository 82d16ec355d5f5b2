mkdir -p gpurun_out/r4
( time python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4/job12_bench.json 2> gpurun_out/r4/job12_bench.err ) 2> gpurun_out/r4/job12_time.txt
cat gpurun_out/r4/job12_time.txt; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r4/job12_bench.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["steady_state"]["ms_per_step"], {k: v["ms_per_step"] for k, v in d["other_configs"].items()}, d["roofline"]["bound"], d["roofline"]["valu_chain"]["bound"])
PY
( timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 )
