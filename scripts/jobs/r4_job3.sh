# round 4, job 3: the bench line with the VALU roofline / exact chain / kernel leg; the node line with per-shard stats; full GPU suite
mkdir -p gpurun_out/r4
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 ) > gpurun_out/r4/job3_pytest.txt
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4/job3_bench.json 2> gpurun_out/r4/job3_bench.err
timeout 600 python bench.py --gpus 2 --devices 0,0 --steps 20 --warmup 5 > gpurun_out/r4/job3_bench_node.json 2> gpurun_out/r4/job3_bench_node.err
cat /proc/cpuinfo | grep "model name" | sort | uniq -c > gpurun_out/r4/job3_cpu.txt; ls /sys/devices/system/node/ >> gpurun_out/r4/job3_cpu.txt; nproc >> gpurun_out/r4/job3_cpu.txt
tail -3 gpurun_out/r4/job3_pytest.txt; python - <<'PY'
import json
for f in ("gpurun_out/r4/job3_bench.json","gpurun_out/r4/job3_bench_node.json"):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "unreadable", e); print(open(f.replace(".json",".err")).read()[-1500:]); continue
    print(f, d["ms_per_step"], d.get("steady_state",{}) and d["steady_state"].get("ms_per_step"))
    r=d["roofline"]; print(" roofline", {k:r.get(k) for k in ("bound","kernel","kernel_ms","frac","traffic","kernel_ms_samples","bound_why")})
    print(" valu", r.get("valu")); print(" valu_chain", r.get("valu_chain"))
    print(" kernel_ms", d.get("kernel_ms"), d.get("kernel_ms_calls"), d.get("kernel_ms_timed_region"))
    print(" exact_chain", d.get("exact_chain"))
    for k,v in (d.get("other_configs") or {}).items():
        rr=v["roofline"]; print(" ",k, v["ms_per_step"], {q:rr.get(q) for q in ("bound","kernel","frac","traffic")}, rr.get("valu"))
    print(" per_gpu", d.get("per_gpu")); print(" cpu", {k:v for k,v in (d.get("cpu_baseline") or {}).items() if k in ("value","cpu","cores","kind")})
PY
