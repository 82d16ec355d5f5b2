timeout 1200 python -m pytest tests/test_hip_fullsize.py -m gpu -x -q -k "node or bench_two or shards" 2>&1 | tail -8
timeout 600 python bench.py --gpus 2 --devices 0,0 --steps 20 --warmup 5 --no-cpu 2>&1 | tail -1 | cut -c1-700
