cd /root/repo
python -m pytest tests -m gpu -x -q 2>&1 | tail -15
python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/r05_bench_b.json; wc -c gpurun_out/r05_bench_b.json; cat gpurun_out/r05_bench_b.json | cut -c1-2600
