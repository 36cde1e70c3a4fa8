mkdir -p gpurun_out/r04
b() { local name=$1; shift; timeout 900 python bench.py "$@" > gpurun_out/r04/bench_$name.json 2> gpurun_out/r04/bench_$name.err; echo "$name: $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/r04/bench_$name.json)"; }
b plain_ref --no-cpu-baseline --no-roofline
b dist --no-cpu-baseline --no-roofline --force-dist
b dist_bf16wire --no-cpu-baseline --no-roofline --force-dist --wire bf16
b plain_ref2 --no-cpu-baseline --no-roofline
timeout 1200 python -m pytest tests/test_dist_gpu.py tests/test_train_gpu.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|error" | tail -3
