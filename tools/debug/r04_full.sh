mkdir -p gpurun_out/r04f; export TMPDIR=/tmp; O=gpurun_out/r04f
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|error|FAILED|assert" | head -20 > $O/pytest.txt
for i in 1 2; do
  timeout 300 python bench.py --no-cpu-baseline --no-roofline --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['value'], d['config']['loss'])" >> $O/bench_ab.txt
done
timeout 300 python bench.py --no-cpu-baseline --no-roofline --steps 20 --warmup 5 --full-loss 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('full', d['ms_per_step'], d['value'], d['config']['loss'])" >> $O/bench_ab.txt
cat $O/pytest.txt $O/bench_ab.txt
