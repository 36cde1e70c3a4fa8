mkdir -p gpurun_out/r04k; export TMPDIR=/tmp; O=gpurun_out/r04k
run() { env $1 timeout 300 python bench.py --no-cpu-baseline --no-roofline --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$1', d['ms_per_step'], d['value'], d['config']['loss'])" >> $O/ab.txt; }
for rep in 1 2; do for e in "ALL=on" "SEGCLIP_REDUCE_SIDE=0"; do run "$e"; done; done
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|FAILED|^E " | head -20 > $O/pytest.txt
run "ALL=on"
cat $O/ab.txt $O/pytest.txt
