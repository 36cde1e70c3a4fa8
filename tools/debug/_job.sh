mkdir -p gpurun_out/r04; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|Error|FAILED|^E " | head -20 > gpurun_out/r04/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -i "smoke" > gpurun_out/r04/smoke.txt
bash tools/profile_round.sh r04 > gpurun_out/r04/profile_round.log 2>&1
cat gpurun_out/r04/pytest_gpu.txt gpurun_out/r04/smoke.txt; head -12 gpurun_out/r04/profile_round.log
