set -x
mkdir -p gpurun_out/r2n
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r2n/pytest.log 2>&1; echo "pytest rc=$?" 
tail -5 gpurun_out/r2n/pytest.log
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/r2n/bench1.json 2> gpurun_out/r2n/bench1.err; tail -1 gpurun_out/r2n/bench1.json | cut -c1-400
timeout 300 python bench.py --no-cpu-baseline --force-dist > gpurun_out/r2n/bench_dist.json 2> gpurun_out/r2n/bench_dist.err; tail -1 gpurun_out/r2n/bench_dist.json | cut -c1-400
timeout 300 python tools/debug/gradsync_cost.py > gpurun_out/r2n/gs_cost.log 2>&1; tail -12 gpurun_out/r2n/gs_cost.log
timeout 300 python tools/bench_gemm.py 2>&1 | grep -v "f32 A\|colsum" 
timeout 200 python tools/bench_attn.py 2>&1 | tail -6
