#!/bin/bash
# tools/final_round.sh <tag>  (run on the GPU box): everything the round's docs quote - GPU test suite, rocprofv3 profile +
# PMC passes (tools/profile_round.sh), the bench configurations, and the micro-benchmark tables.  Results: gpurun_out/<tag>/
TAG=${1:-r02}; OUT=gpurun_out/$TAG; mkdir -p $OUT
F='^RCCL\|^HIP\|^ROCm\|Hostname\|Librccl\|amdgpu.ids\|socket.cpp\|ProcessGroupNCCL'
timeout 1800 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log
bash tools/profile_round.sh $TAG > $OUT/profile_round.log 2>&1; tail -4 $OUT/gemm_traffic.log
b() { local name=$1; shift; timeout 900 python bench.py "$@" > $OUT/bench_$name.json 2> $OUT/bench_$name.err; echo "$name: $(cut -c1-60 $OUT/bench_$name.json | head -1) ... $(grep -o '"ms_per_step": [0-9.]*' $OUT/bench_$name.json) $(grep -o '"value": [0-9.]*' $OUT/bench_$name.json | head -1)"; }
b default
b dist --no-cpu-baseline --force-dist
b dist_ddp --no-cpu-baseline --force-dist --grad-sync ddp
b gb2048 --no-cpu-baseline --global-batch 2048 --steps 5 --warmup 2
b vitl14 --no-cpu-baseline --spec vitl14_336 --batch 128 --steps 5 --warmup 2
b vitl14_nofp8 --no-cpu-baseline --spec vitl14_336 --batch 128 --steps 5 --warmup 2 --attn-fp8 off
b full_loss --no-cpu-baseline --full-loss
timeout 300 python tools/bench_hbm.py 2>&1 | grep -v "$F" > $OUT/hbm_kernels.txt
timeout 300 python tools/bench_gemm.py 2>&1 | grep -v "$F" > $OUT/gemm_shapes.txt
timeout 300 python tools/bench_gemm.py 19712 text 2>&1 | grep -v "$F" >> $OUT/gemm_shapes.txt
timeout 200 python tools/bench_attn.py 2>&1 | grep -v "$F" > $OUT/attn.txt
# main-loop ablations: only meaningful with a library built with `build.sh -DSEGCLIP_P8_ABLATIONS`
if nm -D segclip_amd/libsegclip_hip.so | grep -q segclip_p8_launch_abl1; then
  for a in 0 1 2 3 4; do SEGCLIP_P8_ABL=$a timeout 200 python tools/bench_gemm_abl.py 2>&1 | grep "ABL="; done > $OUT/gemm_ablation.txt
else
  echo "library built without -DSEGCLIP_P8_ABLATIONS: ablations skipped" > $OUT/gemm_ablation.txt
fi
timeout 300 python tools/bench_train_tail.py 2>&1 | grep -v "$F" > $OUT/train_tail.txt
timeout 300 python tools/debug/gradsync_cost.py 2>&1 | grep "ms/step" > $OUT/gradsync_cost.txt
bash tools/pmc_attn.sh $OUT/pmc_attn > $OUT/pmc_attn.log 2>&1
export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_dist -o bench -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --force-dist > $OUT/bench_dist_prof.json 2> $OUT/bench_dist_prof.err
python tools/prof_summary.py $(ls $OUT/prof_dist/*.db | head -1) 40 > $OUT/kernel_stats_dist.txt 2>&1
grep -i "nccl\|rccl\|cast\|AllReduce" $OUT/kernel_stats_dist.txt | cut -c1-150
cat $OUT/hbm_kernels.txt $OUT/attn.txt $OUT/gemm_ablation.txt $OUT/train_tail.txt $OUT/gradsync_cost.txt
