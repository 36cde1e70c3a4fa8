#!/bin/bash
# tools/profile_round.sh <tag>   (run on the GPU box; results under gpurun_out/<tag>/)
# 1. the default bench command, with its own rocprofv3 children kept: bench line + kernel table (the SAME run) + live PMC traffic
# 2. the bench configurations the docs quote (B=2048 strong-scaling base, full loss, --force-dist, ViT-L/14)
# 3. PMC passes of single GEMM shapes (MFMA busy, LDS conflicts, L2 hit rate) for the three operand layouts, and of the
#    attention kernels
TAG=${1:-r03}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
F='^RCCL\|^HIP\|^ROCm\|Hostname\|Librccl\|amdgpu.ids\|socket.cpp\|ProcessGroupNCCL'
SEGCLIP_BENCH_PROFILE_DIR=$OUT/rl timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench_line.json 2> $OUT/bench_line.err
cp $OUT/rl/kernel_stats.txt $OUT/kernel_stats.txt
cp $OUT/rl/intervals.json $OUT/bench_intervals.json 2>/dev/null
DB=$(ls $OUT/rl/trace/*.db $OUT/rl/trace/*/*.db 2>/dev/null | head -1)
python tools/stream_gaps.py $DB 120 > $OUT/stream_gaps.txt 2>&1
python tools/stream_breakdown.py $DB 6 3 > $OUT/stream_breakdown.txt 2>&1
rm -rf $OUT/rl/trace $OUT/rl/pmc_*
b() { local name=$1; shift; timeout 900 python bench.py "$@" > $OUT/bench_$name.json 2> $OUT/bench_$name.err; echo "$name: $(grep -o '"ms_per_step": [0-9.]*' $OUT/bench_$name.json) $(grep -o '"value": [0-9.]*' $OUT/bench_$name.json | head -1)"; }
b gb2048 --no-cpu-baseline --no-traffic --no-parity-leg --global-batch 2048 --steps 5 --warmup 2
b full_loss --no-cpu-baseline --no-traffic --no-parity-leg --full-loss
b resid_bf16 --no-cpu-baseline --no-roofline --no-parity-leg --resid bf16
b plain_ref --no-cpu-baseline --no-roofline --no-parity-leg
b dist --no-cpu-baseline --no-roofline --no-parity-leg --force-dist
b dist_bf16wire --no-cpu-baseline --no-roofline --no-parity-leg --force-dist --wire bf16
b plain_ref2 --no-cpu-baseline --no-roofline --no-parity-leg
b vitl14 --no-cpu-baseline --no-traffic --no-parity-leg --spec vitl14_336 --batch 128 --steps 5 --warmup 2 --attn-fp8 off
b text_trim --no-cpu-baseline --no-roofline --no-parity-leg --text-trim
for bsz in 64 96 128 192 512; do b b$bsz --no-cpu-baseline --no-roofline --no-parity-leg --batch $bsz; done
SEGCLIP_TUNING=1 SEGCLIP_PAD_ROWS=0 b b96_pad_off --no-cpu-baseline --no-roofline --no-parity-leg --batch 96
# same-box A/B of the grouped weight gradients (config.wgrad_group_blocks): one launch per gradient vs the default, twice
# (library kernel-selection switches need SEGCLIP_TUNING=1)
# same-box A/Bs, twice each: the half-tile tail of gemm_bf16_pq.hip (SEGCLIP_PQ_HALF=0 = full tiles only), B = 256 and B = 128
# (at B = 128 the switch also decides whether M = 256 q + 128 runs on that kernel at all); gradient folding on the full loss
for rep in a b; do
  SEGCLIP_TUNING=1 SEGCLIP_PQ_HALF=0 b half_off_$rep --no-cpu-baseline --no-roofline --no-parity-leg --steps 30 --warmup 8
  b half_on_$rep --no-cpu-baseline --no-roofline --no-parity-leg --steps 30 --warmup 8
  SEGCLIP_TUNING=1 SEGCLIP_PQ_HALF=0 b b128_half_off_$rep --no-cpu-baseline --no-roofline --no-parity-leg --batch 128 --steps 30 --warmup 8
  b b128_half_on_$rep --no-cpu-baseline --no-roofline --no-parity-leg --batch 128 --steps 30 --warmup 8
  SEGCLIP_TUNING=1 SEGCLIP_FOLD_GRADS=0 b full_fold_off_$rep --no-cpu-baseline --no-roofline --no-parity-leg --full-loss
  b full_fold_on_$rep --no-cpu-baseline --no-roofline --no-parity-leg --full-loss
done
timeout 400 python tools/bench_pq_half.py 2>&1 | grep -v "$F" | grep -v "^check\|relerr" > $OUT/gemm_half_tile.txt
timeout 300 python tools/debug/center_stage_profile.py 2>&1 | grep -v "$F" | grep -v "Warning\|warn" > $OUT/center_stage.txt
timeout 300 python tools/bench_hbm.py 2>&1 | grep -v "$F" > $OUT/hbm_kernels.txt
timeout 300 python tools/bench_gemm.py 2>&1 | grep -v "$F" > $OUT/gemm_shapes.txt
timeout 300 python tools/bench_gemm.py 19712 text 2>&1 | grep -v "$F" >> $OUT/gemm_shapes.txt
timeout 300 python tools/bench_epi.py 2>&1 | grep -v "$F" > $OUT/gemm_epilogues.txt
timeout 200 python tools/bench_attn.py 2>&1 | grep -v "$F" > $OUT/attn.txt
timeout 300 python tools/bench_pq.py 2>&1 | grep -v "$F" | grep -v "^check" > $OUT/gemm_pq.txt
timeout 300 python tools/bench_pq.py 19712 2>&1 | grep -v "$F" | grep -v "^check" >> $OUT/gemm_pq.txt
timeout 600 python tools/bench_eager.py --classes 2>/dev/null | grep "^{" > $OUT/eager_ab.json
for spec in "50176 768 3072 nt" "50176 3072 768 dgrad" "50176 2304 768 wgrad"; do
  set -- $spec
  timeout 600 bash tools/pmc_gemm.sh $1 $2 $3 $4 $OUT/pmc_gemm_$4 > $OUT/pmc_gemm_$4.txt 2>&1
done
timeout 600 bash tools/pmc_attn.sh $OUT/pmc_attn 256 196 12 0 > $OUT/pmc_attn.txt 2>&1
timeout 300 python tools/bucket_timeline.py $OUT/bucket_timeline.txt > $OUT/bucket_timeline.log 2>&1
timeout 600 python tools/accuracy_b256.py > $OUT/accuracy_b256.txt 2>&1
rm -rf $OUT/pmc_gemm_*/*/ $OUT/pmc_attn/*/ 2>/dev/null
head -30 $OUT/kernel_stats.txt | cut -c1-150; cat $OUT/stream_gaps.txt | head -5; tail -c 600 $OUT/bench_line.json
