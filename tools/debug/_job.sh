OUT=gpurun_out/r04w/step; mkdir -p $OUT
export TMPDIR=/tmp
SEGCLIP_BENCH_PROFILE_DIR=$OUT/rl timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic > $OUT/bench_line.json 2> $OUT/bench_line.err
DB=$(ls $OUT/rl/trace/*.db $OUT/rl/trace/*/*.db 2>/dev/null | head -1)
python tools/stream_gaps.py $DB 45 > $OUT/stream_gaps.txt 2>&1
MS=$(grep -o "of [0-9.]* ms" $OUT/stream_gaps.txt | head -1 | grep -o "[0-9.]*")
python tools/debug/stream_classes.py $DB $MS 30 > $OUT/stream_classes.txt 2>&1
rm -rf $OUT/rl/trace $OUT/rl/pmc_*
