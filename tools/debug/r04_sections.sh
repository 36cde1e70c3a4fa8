mkdir -p gpurun_out/r04e; export TMPDIR=/tmp; O=gpurun_out/r04e
F='^RCCL\|^HIP\|^ROCm\|Hostname\|Librccl\|amdgpu.ids\|socket.cpp\|ProcessGroupNCCL'
cd /tmp && rocprofv3 --kernel-trace -d /tmp/trc -o trc -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-roofline --steps 3 --warmup 3 > /tmp/trc.log 2>&1; cd $GRAFT_REPO_ROOT
DB=$(ls /tmp/trc/*.db /tmp/trc/*/*.db 2>/dev/null | head -1)
python tools/small_kernel_sections.py $DB 45 30 full > $O/sections.txt 2>&1
python tools/stream_gaps.py $DB 45 > $O/gaps.txt 2>&1
timeout 300 python tools/gemm_breakdown.py 2>&1 | grep -v "$F" > $O/gemm_breakdown.txt
timeout 600 python tools/bench_eager.py --classes 2>&1 | grep -v "$F" > $O/eager.txt
head -60 $O/sections.txt; cat $O/gemm_breakdown.txt; tail -3 $O/eager.txt
