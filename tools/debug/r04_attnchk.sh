mkdir -p gpurun_out/r04h
for l in 0 1; do echo "LEAN=$l"; SEGCLIP_ATTN_FWD_LEAN=$l python tools/debug/attn_fwd_check.py 2>&1 | grep "T="; done | tee gpurun_out/r04h/attnchk.txt
