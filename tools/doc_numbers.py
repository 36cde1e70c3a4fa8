"""Rewrite the NUMBER cells (pairs/s, ms/step) of the current round's result tables in README.md / DESIGN.md from
profiles/<tag>_bench_line.json and <tag>_bench_configs.json (run after tools/profiles_from_round.py).  Only the second
and third cell of a row are touched; the notes column and all prose are edited by hand.  A row is found by its first
cell (the current round's table comes first in both files, so the first match is the right one)."""
import json, os, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
L = json.load(open(f"{ROOT}/profiles/{tag}_bench_line.json")); C = json.load(open(f"{ROOT}/profiles/{tag}_bench_configs.json"))
cb = L["cpu_baseline"]
v = lambda k: round(C[k]["value"]); ms = lambda k, n=1: f"{C[k]['ms_per_step']:.{n}f}"


def sub(path, rows):
    s = open(path).read()
    for first_cell, pairs, msstep in rows:
        key = "\n| " + first_cell + " |"
        if key not in s:
            print(f"{os.path.basename(path)}: row not found: {first_cell[:60]}")
            continue
        i = s.index(key) + len(key)
        j = s.index("|", i)            # end of the pairs/s cell
        k = s.index("|", j + 1)        # end of the ms/step cell
        s = s[:i] + f" {pairs} | {msstep} " + s[k:]
    open(path, "w").write(s)


main = (f"**{round(L['value'])}**", f"{L['ms_per_step']:.2f}")
sub(f"{ROOT}/README.md", [
    ("contrastive only (BASELINE configs[1])", *main),
    ("same, global batch 2048 on one GPU (`--global-batch 2048`)", v("gb2048"), ms("gb2048")),
    ("full SegCLIP loss (configs[3])", v("full_loss"), ms("full_loss", 2)),
    ("bf16 residual stream (`--resid bf16`, opt-in)", v("resid_bf16"), ms("resid_bf16", 2)),
    ("through the N>1 code path on one rank (RCCL group + GradSync, fp32 wire; `--force-dist`)", v("dist"), ms("dist", 2)),
    ("ViT-L/14 336^2, B=128 (configs[4], `--spec vitl14_336`; `--attn-fp8 auto` = off)", f"{v('vitl14')} bf16 / {v('vitl14_fp8')} fp8 attention", f"{ms('vitl14')} / {ms('vitl14_fp8')}"),
    ("per-GPU batch 64 / 128 / 512", f"{v('b64')} / {v('b128')} / {v('b512')}", f"{ms('b64')} / {ms('b128')} / {ms('b512')}"),
    ("reference CPU path (oracle, B=4, the box's 16 usable cores)", f"{cb['value']:.1f}", f"{4000 / cb['value']:.0f}"),
])
sub(f"{ROOT}/DESIGN.md", [
    ("contrastive only, B=256 (BASELINE configs[1])", *main),
    ("bf16 residual stream (`--resid bf16`, opt-in, §2)", v("resid_bf16"), ms("resid_bf16", 2)),
    ("same through the N>1 path, 1-rank RCCL group, GradSync fp32 wire (`--force-dist`)", v("dist"), ms("dist", 2)),
    ("contrastive only, global batch 2048 on one GPU (`--global-batch 2048`, SURVEY §8d strong-scaling base)", v("gb2048"), ms("gb2048")),
    ("full SegCLIP loss (configs[3], `--full-loss`)", v("full_loss"), ms("full_loss", 2)),
    ("ViT-L/14 336², B=128 (configs[4], `--spec vitl14_336`; `--attn-fp8 auto` = off)",
     f"{v('vitl14')} (bf16 attention) / {v('vitl14_fp8')} (fp8 forward, `--attn-fp8 on`)", f"{ms('vitl14')} / {ms('vitl14_fp8')}"),
    ("per-GPU batch 64 / 128 / 512", f"{v('b64')} / {v('b128')} / {v('b512')}", f"{ms('b64')} / {ms('b128')} / {ms('b512')}"),
])
print("number cells rewritten from profiles/%s_*; notes and prose are hand-edited" % tag)
