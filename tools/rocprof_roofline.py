"""Kernel-time roofline of one bench step, measured by rocprofv3 (used by bench.py and tools/profile_round.sh).

bench.py re-runs its own workload for a few steps as a child process under `rocprofv3 --kernel-trace`, and once
more under `--pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate, counters-only passes, as MI355X_MICROARCH.md's HBM
section prescribes).  This module starts those children, reads the rocpd sqlite database / counter CSVs they leave
behind, groups the dispatches into kernel classes and turns them into per-step kernel times - the SAME numbers
`rocprofv3 --stats` prints, so the bench line can be recomputed from the summary committed under profiles/.
"""
import csv
import glob
import os
import shutil
import sqlite3
import subprocess
import sys
import tempfile

# kernel class -> substrings of the kernel name
CLASSES = [
    ("gemm_bf16", ("gemm_bf16_pq_kernel", "gemm_bf16_pq_group_kernel", "gemm_bf16_p8_kernel", "gemm_bf16_dma_kernel", "gemm_bf16_kernel")),
    ("splitk_reduce", ("splitk_reduce", "reduce_multi_slabs")),
    ("gemm_f32", ("gemm_f32_kernel",)),
    ("attn_fwd", ("attn_fwd", "attn_smallq_fwd")),
    ("attn_bwd", ("attn_bwd", "attn_smallq_bwd")),
    ("ln_fwd", ("ln_fwd_kernel", "ln_fwd_multi_kernel")),
    ("ln_bwd", ("ln_bwd_kernel", "ln_bwd_multi_kernel")),
    ("row_reductions", ("reduce_rows_kernel", "reduce_multi_rows", "colsum_partial_kernel")),
    ("startup_probe", ("spin_kernel",)),   # torch.cuda._sleep of segclip_amd/streams.py: once per process, not per step
]


def classify(name):
    for cls, pats in CLASSES:
        if any(p in name for p in pats):
            return cls
    return "other"


def rocprofv3_path():
    return shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)


def run_child(child_argv, outdir, pmc=None, timeout=300):
    """Run `python <child_argv>` under rocprofv3.  pmc=None: kernel trace (rocpd database); pmc='FETCH_SIZE': a
    counters-only pass (CSV).  Returns (returncode, stdout, stderr-tail)."""
    exe = rocprofv3_path()
    if exe is None:
        raise RuntimeError("rocprofv3 not found")
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "GROUP_RANK", "MASTER_ADDR", "MASTER_PORT",
              "TORCHELASTIC_RUN_ID", "TORCHELASTIC_RESTART_COUNT", "TORCHELASTIC_MAX_RESTARTS"):
        env.pop(k, None)
    cmd = [exe]
    cmd += ["--pmc", pmc, "--output-format", "csv"] if pmc else ["--kernel-trace", "--stats"]
    cmd += ["-d", outdir, "-o", "rl", "--", sys.executable] + list(child_argv)
    pr = subprocess.run(cmd, env=env, cwd="/tmp", stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout, text=True)
    return pr.returncode, pr.stdout, pr.stderr[-2000:]


def find_db(outdir):
    dbs = sorted(glob.glob(os.path.join(outdir, "**", "*.db"), recursive=True), key=os.path.getsize)
    return dbs[-1] if dbs else None


def kernel_table(db):
    """[(name, calls, total_us, avg_us)] sorted by total time (= rocprofv3 --stats' kernel table)."""
    c = sqlite3.connect(db)
    rows = list(c.execute("select name, total_calls, total_duration, average from top_kernels"))
    c.close()
    return [(n, int(k), float(t), float(a)) for n, k, t, a in rows]


def per_class(table, n_passes):
    """{class: {launches_per_step, time_per_step_ms, avg_launch_us}} from a kernel table that covers n_passes model passes."""
    out = {}
    for name, calls, total_us, _ in table:
        d = out.setdefault(classify(name), {"calls": 0, "total_us": 0.0})
        d["calls"] += calls
        d["total_us"] += total_us
    res = {}
    for cls, d in out.items():
        res[cls] = {"launches_per_step": round(d["calls"] / n_passes, 1),
                    "time_per_step_ms": round(d["total_us"] / n_passes / 1e3, 3),
                    "avg_launch_us": round(d["total_us"] / max(d["calls"], 1), 2)}
    return res


def _merge(iv):
    """merged (start, end) list of a start-sorted interval list"""
    out = []
    cur_s, cur_e = iv[0]
    for s_, e_ in iv[1:]:
        if s_ > cur_e:
            out.append((cur_s, cur_e))
            cur_s, cur_e = s_, e_
        else:
            cur_e = max(cur_e, e_)
    out.append((cur_s, cur_e))
    return out


def class_intervals(db):
    """{class: merged [(start_ns, end_ns)] of its kernel intervals over all streams}, plus "__all__" (any kernel) - what the
    union figures are computed from (kept as profiles/rNN_bench_intervals.json so that they can be recomputed)."""
    c = sqlite3.connect(db)
    rows = list(c.execute("select start, end, name from kernels order by start"))
    c.close()
    by = {}
    for s_, e_, name in rows:
        cls = classify(name)
        by.setdefault(cls, []).append((s_, e_))
        if cls != "startup_probe":
            by.setdefault("__all__", []).append((s_, e_))
    return {cls: _merge(iv) for cls, iv in by.items()}


def class_union_ms(db, n_passes, merged=None):
    """{class: ms per step during which AT LEAST ONE kernel of the class is running} - the union of the class's kernel
    intervals over all streams.  With the two towers on concurrent streams two GEMMs share the chip and each one's duration
    stretches: the sum of durations double-counts the machine, the union is the time the device spends on the class."""
    merged = merged if merged is not None else class_intervals(db)
    return {cls: round(sum(e_ - s_ for s_, e_ in iv) / 1e6 / n_passes, 3) for cls, iv in merged.items()}


def format_table(table, n_passes, header="", top=60):
    tot = sum(t for _, _, t, _ in table)
    lines = [header.rstrip()] if header else []
    lines.append(f"{'kernel':90s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>9s} {'%':>6s} {'ms/step':>8s}")
    for name, calls, total, avg in table[:top]:
        lines.append(f"{name[:90]:90s} {calls:6d} {total / 1e3:10.3f} {avg:9.1f} {100 * total / tot:6.2f} {total / 1e3 / n_passes:8.3f}")
    lines.append(f"{'TOTAL (' + str(n_passes) + ' model passes)':90s} {sum(c for _, c, _, _ in table):6d} {tot / 1e3:10.3f} {'':9s} {'':6s} {tot / 1e3 / n_passes:8.3f}")
    return "\n".join(lines) + "\n"


def pmc_sum(outdir, counter, pats):
    """(sum of `counter` over the dispatches whose kernel name contains one of pats, number of such dispatches)"""
    s, n = 0.0, 0
    for path in glob.glob(os.path.join(outdir, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] == counter and any(p in r["Kernel_Name"] for p in pats):
                s += float(r["Counter_Value"])
                n += 1
    return s, n


def measure(child_argv, n_passes, pmc_argv=None, keep_dir=None, timeout=300):
    """Kernel-trace child running `child_argv` (n_passes model passes) + two PMC children running `pmc_argv` (None: no
    traffic measurement).  Returns dict(classes=..., table=..., traffic=..., notes=[...])."""
    notes = []
    base = os.path.abspath(keep_dir) if keep_dir else tempfile.mkdtemp(prefix="segclip_rl_")   # the children run in /tmp
    os.makedirs(base, exist_ok=True)
    tdir = os.path.join(base, "trace")
    rc, out, err = run_child(child_argv, tdir, None, timeout)
    db = find_db(tdir)
    if rc != 0 or db is None:
        raise RuntimeError(f"rocprofv3 kernel-trace child failed (rc={rc}): {err[-400:]}")
    table = kernel_table(db)
    res = {"classes": per_class(table, n_passes), "table": table, "child_stdout": out, "traffic": None, "notes": notes}
    try:
        merged = class_intervals(db)
        for cls, ms in class_union_ms(db, n_passes, merged).items():
            if cls in res["classes"]:
                res["classes"][cls]["union_ms_per_step"] = ms
        res["union_all_ms_per_step"] = class_union_ms(db, n_passes, merged).get("__all__")
        if keep_dir:   # merged intervals per class, microseconds from the first kernel of the trace: the union is recomputable
            import json
            t0 = min(iv[0][0] for iv in merged.values())
            with open(os.path.join(base, "intervals.json"), "w") as f:
                json.dump({"n_passes": n_passes, "unit": "us from the first kernel of the trace",
                           "note": "per kernel class: merged [start, end] intervals during which at least one kernel of the class "
                                   "was running (all streams); union_ms_per_step = sum(end - start) / n_passes / 1000; "
                                   "__all__ = any kernel except the stream-overlap probe",
                           "classes": {cls: [[round((s_ - t0) / 1e3, 2), round((e_ - t0) / 1e3, 2)] for s_, e_ in iv]
                                       for cls, iv in merged.items()}}, f)
    except Exception as e:
        notes.append(f"union busy time not computed: {e}")
    if pmc_argv:
        try:
            pats = dict(CLASSES)["gemm_bf16"]
            vals = {}
            for counter in ("FETCH_SIZE", "WRITE_SIZE"):
                pdir = os.path.join(base, "pmc_" + counter)
                rc, _, err = run_child(pmc_argv, pdir, counter, timeout)
                vals[counter] = pmc_sum(pdir, counter, pats)
                if rc != 0 or vals[counter][1] == 0:
                    raise RuntimeError(f"--pmc {counter} child: rc={rc}, {vals[counter][1]} GEMM dispatches: {err[-300:]}")
            (f, nf), (w, nw) = vals["FETCH_SIZE"], vals["WRITE_SIZE"]
            # KiB units; FETCH_SIZE doubled: gfx950 reports half of the bytes of wide coalesced streaming reads
            res["traffic"] = {"hbm_bytes_per_launch": round((2 * f * 1024) / nf + (w * 1024) / nw),
                              "fetch_bytes_per_launch_corrected": round(2 * f * 1024 / nf),
                              "write_bytes_per_launch": round(w * 1024 / nw), "launches": nf,
                              "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, two separate counters-only child "
                                        "runs of this command; FETCH_SIZE x2 per MI355X_MICROARCH.md (gfx950)"}
        except Exception as e:  # PMC passes are best effort: the trace numbers stand on their own
            notes.append(f"traffic not measured: {e}")
    if keep_dir is None:
        shutil.rmtree(base, ignore_errors=True)
    return res
