"""Where does each stream's time go?  From a rocprofv3 kernel trace (rocpd database) of the bench child: per stream, the
kernels of the last N passes grouped by a short kernel label, with launches and kernel time per pass - the busiest stream
(the vision tower + everything serial) is the critical path, so its table IS the step.
usage: python tools/stream_breakdown.py <db> <n_passes_in_trace> [passes_to_use]"""
import collections, re, sqlite3, sys

db = sys.argv[1]
npass = int(sys.argv[2])
use = int(sys.argv[3]) if len(sys.argv) > 3 else max(1, npass - 2)
c = sqlite3.connect(db)
rows = [r for r in c.execute("select stream_id, start, end, name from kernels order by start") if "spin_kernel" not in r[3]]
t0, t1 = rows[0][1], rows[-1][2]
w0 = t1 - (t1 - t0) * use / npass


def label(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.match(r"([A-Za-z0-9_:]+)(<[^(]*>)?\(", n)
    if m:
        base, targs = m.group(1), (m.group(2) or "")
        if base.startswith("at::native") or base.startswith("at::cuda"):
            k = re.search(r"at::native::(?:\(anonymous namespace\)::)?([A-Za-z_0-9]+)", n[len(base):])
            return "aten:" + (k.group(1) if k else base.split("::")[-1])
        return base + targs[:24]
    return n[:40]


by = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0]))
for sid, s, e, name in rows:
    if s < w0:
        continue
    a = by[sid][label(name)]
    a[0] += 1
    a[1] += e - s
for sid, tab in sorted(by.items(), key=lambda kv: -sum(v[1] for v in kv[1].values())):
    tot = sum(v[1] for v in tab.values())
    print(f"=== stream {sid}: {sum(v[0] for v in tab.values()) / use:.0f} launches, {tot / use / 1e6:.3f} ms of kernel time per pass")
    for k, (n, t) in sorted(tab.items(), key=lambda kv: -kv[1][1])[:45]:
        print(f"  {t / use / 1e3:9.1f} us  {n / use:6.1f}x  {t / n / 1e3:8.1f} avg  {k}")
