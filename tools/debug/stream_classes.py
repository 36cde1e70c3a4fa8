"""Kernel time per STREAM and kernel name over the last window of a rocprofv3 kernel trace (rocpd database): what the
critical (vision) stream's busy time is made of, separately from the text stream's.
usage: python tools/debug/stream_classes.py <db> <window_ms> [top_n]"""
import collections, sqlite3, sys
db, win_ms = sys.argv[1], float(sys.argv[2])
top = int(sys.argv[3]) if len(sys.argv) > 3 else 25
c = sqlite3.connect(db)
rows = list(c.execute("select stream_id, start, end, name from kernels order by start"))
t1 = max(r[2] for r in rows)
w0 = t1 - win_ms * 1e6
by = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for sid, s, e, name in rows:
    if s >= w0:
        ent = by[sid][name.replace("(anonymous namespace)::", "")[:90]]
        ent[0] += 1; ent[1] += (e - s) / 1e3
for sid, d in sorted(by.items(), key=lambda kv: -sum(v[1] for v in kv[1].values())):
    tot = sum(v[1] for v in d.values())
    print(f"== stream {sid}: {tot / 1e3:.2f} ms busy in the last {win_ms:.0f} ms, {sum(v[0] for v in d.values())} launches")
    for name, (n, us) in sorted(d.items(), key=lambda kv: -kv[1][1])[:top]:
        print(f"   {us / 1e3:8.3f} ms  {n:4d}x  avg {us / n:7.1f} us  {name}")
