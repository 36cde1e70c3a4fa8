"""Per-stream busy time and idle gaps of a rocprofv3 kernel trace (rocpd database): is the step bound by kernels or by
launch gaps?  usage: python tools/stream_gaps.py <db> [n_passes]"""
import sqlite3, sys, collections
db = sys.argv[1]
win_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0   # analyse only the last win_ms milliseconds
c = sqlite3.connect(db)
rows = list(c.execute("select stream_id, queue_id, start, end, name from kernels order by start"))
t0, t1 = rows[0][2], max(r[3] for r in rows)
print(f"{len(rows)} dispatches, span {(t1 - t0) / 1e6:.2f} ms")
by = collections.defaultdict(list)
for sid, qid, s, e, name in rows:
    by[(sid, qid)].append((s, e, name))
w0 = t1 - win_ms * 1e6 if win_ms > 0 else t0 + 0.4 * (t1 - t0)   # default: last 60 % of the span
for key, ks in sorted(by.items(), key=lambda kv: -sum(e - s for s, e, _ in kv[1])):
    ks = [k for k in ks if k[0] >= w0]
    if not ks:
        continue
    busy = sum(e - s for s, e, _ in ks)
    span = ks[-1][1] - ks[0][0]
    gaps = [ks[i + 1][0] - ks[i][1] for i in range(len(ks) - 1)]
    pos = [g for g in gaps if g > 0]
    big = sorted(pos)[-5:]
    print(f"stream {key}: {len(ks)} kernels, busy {busy / 1e6:.2f} ms of {span / 1e6:.2f} ms ({100 * busy / max(span, 1):.1f} %), "
          f"gaps: {len(pos)} totalling {sum(pos) / 1e6:.2f} ms, median {sorted(pos)[len(pos) // 2] / 1e3 if pos else 0:.1f} us, largest {[round(g / 1e3) for g in big]} us")
# union busy over all streams
ev = sorted((s, e) for s, e, _ in [(r[2], r[3], 0) for r in rows if r[2] >= w0])
cur_s, cur_e, tot = ev[0][0], ev[0][1], 0
for s, e in ev[1:]:
    if s > cur_e:
        tot += cur_e - cur_s; cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
tot += cur_e - cur_s
print(f"union of all streams: GPU busy {tot / 1e6:.2f} ms of {(ev[-1][1] - ev[0][0]) / 1e6:.2f} ms")
# context of the largest gaps of the busiest stream
key = max(by, key=lambda k: sum(e - s for s, e, _ in by[k] if s >= w0))
ks = [k for k in by[key] if k[0] >= w0]
gl = sorted(((ks[i + 1][0] - ks[i][1], i) for i in range(len(ks) - 1)), reverse=True)[:12]
for g, i in sorted(gl, key=lambda t: t[1]):
    print(f"gap {g / 1e3:8.1f} us at +{(ks[i][1] - w0) / 1e6:7.2f} ms  after {ks[i][2][:60]}  before {ks[i + 1][2][:60]}")
# optional 3rd argument N: the launches of ALL streams around the N largest gaps of the busiest stream (who is running
# while it idles, and what each side was doing just before)
if len(sys.argv) > 3:
    allk = sorted((r[2], r[3], r[0], r[4]) for r in rows if r[2] >= w0)
    for g, i in sorted(gl[:int(sys.argv[3])], key=lambda t: t[1]):
        gs, ge = ks[i][1], ks[i + 1][0]
        print(f"--- gap {g / 1e3:.1f} us of stream {key[0]} at +{(gs - w0) / 1e6:.2f} ms")
        before = [k for k in allk if k[1] <= gs][-10:]
        during = [k for k in allk if k[0] < ge and k[1] > gs]
        after = [k for k in allk if k[0] >= ge][:4]
        for tag, lst in (("before", before), ("during", during[:6] + ([("...",)] if len(during) > 12 else []) + during[-6:] if len(during) > 12 else during), ("after", after)):
            for k in lst:
                if len(k) == 1:
                    print(f"   {tag:7s} ... {len(during) - 12} more")
                    continue
                print(f"   {tag:7s} s{k[2]} +{(k[0] - gs) / 1e3:9.1f} us  dur {(k[1] - k[0]) / 1e3:7.1f}  {k[3][:90]}")
