"""Where does the main stream spend wall time in runs of SMALL kernels?  Reads a rocprofv3 kernel trace (rocpd database),
takes the busiest stream, cuts its timeline into maximal runs of consecutive kernels shorter than THR us, and prints the
runs by wall time (kernel time + the gaps between them) - the serial sections a faster GEMM cannot shorten.
usage: python tools/small_kernel_sections.py <db> [window_ms] [thr_us]"""
import sqlite3, sys, collections
db = sys.argv[1]
win_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
thr = float(sys.argv[3]) * 1e3 if len(sys.argv) > 3 else 30e3
c = sqlite3.connect(db)
rows = list(c.execute("select stream_id, queue_id, start, end, name from kernels order by start"))
t1 = max(r[3] for r in rows)
w0 = t1 - win_ms * 1e6 if win_ms > 0 else rows[0][2]
by = collections.defaultdict(list)
for sid, qid, s, e, name in rows:
    if s >= w0:
        by[(sid, qid)].append((s, e, name))
key = max(by, key=lambda k: sum(e - s for s, e, _ in by[k]))
ks = by[key]
print(f"stream {key}: {len(ks)} kernels in {(ks[-1][1] - ks[0][0]) / 1e6:.2f} ms")
def short(n):
    n = n.replace("void ", "").replace("(anonymous namespace)::", "").replace("at::native::", "")
    return n.split("(")[0].split("<")[0][:36] or n[:36]


runs, cur = [], []
for k in ks:
    if k[1] - k[0] < thr:
        cur.append(k)
    else:
        if cur: runs.append(cur)
        cur = []
if cur: runs.append(cur)
tot_wall = sum(r[-1][1] - r[0][0] for r in runs); tot_k = sum(sum(e - s for s, e, _ in r) for r in runs)
print(f"{len(runs)} runs of kernels < {thr / 1e3:.0f} us: wall {tot_wall / 1e6:.2f} ms, kernel time {tot_k / 1e6:.2f} ms, "
      f"{sum(len(r) for r in runs)} launches")
for r in sorted(runs, key=lambda r: -(r[-1][1] - r[0][0]))[:14]:
    wall = (r[-1][1] - r[0][0]) / 1e3; kt = sum(e - s for s, e, _ in r) / 1e3
    names = collections.Counter(short(n) for _, _, n in r)
    print(f"  at +{(r[0][0] - ks[0][0]) / 1e6:7.2f} ms: {len(r):3d} launches, wall {wall:7.1f} us, kernels {kt:7.1f} us :: "
          + ", ".join(f"{n} x{c_}" for n, c_ in names.most_common(6)))
if len(sys.argv) > 4:   # full sequence of the longest run
    r = max(runs, key=lambda r: r[-1][1] - r[0][0])
    prev = r[0][0]
    for s_, e_, n in r:
        print(f"    +{(s_ - r[0][0]) / 1e3:8.1f} us  gap {(s_ - prev) / 1e3:6.1f}  dur {(e_ - s_) / 1e3:6.1f}  {short(n)}  {n[n.find('<'):][:70] if 'elementwise' in n or 'reduce' in n else ''}")
        prev = e_
