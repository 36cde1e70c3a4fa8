"""Long kernels on small grids: from a rocprofv3 kernel trace (rocpd database), every dispatch that runs longer than MIN_US on
fewer than MAX_WG workgroups - a single output tile walking a long contraction, a reduction on one workgroup, ... (how the
1.6-ms 8 x 8 weight gradient of the MAE branch was found).  usage: python tools/small_grid_scan.py <db> [min_us=150] [max_wg=64]"""
import collections, sqlite3, sys
db = sys.argv[1]
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 150.0
max_wg = int(sys.argv[3]) if len(sys.argv) > 3 else 64
c = sqlite3.connect(db)
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
gx = [n for n in ("grid_x", "grid_size_x", "grid_size") if n in cols]
wx = [n for n in ("workgroup_x", "workgroup_size_x", "workgroup_size") if n in cols]
if not gx or not wx:
    print("columns:", cols)
    sys.exit(0)
q = f"select name, start, end, {gx[0]}, {wx[0]}" + (", grid_y, grid_z, workgroup_y, workgroup_z" if "grid_y" in cols else "") + " from kernels"
agg = collections.defaultdict(lambda: [0, 0.0, 0])
for r in c.execute(q):
    name, s, e = r[0], r[1], r[2]
    g, w = r[3], max(r[4], 1)
    if len(r) > 5:
        g = g * max(r[5], 1) * max(r[6], 1)
        w = w * max(r[7], 1) * max(r[8], 1)
    nwg = g // w if g >= w else g
    us = (e - s) / 1e3
    if us >= min_us and nwg < max_wg and "spin_kernel" not in name:
        a = agg[(name[:100], nwg)]
        a[0] += 1; a[1] += us; a[2] = max(a[2], us)
for (name, nwg), (n, tot, mx) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{n:4d}x  {tot / n:9.1f} us avg  max {mx:9.1f}  {nwg:4d} workgroups  {name}")
if not agg:
    print("none")
