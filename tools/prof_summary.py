"""(top_kernels view: total_duration and average are in microseconds.)
Summarise a rocprofv3 rocpd sqlite database (kernel-trace) into a per-kernel table (top N)."""
import sqlite3, sys
db, n = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 30
c = sqlite3.connect(db)
rows = list(c.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
tot = sum(r[2] for r in rows)
print(f"{'kernel':90s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>9s} {'%':>6s}")
for name, calls, total, avg, pct in rows[:n]:
    print(f"{name[:90]:90s} {calls:6d} {total/1e3:10.3f} {avg:9.1f} {pct:6.2f}")
print(f"{'TOTAL':90s} {sum(r[1] for r in rows):6d} {tot/1e3:10.3f}")
