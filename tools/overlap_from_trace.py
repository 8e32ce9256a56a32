# rocprofv3 kernel trace (rocpd sqlite) of tools/overlap_probe.py -> how much of the decode kernels' time was spent with a kernel of
# ANOTHER stream running as well.  python tools/overlap_from_trace.py x_results.db [last_ms]
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
qcol = next((k for k in ("stream_id", "queue_id", "stream", "queue") if k in cols), None)
rows = c.execute(f"select name, start, end, {qcol or '0'} from kernels order by start").fetchall()
# keep the last timed section: everything after the largest idle gap in the second half is the `both` run; simpler: take the last
# `last_ms` milliseconds of the trace
last_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 1e9
t_end = max(r[2] for r in rows)
rows = [r for r in rows if r[1] >= t_end - last_ms * 1e6]
queues = sorted({r[3] for r in rows})
ev = []
for name, s, e, q in rows:
    ev.append((s, 1, q)); ev.append((e, -1, q))
ev.sort()
busy = both = 0
live = {}
prev = ev[0][0]
for t, d, q in ev:
    n_q = sum(1 for v in live.values() if v > 0)
    if n_q >= 1: busy += t - prev
    if n_q >= 2: both += t - prev
    live[q] = live.get(q, 0) + d
    prev = t
span = ev[-1][0] - ev[0][0]
print(f"column {qcol}: {len(queues)} queues, {len(rows)} kernels over {span / 1e6:.1f} ms; some kernel running {busy / 1e6:.1f} ms "
      f"({100.0 * busy / span:.0f} % of the span); kernels of two or more queues running at once {both / 1e6:.1f} ms = "
      f"{100.0 * both / max(busy, 1):.0f} % of the busy time")
by = {}
for name, s, e, q in rows:
    k = name.split("(")[0][-48:]
    by.setdefault(k, []).append((e - s) / 1e3)
for k, v in sorted(by.items(), key=lambda kv: -sum(kv[1]))[:8]:
    print(f"   {k:50s} {len(v):6d} x {sum(v) / len(v):7.2f} us")
