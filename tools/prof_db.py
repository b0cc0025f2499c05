"""per-kernel time of a rocprofv3 --kernel-trace run (rocpd sqlite output):  python tools/prof_db.py <results.db> [steps-kernel-name]"""
import collections, re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
rows = db.execute(f"select s.kernel_name, d.end-d.start, d.start from {kd} d join {ks} s on d.kernel_id=s.id order by d.start").fetchall()
rows = rows[len(rows) // 3:]                      # steady state
agg = collections.defaultdict(lambda: [0, 0])
for n, dur, _ in rows:
    n = re.sub(r'\(.*', '', n)
    n = re.sub(r'^_ZN3eeg\d+', '', n)[:80]
    agg[n][0] += dur
    agg[n][1] += 1
marker = sys.argv[2] if len(sys.argv) > 2 else "tsconv_bwd_w_kernel"
steps = max(1, sum(c for n, (t, c) in agg.items() if marker in n and "reduce" not in n))
tot = sum(v[0] for v in agg.values())
print(f"steps {steps}   kernel time per step {tot / steps / 1e3:.1f} us   span per step {(rows[-1][2] - rows[0][2]) / steps / 1e3:.1f} us")
for n, (t, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:int(sys.argv[3]) if len(sys.argv) > 3 else 40]:
    print(f"{t / tot * 100:5.1f}%  {t / steps / 1e3:7.1f} us/step  {t / c / 1e3:8.1f} us x {c / steps:5.2f}  {n}")
