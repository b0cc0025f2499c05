"""summary of one prior training step out of a rocprofv3 kernel trace of tools/bench_prior_train.py:  python tools/prior_trace_summary.py <kernel_trace.csv> [--list]"""
import collections, csv, re, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r['Queue_Id'], int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X']))))
rows.sort()
idx = [i for i, r in enumerate(rows) if 'adamw' in r[2]]
clusters = []
for i in idx:
    if clusters and i - clusters[-1][-1] <= 3:
        clusters[-1].append(i)
    else:
        clusters.append([i])
a, b = clusters[-3][-1] + 1, clusters[-2][-1] + 1
step = rows[a:b]
t0 = step[0][0]
print("launches in step:", len(step), "span us:", round((step[-1][1] - t0) / 1e3, 1))
agg = collections.defaultdict(lambda: [0, 0])
for s, e, n, q, g in step:
    k = re.sub(r'<.*', '', n).replace('void eeg::', '').replace('eeg::', '')[:44]
    agg[k][0] += 1
    agg[k][1] += e - s
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{k:46s} {v[0]:3d} {v[1] / 1e3:8.1f} us  avg {v[1] / v[0] / 1e3:6.2f}")
qs = collections.defaultdict(int)
for s, e, n, q, g in step:
    qs[q] += e - s
print("kernel time per queue (us):", {k: round(v / 1e3, 1) for k, v in qs.items()})
if "--list" in sys.argv:
    for s, e, n, q, g in step:
        print(f"{(s - t0) / 1e3:8.1f} {(e - s) / 1e3:7.1f} us q{q} wgs {g:5d} {re.sub(r'<.*', '', n).replace('void eeg::', '')[:50]}")
