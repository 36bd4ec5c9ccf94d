#!/usr/bin/env python3
"""Per-traversal accounting of a rocprofv3 kernel trace of `bench.py` (batched submission): the timed batch's traversals
(delimited by k_bfs_fused_init), per kernel type: launches and us per traversal; push / build launches by duration class.
   python tools/trace_batch_stats.py <trace dir> [steps]"""
import collections, csv, glob, sys
d = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 64
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
def short(n):
    n = n.replace('void mgx::', '').replace('mgx::', '')
    return n.split('(')[0][:40]
inits = [i for i, r in enumerate(rows) if 'k_bfs_fused_init' in r['Kernel_Name']]
# the timed batch: the longest run of `steps` consecutive inits whose gaps are small (no host work between them)
best = None
for k in range(len(inits) - steps + 1):
    span = int(rows[inits[k + steps - 1]]['Start_Timestamp']) - int(rows[inits[k]]['Start_Timestamp'])
    if best is None or span < best[0]:
        best = (span, k)
k0 = best[1]
i0, i1 = inits[k0], (inits[k0 + steps] if k0 + steps < len(inits) else len(rows))
seg = rows[i0:i1]
# cut at the publish kernel behind the last traversal
for j, r in enumerate(seg):
    if 'k_bfs_publish' in r['Kernel_Name'] and j > len(seg) - 40:
        seg = seg[:j + 1]; break
t0, t1 = int(seg[0]['Start_Timestamp']), int(seg[-1]['End_Timestamp'])
print("timed batch: %d launches, %.1f us per traversal (span / %d)" % (len(seg), (t1 - t0) / 1e3 / steps, steps))
tot = collections.defaultdict(lambda: [0, 0.0])
classes = {"k_bfs_push": collections.defaultdict(lambda: [0, 0.0]), "k_bfs_build2": collections.defaultdict(lambda: [0, 0.0])}
busy = 0.0
for r in seg:
    du = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    n = short(r['Kernel_Name'])
    tot[n][0] += 1; tot[n][1] += du; busy += du
    for key in classes:
        if n.startswith(key):
            b = 4 if du < 4 else 8 if du < 8 else 16 if du < 16 else 32 if du < 32 else 64 if du < 64 else 999
            classes[key][b][0] += 1; classes[key][b][1] += du
print("busy %.1f us per traversal, gaps %.1f" % (busy / steps, ((t1 - t0) / 1e3 - busy) / steps))
for n, (c, u) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print("  %-42s %6.2f launches  %7.1f us per traversal  (avg %6.1f us)" % (n, c / steps, u / steps, u / c))
for key, cl in classes.items():
    print("  %s by duration:" % key)
    for b in sorted(cl):
        c, u = cl[b]
        print("      < %3s us: %6.2f launches  %7.1f us per traversal" % (b if b < 999 else "inf", c / steps, u / steps))
