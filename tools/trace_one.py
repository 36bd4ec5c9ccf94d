#!/usr/bin/env python3
"""Print the kernel sequence (duration, gap) of one BFS from a rocprofv3 kernel trace directory."""
import csv, glob, sys
d = sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/trace_quick'
which = int(sys.argv[2]) if len(sys.argv) > 2 else 10
f = glob.glob(d + '/trace/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
seq = [(r['Kernel_Name'].replace('void mgx::', '').replace('mgx::', '')[:34], (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in rows if 'mgx' in r['Kernel_Name'] or 'Buffer' in r['Kernel_Name']]
seq.sort(key=lambda x: x[2])
inits = [i for i, x in enumerate(seq) if 'fused_init' in x[0]]
i0, i1 = inits[which], inits[which + 1]
prev = None
for name, dur, st, en in seq[i0:i1 + 1]:
    print("%-36s dur %8.1f us  gap %6.1f us" % (name, dur, (st - prev) / 1e3 if prev else 0)); prev = en
print("span us:", (seq[i1][2] - seq[i0][2]) / 1e3)
