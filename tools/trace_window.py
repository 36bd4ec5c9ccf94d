#!/usr/bin/env python3
"""Print a window of the kernel sequence (name, duration, gap to the previous kernel's end) of a rocprofv3 kernel trace:
   python tools/trace_window.py <dir> <first-kernel-substring> <occurrence> <count>"""
import csv, glob, sys
d, key, occ, cnt = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
hits = [i for i, r in enumerate(rows) if key in r['Kernel_Name']]
i0 = hits[occ]
prev = None
t0 = int(rows[i0]['Start_Timestamp'])
for r in rows[i0:i0 + cnt]:
    st, en = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    name = r['Kernel_Name'].replace('void mgx::', '').replace('mgx::', '').replace('gunrock::', '')[:70]
    print("%9.1f us  %-70s dur %8.1f us  gap %7.1f us" % ((st - t0) / 1e3, name, (en - st) / 1e3, (st - prev) / 1e3 if prev else 0))
    prev = en
