#!/usr/bin/env python3
"""Per-launch timeline of one iteration from a rocprofv3 --kernel-trace csv: kernel, duration, and the item count the launch
worked on estimated from the traversal kernel before it (k_trace_closest is linear in rays at full grids).
    python3 tools/round_timeline.py <kernel_trace.csv> [kernel substring] """
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
want = sys.argv[2] if len(sys.argv) > 2 else "k_camera_shade"
def short(n): return n.split('(')[0].replace('void ', '').replace('etxd::', '')
ev = sorted((int(r['Start_Timestamp']), short(r['Kernel_Name']), (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3) for r in rows)
idx = max(i for i, e in enumerate(ev) if e[1] == 'k_camera_generate')
last_trace = 0.0
for t, k, d in ev[idx:]:
    if k.startswith('k_trace_closest'):
        last_trace = d
    if want in k:
        print('%-30s %8.1f us   (trace before it %6.1f us)' % (k, d, last_trace))
