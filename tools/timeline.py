"""Per-step timeline of the two streams from a rocprofv3 --kernel-trace CSV (test infrastructure).
    python tools/timeline.py <kernel_trace.csv> [step_index_from_end=2] [marker=pack_input_kernel]
Splits the trace into steps at launches of the marker kernel (scene inference: gather_tiles_kernel), then prints for one step every kernel with its queue, start
offset, duration, and the idle gap since the previous kernel of the SAME queue; plus per-queue busy time and union busy time."""
import csv, sys, collections
path = sys.argv[1]
which = int(sys.argv[2]) if len(sys.argv) > 2 else 2
marker = sys.argv[3] if len(sys.argv) > 3 else 'pack_input_kernel'
rows = []
for r in csv.DictReader(open(path)):
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r.get('Queue_Id', '0'), r['Kernel_Name'].split('(')[0].replace('void ', '')))
rows.sort()
starts = [i for i, r in enumerate(rows) if r[3].startswith(marker)]
i0, i1 = starts[-which - 1], starts[-which]
step = rows[i0:i1]
t0 = step[0][0]
last_end = {}
busy = collections.Counter()
print(f'step of {len(step)} kernels, {(step[-1][1] - t0) / 1e3:.1f} us from first start to last end')
queues = sorted({r[2] for r in step})
for s, e, q, name in step:
    gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
    last_end[q] = max(e, last_end.get(q, 0))
    busy[q] += e - s
    col = queues.index(q)
    print(f'{(s - t0) / 1e3:9.1f} {" " * 8 * col}{(e - s) / 1e3:7.1f}  gap {gap:6.1f}  q{col} {name[:70]}')
# union of busy intervals
iv = sorted((s, e) for s, e, _, _ in step)
tot, cs, ce = 0, iv[0][0], iv[0][1]
for s, e in iv[1:]:
    if s > ce: tot += ce - cs; cs, ce = s, e
    else: ce = max(ce, e)
tot += ce - cs
print({f'q{queues.index(q)}_busy_us': round(v / 1e3, 1) for q, v in busy.items()}, 'union_busy_us', round(tot / 1e3, 1))
