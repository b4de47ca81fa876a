"""Span, union busy time and idle time of the replays in a rocprofv3 --kernel-trace CSV of tools/infer100_probe.py (all REPS
replays together: under the profiler the synchronize between two replays leaves no reliable gap), per replay, and the
kernel families by time.
  python tools/infer_timeline.py <kernel_trace.csv> [replays=6]"""
import csv, sys, re, collections
rows = list(csv.DictReader(open(sys.argv[1])))
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
g = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r['Queue_Id']) for r in rows)
# drop everything before the first RoIAlign of the last `reps` replays (warm-up, weight packing, capture)
starts = [i for i, e in enumerate(g) if 'roi_align_tile_kernel' in e[2]]
per = max(len(starts) // max(reps, 1), 1) if len(starts) >= reps else 1
g = g[starts[-reps * per]:] if len(starts) >= reps * per else g
t0, t1 = g[0][0], max(e[1] for e in g)
pts = sorted([(s, 1) for s, e, n, q in g] + [(e, -1) for s, e, n, q in g])
busy, act, last = 0, 0, pts[0][0]
for t, d in pts:
    if act > 0: busy += t - last
    act += d; last = t
host_gaps = 0
print(f'{reps} replays: {len(g)} launches on {len(set(e[3] for e in g))} queues; per replay: {len(g) / reps:.0f} launches, '
      f'union busy {busy / 1e3 / reps:.1f} us, sum of kernel durations {sum(e[1] - e[0] for e in g) / 1e3 / reps:.1f} us '
      f'(span incl. the host\'s synchronize between replays {(t1 - t0) / 1e3 / reps:.1f} us)')
fam = collections.defaultdict(lambda: [0, 0.0])
for s, e, n, q in g:
    k = re.sub(r'\(anonymous namespace\)::', '', n); k = re.sub(r'^void ', '', k).split('(')[0][:50]
    fam[k][0] += 1; fam[k][1] += (e - s) / 1e3
print('per replay:')
for k, (c, us) in sorted(fam.items(), key=lambda kv: -kv[1][1])[:16]:
    print(f'  {us / reps:8.1f} us  x{c / reps:4.1f}  {k}')
if len(sys.argv) > 3 and sys.argv[3] == 'order':
    # the LAST replay launch by launch: start (us from the replay's first launch), duration, queue, workgroups, kernel
    last = g[-int(round(len(g) / reps)):]
    by = {(int(r['Start_Timestamp']), r['Kernel_Name']): r for r in rows}
    z = last[0][0]
    qs = sorted(set(e[3] for e in last))
    print(f'last replay, launch by launch ({len(last)} launches, span {(max(e[1] for e in last) - z) / 1e3:.1f} us):')
    for s, e, n, q in last:
        r = by[(s, n)]
        wgs = 1
        for ax in 'XYZ':
            wgs *= max(int(r.get(f'Grid_Size_{ax}', 1) or 1) // max(int(r.get(f'Workgroup_Size_{ax}', 1) or 1), 1), 1)
        k = re.sub(r'\(anonymous namespace\)::', '', n); k = re.sub(r'^void ', '', k).split('(')[0][:46]
        print(f'  {(s - z) / 1e3:8.1f} +{(e - s) / 1e3:7.1f} us  q{qs.index(q)}  {wgs:6d} wg  lds {r.get("LDS_Block_Size", "?"):>6}  {k}')
