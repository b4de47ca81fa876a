"""Timeline of one training step from a rocprofv3 --kernel-trace CSV: busy time per queue, union busy time, and
how long each kernel family runs ALONE (nothing else on the GPU) -- the part of it no second stream can hide.
  python tools/timeline.py gpurun_out/prof_tl/*/*kernel_trace.csv [steps=1]"""
import csv, gzip, sys, collections, re

f = sys.argv[1]
op = gzip.open if f.endswith('.gz') else open
rows = list(csv.DictReader(op(f, 'rt')))
ev = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r['Queue_Id']) for r in rows]
ev.sort()
# steps are delimited by the SGD kernel
ends = [i for i, e in enumerate(ev) if 'sgd' in e[2]]
if len(ends) < 2:
    sys.exit('need at least two optimizer steps in the trace')
lo, hi = ends[-2] + 1, ends[-1] + 1
step = ev[lo:hi]
t0, t1 = step[0][0], max(e[1] for e in step)
print(f'last step: {len(step)} launches, span {(t1 - t0) / 1e6:.3f} ms')


def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    n = re.sub(r'^void ', '', n)
    return n.split('(')[0][:60]


per_q = collections.defaultdict(float)
for s, e, n, q in step:
    per_q[q] += (e - s) / 1e6
print('busy per queue (ms):', {q: round(v, 2) for q, v in sorted(per_q.items())}, ' sum', round(sum(per_q.values()), 2))
# sweep: union busy, and exclusive time per kernel family
pts = []
for i, (s, e, n, q) in enumerate(step):
    pts.append((s, 1, i))
    pts.append((e, -1, i))
pts.sort()
active = set()
last = pts[0][0]
union = 0
alone = collections.defaultdict(float)
shared = collections.defaultdict(float)
for t, d, i in pts:
    if active:
        union += t - last
        if len(active) == 1:
            alone[short(step[next(iter(active))][2])] += (t - last) / 1e6
        else:
            for j in active:
                shared[short(step[j][2])] += (t - last) / 1e6
    last = t
    if d == 1:
        active.add(i)
    else:
        active.discard(i)
print(f'union busy {union / 1e6:.3f} ms, idle {(t1 - t0 - union) / 1e6:.3f} ms, alone {sum(alone.values()):.2f} ms')
print('running alone (ms) | overlapped (ms)')
for n, v in sorted(alone.items(), key=lambda kv: -kv[1])[:28]:
    print(f'{v:7.3f} | {shared.get(n, 0.0):7.3f}  {n}')

# per queue: the kernel families that keep it busy (queue 1 is the chain of the step)
per = collections.defaultdict(lambda: collections.defaultdict(float))
for s_, e_, n_, q_ in step:
    per[q_][short(n_)] += (e_ - s_) / 1e6
for q_ in sorted(per):
    top = sorted(per[q_].items(), key=lambda kv: -kv[1])[:14]
    print(f'queue {q_}: ' + ', '.join(f'{n_} {v:.2f}' for n_, v in top))
