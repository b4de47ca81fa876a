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

# ---- the chain: queue with the most busy time.  Its kernels in order, the gap in front of each (time the queue sat
# empty: launch latency or a cross-stream wait), and what ran elsewhere during the longest gaps.
if len(sys.argv) > 2 and sys.argv[2] == 'chain':
    chain_q = max(per_q, key=per_q.get)
    ch = [(s, e, n) for s, e, n, q in step if q == chain_q]
    gaps = []
    prev_end = ch[0][0]
    tot_gap = 0
    for s, e, n in ch:
        g = max(0, s - prev_end)
        tot_gap += g
        gaps.append((g, s, n))
        prev_end = max(prev_end, e)
    print(f'\nchain = queue {chain_q}: {len(ch)} kernels, busy {per_q[chain_q]:.2f} ms, gaps in front of its kernels {tot_gap / 1e6:.2f} ms '
          f'(first start to last end {(ch[-1][1] - ch[0][0]) / 1e6:.2f} ms)')
    hist = collections.Counter(min(int(g / 1e3) // 5 * 5, 100) for g, _, _ in gaps)
    print('gap histogram (us bucket: count):', dict(sorted(hist.items())))
    print('largest gaps on the chain (us | at ms | kernel that follows | kernels running elsewhere during the gap):')
    for g, s, n in sorted(gaps, reverse=True)[:25]:
        others = sorted({short(nn) for ss, ee, nn, qq in step if qq != chain_q and ss < s and ee > s - g})
        print(f'{g / 1e3:8.1f} | {(s - t0) / 1e6:7.3f} | {short(n)[:50]:50s} | {", ".join(o[:28] for o in others[:4])}')
    # per-kernel-family: time on the chain, solo-equivalent unknown; print the chain's sequence compactly
    if len(sys.argv) > 3 and sys.argv[3] == 'seq':
        for s, e, n in ch:
            print(f'{(s - t0) / 1e6:8.3f} +{(e - s) / 1e3:7.1f} us  {short(n)}')
