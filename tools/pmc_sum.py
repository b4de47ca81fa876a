"""Summarise rocprofv3 --pmc counter_collection CSVs: per kernel, mean counter value per launch."""
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0][-40:]
        acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
flt = sys.argv[2] if len(sys.argv) > 2 else ''
for k, d in acc.items():
    if flt in k:
        for c, v in sorted(d.items()):
            print(f'{k:42s} {c:34s} n={len(v):3d} mean={sum(v)/len(v):.4g}')
