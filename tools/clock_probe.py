"""Effective shader clock per kernel family of the training step: GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / duration,
from two rocprofv3 runs of the serialized step (DM_TRAIN_SIDE_STREAM=0: one kernel at a time) -- a --pmc GRBM_GUI_ACTIVE
pass and a --kernel-trace pass (MI355X_MICROARCH.md, "DVFS give-back").  The quotient reads high on dispatches shorter than
~0.3 ms; families are listed with their mean duration so that those can be told apart.
  python tools/clock_probe.py <pmc dir> <kernel-trace dir>"""
import csv, glob, sys, collections, re


def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    n = re.sub(r'^void ', '', n)
    return n.split('(')[0][:60]


gui = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] == 'GRBM_GUI_ACTIVE':
            gui[short(r['Kernel_Name'])].append(float(r['Counter_Value']))
dur = collections.defaultdict(list)
for f in glob.glob(sys.argv[2] + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        dur[short(r['Kernel_Name'])].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
rows = []
for k in gui:
    if k in dur and dur[k]:
        g = sum(gui[k]) / len(gui[k])
        d = sum(dur[k]) / len(dur[k])          # ns
        rows.append((sum(dur[k]), k, len(dur[k]), d / 1e3, g / 8 / d))
rows.sort(reverse=True)
print(f'{"kernel":60s} {"calls":>6s} {"mean us":>9s} {"GHz":>6s}')
for tot, k, n, us, ghz in rows[:45]:
    print(f'{k:60s} {n:6d} {us:9.1f} {ghz:6.2f}')
