"""Aggregate rocprofv3 --pmc counter_collection csv per kernel (short names).
    python tools/pmc_summary.py <counter_collection.csv> [--steps N]"""
import csv, sys
from collections import defaultdict
sys.path.insert(0, __file__.rsplit('/', 1)[0])
from prof_summary import short

path = sys.argv[1]
steps = int(sys.argv[sys.argv.index('--steps') + 1]) if '--steps' in sys.argv else 1
agg = defaultdict(lambda: defaultdict(float))
calls = defaultdict(set)
for r in csv.DictReader(open(path)):
    k = short(r['Kernel_Name'])
    agg[k][r['Counter_Name']] += float(r['Counter_Value'])
    calls[k].add(r['Dispatch_Id'])
names = sorted({c for v in agg.values() for c in v})
print('kernel'.ljust(40), 'calls/step'.rjust(10), *[n[-18:].rjust(19) for n in names])
tot = lambda k: sum(agg[k].values())
for k in sorted(agg, key=lambda k: -agg[k].get(names[0], 0)):
    print(k[:40].ljust(40), f'{len(calls[k]) / steps:10.1f}', *[f'{agg[k].get(n, 0) / steps:19.4g}' for n in names])
