"""Compact summary of a rocprofv3 kernel trace (csv): per kernel (short name) and optionally per grid size.
    python tools/prof_summary.py gpurun_out/prof2/r2_kernel_trace.csv [--by-grid SUBSTR] [--steps N | --steady]"""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r'\(.*', '', name)
    name = re.sub(r'^void ', '', name)
    name = name.replace('at::native::(anonymous namespace)::', 'at::').replace('at::native::', 'at::')
    if name.startswith('_ZN3tfx'):
        m = re.match(r'_ZN3tfx\d+([a-z_0-9]+?)I', name)
        name = 'tfx::' + (m.group(1) if m else name)
    return name[:60]


def main():
    path = sys.argv[1]
    by_grid = sys.argv[sys.argv.index('--by-grid') + 1] if '--by-grid' in sys.argv else None
    steps = int(sys.argv[sys.argv.index('--steps') + 1]) if '--steps' in sys.argv else 1
    rows = list(csv.DictReader(open(path)))
    if '--steady' in sys.argv:
        # steady state only: the kernels between the first and the last optimizer launch (`adam_k` closes a training step) - model construction,
        # warm-up allocation and the synthetic-batch generator (hundreds of tiny at:: kernels) stay out of the per-step figures
        rows.sort(key=lambda r: int(r['Start_Timestamp']))
        marks = [i for i, r in enumerate(rows) if 'adam_k' in r['Kernel_Name']]
        assert len(marks) >= 2, 'need at least two optimizer launches for --steady'
        spans = [(a + 1, b + 1) for a, b in zip(marks[:-1], marks[1:])]
        small = min(b - a for a, b in spans)
        spans = [(a, b) for a, b in spans if b - a <= 1.5 * small]          # drop bench.py's structure-miss step (thousands of tiny upload kernels)
        rows, steps = [r for a, b in spans for r in rows[a:b]], len(spans)
    agg = defaultdict(lambda: [0, 0.0, 1e30, 0.0])
    for r in rows:
        nm = short(r['Kernel_Name'])
        dur = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        key = nm
        if by_grid and by_grid in nm:
            key = f"{nm} grid={int(r['Grid_Size_X']) // max(int(r['Workgroup_Size_X']), 1)}x{r['Grid_Size_Y']}x{r['Grid_Size_Z']}"
        a = agg[key]
        a[0] += 1; a[1] += dur; a[2] = min(a[2], dur); a[3] = max(a[3], dur)
    tot = sum(a[1] for a in agg.values())
    print(f'total kernel time {tot / 1e3:.2f} ms over {steps} step(s) = {tot / 1e3 / steps:.2f} ms/step')
    print(f'{"ms/step":>9} {"%":>6} {"calls/step":>10} {"avg us":>9} {"min us":>9} {"max us":>9}  kernel')
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        if a[1] / tot < 0.002 and not (by_grid and by_grid in k):
            continue
        print(f'{a[1] / 1e3 / steps:9.3f} {100 * a[1] / tot:6.2f} {a[0] / steps:10.1f} {a[1] / a[0]:9.1f} {a[2]:9.1f} {a[3]:9.1f}  {k}')


if __name__ == '__main__':
    main()
