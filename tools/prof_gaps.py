"""Idle time BETWEEN kernels in a rocprofv3 kernel trace (csv) of bench.py on ONE stream (TFX_SIDE_STREAM=0): for the last `--steps` training
steps (delimited by the optimizer kernel `adam_k`), wall time of the step on the GPU, sum of kernel durations, and the distribution of the gaps
between one kernel's end and the next kernel's start.  Tells what a hipGraph replay of the training step could win at most.
    python tools/prof_gaps.py trace.csv [--steps 2]"""
import csv
import sys


def main():
    path = sys.argv[1]
    steps = int(sys.argv[sys.argv.index('--steps') + 1]) if '--steps' in sys.argv else 2
    rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r['Start_Timestamp']))
    ends = [i for i, r in enumerate(rows) if 'adam_k' in r['Kernel_Name']]
    assert len(ends) > steps, 'not enough optimizer launches in the trace'
    for s in range(steps):
        lo, hi = ends[-s - 2] + 1, ends[-s - 1]
        ks = rows[lo:hi + 1]
        wall = (int(ks[-1]['End_Timestamp']) - int(ks[0]['Start_Timestamp'])) / 1e3
        busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in ks) / 1e3
        gaps = [(int(b['Start_Timestamp']) - int(a['End_Timestamp'])) / 1e3 for a, b in zip(ks[:-1], ks[1:])]
        pos = [g for g in gaps if g > 0]
        big = sorted(((g, ks[i]['Kernel_Name'][:40], ks[i + 1]['Kernel_Name'][:40]) for i, g in enumerate(gaps)), reverse=True)[:6]
        print(f'step -{s + 1}: {len(ks)} kernels, wall {wall / 1e3:.2f} ms, busy {busy / 1e3:.2f} ms, idle {sum(pos) / 1e3:.2f} ms '
              f'({len(pos)} gaps, median {sorted(pos)[len(pos) // 2]:.1f} us, mean {sum(pos) / max(len(pos), 1):.1f} us)')
        for g, a, b in big:
            print(f'    {g:8.1f} us between {a} -> {b}')


if __name__ == '__main__':
    main()
