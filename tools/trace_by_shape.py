"""Aggregates a rocprofv3 kernel-trace CSV per (kernel, grid, workgroup): the per-shape launch durations a --stats summary folds together.
usage: python tools/trace_by_shape.py <kernel_trace.csv> [out.json]"""
import collections
import csv
import json
import re
import sys


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'\(.*$', '', name)
    return name.replace('void ', '')[:90]


def main():
    rows = collections.defaultdict(list)
    with open(sys.argv[1]) as f:
        for r in csv.DictReader(f):
            key = (short(r['Kernel_Name']), int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X'])), int(r.get('Grid_Size_Y', 1)),
                   int(r['Workgroup_Size_X']))
            rows[key].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    out = []
    for (name, gx, gy, wg), d in rows.items():
        d.sort()
        out.append(dict(kernel=name, blocks_x=gx, grid_y=gy, threads=wg, launches=len(d), median_us=round(d[len(d) // 2], 2),
                        mean_us=round(sum(d) / len(d), 2), total_us=round(sum(d), 1)))
    out.sort(key=lambda e: -e['total_us'])
    tot = sum(e['total_us'] for e in out)
    for e in out[:70]:
        print(f"{e['total_us'] / tot * 100:5.1f}%  {e['launches']:6d} x {e['median_us']:8.2f} us  grid {e['blocks_x']:5d} x {e['grid_y']:3d} x {e['threads']:4d}  {e['kernel']}")
    if len(sys.argv) > 2:
        json.dump(out, open(sys.argv[2], 'w'), indent=1)
    return 0


if __name__ == '__main__':
    sys.exit(main())
