"""Prints a bench.py --kernel-table JSON: python tools/print_kt.py <kernel_table.json> [substring] [top_n]."""
import json, sys
kt = json.load(open(sys.argv[1]))
sub = sys.argv[2] if len(sys.argv) > 2 else ''
top = int(sys.argv[3]) if len(sys.argv) > 3 else 100
print('ms_per_step', kt.get('ms_per_step'))
rows = [(k, v) for k, v in kt['kernels'].items() if sub in k]
for k, v in sorted(rows, key=lambda kv: -kv[1]['total_us'])[:top]:
    print(f"{k:36s} {v['launches']:3d} x {v['avg_us']:8.2f} us = {v['total_us']:8.1f} us   frac {v.get('frac', 0):.3f}")
