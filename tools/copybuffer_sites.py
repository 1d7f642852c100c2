"""Which launches surround the __amd_rocclr_copyBuffer kernels of a traced run: histogram of (previous kernel, next kernel) per copy.
    python tools/copybuffer_sites.py <kernel_trace.csv>"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
short = lambda n: n.split('(')[0][-60:]
hist = collections.Counter()
sizes = collections.Counter()
for i, r in enumerate(rows):
    if 'copyBuffer' in r['Kernel_Name']:
        prev = short(rows[i - 1]['Kernel_Name']) if i else '-'
        nxt = short(rows[i + 1]['Kernel_Name']) if i + 1 < len(rows) else '-'
        hist[(prev, nxt)] += 1
        sizes[(r.get('Grid_Size_X', r.get('Grid_Size', '?')), r.get('Workgroup_Size_X', r.get('Workgroup_Size', '?')))] += 1
print('copyBuffer launches:', sum(hist.values()), 'of', len(rows))
for (p, n), c in hist.most_common(40):
    print(f'{c:5d}  after {p:62s} before {n}')
print('grid / workgroup sizes:', sizes.most_common(10))
