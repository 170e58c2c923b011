#!/usr/bin/env python
"""Group sums of the A/B per-launch tables written by tools/ab_layers.sh:  python tools/ab_report.py outdir"""
import collections, glob, json, sys
def cls(r):
    if r['kind'] != 0: return 'other'
    if r['k'] == 1: return '1x1'
    if r['s'] == 2: return '3x3s2'
    if 'band' in r.get('name', ''): return 'band%d' % r['hout']
    return '3x3s1gen'
tabs = {}
for f in sorted(glob.glob(sys.argv[1] + '/*.json')):
    d = json.load(open(f)); g = collections.defaultdict(float)
    for r in d['per_launch']: g[cls(r)] += r['ms']
    g['TOTAL'] = d['sum_ms']; tabs[f.split('/')[-1][:-5]] = g
keys = sorted({k for g in tabs.values() for k in g})
print('%-10s' % '', ' '.join('%8s' % t for t in tabs))
for k in keys: print('%-10s' % k, ' '.join('%8.4f' % tabs[t][k] for t in tabs))
