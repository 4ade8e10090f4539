#!/usr/bin/env python3
"""Per-layer table (min over rounds, us) of a tools/ab_conv.sh log: python tools/ab_table.py LOG"""
import collections
import re
import sys

d, names, cur = collections.OrderedDict(), [], None
for line in open(sys.argv[1]):
    m = re.match(r'== (\S+) B=', line)
    if m:
        cur = m.group(1)
        if cur not in names:
            names.append(cur)
        continue
    m = re.match(r'(\S.*?)\s+M=.*?([\d.]+) us', line) or re.match(r'(conv stack), B=\d+: ([\d.]+) us', line)
    if m:
        d.setdefault(m.group(1), collections.OrderedDict()).setdefault(cur, []).append(float(m.group(2)))
print('%-28s' % 'layer (us, min of rounds)' + ''.join('%10s' % n for n in names) + ''.join('%9s' % ('%s/%s' % (n[:3], names[0][:3])) for n in names[1:]))
for k, v in d.items():
    mins = [min(v[n]) for n in names]
    print('%-28s' % k + ''.join('%10.1f' % x for x in mins) + ''.join('%9.3f' % (x / mins[0]) for x in mins[1:]))
