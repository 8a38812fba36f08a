import sys, re, collections
tab = collections.OrderedDict(); cols = []
cur = None
for l in open(sys.argv[1]):
    m = re.match(r'== gemm_ablate_(\S+) epi=(\d) tile=(\S+)', l)
    if m:
        cur = f'{m.group(1)}@{m.group(3)}'; cols.append(cur); continue
    m = re.match(r'(\S+ \S+)\s+M=.*?:\s+([\d.]+) us', l)
    if m and cur: tab.setdefault(m.group(1), {})[cur] = float(m.group(2))
print(' ' * 10 + ' '.join(f'{c[-18:]:>18s}' for c in cols))
for k, v in tab.items():
    print(f'{k:10s}' + ' '.join(f'{v.get(c, 0):18.1f}' for c in cols))
