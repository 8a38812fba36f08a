"""Condense gemm_trace.txt (trace_summary of gemm_bench.py under rocprofv3) into one line per shape:
library GEMM | own bias | own +gelu | own dgelu (device-side medians, us)."""
import sys, re
rows = [l for l in open(sys.argv[1]) if 'median' in l and 'grid' in l]
names = ['qkv s1', 'proj s1', 'fc1 s1', 'fc2 s1', 'merge s1', 'qkv s2', 'proj s2', 'fc1 s2', 'fc2 s2', 'merge s2', 'qkv s3',
         'proj s3', 'fc1 s3', 'fc2 s3', 'qkv fu', 'out fu', 'fc1 fu', 'fc2 fu', 'qkv bert', 'fc1 bert', 'fc2 bert']
# per shape the bench runs: lib, own(1), [lib, gelu], own(2), [lib, dgelu-kernel], own(3): keep the first library GEMM and the three own kernels
out, cur = [], None
for l in rows:
    med = float(re.search(r'median\s+([\d.]+)', l).group(1))
    if 'gemm_nt_' in l:
        epi = int(re.search(r'gemm_nt_\w*kernel<[\d, ]*?(\d)>', l).group(1))
        if cur is not None:
            cur[epi] = med
            if epi == 3:
                out.append(cur); cur = None
    elif 'gelu' not in l and cur is None:
        cur = {'lib': med}
for n, r in zip(names, out):
    print(f"{n:9s} lib {r['lib']:6.1f} | own {r.get(1, 0):6.1f} | +gelu {r.get(2, 0):6.1f} | dgelu {r.get(3, 0):6.1f}")
