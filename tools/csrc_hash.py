"""sha256 (first 16 hex digits) over the kernel sources a profile describes: clover_amd/csrc/*.hip, common.hpp, the Makefile and
include/clover_hip.h.  The profiling scripts store it in a profile's meta; bench.py recomputes it and flags counters taken
from other kernel sources as stale (the GPU box has no .git to ask)."""
import glob
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def csrc_sha16(root=ROOT):
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(root, 'clover_amd', 'csrc', '*.hip'))) + [
        os.path.join(root, 'clover_amd', 'csrc', 'common.hpp'), os.path.join(root, 'clover_amd', 'csrc', 'Makefile'),
        os.path.join(root, 'include', 'clover_hip.h')]
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, 'rb').read())
    return h.hexdigest()[:16]


if __name__ == '__main__':
    print(csrc_sha16())
