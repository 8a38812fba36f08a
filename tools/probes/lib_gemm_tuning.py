"""Probe helper: tuned LIBRARY GEMMs (PyTorch TunableOp over hipBLASLt / rocBLAS) as the comparison baseline of the GEMM
probes (gemm_sweep.py, gemm_tiles.py, gemm_bench.py: "own kernel vs the library's best on this shape").

No product code uses it: since round 4 no library GEMM is left on the step.  ``tools/probes/tuning/*.csv`` holds the
per-shape winners measured once on an MI355X — the library's default heuristics pick poor kernels for several of this
workload's tall-skinny shapes, and an untuned baseline would flatter the own kernels.  torch validates the file header
(torch / hipBLASLt / rocBLAS versions, gfx arch) and ignores a file that does not match."""
import os
import tempfile

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_FILE = os.path.join(_HERE, 'tuning', 'gemm_gfx950_bench_b8.csv')


def enable_tuned_gemms(path=None, tune_missing=False, out_path=None):
    """Turn TunableOp on with the committed selections.  tune_missing=True additionally benchmarks every
    GEMM shape met for the first time (tens of seconds on the first step) and writes the merged table to
    `out_path` at exit.  Returns the number of selections loaded."""
    t = torch.cuda.tunable
    t.enable(True)
    t.tuning_enable(bool(tune_missing))
    if tune_missing:
        t.set_max_tuning_duration(150)
        t.set_max_tuning_iterations(200)
        t.set_rotating_buffer_size(512)
        if out_path:
            t.set_filename(out_path)
    else:                                   # torch rewrites its table at exit: keep that out of the cwd
        t.set_filename(os.path.join(tempfile.gettempdir(), 'clover_tunableop.csv'))
    path = path or DEFAULT_FILE
    ok = os.path.exists(path) and t.read_file(path)
    return len(t.get_results()) if ok else 0


def save_results(path):
    """Write the current selections (loaded + newly tuned) in TunableOp's CSV format."""
    t = torch.cuda.tunable
    lines = ['Validator,%s,%s' % (k, v) for k, v in t.get_validators()]
    lines += ['%s,%s,%s,%s' % tuple(r) for r in t.get_results()]
    with open(path, 'w') as f:
        f.write('\n'.join(lines) + '\n')
    return len(lines)
