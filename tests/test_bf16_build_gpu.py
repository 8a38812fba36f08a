"""The bf16 build of the kernels (libclover_hip.so, CLOVER_HALF=bf16).  The element type is chosen when clover_amd is first
imported, so the default test process runs the fp16 build (libclover_hip_f16.so) throughout; this file runs the kernel tests
and the step-level parity tests once more in a child process with CLOVER_HALF=bf16 — the same test files, the bf16 tolerance
table of tests/test_step_gpu.py."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args):
    env = dict(os.environ, CLOVER_HALF='bf16')
    r = subprocess.run([sys.executable, '-m', 'pytest', '-x', '-q', '-m', 'gpu'] + args, env=env, capture_output=True,
                       text=True, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-4000:] + r.stderr[-2000:]
    return r.stdout


@pytest.mark.skipif(os.environ.get('CLOVER_HALF', 'f16').lower() == 'bf16', reason='this process already runs the bf16 build')
def test_kernels_in_the_bf16_build():
    out = _run(['tests/test_kernels_gpu.py', '--deselect',
                'tests/test_kernels_gpu.py::test_grouped_weight_gradients_shape_fitted_tiles'])
    assert ' passed' in out


@pytest.mark.skipif(os.environ.get('CLOVER_HALF', 'f16').lower() == 'bf16', reason='this process already runs the bf16 build')
def test_step_parity_in_the_bf16_build():
    out = _run(['tests/test_step_gpu.py', 'tests/test_engine_gpu.py', '-k',
                'step_losses_and_grads or mid_ or bench_shapes or full_size_step_matches_oracle or train_mode_step or '
                'engine_matches_torch_adamw or graph_capture_equals_eager or own_decoder'])
    assert ' passed' in out


@pytest.mark.skipif(os.environ.get('CLOVER_HALF', 'f16').lower() == 'bf16', reason='this process already runs the bf16 build')
def test_fp8_forward_path_in_the_bf16_build():
    out = _run(['tests/test_fp8_gpu.py'])
    assert ' passed' in out
