"""The convolution dispatcher hands small problems to the direct / F(2x2) kernels, so the operator-level cases of
test_hip_ops.py reach the F(4x4,3x3) and the upsample kernels only at benchmark-size shapes (test_hip_configs.py).
This test re-runs the convolution and Block cases ONCE in a child process whose dispatch thresholds are lowered
(the thresholds are read once per process), so that every small edge case also goes through those kernels.  The same
child runs the LinearAttention / Attention cases with the fused kernels switched off: the unfused chains (pre-norm, 1x1
GEMM with its fused RMSNorm epilogue at 64 channels, attention cores) stay covered although no default shape takes them."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_small_shapes_through_the_large_shape_kernels():
    env = dict(os.environ, DM_WINO4_MIN_WGS="1", DM_WINO4_MIN_K="1", DM_UPWINO_MIN_WGS="1", DM_UPWINO_MIN_K="1",
               DM_NO_FUSED_LINATTN="1", DM_NO_ATTN16="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_hip_ops.py"), "-q", "-x",
                        "-m", "gpu", "-k", "conv2d or block or attention", "-p", "no:cacheprovider"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-4000:] + r.stderr[-2000:]
