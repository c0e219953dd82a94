"""The convolution dispatcher hands small problems to the direct / F(2x2) kernels, so the operator-level cases of
test_hip_ops.py reach the F(4x4,3x3) and the upsample kernels only at benchmark-size shapes (test_hip_configs.py).
This test re-runs the convolution and Block cases ONCE in a child process whose dispatch thresholds are lowered
(the thresholds are read once per process), so that every small edge case also goes through those kernels.  The same
child runs the LinearAttention / Attention cases with the fused kernels switched off: the unfused chains (pre-norm, 1x1
GEMM with its fused RMSNorm epilogue at 64 channels, attention cores -- the softmax core in its tiled long-sequence form)
stay covered although no default shape takes them."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_small_shapes_through_the_large_shape_kernels():
    env = dict(os.environ, DM_WINO4_MIN_WGS="1", DM_WINO4_MIN_K="1", DM_UPWINO_MIN_WGS="1", DM_UPWINO_MIN_K="1",
               DM_NO_FUSED_LINATTN="1", DM_NO_ATTN16="1", DM_ATTN_TILED="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_hip_ops.py"), "-q", "-x",
                        "-m", "gpu", "-k", "conv2d or block or attention", "-p", "no:cacheprovider"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-4000:] + r.stderr[-2000:]


def test_operator_cases_on_the_small_tile_kernel_forms():
    """Round 4: the 1x1 GEMM kernel with 16-pixel wave tiles (RT = 1) and the F(2x2) Winograd kernel with 32-cout workgroups
    (Q = 1) are chosen for grids that leave CUs idle; with the thresholds raised they take EVERY eligible shape of the operator
    tests (ragged pixel blocks, concat inputs, K splits, residual / partial epilogues), at the unchanged tolerances.  Also the
    other order of the 1x1 plan (K splits before smaller tiles) and RT = 2."""
    for extra in (dict(DM_PW_RT_TARGET_WGS="1000000", DM_WINO_Q_TARGET_WGS="1000000"),
                  dict(DM_PW_RT_TARGET_WGS="1000000", DM_PW_RT_MIN="2", DM_PW_RT_FIRST="0")):
        env = dict(os.environ, **extra)
        r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_hip_ops.py"),
                            os.path.join(ROOT, "tests", "test_hip_train_ops.py"), "-q", "-x", "-m", "gpu", "-k",
                            "conv2d or block or attention or downsample", "-p", "no:cacheprovider"],
                           cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-4000:] + r.stderr[-2000:]


MODEL_CASES = "test_unet_full_forward or test_unet_latent_and_text_full or test_full_samplers or test_unet_text_variants"


def _run_models(env_extra):
    env = dict(os.environ, **env_extra)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_hip_model.py"),
                        os.path.join(ROOT, "tests", "test_hip_r3.py"), "-q", "-x", "-m", "gpu", "-k",
                        MODEL_CASES + " or test_ldm_coco_text_shapes or test_ldm_cifar_shapes", "-p", "no:cacheprovider"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-4000:] + r.stderr[-2000:]


def test_model_goldens_with_the_alternative_kernels():
    """The model-level goldens of the reference (full 32x32 U-Net, latent / text variants, DDPM + DDIM loops, the LDM YAML
    shapes) with every dispatch alternative forced: no F(4x4) Winograd and no 1x1 GEMM kernel (the F(2x2) / direct
    kernels take their layers), the general CrossAttention path instead of the one-token algebra, separate res_conv
    landings, no fused attention kernels (and the VALU forms of the unfused LinearAttention core), the folded instead of the
    9-multiply upsample conv."""
    _run_models(dict(DM_NO_WINO4="1", DM_NO_PW="1", DM_NO_CROSS1="1", DM_NO_RES_MERGE="1", DM_NO_UPWINO="1",
                     DM_NO_FUSED_LINATTN="1", DM_NO_ATTN16="1", DM_NO_INIT7="1", DM_LINATTN_VALU="1"))


def test_model_goldens_on_the_small_tile_kernel_forms():
    """The model-level goldens with the small-tile forms of the 1x1 and F(2x2) kernels forced everywhere (see above)."""
    _run_models(dict(DM_PW_RT_TARGET_WGS="1000000", DM_WINO_Q_TARGET_WGS="1000000"))


def test_model_goldens_with_the_forked_step():
    """... and with the second-stream fork of the step switched on (res_conv next to block1, the time MLP next to
    init_conv; off by default because it measured slower, dm_api.hip: par_policy): eager and graph-replayed loops."""
    _run_models(dict(DM_PAR="1"))


def test_model_goldens_without_any_winograd():
    """... and with every Winograd-family kernel off (the direct implicit-GEMM kernel takes all 3x3 layers)."""
    _run_models(dict(DM_NO_WINOGRAD="1"))


TRAIN_CASES = ("test_loss_and_all_gradients_vs_reference_autograd and (small_d32 or mid_d64 or full_b8) or "
               "test_training_steps_vs_torch_adam or test_checkpoint_round_trip or test_text_conditional_training_step")


def _run_training(env_extra):
    env = dict(os.environ, **env_extra)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_hip_train.py"), "-q", "-x", "-m",
                        "gpu", "-k", TRAIN_CASES, "-p", "no:cacheprovider"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-4000:] + r.stderr[-2000:]


def test_training_goldens_with_one_launch_per_layer():
    """The gradient goldens of the reference's autograd, the three-iteration Adam loop and the checkpoint round trip with the
    per-layer forms of everything the training step batches by default: one weight-gradient launch and one split-K sum per
    layer (no grouped launch, no deferred reductions), the direct 3x3 weight gradient instead of the Winograd-domain one,
    lazy instead of grouped re-packing; round 4: separate landing passes, the VALU Linear kernels, the attention backward
    without its score cache, the LinearAttention backward recomputing the key statistics instead of reading the tape's, final_conv's three
    gradients as three launches, init_conv's weight gradient and the LinearAttention backward on the VALU, the rotated weights materialised before packing."""
    _run_training(dict(DM_WGRAD_NO_DEFER="1", DM_WGRAD_NO_WINO="1", DM_NO_BATCH_REPACK="1", DM_TRAIN_NO_LANDING_FUSE="1",
                       DM_NO_SMALL_GEMM="1", DM_ATTN_BWD_NO_CACHE="1", DM_LINATTN_NO_KSTATS="1",
                       DM_TRAIN_NO_FINAL_FUSE="1", DM_WGRAD_INIT_VALU="1", DM_LINATTN_BWD_VALU="1",
                       DM_REPACK_ROT_TMP="1"))


def test_attention_backward_tiled_form_on_every_shape():
    """The operator-level attention backward cases and the text-conditional training step (mid_attn + CrossAttention) with the
    tiled kernels forced on the short sequences the LDS-resident kernel normally takes; and the LDS-resident kernel without
    its score cache (the form longer sequences take)."""
    for extra in (dict(DM_ATTN_BWD_NO_CACHE="1"), dict(DM_ATTN_BWD_NO_PAIRS="1")):  # thread-per-query: uncached, cached
        env0 = dict(os.environ, **extra)
        r0 = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_hip_train_ops.py"), "-q", "-x",
                             "-m", "gpu", "-k", "test_attention_bwd", "-p", "no:cacheprovider"],
                            cwd=ROOT, env=env0, capture_output=True, text=True, timeout=600)
        assert r0.returncode == 0, r0.stdout[-4000:] + r0.stderr[-2000:]
    env = dict(os.environ, DM_ATTN_BWD_TILED="1", DM_ATTN_TILED="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_hip_train_ops.py"),
                        os.path.join(ROOT, "tests", "test_hip_train.py"), "-q", "-x", "-m", "gpu", "-k",
                        "test_attention_bwd or test_text_conditional_training_step", "-p", "no:cacheprovider"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-4000:] + r.stderr[-2000:]


def test_training_goldens_grouped_without_winograd():
    """... and the grouped launches with the direct (row-split) 3x3 weight gradient."""
    _run_training(dict(DM_WGRAD_NO_WINO="1"))
