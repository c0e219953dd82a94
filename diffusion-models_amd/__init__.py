"""MI355X-native sampling path for lbarseghyan/diffusion-models.

The directory name carries a hyphen (it is the name the build contract asks
for); import it as ``diffusion_models_amd`` through the shim at the repo root.
"""
from .spec import (  # noqa: F401
    DecoderConfig,
    UnetConfig,
    ddim_time_pairs,
    decoder_param_spec,
    encoder_param_spec,
    make_schedule,
    unet_param_spec,
)
from .synth import synth_state_dict, synth_tensor  # noqa: F401
from .unet import Unet  # noqa: F401
from .diffusion import (  # noqa: F401
    DenoisingDiffusion,
    ImageConditionalDenoisingDiffusion,
    ImageConditionalLatentDiffusion,
    LatentDiffusion,
    ModelPrediction,
    TextConditionalDenoisingDiffusion,
    TextConditionalLatentDiffusion,
)
from .vae import VQDecoder, VQEncoder, VQModel  # noqa: F401
from .dist import gather_shards, sample_global, sample_sharded, shard_bounds, shared_seed  # noqa: F401
from .checkpoint import load_trainer_checkpoint, load_vae_checkpoint  # noqa: F401
from .train import EMA, diffusion_state_dict, load_checkpoint, save_checkpoint, train_step  # noqa: F401
from .inception import FIDEvaluation, InceptionScoreEvaluation, InceptionV3, calculate_frechet_distance  # noqa: F401
from .inception_spec import inception_param_spec  # noqa: F401

__all__ = [
    "Unet",
    "DenoisingDiffusion",
    "TextConditionalDenoisingDiffusion",
    "ImageConditionalDenoisingDiffusion",
    "ImageConditionalLatentDiffusion",
    "LatentDiffusion",
    "TextConditionalLatentDiffusion",
    "VQDecoder",
    "VQEncoder",
    "VQModel",
    "InceptionV3",
    "FIDEvaluation",
    "InceptionScoreEvaluation",
    "sample_sharded",
    "sample_global",
    "gather_shards",
    "shard_bounds",
    "UnetConfig",
    "DecoderConfig",
    "unet_param_spec",
    "decoder_param_spec",
    "make_schedule",
    "ddim_time_pairs",
    "synth_state_dict",
    "synth_tensor",
]
