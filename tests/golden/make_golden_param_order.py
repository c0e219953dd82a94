#!/usr/bin/env python3
"""``[name for name, _ in Unet(...).named_parameters()]`` of the REFERENCE for three U-Net variants (build container only):
``python tests/golden/make_golden_param_order.py``  ->  ``tests/golden/param_order.json`` (+ ``api_surface.json``: the public
method names of the reference classes this package mirrors, for tests/test_host_logic.py).

``torch.optim.Adam(model.parameters())`` numbers its state by this order (``Trainer.save`` stores it under ``'opt'``,
DD/denoising_diffusion.py:1006, :1107); ``Unet.optimizer_state_dict()`` must use the same one.  Only DATA is written."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

from make_golden import import_reference  # noqa: E402


def main():
    dd, ddt, _ = import_reference()
    out = {
        "unet_d64": [n for n, _ in dd.Unet(dim=64, dim_mults=(1, 2, 4, 8), channels=3).named_parameters()],
        "unet_d32_selfcond": [n for n, _ in dd.Unet(dim=32, dim_mults=(1, 2), channels=3, self_condition=True).named_parameters()],
        "unet_text_cross": [n for n, _ in ddt.Unet(dim=64, dim_mults=(1, 2, 4, 8), channels=3, text_condition=True,
                                                     use_cross_attn=True).named_parameters()],
        "unet_text_concat": [n for n, _ in ddt.Unet(dim=64, dim_mults=(1, 2, 4, 8), channels=3,
                                                      text_condition=True).named_parameters()],
    }
    with open(os.path.join(HERE, "param_order.json"), "w") as f:
        json.dump(out, f)
    print({k: len(v) for k, v in out.items()})

    # public method / property names of the mirrored classes (everything defined below nn.Module / LightningModule)
    import denoising_diffusion.denoising_diffusion_image_conditional as ddi
    from make_golden_configs import import_ldm

    ae, ld = import_ldm()
    import ldm.models.latent_diffusion_image_conditional as ldi
    import ldm.models.latent_diffusion_text_conditional as ldt

    def methods(c):
        names = set()
        for k in c.__mro__:
            if k.__name__ in ("Module", "LightningModule", "object"):
                break
            names |= {n for n, v in vars(k).items() if callable(v) or isinstance(v, property)}
        return sorted(n for n in names if not n.startswith("__"))

    api = {c.__name__: methods(c) for c in (dd.Unet, dd.DenoisingDiffusion, ddt.TextConditionalDenoisingDiffusion,
                                             ddi.ImageConditionalDenoisingDiffusion, ld.LatentDiffusion,
                                             ldt.TextConditionalLatentDiffusion, ldi.ImageConditionalLatentDiffusion,
                                             ae.VQModel)}
    with open(os.path.join(HERE, "api_surface.json"), "w") as f:
        json.dump(api, f, indent=0)
    print({k: len(v) for k, v in api.items()})


if __name__ == "__main__":
    main()
