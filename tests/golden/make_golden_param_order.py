#!/usr/bin/env python3
"""``[name for name, _ in Unet(...).named_parameters()]`` of the REFERENCE for three U-Net variants (build container only):
``python tests/golden/make_golden_param_order.py``  ->  ``tests/golden/param_order.json``.

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


if __name__ == "__main__":
    main()
