"""`inference(groups, model, device)` with the reference's signature and return value
(/root/reference/src/dust3r/inference.py:219-239): moves every tensor of every view dict to `device` (mutating the
dicts, as the reference does), runs the model with ret_state=True and returns (dict(views=..., pred=...), state_args).
"""
from __future__ import annotations

import torch

_IGNORE = {"depthmap", "dataset", "label", "instance", "idx", "true_shape", "rng"}


@torch.no_grad()
def inference(groups, model, device, verbose=False):
    for view in groups:
        for name in view.keys():
            if name in _IGNORE:
                continue
            if isinstance(view[name], (tuple, list)):
                view[name] = [x.to(device, non_blocking=True) for x in view[name]]
            else:
                view[name] = view[name].to(device, non_blocking=True)
    if verbose:
        print(f">> Inference with model on {len(groups)} image/raymaps")
    output, state_args = model(groups, ret_state=True)
    return dict(views=output.views, pred=output.ress), state_args
