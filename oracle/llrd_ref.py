"""TEST INFRASTRUCTURE (oracle): restatement of the reference's torch param-group construction
(mmgpt/utils/llrd_utils.py:26-79, get_param_groups), group order included.  Pinned to goldens captured from the reference
(oracle/make_llrd_golden.py -> tests/golden/llrd_groups.json); used by tests to drive torch.optim.AdamW as the comparison for the
fused arena optimizer.  Never imported by merlin_amd/."""


def param_groups(named_parameters, lr, weight_decay, lr_scale_fn=None):
    """The reference's grouping (llrd_utils.py:26-79): trainable parameters split by (decayed or not: biases and 1-D
    tensors are not) x (lr multiplier), in the reference's group order.  Returns a list of dicts
    {"names": [...], "weight_decay": wd, "lr": lr * mult}."""
    wd_plain, wd_scaled, nowd_plain, nowd_scaled = [], {}, [], {}
    for name, prm in named_parameters:
        if not prm.requires_grad:
            continue
        no_wd = name.endswith(".bias") or prm.dim() == 1
        mult = lr_scale_fn(name) if lr_scale_fn is not None else 1
        scaled = mult != 1
        if not no_wd and not scaled:
            wd_plain.append(name)
        elif not no_wd:
            wd_scaled.setdefault(mult, []).append(name)
        elif not scaled:
            nowd_plain.append(name)
        else:
            nowd_scaled.setdefault(mult, []).append(name)
    groups = []
    if wd_plain:
        groups.append({"names": wd_plain, "weight_decay": weight_decay, "lr": lr})
    for mult, names in wd_scaled.items():
        groups.append({"names": names, "weight_decay": weight_decay, "lr": lr * mult})
    if nowd_plain:
        groups.append({"names": nowd_plain, "weight_decay": 0.0, "lr": lr})
    for mult, names in nowd_scaled.items():
        groups.append({"names": names, "weight_decay": 0.0, "lr": lr * mult})
    return groups
