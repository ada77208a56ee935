"""Generate golden vectors by running the REAL reference (Ahnsun/merlin) on CPU, fp32.

Run in the build container only (needs /root/reference and transformers):
    python oracle/make_golden.py [tiny] [medium] [full]
Writes tests/golden/<case>.npz.  The reference source never leaves /root/reference: only
inputs and outputs (data) are stored.  Weights are NOT stored; they are regenerated from
merlin_amd/weights.py (identical bits anywhere).

Import recipe: the reference does not import as shipped (loguru and
torchvision missing here; mmgpt/utils/constants.py has an unclosed '(' at line 25), so three
stubs are installed: `loguru`, `torchvision(.transforms(.functional))`, and a
`mmgpt.utils.constants` module exec'd from lines 1-23 of the real file.
"""
from __future__ import annotations

import logging
import os
import sys
import tempfile
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference"
sys.dont_write_bytecode = True

from merlin_amd import weights as W  # noqa: E402
from oracle import cases as C  # noqa: E402


def _install_reference():
    import transformers
    from transformers import CLIPVisionModel, CLIPImageProcessor, LlamaForCausalLM  # noqa: F401 (touch lazies first)

    lg = types.ModuleType("loguru")
    lg.logger = logging.getLogger("ref")
    sys.modules["loguru"] = lg
    for name in ("torchvision", "torchvision.transforms", "torchvision.transforms.functional"):
        m = types.ModuleType(name)

        class InterpolationMode:  # noqa: D401
            BICUBIC = "bicubic"

        m.InterpolationMode = InterpolationMode
        sys.modules[name] = m
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    sys.modules["torchvision.transforms"].functional = sys.modules["torchvision.transforms.functional"]
    sys.path.insert(0, REF)
    import mmgpt  # namespace package
    import mmgpt.utils  # noqa: F401

    const = types.ModuleType("mmgpt.utils.constants")
    with open(os.path.join(REF, "mmgpt/utils/constants.py")) as f:
        head = "".join(f.readlines()[:23])
    exec(compile(head, "constants_head", "exec"), const.__dict__)
    sys.modules["mmgpt.utils.constants"] = const
    from mmgpt.model.mmgpt.llama_mmgpt import MMGPTConfig, MMGPTLlamaForCausalLM

    return transformers, MMGPTConfig, MMGPTLlamaForCausalLM


class _Tok:
    """Minimal tokenizer stand-in for build_vision_tokenizer (base_mmgpt.py:55-63)."""

    def __init__(self, n):
        self.vocab = {i: i for i in range(n)}
        self.n = n
        self.names = {}

    def add_tokens(self, toks, special_tokens=True):
        k = 0
        for t in toks:
            if t not in self.names:
                self.names[t] = self.n
                self.n += 1
                k += 1
        return k

    def __len__(self):
        return self.n

    def convert_tokens_to_ids(self, toks):
        return [self.names[t] for t in toks]


def canonical_name(n: str) -> str:
    """transformers 5.x flattened CLIP names -> the 4.31 / released-checkpoint names."""
    pre = "model.vision_tower.vision_tower."
    if n.startswith(pre) and not n.startswith(pre + "vision_model."):
        n = pre + "vision_model." + n[len(pre):]
    return n


def build_reference_model(cfg, seed=0):
    transformers, MMGPTConfig, MMGPTLlamaForCausalLM = _install_reference()
    from transformers import CLIPVisionConfig, CLIPVisionModel, CLIPImageProcessor

    base_vocab = cfg.vocab_size - 3
    hf = MMGPTConfig(
        vocab_size=base_vocab, hidden_size=cfg.hidden_size, intermediate_size=cfg.intermediate_size,
        num_hidden_layers=cfg.num_hidden_layers, num_attention_heads=cfg.num_attention_heads,
        num_key_value_heads=cfg.num_attention_heads, rms_norm_eps=cfg.rms_norm_eps,
        max_position_embeddings=8192, rope_theta=cfg.rope_theta, tie_word_embeddings=False,
        attn_implementation="eager", use_cache=False,
    )
    tmp = tempfile.mkdtemp(prefix="goldclip_")
    vcfg = CLIPVisionConfig(
        hidden_size=cfg.v_hidden_size, intermediate_size=cfg.v_intermediate_size,
        num_hidden_layers=cfg.v_num_hidden_layers, num_attention_heads=cfg.v_num_attention_heads,
        image_size=cfg.v_image_size, patch_size=cfg.v_patch_size, attn_implementation="eager",
    )
    with torch.device("meta"):
        vm = CLIPVisionModel(vcfg)
    vm = vm.to_empty(device="cpu")
    vm.save_pretrained(tmp)
    CLIPImageProcessor(size={"shortest_edge": cfg.v_image_size}, crop_size=cfg.v_image_size).save_pretrained(tmp)
    with torch.device("meta"):
        model = MMGPTLlamaForCausalLM(hf)
    model = model.to_empty(device="cpu")
    margs = types.SimpleNamespace(
        vision_tower=tmp, vision_select_layer=cfg.vision_select_layer, vision_select_feature=cfg.vision_select_feature,
        freeze_vision_tower=False, conv_stride=cfg.conv_stride, model_name_or_path=tmp, projector=cfg.projector,
        freeze_projector=False, use_im_start_end=True, freeze_lm_model=False,
    )
    dargs = types.SimpleNamespace(use_beam_search=False)
    targs = types.SimpleNamespace(device="cpu")
    tok = _Tok(base_vocab)
    model.build_vision_tokenizer(margs, dargs, targs, tok)
    assert (model.im_patch_token, model.im_start_token, model.im_end_token) == (cfg.im_patch_token, cfg.im_start_token, cfg.im_end_token)
    assert dargs.image_token_len == cfg.num_patches, (dargs.image_token_len, cfg.num_patches)
    model = model.float()
    # fill every parameter from the build's generator (overwrites the resize/mean-init rows too)
    from oracle.ref_cpu import param_shapes

    shapes = param_shapes(cfg)
    seen = set()
    with torch.no_grad():
        for n, p in model.named_parameters():
            cn = canonical_name(n)
            assert cn in shapes and tuple(p.shape) == tuple(shapes[cn]), (n, cn, tuple(p.shape), shapes.get(cn))
            p.copy_(torch.from_numpy(W.generate(cn, tuple(p.shape), seed)))
            seen.add(cn)
    missing = set(shapes) - seen
    assert not missing, missing
    # non-persistent buffers (rotary inv_freq, CLIP position_ids) are lost by to_empty(): rebuild
    for mod in model.modules():
        if hasattr(mod, "inv_freq") and hasattr(mod, "original_inv_freq"):
            hd = cfg.head_dim
            inv = 1.0 / (cfg.rope_theta ** (torch.arange(0, hd, 2, dtype=torch.float32) / hd))
            mod.inv_freq = inv
            mod.original_inv_freq = inv.clone()
        if hasattr(mod, "position_ids") and isinstance(getattr(mod, "position_ids"), torch.Tensor):
            n = mod.position_ids.shape[-1]
            mod.position_ids = torch.arange(n).expand((1, -1))
    model.config.use_cache = False
    return model


def grad_digest(g: np.ndarray) -> dict:
    f = g.reshape(-1).astype(np.float64)
    stride = max(1, f.size // 257)
    return {"norm": np.float64(np.sqrt((f * f).sum())), "sum": np.float64(f.sum()), "head": f[:64].astype(np.float32),
            "strided": f[::stride][:512].astype(np.float32)}


def run_case(name: str, want_grads: bool = True, logits_slice=None):
    cfg, batch = C.get_case(name)
    print(f"[{name}] building reference model ...", flush=True)
    model = build_reference_model(cfg, seed=0)
    model.train()
    out = model(input_ids=batch["input_ids"], attention_mask=batch["attention_mask"], labels=batch["labels"],
                images=batch["images"], return_dict=True)
    logits = out.logits.detach().float().numpy()
    loss = float(out.loss.detach())
    rec = {"loss": np.float64(loss), "input_ids": batch["input_ids"].numpy(), "attention_mask": batch["attention_mask"].numpy(),
           "labels": batch["labels"].numpy()}
    if logits_slice is None:
        rec["logits"] = logits.astype(np.float32)
    else:
        rec["logits_slice"] = logits[logits_slice].astype(np.float32)
        rec["logits_lse"] = torch.logsumexp(out.logits.detach().float(), dim=-1).numpy().astype(np.float32)
        rec["logits_absmax"] = np.float64(np.abs(logits).max())
    for i, im in enumerate(batch["images"]):
        if im.numel() <= 4 * 3 * 56 * 56:
            rec[f"image_{i}"] = im.numpy()
    if want_grads:
        out.loss.backward()
        for n, p in model.named_parameters():
            cn = canonical_name(n)
            if p.grad is None:
                continue
            for k, v in grad_digest(p.grad.numpy()).items():
                rec[f"grad/{cn}/{k}"] = v
    path = os.path.join(ROOT, "tests", "golden", f"{name}.npz")
    np.savez_compressed(path, **rec)
    print(f"[{name}] loss={loss:.6f} logits{logits.shape} |max|={np.abs(logits).max():.4f} -> {path} ({os.path.getsize(path)/1e3:.0f} kB)", flush=True)


if __name__ == "__main__":
    torch.manual_seed(0)
    groups = sys.argv[1:] or ["tiny"]
    for g in groups:
        if g == "tiny":
            for n in C.TINY_CASES:
                run_case(n)
        elif g == "medium":
            run_case("medium_cfg1", want_grads=True, logits_slice=(slice(None), slice(None, None, 8), slice(0, 512)))
        elif g == "released":
            run_case("released_conv448", want_grads=True, logits_slice=(slice(None), slice(None, None, 4), slice(0, 512)))
        elif g == "full":
            run_case("full_cfg1", want_grads=False, logits_slice=(slice(None), slice(None, None, 16), slice(0, 256)))
        else:
            run_case(g)
