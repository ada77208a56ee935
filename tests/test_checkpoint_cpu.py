"""Checkpoint I/O in the reference's layout (SURVEY §8b state-dict keys, §8f N4): the builder flow
`from_pretrained(path)` + `build_vision_tokenizer(model_args with model_name_or_path=path)` (builder.py:70-74,135-140;
clip_encoder.py:26-62; base_projector.py:12-48) must restore every tensor written by `save_pretrained`, single-file
and sharded; key names are the released checkpoint's."""
import json
import os
import types

import pytest
import torch

from merlin_amd import checkpoint as CK


def _tiny(projector="conv", stride=2):
    from oracle import cases as C
    from merlin_amd.model.llama_mmgpt import build_synthetic_model

    cfg, _ = C.get_case("tiny_conv2" if projector == "conv" else "tiny_1img")
    llama = dict(vocab_size=cfg.vocab_size - 3, hidden_size=cfg.hidden_size, intermediate_size=cfg.intermediate_size,
                 num_hidden_layers=cfg.num_hidden_layers, num_attention_heads=cfg.num_attention_heads, rms_norm_eps=cfg.rms_norm_eps,
                 rope_theta=cfg.rope_theta, max_position_embeddings=8192)
    vision = dict(hidden_size=cfg.v_hidden_size, intermediate_size=cfg.v_intermediate_size, num_hidden_layers=cfg.v_num_hidden_layers,
                  num_attention_heads=cfg.v_num_attention_heads, image_size=cfg.v_image_size, patch_size=cfg.v_patch_size,
                  layer_norm_eps=cfg.v_layer_norm_eps)
    m = build_synthetic_model(llama, vision, projector=cfg.projector, conv_stride=cfg.conv_stride, dtype=torch.bfloat16, device="cpu", seed=0)
    return m, cfg, vision


@pytest.mark.parametrize("sharded", [False, True])
@pytest.mark.parametrize("projector", ["conv", "mlp"])
def test_save_then_builder_flow_restores_everything(tmp_path, sharded, projector):
    from merlin_amd.model.llama_mmgpt import MMGPTLlamaForCausalLM, _SynthTokenizer

    m, cfg, vision = _tiny(projector)
    path = str(tmp_path / "ckpt")
    m.config.save_pretrained(path)
    CK.save_state_dict(m, path, max_shard_numel=(200_000 if sharded else 2_500_000_000))
    files = sorted(os.listdir(path))
    if sharded:
        idx = json.load(open(os.path.join(path, "pytorch_model.bin.index.json")))
        assert set(idx["weight_map"]) == set(m.state_dict())
        assert len({f for f in idx["weight_map"].values()}) > 1
    else:
        assert "pytorch_model.bin" in files
    sd = m.state_dict()
    # the released checkpoint's key names
    for k in ("model.embed_tokens.weight", "model.layers.0.self_attn.q_proj.weight", "model.layers.1.mlp.down_proj.weight",
              "model.layers.0.post_attention_layernorm.weight", "model.norm.weight", "lm_head.weight",
              "model.vision_tower.vision_tower.vision_model.embeddings.class_embedding",
              "model.vision_tower.vision_tower.vision_model.embeddings.patch_embedding.weight",
              "model.vision_tower.vision_tower.vision_model.embeddings.position_embedding.weight",
              "model.vision_tower.vision_tower.vision_model.pre_layrnorm.weight",
              "model.vision_tower.vision_tower.vision_model.encoder.layers.0.self_attn.out_proj.bias",
              "model.vision_tower.vision_tower.vision_model.encoder.layers.2.mlp.fc2.weight",
              "model.vision_tower.vision_tower.vision_model.post_layernorm.bias",
              "model.projector.projector.weight", "model.projector.projector.bias"):
        assert k in sd, k
    # builder flow
    m2 = MMGPTLlamaForCausalLM.from_pretrained(path, torch_dtype=torch.bfloat16)
    margs = types.SimpleNamespace(vision_tower="synthetic", vision_select_layer=-2, vision_select_feature="patch", freeze_vision_tower=False,
                                  conv_stride=cfg.conv_stride, model_name_or_path=path, projector=cfg.projector, freeze_projector=False,
                                  use_im_start_end=True, freeze_lm_model=False)
    dargs = types.SimpleNamespace()
    # the checkpoint's own tokenizer already holds the three special tokens (add_tokens then returns 0 and the
    # embedding rows are NOT re-initialised, base_mmgpt.py:60-76)
    from merlin_amd.model.llama_mmgpt import DEFAULT_IM_PATCH_TOKEN, DEFAULT_IM_START_TOKEN, DEFAULT_IM_END_TOKEN
    tok = _SynthTokenizer(cfg.vocab_size - 3)
    tok.add_tokens([DEFAULT_IM_PATCH_TOKEN, DEFAULT_IM_START_TOKEN, DEFAULT_IM_END_TOKEN])
    from merlin_amd.model.config import CLIPVisionConfig
    m2.build_vision_tokenizer(margs, dargs, None, tok, vision_config=CLIPVisionConfig(**vision))
    sd2 = m2.state_dict()
    assert set(sd2) == set(sd)
    for k in sd:
        assert sd2[k].shape == sd[k].shape and torch.equal(sd2[k], sd[k]), k
    assert dargs.image_token_len == (cfg.v_image_size // cfg.v_patch_size // cfg.conv_stride) ** 2
    assert (m2.im_patch_token, m2.im_start_token, m2.im_end_token) == (cfg.im_patch_token, cfg.im_start_token, cfg.im_end_token)


def test_interpolate_pos_embed():
    g, d = 4, 8
    torch.manual_seed(0)
    pe = torch.randn(1 + g * g, d)
    out = CK.interpolate_pos_embed(pe, factor=2)
    assert out.shape == (1 + (2 * g) ** 2, d)
    assert torch.equal(out[0], pe[0])                         # class-token row kept
    grid_old, grid_new = pe[1:].view(g, g, d), out[1:].view(2 * g, 2 * g, d)
    for (a, b) in (((0, 0), (0, 0)), ((0, g - 1), (0, 2 * g - 1)), ((g - 1, 0), (2 * g - 1, 0)), ((g - 1, g - 1), (2 * g - 1, 2 * g - 1))):
        assert torch.allclose(grid_old[a], grid_new[b], atol=1e-5)   # align_corners=True keeps the corners
    import torch.nn.functional as F
    ref = F.interpolate(pe[1:].view(1, g, g, d).permute(0, 3, 1, 2), size=(2 * g, 2 * g), mode="bicubic", align_corners=True)
    assert torch.allclose(grid_new, ref[0].permute(1, 2, 0), atol=1e-6)
    bf = CK.interpolate_pos_embed(pe.bfloat16(), factor=2)
    assert bf.dtype == torch.bfloat16 and bf.shape == out.shape
