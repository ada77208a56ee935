"""Auto-class registration (llama_mmgpt.py:27-28,137-138 of the reference): a checkpoint directory whose config.json says
model_type "mmgpt" resolves through transformers' AutoConfig / AutoModelForCausalLM to merlin_amd's classes, and the config
is a real LlamaConfig."""
import json
import os

import torch


def test_auto_classes_resolve_mmgpt(tmp_path):
    from transformers import AutoConfig, AutoModelForCausalLM, LlamaConfig

    from merlin_amd import hf_compat
    from merlin_amd.model.llama_mmgpt import MMGPTLlamaForCausalLM as HipModel

    d = str(tmp_path)
    cfg = dict(model_type="mmgpt", architectures=["MMGPTLlamaForCausalLM"], vocab_size=103, hidden_size=64, intermediate_size=128,
               num_hidden_layers=2, num_attention_heads=2, num_key_value_heads=2, rms_norm_eps=1e-5, max_position_embeddings=512)
    json.dump(cfg, open(os.path.join(d, "config.json"), "w"))
    c = AutoConfig.from_pretrained(d)
    assert isinstance(c, hf_compat.MMGPTConfig) and isinstance(c, LlamaConfig) and c.model_type == "mmgpt"
    # a reference-layout weight file next to it
    m0 = hf_compat.MMGPTLlamaForCausalLM(c)
    with torch.no_grad():
        for i, p in enumerate(m0.parameters()):
            p.copy_(torch.full_like(p, 0.001 * (i + 1)))
    torch.save(m0.state_dict(), os.path.join(d, "pytorch_model.bin"))
    m = AutoModelForCausalLM.from_pretrained(d)
    assert isinstance(m, HipModel) and type(m).__name__ == "MMGPTLlamaForCausalLM" and m.config is not None
    assert isinstance(m.config, LlamaConfig) and m.config.rms_norm_eps == 1e-5
    assert set(m.state_dict()) == set(m0.state_dict())
    for k, v in m0.state_dict().items():
        assert torch.equal(m.state_dict()[k], v), k
    assert m.get_model().layers[1].mlp.down_proj.weight.shape == (64, 128)
    # config round trip in the reference's format
    m.config.save_pretrained(d)
    assert json.load(open(os.path.join(d, "config.json")))["model_type"] == "mmgpt"
    # builder.py:98: resize through the HF config object
    m.resize_token_embeddings(106)
    assert m.config.vocab_size == 106 and m.lm_head.weight.shape[0] == 106


def test_plain_config_twin_has_the_same_fields():
    from merlin_amd import hf_compat
    from merlin_amd.model.config import MMGPTConfig as Plain
    from merlin_amd.model.llama_mmgpt import rope_theta_of

    kw = dict(vocab_size=103, hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=2, rms_norm_eps=1e-5)
    a, b = Plain(**kw), hf_compat.MMGPTConfig(**kw)
    for f in ("vocab_size", "hidden_size", "intermediate_size", "num_hidden_layers", "num_attention_heads", "rms_norm_eps", "head_dim",
              "eos_token_id", "bos_token_id", "use_cache", "use_return_dict"):
        assert getattr(a, f) == getattr(b, f), f
    assert rope_theta_of(a) == rope_theta_of(b) == 10000.0
