"""transformers Auto-class registration for the HIP model (SURVEY §8a A14, §8b "Construction").

The reference does, at import time of its model module (mmgpt/model/mmgpt/llama_mmgpt.py:27-28, 137-138):

    class MMGPTConfig(LlamaConfig): model_type = "mmgpt"
    AutoConfig.register("mmgpt", MMGPTConfig)
    AutoModelForCausalLM.register(MMGPTConfig, MMGPTLlamaForCausalLM)

so that `AutoConfig.from_pretrained(ckpt)` / `AutoModelForCausalLM.from_pretrained(ckpt)` resolve checkpoints whose config.json
says `"model_type": "mmgpt"` (the released Kangheng/Merlin weights, README.md:79-80).  Importing THIS module does the same
for merlin_amd: `MMGPTConfig` here IS a `transformers.LlamaConfig` (usable wherever a LlamaConfig is expected: HF Trainer,
generation config plumbing, save_pretrained), and the registered model class is the HIP-engine model, which reads its
hyper-parameters from either config flavour.  The hot path itself never imports transformers (merlin_amd/model/config.py is
the dependency-free twin used by bench.py / synthetic runs)."""
from __future__ import annotations

from transformers import AutoConfig, AutoModelForCausalLM, LlamaConfig

from .model.llama_mmgpt import MMGPTLlamaForCausalLM as _HipModel


class MMGPTConfig(LlamaConfig):
    model_type = "mmgpt"


class MMGPTLlamaForCausalLM(_HipModel):
    """The HIP-engine model under the reference's class name with the HF config class attached (what the Auto classes need)."""

    config_class = MMGPTConfig


AutoConfig.register("mmgpt", MMGPTConfig, exist_ok=True)
AutoModelForCausalLM.register(MMGPTConfig, MMGPTLlamaForCausalLM, exist_ok=True)
