"""End-to-end generate() at Llama-7B size: greedy, NEW tokens after a P-token prompt, with the decode step replayed as a HIP graph
(generate's default) or launched eagerly - what a caller of the reference's eval scripts sees (run on the GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from merlin_amd.model.llama_mmgpt import build_synthetic_model

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
P = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
NEW = 128
llama = dict(vocab_size=32000, hidden_size=4096, intermediate_size=11008, num_hidden_layers=32, num_attention_heads=32,
             rms_norm_eps=1e-6, rope_theta=10000.0, max_position_embeddings=8192)
vision = dict(hidden_size=1024, intermediate_size=4096, num_hidden_layers=24, num_attention_heads=16, image_size=336, patch_size=14, layer_norm_eps=1e-5)
model = build_synthetic_model(llama, vision, projector="mlp", conv_stride=1, dtype=torch.bfloat16, device="cuda", seed=0)
ids = torch.randint(3, 32000, (B, P), generator=torch.Generator().manual_seed(0)).cuda()
outs = {}
for rep in range(2):
    for graph in (True, False):
        for fp8 in (False, True):
            torch.cuda.synchronize(); t0 = time.time()
            out = model.generate(ids, max_new_tokens=NEW, eos_token_id=-1, use_graph=graph, fp8_weights=fp8)
            torch.cuda.synchronize(); dt = time.time() - t0
            outs[(graph, fp8)] = out
            if rep:
                print(f"B={B} prompt {P} + {NEW} new, graph={graph} fp8_weights={fp8}: {dt * 1e3:.0f} ms total, {(dt * 1e3) / NEW:.2f} ms/token incl. prefill", flush=True)
print("graph == eager tokens:", bool((outs[(True, False)] == outs[(False, False)]).all()))
