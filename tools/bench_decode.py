"""Decode-step timing of the KV-cache path at Llama-7B size (run on the GPU box): prefill a cfg-3 style prompt, then time
decode steps eagerly and (optionally) as a replayed HIP graph.  Reports ms/token and the HBM rate the step sustains
(weights + K/V cache streamed once per step)."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from merlin_amd.model.llama_mmgpt import build_synthetic_model
from merlin_amd import synth

from merlin_amd import ops as _O
if os.environ.get("MH_GEMV_MFMA_MIN"):  # A/B: 17 = never use the MFMA GEMV
    _O.gemv_mfma_min_rows(int(os.environ["MH_GEMV_MFMA_MIN"]))
if os.environ.get("MH_GEMV_PAIR_MIN"):  # A/B: "rows16,rows_fp8" from which the SwiGLU / RoPE projections use the MFMA form
    _O.gemv_mfma_pair_min_rows(*[int(a) for a in os.environ["MH_GEMV_PAIR_MIN"].split(",")])
if os.environ.get("MH_GEMV_MFMA_WIDE"):  # A/B: 0 = 8 waves per block in the small-N MFMA GEMV
    _O.gemv_mfma_wide(os.environ["MH_GEMV_MFMA_WIDE"] != "0")
if os.environ.get("MH_GEMV_KSPLIT"):  # A/B: 0 = one wave per row pair in the small-N GEMV
    _O.gemv_ksplit(os.environ["MH_GEMV_KSPLIT"] != "0")
if os.environ.get("MH_DECODE_FUSED_MERGE"):  # A/B: 1 = split-KV partials merged by the last block of a (b, h) instead of a second launch
    _O.attn_decode_fused_merge(os.environ["MH_DECODE_FUSED_MERGE"] != "0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
S = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
NEW = 160
dev = torch.device("cuda:0")
llama = dict(vocab_size=32000, hidden_size=4096, intermediate_size=11008, num_hidden_layers=32, num_attention_heads=32,
             rms_norm_eps=1e-6, rope_theta=10000.0, max_position_embeddings=8192)
vision = dict(hidden_size=1024, intermediate_size=4096, num_hidden_layers=24, num_attention_heads=16, image_size=336, patch_size=14, layer_norm_eps=1e-5)
model = build_synthetic_model(llama, vision, projector="mlp", conv_stride=1, dtype=torch.bfloat16, device="cuda", seed=0)
g = torch.Generator().manual_seed(0)
ids = torch.randint(3, 32000, (B, S), generator=g).to(dev)
with torch.no_grad():
    t0 = time.time()
    logits, cache = model.engine.prefill(ids, None, None, NEW + 8)
    torch.cuda.synchronize()
    print(f"prefill B={B} S={S}: {(time.time()-t0)*1e3:.1f} ms (first call, incl. arena setup)", flush=True)
    tok = logits.argmax(-1)
    for _ in range(3):
        logits = model.engine.decode_step(tok, cache); tok = logits.argmax(-1)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    n = 32
    for _ in range(n):
        logits = model.engine.decode_step(tok, cache); tok = logits.argmax(-1)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    wbytes = sum(p.numel() for n_, p in model.named_parameters() if n_.startswith("model.layers") or n_.startswith("lm_head") or n_ == "model.norm.weight") * 2
    cbytes = 2 * 32 * B * (S + 20) * 4096 * 2
    print(json.dumps({"decode": "eager", "B": B, "context": S, "ms_per_step": round(ms, 3), "tokens_per_s": round(B * 1e3 / ms, 1),
                      "hbm_gb_per_step": round((wbytes + cbytes) / 1e9, 2), "hbm_tb_s": round((wbytes + cbytes) / ms / 1e9, 2)}), flush=True)
    # the same step as one replayed HIP graph
    g, gtok, glog = model.engine.capture_decode_graph(cache)
    gtok.copy_(tok)
    for _ in range(3):
        g.replay(); gtok.copy_(glog.argmax(-1))
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        g.replay(); gtok.copy_(glog.argmax(-1))
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print(json.dumps({"decode": "hip-graph", "B": B, "context": S, "ms_per_step": round(ms, 3), "tokens_per_s": round(B * 1e3 / ms, 1),
                      "hbm_gb_per_step": round((wbytes + cbytes) / 1e9, 2), "hbm_tb_s": round((wbytes + cbytes) / ms / 1e9, 2)}), flush=True)
    # fp8 weights (e4m3, per-128-block scales), graph replay
    model.engine.quantize_decode_weights()
    g8, gtok8, glog8 = model.engine.capture_decode_graph(cache, fp8=True)
    gtok8.copy_(tok)
    for _ in range(3):
        g8.replay(); gtok8.copy_(glog8.argmax(-1))
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        g8.replay(); gtok8.copy_(glog8.argmax(-1))
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    w8 = wbytes // 2 + wbytes // 2 // 64  # fp8 bytes + fp32 scales (1 per 128)
    print(json.dumps({"decode": "hip-graph fp8 weights", "B": B, "context": S, "ms_per_step": round(ms, 3), "tokens_per_s": round(B * 1e3 / ms, 1),
                      "hbm_gb_per_step": round((w8 + cbytes) / 1e9, 2), "hbm_tb_s": round((w8 + cbytes) / ms / 1e9, 2)}), flush=True)
