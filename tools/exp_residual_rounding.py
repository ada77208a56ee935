"""Attribution experiment (CPU, full 7B cfg 1): which 16-bit roundings account for the full-depth logits error of the
HIP path?  Runs the fp32 oracle with (a) nothing rounded, (b) only the decoder's residual stream rounded to fp16 after
each add, (c) only Linear outputs rounded to fp16 (residual stream fp32), (d) both.  Reports max|d|/max|ref| on the
golden slice.  Build-container only (28 GB RAM)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import torch.nn.functional as F
from oracle import cases as C, ref_cpu as R

torch.set_num_threads(8)
cfg, batch = C.get_case("full_cfg1")
t0 = time.time(); P = R.make_params(cfg, seed=0); print("params", time.time() - t0, flush=True)
g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "full_cfg1.npz"))
print({k: g[k].shape for k in g.files}, flush=True)
DT = torch.float16
r16 = lambda t: t.to(DT).float()
orig_linear = F.linear

def llama_forward(P, cfg, x, attention_mask, round_resid):
    B, S, _ = x.shape
    cos, sin = R.rope_tables(S, cfg.head_dim, cfg.rope_theta)
    neg = torch.finfo(torch.float32).min
    add_mask = torch.full((S, S), neg).triu(1)[None, None].expand(B, 1, S, S)
    rr = r16 if round_resid else (lambda t: t)
    x = rr(x)
    for i in range(cfg.num_hidden_layers):
        p = f"model.layers.{i}."
        x = rr(x + R.llama_attention(P, cfg, p + "self_attn.", R.rms_norm(x, P[p + "input_layernorm.weight"], cfg.rms_norm_eps), cos, sin, add_mask))
        h = R.rms_norm(x, P[p + "post_attention_layernorm.weight"], cfg.rms_norm_eps)
        h = F.silu(R.F.linear(h, P[p + "mlp.gate_proj.weight"])) * R.F.linear(h, P[p + "mlp.up_proj.weight"])
        x = rr(x + R.F.linear(h, P[p + "mlp.down_proj.weight"]))
    return R.rms_norm(x, P["model.norm.weight"], cfg.rms_norm_eps)

def run(round_resid, round_linear):
    R.F.linear = (lambda x, w, b=None: r16(orig_linear(x, w, b))) if round_linear else orig_linear
    try:
        with torch.no_grad():
            feats = R.encode_images(P, cfg, batch["images"])
            x = R.splice_image_features(P, cfg, batch["input_ids"], feats)
            h = llama_forward(P, cfg, x, batch["attention_mask"], round_resid)
            logits = orig_linear(h, P["lm_head.weight"])
    finally:
        R.F.linear = orig_linear
    return logits

ref = None
for name, rr, rl in (("fp32", False, False), ("resid16", True, False), ("linear16", False, True), ("both16", True, True)):
    t0 = time.time(); lg = run(rr, rl)
    sl = lg[0, ::16, :256].numpy()
    if ref is None:
        ref = lg
        print(name, "vs golden slice:", float(np.abs(sl - g["logits_slice"]).max() / np.abs(g["logits_slice"]).max()) if "logits_slice" in g.files else "n/a", time.time() - t0, flush=True)
    else:
        print(name, "max|d|/max|ref| =", float((lg - ref).abs().max() / ref.abs().max()), time.time() - t0, flush=True)
