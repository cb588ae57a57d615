"""Where does a batch stop being bit-equal to its images run alone?  (GPU box; diagnostic.)

Runs TransVAE-Large's encoder on 2 images and on the first image alone, reports the first stage whose output differs, then
takes the stage-2 block apart (row norm, QKV with and without the RoPE epilogue, attention branch, FFN branch) per tile
configuration and per epilogue mode (register forms vs the generic LDS loop).  Written when an fma contraction that differed
between two tile-shape instantiations of the RoPE epilogue broke tests/test_model_gpu.py::test_large_f16d32_256_full_size_
properties_and_oracle."""
import os, sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/deepl-project_amd")
import bench
from transvae import TransVAE
from transvae.hip import _lib as L, ops
DEV = "cuda:0"
with torch.device(DEV):
    m = TransVAE(variant="large", compression_ratio=16, latent_dim=32)
bench.init_scaled_(m, seed=3)
m.eval()
g = torch.Generator().manual_seed(11)
x = torch.rand(2, 3, 256, 256, generator=g).to(DEV)
t = {}
with torch.no_grad():
    m.encoder.forward_nhwc(x, taps=t)
h = t["encoder.stages.2.0"]            # [2, 64, 64, 384] input of block 2.1
blk = m.encoder.stages[2][1]
lib = L.load()
bf = torch.bfloat16
tok2 = h.reshape(2 * 4096, 384).contiguous()
tok1 = tok2[:4096].contiguous()
def cmp(name, f):
    with torch.no_grad():
        a2, a1 = f(tok2), f(tok1)
    same = torch.equal(a2[: a1.shape[0]], a1)
    d = (a2[: a1.shape[0]].float() - a1.float()).abs().max().item()
    print(f"{name:40s} {'same' if same else 'DIFF max abs %.4g' % d}", flush=True)
w = torch.randn(1536, 384, device=DEV) * 384 ** -0.5
b = torch.randn(1536, device=DEV) * 0.1
wq = torch.randn(1152, 384, device=DEV) * 384 ** -0.5
w2 = torch.randn(384, 384, device=DEV) * 384 ** -0.5
for cfg in ((0, 0, 0, 0), (256, 192, 0, 0), (128, 128, 0, 0), (256, 256, 0, 0)):
    lib.tv_set_igemm_config(*cfg)
    print("config", cfg)
    cmp("linear 384->1536 plain", lambda tk: ops.conv_forward(tk, w, b, None, "linear", L.ACT_NONE, False)[0])
    cmp("linear 384->1536 gelu", lambda tk: ops.conv_forward(tk, w, b, None, "linear", L.ACT_GELU, False)[0])
    cmp("linear 384->384 +res", lambda tk: ops.conv_forward(tk, w2, None, tk, "linear", L.ACT_NONE, False)[0])
    cmp("linear 384->1152 plain", lambda tk: ops.conv_forward(tk, wq, None, None, "linear", L.ACT_NONE, False)[0])
lib.tv_set_igemm_config(0, 0, 0, 0)
h1 = h[:1].contiguous()
def both(f):
    with torch.no_grad():
        return f(h, 2), f(h1, 1)
def report(name, a2, a1):
    a2 = a2.reshape(2, -1)[:1]; a1 = a1.reshape(1, -1)
    same = torch.equal(a2, a1)
    print(f"{name:32s} {'same' if same else 'DIFF max abs %.4g' % (a2.float() - a1.float()).abs().max().item()}", flush=True)
att, ffn = blk.attn, blk.ffn
from transvae.hip import fused
def attn_parts(hh, B):
    t = hh.view(B * 4096, 384)
    w, b = att.folded_qkv()
    tab = att.rope.table(64, 64)
    r = fused.rownorm_fwd(t, blk.norm1.weight, 1, blk.norm1.eps, att.norm_q.eps)
    qkv_plain = ops.conv_forward(r, w, b, None, "linear", L.ACT_NONE, False)[0]
    qkv_rope = ops.conv_forward(r, w, b, None, "linear", L.ACT_NONE, False, rope=(tab, 4096, 768))[0]
    return r, qkv_plain, qkv_rope
(r2, qp2, qr2), (r1, qp1, qr1) = both(attn_parts)
report("rownorm", r2, r1); report("qkv plain", qp2, qp1); report("qkv rope", qr2, qr1)
with torch.no_grad():
    for cfg in ((256, 192, 0, 0), (128, 128, 0, 0)):
        lib.tv_set_igemm_config(*cfg)
        (_, _, a2), (_, _, a1) = both(attn_parts)
        report("qkv rope cfg %s" % (cfg,), a2, a1)
        lib.tv_set_igemm_epilogue(0)
        (_, _, g2), (_, _, g1) = both(attn_parts)
        lib.tv_set_igemm_epilogue(1)
        report("   generic epilogue", g2, g1)
        print("   register form == generic form (B=2):", torch.equal(a2, g2), " (B=1):", torch.equal(a1, g1))
    lib.tv_set_igemm_config(0, 0, 0, 0)
    t2 = att.forward_tokens(h.view(-1, 384), 2, 64, 64, blk.norm1.weight, blk.norm1.eps)
    t1 = att.forward_tokens(h1.view(-1, 384), 1, 64, 64, blk.norm1.weight, blk.norm1.eps)
    report("attention branch", t2, t1)
    f2 = ffn.forward_tokens(t2, 2, 64, 64, blk.norm2.weight, blk.norm2.eps)
    f1 = ffn.forward_tokens(t2[:4096].contiguous(), 1, 64, 64, blk.norm2.weight, blk.norm2.eps)
    report("ffn branch (same input)", f2, f1)
