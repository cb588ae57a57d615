"""Host-side (no GPU) checks of the drop-in boundary: constructor styles, state_dict schema,
parameter counts, error behaviour (R/test_installation.py:10-27,144-175)."""
import json
import os

import pytest
import torch

from oracle import filler
from oracle import transvae_oracle as O


@pytest.fixture(scope="module")
def schemas(golden_dir):
    with open(os.path.join(golden_dir, "state_dict_schemas.json")) as f:
        return json.load(f)


def test_import_surface():
    import transvae
    assert hasattr(transvae, "TransVAE") and hasattr(transvae, "create_transvae")


def test_config_style_constructor_matches_reference_schema(schemas):
    from transvae import TransVAE
    m = TransVAE(config=dict(O.MICRO), variant="micro", compression_ratio=16, latent_dim=4, input_channels=3,
                 use_rope=True, use_conv_ffn=True, use_dc_path=True)  # the keyword set of R/train.py:116-124
    sd = m.state_dict()
    assert list(sd) == list(schemas["micro"])
    for k, s in schemas["micro"].items():
        assert tuple(sd[k].shape) == tuple(s), k
    assert (m.variant, m.compression_ratio, m.latent_dim) == ("micro", 16, 4)


@pytest.mark.parametrize("variant,key", [("tiny", "tiny_f16d32"), ("large", "large_f16d32")])
def test_variant_style_constructor(schemas, golden_dir, variant, key):
    from transvae import TransVAE, create_transvae
    with torch.device("meta"):
        m = TransVAE(variant=variant, compression_ratio=16, latent_dim=32)  # README / test_installation style
        m2 = create_transvae(variant=variant)
    assert list(m.state_dict()) == list(schemas[key]) == list(m2.state_dict())
    for k, s in schemas[key].items():
        assert tuple(m.state_dict()[k].shape) == tuple(s), k
    with open(os.path.join(golden_dir, "param_counts.json")) as f:
        counts = json.load(f)
    n = m.get_num_params()
    assert n["total"] == counts[key] and n["encoder"] + n["decoder"] < n["total"]


def test_f8_variant_has_three_downsamples():
    from transvae import TransVAE
    with torch.device("meta"):
        m = TransVAE(variant="large", compression_ratio=8, latent_dim=16)
    assert len(m.encoder.downsamples) == 3 and m.conv_mu.out_channels == 16


def test_unknown_variant_raises_value_error():
    from transvae import TransVAE
    with pytest.raises(ValueError, match="Unknown variant"):
        TransVAE(variant="small", compression_ratio=16, latent_dim=32)


def test_state_dict_round_trip_and_public_attributes():
    from transvae import TransVAE
    m = TransVAE(config=dict(O.MICRO), variant="micro", latent_dim=4)
    sd = filler.fill_state_dict(O.state_dict_schema(O.MICRO, latent_dim=4))
    missing = m.load_state_dict(sd)
    assert not missing.missing_keys and not missing.unexpected_keys
    out = m.state_dict()
    for k, v in sd.items():
        assert torch.equal(out[k], v), k
    assert m.get_last_layer() is m.decoder.conv_out.weight
    assert len(list(m.encoder.parameters())) > 0
    m.enable_gradient_checkpointing()
    assert m.encoder.gradient_checkpointing and m.decoder.gradient_checkpointing
    # conv weights are stored channels_last (== the kernels' [Cout,KH,KW,Cin]) without changing the schema
    assert m.encoder.stages[0][0].conv1.weight.permute(0, 2, 3, 1).is_contiguous()


def test_init_distributions_follow_reference():
    from transvae import TransVAE
    torch.manual_seed(0)
    m = TransVAE(config=dict(O.MICRO), variant="micro", latent_dim=4)
    w = m.decoder.stages[3][0].conv1.weight  # 32->32 3x3: kaiming fan_out/relu => std = sqrt(2/(32*9))
    assert abs(float(w.std()) - (2.0 / (32 * 9)) ** 0.5) < 0.015
    lw = m.encoder.stages[2][0].attn.to_q.weight
    assert abs(float(lw.std()) - 0.02) < 0.004
    assert float(m.decoder.norm_out.weight.min()) == 1.0 and float(m.conv_mu.bias.abs().max()) == 0.0


def test_forward_without_gpu_fails_loudly():
    from transvae import TransVAE
    m = TransVAE(config=dict(O.MICRO), variant="micro", latent_dim=4)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.rand(1, 3, 64, 64))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m.decode(torch.zeros(1, 4, 4, 4))


def test_from_pretrained_parses_name():
    from transvae import TransVAE
    with torch.device("meta"):
        m = TransVAE.from_pretrained("transvae-tiny-f16d32")
    assert m.variant == "tiny" and m.latent_dim == 32 and m.compression_ratio == 16


def test_checkpoint_reference_format_round_trip(tmp_path):
    """the reference's checkpoint dictionary (R/train.py:753-769) written and read back, including a file
    as the reference itself would write it (contiguous OIHW tensors, contiguous optimizer moments)."""
    from transvae import TransVAE
    from transvae.checkpoint import load_checkpoint, save_checkpoint
    torch.manual_seed(0)
    m = TransVAE(config=dict(O.MICRO), variant="micro", latent_dim=4)
    opt = torch.optim.AdamW(m.parameters(), lr=1e-4, betas=(0.9, 0.95), weight_decay=0.0)
    for p in m.parameters():
        p.grad = torch.randn_like(p) * 1e-2
    opt.step()
    path = str(tmp_path / "ckpt.pth")
    save_checkpoint(m, opt, epoch=3, global_step=1234, path=path, args={"lr": 1e-4, "variant": "micro"})
    raw = torch.load(path, weights_only=True)
    assert set(raw) == {"epoch", "global_step", "model_state_dict", "optimizer_state_dict", "args"}
    assert all(v.is_contiguous() for v in raw["model_state_dict"].values())
    m2 = TransVAE(config=dict(O.MICRO), variant="micro", latent_dim=4)
    opt2 = torch.optim.AdamW(m2.parameters(), lr=1e-4, betas=(0.9, 0.95), weight_decay=0.0)
    meta = load_checkpoint(path, m2, opt2)
    assert meta["epoch"] == 3 and meta["global_step"] == 1234 and meta["args"]["variant"] == "micro"
    for (k, a), (_, b) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(a, b), k
    for pa, pb in zip(m.parameters(), m2.parameters()):
        sa, sb = opt.state[pa], opt2.state[pb]
        assert torch.equal(sa["exp_avg"], sb["exp_avg"]) and torch.equal(sa["exp_avg_sq"], sb["exp_avg_sq"])
        assert sb["exp_avg"].stride() == pb.stride()       # moments follow the (channels_last) parameter layout
    # a reference-written file: same keys, plain contiguous tensors
    ref_like = {"epoch": 0, "model_state_dict": {k: v.clone().contiguous() for k, v in m.state_dict().items()},
                "optimizer_state_dict": opt.state_dict(), "args": {}}
    torch.save(ref_like, path)
    meta = load_checkpoint(path, m2, opt2)
    assert meta["global_step"] == 0


def test_reference_written_checkpoint_loads_and_round_trips(golden_dir, tmp_path):
    """SURVEY 8f-4 with a file the REFERENCE wrote (tests/golden/ref_checkpoint_nano.pth: its own model class, one AdamW
    step, the dictionary of R/train.py:753-769; minted by `oracle/make_goldens.py --checkpoint`): it loads into this build's
    model and optimizer key for key and bit for bit, a file written back by this build holds the same tensors, and one more
    optimizer step through the fp32 ORACLE from the loaded state lands on the reference's own second step."""
    import json
    from oracle import filler
    from transvae import TransVAE
    from transvae.checkpoint import load_checkpoint, save_checkpoint
    path = os.path.join(golden_dir, "ref_checkpoint_nano.pth")
    raw = torch.load(path, weights_only=True)
    assert set(raw) == {"epoch", "global_step", "model_state_dict", "optimizer_state_dict", "args"}
    cfg = dict(depths=[1, 1, 1], base_dims=[32, 32, 64], mlp_ratio=1.0, head_dim=64)
    m = TransVAE(config=dict(cfg), variant="nano", compression_ratio=4, latent_dim=4)
    opt = torch.optim.AdamW(m.parameters(), lr=1e-3, betas=(0.9, 0.95), weight_decay=0.01)
    meta = load_checkpoint(path, m, opt)
    assert meta["epoch"] == 0 and meta["global_step"] == 1 and meta["args"]["variant"] == "nano"
    assert list(m.state_dict()) == list(raw["model_state_dict"])
    for k, v in m.state_dict().items():
        assert torch.equal(v, raw["model_state_dict"][k]), k
    for i, p in enumerate(m.parameters()):
        st, rs = opt.state[p], raw["optimizer_state_dict"]["state"][i]
        assert torch.equal(st["exp_avg"], rs["exp_avg"]) and torch.equal(st["exp_avg_sq"], rs["exp_avg_sq"])
        assert float(st["step"]) == 1.0 and st["exp_avg"].stride() == p.stride()
    out = str(tmp_path / "back.pth")
    save_checkpoint(m, opt, epoch=0, global_step=1, path=out, args=raw["args"])
    back = torch.load(out, weights_only=True)
    for k, v in raw["model_state_dict"].items():
        assert torch.equal(back["model_state_dict"][k], v) and back["model_state_dict"][k].is_contiguous(), k
    for i, rs in raw["optimizer_state_dict"]["state"].items():
        assert torch.equal(back["optimizer_state_dict"]["state"][i]["exp_avg"], rs["exp_avg"])
    assert back["optimizer_state_dict"]["param_groups"][0]["params"] == raw["optimizer_state_dict"]["param_groups"][0]["params"]
    # resume: the reference's second step, reproduced by the oracle + torch.optim.AdamW from the loaded file
    with open(os.path.join(golden_dir, "ref_checkpoint_nano_expect.json")) as f:
        exp = json.load(f)
    sd = {k: v.clone().requires_grad_(not k.endswith("inv_freq")) for k, v in raw["model_state_dict"].items()}
    names = [k for k in sd if sd[k].requires_grad]
    assert names == exp["param_order"]
    params = [sd[k] for k in names]
    o2 = torch.optim.AdamW(params, lr=1e-3, betas=(0.9, 0.95), weight_decay=0.01)
    o2.load_state_dict(raw["optimizer_state_dict"])
    x = filler.rand_input("nano.x1", (2, 3, 32, 32))
    eps = filler.randn_input("nano.eps1", (2, 4, 8, 8))
    recon, mu, logvar = O.forward(x, sd, cfg, eps)
    loss = O.bench_loss(recon, x, mu, logvar)
    loss.backward()
    torch.nn.utils.clip_grad_norm_(params, 1.0)
    o2.step()
    assert abs(float(loss) - exp["loss_step1"]) < 1e-5 * exp["loss_step1"]
    for k, e in exp["params"].items():
        got = sd[k].detach().flatten()[e["idx"]]
        assert torch.allclose(got, torch.tensor(e["val"]), rtol=1e-4, atol=1e-6), k


def test_batch_chunk_plan_for_tensors_beyond_the_launch_limit():
    """ops._batch_chunks (host logic): tensors below 2 GiB go out in one launch; larger ones in even chunks of whole rows /
    whole images (`align`), every chunk's slice below the limit."""
    import torch
    from transvae.hip import ops
    small = torch.empty((64, 4), dtype=torch.bfloat16)
    assert ops._batch_chunks(64, (small, None)) is None
    meta = torch.empty((128, 256, 256, 192), dtype=torch.bfloat16, device="meta")     # micro-batch 128 at 256 x 256: 3.2 GB
    ch = ops._batch_chunks(128, (meta,))
    assert ch == [(0, 64), (64, 64)]
    tok = torch.empty((96 * 4096, 1536 * 2), dtype=torch.bfloat16, device="meta")     # 96 images x 4096 tokens, 2.4 GB
    ch = ops._batch_chunks(96 * 4096, (tok,), align=4096)
    assert sum(c for _, c in ch) == 96 * 4096 and all(s % 4096 == 0 and c % 4096 == 0 for s, c in ch)
    assert all(c * 1536 * 2 * 2 < (1 << 31) for _, c in ch) and len(ch) == 2
