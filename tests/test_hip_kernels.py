"""Kernel-level parity: every C-ABI entry point against a plain PyTorch fp32 computation of the
same op on the same (bf16-rounded) inputs, run on the host.  `-m gpu` only.

Tolerances (stated per assert): outputs are bf16, so each value carries up to 2^-9 relative
rounding on top of fp32-accumulated sums of bf16 products; we require
max|hip - ref| <= 1e-2 * max|ref| (BASELINE's bf16 tier) and usually see ~3e-3.
"""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

BF = torch.bfloat16


def dev():
    return torch.device("cuda:0")


def rel(a, b):
    a = a.detach().float().cpu()
    b = b.detach().float().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-20))


def r16(t):
    return t.to(BF).float()


def gen(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def ref_conv(x, w, bias, mode, act=None, residual=None):
    """x NHWC fp32 (already bf16-rounded), w [Cout,KH,KW,Cin] fp32 (rounded) -> NHWC fp32."""
    if mode == "linear":
        y = F.linear(x, w, bias)
    else:
        xn = x.permute(0, 3, 1, 2)
        wn = w.permute(0, 3, 1, 2)
        if mode == "c3s1":
            y = F.conv2d(xn, wn, bias, padding=1)
        elif mode == "c3s2":
            y = F.conv2d(xn, wn, bias, stride=2, padding=1)
        elif mode == "c3up":
            y = F.conv2d(F.interpolate(xn, scale_factor=2, mode="nearest"), wn, bias, padding=1)
        elif mode == "unshuf":
            y = F.conv2d(xn, wn, bias, stride=2)
        elif mode == "shuf":
            y = F.conv2d(xn, wn, bias)
            B, C4, H, W = y.shape
            cq = C4 // 4
            y = y.view(B, 2, 2, cq, H, W).permute(0, 3, 4, 1, 5, 2).reshape(B, cq, 2 * H, 2 * W)
        y = y.permute(0, 2, 3, 1)
    if act == "gelu":
        y = F.gelu(y)
    elif act == "silu":
        y = F.silu(y)
    if residual is not None:
        y = y + residual
    return y


CONV_CASES = [
    # mode, x shape, (Cout,KH,KW), act, residual
    ("linear", (200, 64), (128,), None, False),
    ("linear", (130, 96), (192,), "gelu", True),      # BK=32 path, BN=192 tile, M edge
    ("linear", (256, 192), (32,), None, False),       # BN=32 tile
    ("linear", (77, 128), (64,), "silu", False),      # BN=64 tile
    ("linear", (300, 256), (384,), None, True),       # 3 N tiles of 128
    ("linear", (64, 32), (160,), None, False),        # N edge inside the second 128 tile
    ("c3s1", (2, 9, 7, 64), (64, 3, 3), None, True),
    ("c3s1", (1, 16, 16, 192), (192, 3, 3), "gelu", False),
    ("c3s1", (2, 8, 8, 32), (96, 3, 3), "silu", False),
    ("c3s2", (2, 12, 8, 64), (128, 3, 3), None, True),    # even size: data gradient by output parity
    ("c3s2", (1, 16, 24, 96), (64, 3, 3), "silu", False),
    ("c3up", (2, 5, 6, 128), (64, 3, 3), None, False),
    ("c3up", (1, 8, 16, 64), (96, 3, 3), "silu", True),    # polyphase form with activation, saved pre-activation, residual
    ("unshuf", (2, 8, 12, 64), (128, 2, 2), None, True),
    ("shuf", (2, 6, 5, 128), (256, 1, 1), None, False),
]


def _mk(mode, xs, ws, seed):
    Cin = xs[-1]
    wshape = (ws[0], Cin) if mode == "linear" else (ws[0], ws[1], ws[2], Cin)
    fan = Cin * (1 if mode == "linear" else ws[1] * ws[2])
    x = r16(gen(*xs, seed=seed))
    w = r16(gen(*wshape, seed=seed + 1, scale=fan ** -0.5))
    b = gen(ws[0], seed=seed + 2, scale=0.1)
    return x, w, b


@pytest.mark.parametrize("dma", [1, 0])
@pytest.mark.parametrize("case", CONV_CASES, ids=[f"{c[0]}-{'x'.join(map(str, c[1]))}-{c[2][0]}" for c in CONV_CASES])
def test_conv_forward_backward(case, dma):
    from transvae.hip import ops, _lib
    mode, xs, ws, act, use_res = case
    _lib.load().tv_set_dma(dma)
    try:
        x, w, b = _mk(mode, xs, ws, seed=10 * CONV_CASES.index(case))
        xr = x.clone().requires_grad_(True)
        wr = w.clone().requires_grad_(True)
        br = b.clone().requires_grad_(True)
        y0 = ref_conv(xr, wr, br, mode, act, None)
        res = r16(gen(*y0.shape, seed=5)) if use_res else None
        rr = res.clone().requires_grad_(True) if use_res else None
        yref = ref_conv(xr, wr, br, mode, act, rr)
        gy = r16(gen(*yref.shape, seed=6))
        yref.backward(gy)

        xd = x.to(dev(), BF).requires_grad_(True)
        wd = w.to(dev()).requires_grad_(True)
        bd = b.to(dev()).requires_grad_(True)
        rd = res.to(dev(), BF).requires_grad_(True) if use_res else None
        y = ops.conv(xd, wd, bd, rd, mode=mode, act=act)
        assert y.dtype == BF and tuple(y.shape) == tuple(yref.shape)
        assert rel(y, yref) < 1e-2
        y.backward(gy.to(dev(), BF))
        torch.cuda.synchronize()
        assert rel(xd.grad, xr.grad) < 1e-2, "dgrad"
        assert rel(wd.grad, wr.grad) < 1e-2, "wgrad"
        assert rel(bd.grad, br.grad) < 1e-2, "bias grad"
        if use_res:
            assert rel(rd.grad, rr.grad) < 1e-2
    finally:
        _lib.load().tv_set_dma(1)


CHUNK_CASES = [
    ("linear", (6 * 64, 64), (128,), "gelu", True),          # 6 "images" of 64 tokens
    ("c3s1", (5, 16, 64, 192), (192, 3, 3), None, True),      # the kx-triple weight gradient, chunk by chunk
    ("c3s2", (4, 12, 8, 64), (128, 3, 3), "silu", False),
    ("c3up", (3, 8, 8, 64), (96, 3, 3), "silu", True),
    ("unshuf", (4, 8, 12, 64), (128, 2, 2), None, True),
    ("shuf", (3, 6, 5, 128), (256, 1, 1), None, False),
]


@pytest.mark.parametrize("case", CHUNK_CASES, ids=[c[0] for c in CHUNK_CASES])
def test_launches_in_batch_chunks_equal_one_launch(case, monkeypatch):
    """ops._batch_chunks: an activation tensor of 2 GiB or more (micro-batch 128 at 256 x 256: 3.2 GB per 192-channel tensor)
    is launched in batch chunks -- forward and data gradient chunk by chunk, the weight gradient with the later chunks adding.
    With the launch limit lowered so that these small layers split into 2-3 chunks, outputs and activation gradients must be
    bit-identical to the single launch (images are independent) and the weight / bias gradients equal to fp32 summation order."""
    from transvae.hip import ops
    mode, xs, ws, act, use_res = case
    x, w, b = _mk(mode, xs, ws, seed=91)

    def run():
        xd = x.to(dev(), BF).requires_grad_(True)
        wd = w.to(dev()).requires_grad_(True)
        bd = b.to(dev()).requires_grad_(True)
        y0 = ops.conv(xd, wd, bd, None, mode=mode, act=act)
        rd = r16(gen(*y0.shape, seed=7)).to(dev(), BF).requires_grad_(True) if use_res else None
        y = ops.conv(xd, wd, bd, rd, mode=mode, act=act)
        y.backward(r16(gen(*y.shape, seed=8)).to(dev(), BF))
        torch.cuda.synchronize()
        return y.detach(), xd.grad, wd.grad, bd.grad
    one = run()
    rows = xs[0]
    per_row = max(x.numel() // rows, one[0].numel() // rows) * 2
    monkeypatch.setattr(ops, "_LAUNCH_BYTES", per_row * (rows // 2) + 1)     # at most half of the rows per launch
    assert ops._batch_chunks(rows, (x.to(BF), one[0])) is not None
    many = run()
    assert torch.equal(one[0], many[0]) and torch.equal(one[1], many[1])
    assert rel(many[2], one[2]) < 1e-5 and rel(many[3], one[3]) < 1e-5


WGRAD3_CASES = [
    # (B, H, W, Cin, Cout, ring): 3x3 / stride-1 weight gradient through the kx-triple kernel (csrc/wgrad_kx3.hip: image rows of
    # >= 64 pixels; one K-step = 64 pixels of one image row + its two neighbours) -- both tile shapes, one / several K-steps
    # per image row, image borders in x and y, several images, split-K chunks, XCD-grouped order, both ring depths
    (1, 16, 64, 192, 192, 0),     # 192 x 96 x 3 tiles: one K-step = one image row (both neighbours out of range)
    (2, 8, 128, 96, 192, 3),      # two K-steps per image row, ring 3
    (1, 64, 256, 128, 128, 0),    # 128 x 128 x 3 tiles, four K-steps per image row
    (3, 16, 64, 128, 256, 3),     # 128-wide tiles, batch 3, ring 3
    (4, 64, 64, 192, 384, 0),     # two co tiles x two ci tiles; split-K chunks
    (5, 128, 128, 192, 192, 0),   # 81 920 pixels: split over many chunks, XCD-grouped order, ragged last chunk
    (2, 64, 64, 384, 384, 0),     # the Conv-FFN 3x3 of stage 2 (384 = 2 x 192 = 4 x 96)
    (1, 16, 32, 192, 192, 0),     # W = 32: two 32-pixel segments per K-step (40 tile rows each)
    (2, 16, 16, 192, 96, 0),      # W = 16: four 16-pixel segments per K-step (24 tile rows each), ring of 3
    (3, 32, 32, 128, 256, 0),     # W = 32, 128-wide tiles, ring of 4
    (2, 16, 16, 256, 128, 0),     # W = 16, 128-wide tiles: the x slots exceed the 16-bit offset field (second base registers)
    (1, 8, 16, 128, 128, 3),      # W = 16, 8 rows: 128 pixels = 2 K-steps per image, ring of 3
    (2, 8, 8, 192, 192, 0),       # W = 8: not this kernel's -- falls through to the single-tap kernel
    # the schedule variants (ring + 10 x variant, csrc/wgrad_kx3.hip): reads threaded into the MFMA phase, DMA pieces in either phase
    (2, 32, 128, 192, 192, 14), (2, 32, 128, 192, 192, 24), (2, 32, 128, 192, 192, 34), (2, 32, 128, 192, 192, 44),
    (3, 64, 64, 256, 128, 14), (3, 64, 64, 256, 128, 24), (3, 64, 64, 256, 128, 34), (3, 64, 64, 256, 128, 44),
    (1, 8, 64, 96, 192, 14), (1, 8, 64, 96, 192, 24),     # 8 K-steps in all: prologue / tail of the ring
]


@pytest.mark.parametrize("case", WGRAD3_CASES, ids=["x".join(map(str, c)) for c in WGRAD3_CASES])
def test_wgrad_kx_triple(case):
    """tv_wgrad_tn on 3x3 / stride-1 layers (kx-triple kernel: the three kx taps share the staged tiles, x-padding from the
    neighbour rows of each 64-pixel segment) against fp32 PyTorch on the same bf16-rounded operands, against the single-tap
    kernel on the same inputs, and in accumulate mode (tv_wgrad_tn_acc)."""
    from transvae.hip import ops, _lib
    B, H, W, Ci, Co, ring = case
    g = torch.Generator().manual_seed(sum(case))
    x = r16(torch.randn(B, H, W, Ci, generator=g))
    gy = r16(torch.randn(B, H, W, Co, generator=g))
    w = torch.zeros(Co, 3, 3, Ci)
    xn = x.permute(0, 3, 1, 2).clone().requires_grad_(False)
    wn = w.permute(0, 3, 1, 2).clone().requires_grad_(True)
    bn = torch.zeros(Co, requires_grad=True)
    F.conv2d(xn, wn, bn, padding=1).backward(gy.permute(0, 3, 1, 2))
    ref_dw = wn.grad.permute(0, 2, 3, 1)
    xd, gd, wd = x.to(dev(), BF), gy.to(dev(), BF), w.to(dev())
    geo = ops._Geo("c3s1", xd, wd)
    lib = _lib.load()
    outs = []
    for single in (False, True):
        lib.tv_set_wgrad_kx3(0 if single else 1, ring, 0)
        try:
            dw, db = ops.conv_wgrad(geo, wd, xd, gd, True)
            if not single:      # accumulate mode on top of a known tensor
                dw2 = torch.full_like(dw, 0.5)
                db2 = torch.full_like(db, -0.25)
                ops.wgrad_acc(geo.fwd_desc(0), xd, gd, dw2, db2)
            torch.cuda.synchronize()
        finally:
            lib.tv_set_wgrad_kx3(2, 0, 0)
        outs.append((dw.cpu(), db.cpu()))
        assert rel(dw, ref_dw) < 1e-2, ("single-tap" if single else "triple", rel(dw, ref_dw))
        assert rel(db, bn.grad) < 1e-2
    assert rel(outs[0][0], outs[1][0]) < 1e-4       # same bf16 products, fp32 sums in a different order
    assert rel(outs[0][1], outs[1][1]) < 1e-4
    assert rel(dw2 - 0.5, outs[0][0]) < 1e-4 and rel(db2 + 0.25, outs[0][1]) < 1e-4


@pytest.mark.parametrize("shape", [(1024, 192, 384), (4096, 384, 192), (640, 768, 192)], ids=lambda s: "x".join(map(str, s)))
def test_wgrad_linear_one_tap_instantiation(shape):
    """The one-tap instantiation of the kx3 loop on linear layers (tv_set_wgrad_kx3(2, 0, 0); channels multiples of 192,
    tokens a multiple of 64) against fp32 PyTorch and against the single-tap kernel."""
    from transvae.hip import ops, _lib
    T, Ci, Co = shape
    g = torch.Generator().manual_seed(T + Ci)
    x = r16(torch.randn(T, Ci, generator=g))
    gy = r16(torch.randn(T, Co, generator=g))
    ref_dw = gy.t() @ x
    ref_db = gy.sum(0)
    xd, gd, wd = x.to(dev(), BF), gy.to(dev(), BF), torch.zeros(Co, Ci, device=dev())
    geo = ops._Geo("linear", xd, wd)
    lib = _lib.load()
    outs = []
    for mode in (2, 1):
        lib.tv_set_wgrad_kx3(mode, 0, 0)
        try:
            dw, db = ops.conv_wgrad(geo, wd, xd, gd, True)
            torch.cuda.synchronize()
        finally:
            lib.tv_set_wgrad_kx3(2, 0, 0)
        assert rel(dw, ref_dw) < 1e-2 and rel(db, ref_db) < 1e-2, (mode, rel(dw, ref_dw), rel(db, ref_db))
        outs.append((dw.cpu(), db.cpu()))
    assert rel(outs[0][0], outs[1][0]) < 1e-4 and rel(outs[0][1], outs[1][1]) < 1e-4


DERIV_CASES = [
    ("linear", (300, 256), (384,), "gelu", False),
    ("linear", (130, 96), (192,), "silu", True),
    ("c3s1", (2, 16, 16, 192), (192, 3, 3), "gelu", True),
    ("c3up", (1, 8, 16, 64), (96, 3, 3), "silu", False),
]


@pytest.mark.parametrize("case", DERIV_CASES, ids=[f"{c[0]}-{c[3]}" for c in DERIV_CASES])
def test_saved_activation_derivative(case):
    """want_pre="deriv" (TV_ACT_SAVE_DERIV): the saved tensor is act'(pre-activation), the output is unchanged, and the
    data gradient with aux_act = TV_ACT_DERIV equals the one computed from the saved pre-activation."""
    from transvae.hip import ops, _lib as L
    mode, xs, ws, act, use_res = case
    act_id = {"gelu": L.ACT_GELU, "silu": L.ACT_SILU}[act]
    x, w, b = _mk(mode, xs, ws, seed=77)
    xd, wd, bd = x.to(dev(), BF), w.to(dev()), b.to(dev())
    y_pre, pre, g, _ = ops.conv_forward(xd, wd, bd, None, mode, act_id, True)
    res = r16(gen(*y_pre.shape, seed=5)).to(dev(), BF) if use_res else None
    y_pre, pre, g, _ = ops.conv_forward(xd, wd, bd, res, mode, act_id, True)
    y_der, der, _, _ = ops.conv_forward(xd, wd, bd, res, mode, act_id, "deriv")
    assert torch.equal(y_pre, y_der)
    z = pre.float().cpu().requires_grad_(True)
    (F.gelu(z) if act == "gelu" else F.silu(z)).sum().backward()
    # the kernel derives from the unrounded pre-activation, the reference from its bf16 rounding
    assert rel(der, z.grad) < 8e-3
    assert (der.float().cpu() - z.grad).abs().max() < 0.03
    # consumer: data gradient of a following linear layer, multiplied by the saved tensor
    C = y_pre.shape[-1]
    T = y_pre.numel() // C
    w2 = r16(gen(64, C, seed=9, scale=C ** -0.5)).to(dev())
    gz = r16(gen(T, 64, seed=10)).to(dev(), BF)
    g2 = ops._Geo("linear", y_pre.view(T, C), w2)
    d_pre = ops.conv_dgrad(g2, w2, gz, (T, C), None, pre.view(T, C), act_id)
    d_der = ops.conv_dgrad(g2, w2, gz, (T, C), None, der.view(T, C), L.ACT_DERIV)
    assert rel(d_der, d_pre) < 8e-3


# 3x3 stride-1 halo-tile kernel (csrc/igemm_nt.hip: conv3x3_halo_kernel): every tile shape, image borders inside and
# between tiles, several channel chunks, non-power-of-two grids; (bm, bn) forces the tile through the tuning hook.
HALO_CASES = [
    # x shape, Cout, (bm, bn), act, residual
    ((2, 16, 16, 192), 192, (256, 0), "silu", True),    # 256x192 tile, 3 chunks
    ((1, 32, 48, 64), 192, (128, 0), None, False),      # 128x192 tile, 8x16 spatial tiles, W = 3 tiles
    ((2, 16, 32, 256), 256, (0, 256), "gelu", True),    # 256x256 tile, 4 chunks
    ((3, 24, 16, 128), 256, (0, 0), None, False),       # h % 16 != 0: 128x128 tiles
    ((1, 48, 32, 64), 128, (256, 128), None, True),     # 256x128 tile
    ((2, 8, 16, 320), 160, (0, 0), "silu", False),      # N edge in the second 128 tile, 5 chunks, one tile per image
]


@pytest.mark.parametrize("case", HALO_CASES, ids=[f"{'x'.join(map(str, c[0]))}-{c[1]}-bm{c[2][0]}bn{c[2][1]}" for c in HALO_CASES])
def test_conv3x3_halo(case):
    from transvae.hip import ops, _lib
    xs, cout, (bm, bn), act, use_res = case
    lib = _lib.load()
    x, w, b = _mk("c3s1", xs, (cout, 3, 3), seed=77 + HALO_CASES.index(case))
    xr, wr, br = (t.clone().requires_grad_(True) for t in (x, w, b))
    res = r16(gen(xs[0], xs[1], xs[2], cout, seed=5)) if use_res else None
    rr = res.clone().requires_grad_(True) if use_res else None
    yref = ref_conv(xr, wr, br, "c3s1", act, rr)
    gy = r16(gen(*yref.shape, seed=6))
    yref.backward(gy)
    outs = {}
    try:
        for halo in (1, 0):
            lib.tv_set_igemm_halo(halo)
            lib.tv_set_igemm_config(bm, bn, 0, 0)
            xd = x.to(dev(), BF).requires_grad_(True)
            wd = w.to(dev()).requires_grad_(True)
            bd = b.to(dev()).requires_grad_(True)
            rd = res.to(dev(), BF).requires_grad_(True) if use_res else None
            y = ops.conv(xd, wd, bd, rd, mode="c3s1", act=act)
            y.backward(gy.to(dev(), BF))
            torch.cuda.synchronize()
            assert rel(y, yref) < 1e-2, f"halo={halo} forward"
            assert rel(xd.grad, xr.grad) < 1e-2, f"halo={halo} dgrad"
            assert rel(wd.grad, wr.grad) < 1e-2
            outs[halo] = (y.float().cpu(), xd.grad.float().cpu())
        # same math, different summation order: the two kernels agree to about one bf16 rounding step (2^-8)
        assert rel(outs[1][0], outs[0][0]) < 8e-3
        assert rel(outs[1][1], outs[0][1]) < 8e-3
    finally:
        lib.tv_set_igemm_halo(1)
        lib.tv_set_igemm_config(0, 0, 0, 0)


# Tuning variants that the heuristics only pick at full size: forced here on small problems so that their indexing is
# covered -- XCD-grouped block orders with tile counts that are not multiples of 8, the 3-deep wgrad ring, the 4-wave
# 256x192 halo tile, the 3-deep halo weight ring on a multi-chunk problem.
VARIANT_CASES = [
    # name, setup(lib), x shape, Cout, mode
    ("wgrad-xcd-split11", lambda lib: lib.tv_set_wgrad_config(0, 0, 9 * 11), (3, 32, 32, 64), 64, "c3s1"),
    ("wgrad-xcd-split5-linear", lambda lib: lib.tv_set_wgrad_config(0, 0, 2 * 5), (5 * 640, 128), 256, "linear"),
    ("wgrad-ring3", lambda lib: (lib.tv_set_wgrad_stages(3), lib.tv_set_wgrad_config(0, 8, 9 * 3)), (2, 32, 32, 192), 192, "c3s1"),
    ("wgrad-plain-order", lambda lib: (lib.tv_set_wgrad_stages(12), lib.tv_set_wgrad_config(0, 0, 9 * 6)), (2, 32, 32, 64), 128, "c3s1"),
    ("halo-4wave", lambda lib: (lib.tv_set_igemm_halo(101), lib.tv_set_igemm_config(256, 0, 0, 0)), (2, 32, 16, 192), 192, "c3s1"),
    ("halo-ring3-all", lambda lib: lib.tv_set_igemm_halo(4), (2, 16, 32, 128), 384, "c3s1"),
    ("igemm-plain-order", lambda lib: lib.tv_set_igemm_halo(11), (5 * 256 + 40, 128), 640, "linear"),
    ("igemm-xcd-order", lambda lib: None, (11 * 128 + 7, 64), 320, "linear"),
]


@pytest.mark.parametrize("case", VARIANT_CASES, ids=[c[0] for c in VARIANT_CASES])
def test_kernel_variants(case):
    from transvae.hip import ops, _lib
    name, setup, xs, cout, mode = case
    lib = _lib.load()
    ws = (cout,) if mode == "linear" else (cout, 3, 3)
    x, w, b = _mk(mode, xs, ws, seed=300 + VARIANT_CASES.index(case))
    xr, wr, br = (t.clone().requires_grad_(True) for t in (x, w, b))
    yref = ref_conv(xr, wr, br, mode, None, None)
    gy = r16(gen(*yref.shape, seed=6))
    yref.backward(gy)
    try:
        setup(lib)
        xd = x.to(dev(), BF).requires_grad_(True)
        wd = w.to(dev()).requires_grad_(True)
        bd = b.to(dev()).requires_grad_(True)
        y = ops.conv(xd, wd, bd, None, mode=mode)
        y.backward(gy.to(dev(), BF))
        torch.cuda.synchronize()
        assert rel(y, yref) < 1e-2, "forward"
        assert rel(xd.grad, xr.grad) < 1e-2, "dgrad"
        assert rel(wd.grad, wr.grad) < 1e-2, "wgrad"
        assert rel(bd.grad, br.grad) < 1e-2, "bias grad"
    finally:
        lib.tv_set_igemm_halo(1)
        lib.tv_set_igemm_config(0, 0, 0, 0)
        lib.tv_set_wgrad_config(0, 0, 0)
        lib.tv_set_wgrad_stages(0)


def test_packed_weight_cache():
    """bf16 repacks are reused only for (views of) parameters, and an in-place update invalidates them."""
    from transvae.hip import ops
    conv = torch.nn.Conv2d(64, 32, 3, padding=1).to(dev()).to(memory_format=torch.channels_last)

    def view():
        return conv.weight.permute(0, 2, 3, 1).view(32, 9, 64)
    with ops.packed_weight_cache():
        a, _ = ops.pack_weight(view(), True, False, False)
        b, _ = ops.pack_weight(view(), True, False, False)
        assert a.data_ptr() == b.data_ptr()
        _, at = ops.pack_weight(view(), False, True, True)          # different request: its own entry
        assert at is not None and at.shape == (64, 9, 32)
        with torch.no_grad():
            conv.weight.add_(1.0)
        c, _ = ops.pack_weight(view(), True, False, False)
        assert c.data_ptr() != a.data_ptr()
        assert torch.equal(c.float().cpu(), view().detach().to(BF).float().cpu())
        t = torch.randn(32, 9, 64, device=dev())                   # not a parameter: never cached
        p1, _ = ops.pack_weight(t, True, False, False)
        p2, _ = ops.pack_weight(t, True, False, False)
        assert p1.data_ptr() != p2.data_ptr()
    d1, _ = ops.pack_weight(view(), True, False, False)             # outside the block: no cache
    d2, _ = ops.pack_weight(view(), True, False, False)
    assert d1.data_ptr() != d2.data_ptr()


@torch.no_grad()   # (as inside an autograd Function's forward)
def test_derived_weight_cache():
    """Polyphase / parity operands derived from a parameter are built once per version of the parameter inside a
    packed_weight_cache() block, and the cached operand gives the same convolution as a fresh one."""
    from transvae.hip import ops, _lib as L
    conv = torch.nn.Conv2d(64, 32, 3, padding=1).to(dev()).to(memory_format=torch.channels_last)
    x = r16(gen(2, 6, 8, 64, seed=1)).to(dev(), BF)

    def w():
        return conv.weight.permute(0, 2, 3, 1)
    y_ref = ops.conv_forward(x, w(), None, None, "c3up", L.ACT_NONE, False)[0]
    with ops.packed_weight_cache():
        n0 = len(ops._pack_cache)
        y1 = ops.conv_forward(x, w(), None, None, "c3up", L.ACT_NONE, False)[0]
        n1 = len(ops._pack_cache)
        y2 = ops.conv_forward(x, w(), None, None, "c3up", L.ACT_NONE, False)[0]
        assert n1 == n0 + 1 and len(ops._pack_cache) == n1            # second call: a hit
        assert torch.equal(y1, y_ref) and torch.equal(y2, y_ref)
        conv.weight.mul_(0.5)
        y3 = ops.conv_forward(x, w(), None, None, "c3up", L.ACT_NONE, False)[0]
        assert len(ops._pack_cache) == n1 + 1                          # new version: rebuilt
        assert rel(y3, 0.5 * y_ref.float()) < 1e-2


def test_pack_weight():
    from transvae.hip import ops
    w = gen(40, 9, 72, seed=3)
    d, dt = ops.pack_weight(w.to(dev()), True, True, True)
    assert torch.equal(d.cpu(), w.to(BF))
    assert torch.equal(dt.cpu(), w.flip(1).permute(2, 1, 0).contiguous().to(BF))
    _, dt2 = ops.pack_weight(w.to(dev()), False, True, False)
    assert torch.equal(dt2.cpu(), w.permute(2, 1, 0).contiguous().to(BF))


@pytest.mark.parametrize("shape", [(2, 16, 16, 64), (3, 7, 9, 192), (1, 32, 32, 32), (2, 24, 24, 320)])
def test_groupnorm_silu(shape):
    from transvae.hip import ops
    B, H, W, Cc = shape
    x = r16(gen(*shape, seed=1) * 1.5 + 0.3)
    ga = 1 + 0.1 * gen(Cc, seed=2)
    be = 0.1 * gen(Cc, seed=3)
    xr, gr, br = x.clone().requires_grad_(True), ga.clone().requires_grad_(True), be.clone().requires_grad_(True)
    yref = F.silu(F.group_norm(xr.permute(0, 3, 1, 2), 32, gr, br, eps=1e-5)).permute(0, 2, 3, 1)
    gy = r16(gen(*shape, seed=4))
    yref.backward(gy)
    xd = x.to(dev(), BF).requires_grad_(True)
    gd, bd = ga.to(dev()).requires_grad_(True), be.to(dev()).requires_grad_(True)
    y = ops.group_norm_silu(xd, gd, bd)
    assert rel(y, yref) < 1e-2
    y.backward(gy.to(dev(), BF))
    assert rel(xd.grad, xr.grad) < 1e-2
    assert rel(gd.grad, gr.grad) < 1e-2
    assert rel(bd.grad, br.grad) < 1e-2


def test_groupnorm_statistics_with_a_large_mean():
    """|mean| >> std (a drifted residual stream): the one-pass E[x^2] - mean^2 form loses the variance to cancellation in
    fp32; the kernels sum about a per-channel pivot and merge channels with Chan's update, so mean / rstd must agree with
    a float64 two-pass computation.  Per-channel offsets differ inside a group (the merge term), 65 536 pixels per image."""
    from transvae.hip import _lib as L
    from transvae.hip import fused
    B, H, W, Cc, G = 2, 256, 256, 64, 32
    g = torch.Generator().manual_seed(0)
    offs = 50.0 + 3.0 * torch.randn(Cc, generator=g)                      # mean / std ~ 100 at std 0.5 (bf16 ulp 0.25)
    x = (0.5 * torch.randn(B, H, W, Cc, generator=g) + offs).to(BF)
    xd = x.to(dev())
    ga, be = torch.ones(Cc, device=dev()), torch.zeros(Cc, device=dev())
    y, mr = fused.gn_silu_fwd(xd, ga, be, G, 1e-5)
    xg = x.double().view(B, H * W, G, Cc // G)
    mean = xg.mean(dim=(1, 3))
    var = xg.var(dim=(1, 3), unbiased=False)
    rstd = (var + 1e-5).rsqrt()
    mr = mr.cpu().double()
    assert float((mr[..., 0] - mean).abs().max()) < 1e-4 * float(mean.abs().max())
    assert float(((mr[..., 1] - rstd) / rstd).abs().max()) < 1e-3, float(((mr[..., 1] - rstd) / rstd).abs().max())
    yref = F.silu(F.group_norm(x.double().permute(0, 3, 1, 2), G, eps=1e-5)).permute(0, 2, 3, 1)
    assert rel(y, yref) < 1e-2


@pytest.mark.parametrize("T,Cc", [(50, 64), (300, 384), (40, 768), (17, 1536), (9, 2560)])
def test_rownorm(T, Cc):
    from transvae.hip import ops
    x = r16(gen(T, Cc, seed=1) * 2.0)
    w = 1 + 0.1 * gen(Cc, seed=2)
    gy = r16(gen(T, Cc, seed=3))
    # mode 0
    xr = x.clone().requires_grad_(True)
    yref = xr * torch.rsqrt((xr * xr).mean(-1, keepdim=True) + 1e-6)
    yref.backward(gy)
    xd = x.to(dev(), BF).requires_grad_(True)
    y = ops.rms_hat(xd)
    assert rel(y, yref) < 1e-2
    y.backward(gy.to(dev(), BF))
    assert rel(xd.grad, xr.grad) < 1e-2
    # mode 1
    xr = x.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    u = xr * torch.rsqrt((xr * xr).mean(-1, keepdim=True) + 1e-6) * wr
    yref = F.layer_norm(u, (Cc,), eps=1e-5)
    yref.backward(gy)
    xd = x.to(dev(), BF).requires_grad_(True)
    wd = w.to(dev()).requires_grad_(True)
    y = ops.rms_ln_hat(xd, wd)
    assert rel(y, yref) < 1e-2
    y.backward(gy.to(dev(), BF))
    assert rel(xd.grad, xr.grad) < 1.5e-2
    assert rel(wd.grad, wr.grad) < 1e-2


def test_layout_and_im2col():
    from transvae.hip import ops
    x = gen(2, 3, 10, 12, seed=1)
    y = ops.to_nhwc(x.to(dev()), 32)
    ref = torch.zeros(2, 10, 12, 32)
    ref[..., :3] = x.permute(0, 2, 3, 1)
    assert torch.equal(y.cpu().float(), r16(ref))
    t = r16(gen(2, 6, 5, 64, seed=2))
    z = ops.to_nchw(t.to(dev(), BF), 32, 16)
    assert torch.equal(z.cpu(), t[..., 32:48].permute(0, 3, 1, 2))
    col = ops.im2col3x3(x.to(dev()), 32).cpu().float().view(2, 10, 12, 32)
    pat = F.unfold(x, 3, padding=1).view(2, 3, 9, 10, 12).permute(0, 3, 4, 2, 1).reshape(2, 10, 12, 27)
    assert torch.equal(col[..., :27], r16(pat)) and float(col[..., 27:].abs().max()) == 0.0


@pytest.mark.parametrize("R,Cc,n,bias", [(384, 384, 3, True), (1536, 384, 1, False), (100, 72, 3, True)])
def test_parameter_fold_matches_autograd(R, Cc, n, bias):
    """fused.fold (tv_fold_cols / tv_fold_cols_bwd): Wf = cat(W_s * gamma_s), bf = cat(W_s @ beta_s) and the gradients onto
    W, gamma, beta -- against the PyTorch formulation it replaces (R/transvae/modules/attention.py:39-48,71-78)."""
    from transvae.hip import fused
    Ws = [gen(R, Cc, seed=10 + s).to(dev()).requires_grad_(True) for s in range(n)]
    gs = [(1 + 0.1 * gen(Cc, seed=20 + s)).to(dev()).requires_grad_(True) for s in range(n)]
    bs = [(0.1 * gen(Cc, seed=30 + s)).to(dev()).requires_grad_(True) for s in range(n)] if bias else None
    Wf, bf = fused.fold(Ws, gs, bs)
    ref_W = torch.cat([w * g[None, :] for w, g in zip(Ws, gs)], 0)
    assert torch.allclose(Wf, ref_W, rtol=1e-6, atol=1e-7)
    gW = gen(n * R, Cc, seed=40).to(dev())
    loss = (Wf * gW).sum()
    ref_loss = (ref_W * gW).sum()
    if bias:
        ref_b = torch.cat([w @ b for w, b in zip(Ws, bs)], 0)
        assert torch.allclose(bf, ref_b, rtol=1e-4, atol=1e-5)
        gb = gen(n * R, seed=41).to(dev())
        loss = loss + (bf * gb).sum()
        ref_loss = ref_loss + (ref_b * gb).sum()
    else:
        assert bf is None
    leaves = Ws + gs + (bs or [])
    got = torch.autograd.grad(loss, leaves)
    ref = torch.autograd.grad(ref_loss, leaves)
    for a, b in zip(got, ref):
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-5), float((a - b).abs().max())


def test_unsupported_shape_raises():
    from transvae.hip import ops
    x = torch.zeros(4, 40, dtype=BF, device=dev())       # c_in not a multiple of 32
    w = torch.zeros(32, 40, device=dev())
    with pytest.raises(RuntimeError, match="c_in"):
        ops.linear(x, w)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.linear(torch.zeros(4, 64, dtype=BF), torch.zeros(32, 64))


def _rope_tab(H, W):
    from oracle import transvae_oracle as O
    from oracle import filler
    c1, s1, c2, s2 = O.rope_tables(H, W, filler.inv_freq(64))
    return torch.stack([c1, s1, c2, s2], dim=1).contiguous()  # [N,4,32]


@pytest.mark.parametrize("H,W,heads", [(4, 6, 2), (16, 16, 1)])
def test_rope_kernel(H, W, heads):
    from oracle import transvae_oracle as O
    from oracle import filler
    from transvae.hip import ops, _lib
    import ctypes as C
    B, N = 2, H * W
    qkv = r16(gen(B, N, 3, heads, 64, seed=7))
    tab = _rope_tab(H, W)
    tabs = O.rope_tables(H, W, filler.inv_freq(64))
    ref = qkv.clone()
    for which in (0, 1):
        ref[:, :, which] = O.rope_apply(qkv[:, :, which].permute(0, 2, 1, 3), tabs).permute(0, 2, 1, 3)
    d = qkv.to(dev(), BF).contiguous()
    lib = _lib.load()
    _lib.check(lib.tv_rope_qk(C.c_void_p(d.data_ptr()), C.c_void_p(tab.to(dev()).data_ptr()), B, N, heads, 0, None))
    torch.cuda.synchronize()
    assert rel(d, ref) < 1e-2
    assert torch.equal(d[:, :, 2].cpu().float(), qkv[:, :, 2])  # v untouched
    # adjoint: <R x, y> == <x, R^T y>
    y = r16(gen(B, N, 3, heads, 64, seed=8))
    dy = y.to(dev(), BF).contiguous()
    _lib.check(lib.tv_rope_qk(C.c_void_p(dy.data_ptr()), C.c_void_p(tab.to(dev()).data_ptr()), B, N, heads, 1, None))
    xin = qkv[:, :, :2].clone().requires_grad_(True)   # [B,N,2,h,64]
    out = O.rope_apply(xin.permute(0, 2, 3, 1, 4), tabs)  # [B,2,h,N,64]
    out.backward(y[:, :, :2].permute(0, 2, 3, 1, 4))
    assert rel(dy[:, :, :2], xin.grad) < 1e-2
    assert torch.equal(dy[:, :, 2].cpu().float(), y[:, :, 2])


@pytest.mark.parametrize("B,H,W,heads", [(2, 8, 8, 2), (1, 16, 12, 1), (1, 5, 12, 3), (2, 32, 32, 6)])
def test_qkv_projection_with_rope_in_the_epilogue_and_adjoint_in_attention_backward(B, H, W, heads):
    """tv_igemm_nt_rope == projection followed by the reference RoPE on q and k (fp32, oracle formula; v untouched), and
    tv_attn_bwd(rope_tab) == tv_attn_bwd(NULL) followed by the adjoint rotation (tv_rope_qk transpose), on the same inputs."""
    from oracle import transvae_oracle as O
    from oracle import filler
    from transvae.hip import ops, _lib
    import ctypes as C
    N, Cc = H * W, heads * 64
    lib = _lib.load()
    x = r16(gen(B * N, Cc, seed=1))
    w = r16(gen(3 * Cc, Cc, seed=2) * Cc ** -0.5)
    b = gen(3 * Cc, seed=3) * 0.1
    tab = _rope_tab(H, W)
    tabs = O.rope_tables(H, W, filler.inv_freq(64))
    ref = F.linear(x, w, b).view(B, N, 3, heads, 64)
    ref_rot = ref.clone()
    for which in (0, 1):
        ref_rot[:, :, which] = O.rope_apply(ref[:, :, which].permute(0, 2, 1, 3), tabs).permute(0, 2, 1, 3)
    tabd = tab.to(dev())
    out, _, geo, _ = ops.conv_forward(x.to(dev(), BF), w.to(dev()), b.to(dev()), None, "linear", _lib.ACT_NONE, False,
                                      rope=(tabd, N, 2 * Cc))
    torch.cuda.synchronize()
    got = out.view(B, N, 3, heads, 64)
    assert rel(got, ref_rot) < 1e-2, rel(got, ref_rot)
    assert rel(got[:, :, 2], ref[:, :, 2]) < 1e-2                      # v: plain projection
    # backward: fused adjoint vs separate pass
    qkv = got.contiguous()
    o = torch.empty((B, N, Cc), dtype=BF, device=dev())
    lse = torch.empty((B, heads, N), dtype=torch.float32, device=dev())
    _lib.check(lib.tv_attn_fwd(ops._p(qkv), ops._p(o), ops._p(lse), B, N, heads, 0.125, None))
    do = gen(B, N, Cc, seed=4).to(dev(), BF)
    res = []
    for fused_adjoint in (True, False):
        delta = torch.empty((2, B, heads, N), dtype=torch.float32, device=dev())
        dqkv = torch.empty_like(qkv)
        _lib.check(lib.tv_attn_bwd(ops._p(qkv), ops._p(o), ops._p(do), ops._p(lse), ops._p(delta), ops._p(tabd) if fused_adjoint else None,
                                   ops._p(dqkv), B, N, heads, 0.125, None))
        if not fused_adjoint:
            _lib.check(lib.tv_rope_qk(ops._p(dqkv), ops._p(tabd), B, N, heads, 1, None))
        torch.cuda.synchronize()
        res.append(dqkv.float().cpu())
    assert rel(res[0], res[1]) < 1e-2, rel(res[0], res[1])
    assert torch.equal(res[0][:, :, 2], res[1][:, :, 2])               # dv is not rotated


@pytest.mark.parametrize("B,H,W,heads,rope", [(2, 8, 8, 2, True), (1, 16, 12, 1, True), (2, 4, 4, 3, False),
                                              (1, 16, 16, 2, True), (1, 15, 20, 1, True), (1, 32, 32, 2, True),
                                              (1, 48, 45, 1, True), (2, 64, 32, 1, False)])   # N >= 2048: the 64-queries-per-wave forward kernel (ragged / whole)
def test_attention_fwd_bwd(B, H, W, heads, rope):
    from oracle import transvae_oracle as O
    from oracle import filler
    from transvae.hip import ops
    N, C = H * W, heads * 64
    qkv = r16(gen(B, N, 3 * C, seed=11))
    go = r16(gen(B, N, C, seed=12))
    scale = 64 ** -0.5
    x = qkv.clone().requires_grad_(True)
    q, k, v = [t.view(B, N, heads, 64).transpose(1, 2) for t in x.split(C, dim=-1)]
    if rope:
        tabs = O.rope_tables(H, W, filler.inv_freq(64))
        q, k = O.rope_apply(q, tabs), O.rope_apply(k, tabs)
    s = (q @ k.transpose(-1, -2)) * scale
    oref = (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(B, N, C)
    oref.backward(go)
    xd = qkv.to(dev(), BF).requires_grad_(True)
    tab = _rope_tab(H, W).to(dev()) if rope else None
    o = ops.attention(xd.clone(), tab, heads, scale)   # clone: RoPE works in place on its input
    assert rel(o, oref) < 1e-2
    o.backward(go.to(dev(), BF))
    g = xd.grad.cpu().float()
    for i, nm in enumerate("qkv"):
        assert rel(g[..., i * C:(i + 1) * C], x.grad[..., i * C:(i + 1) * C]) < 2e-2, f"d{nm}"


@pytest.mark.gpu
@pytest.mark.parametrize("N", [512, 2304])
def test_attention_deferred_rescale_is_exercised_by_spiked_keys(N):
    """The forward kernels rescale their running sums only when a row's maximum grows by more than 2^8 in the exponent
    (attention.hip, TV_ATTN_LAZY).  Bounded random scores never take that branch after the first key block, so this input
    forces it: scores of a wide range, and a few keys late in the sequence aligned with a few queries so that those rows'
    maxima jump by hundreds of exponent units at chosen key blocks (and stay put for the others in the same wave).  Output
    and all three gradients against an fp32 reference; both kernels (32 and 64 queries per wave)."""
    from transvae.hip import ops
    heads, B = 2, 1
    C = heads * 64
    g = torch.Generator().manual_seed(23)
    q = torch.randn(B, N, heads, 64, generator=g) * 2.0
    k = torch.randn(B, N, heads, 64, generator=g) * 2.0
    v = torch.randn(B, N, heads, 64, generator=g)
    for (kj, qi, gain) in ((N // 2 + 3, 5, 6.0), (N - 70, 40, 9.0), (N - 1, 130, 12.0), (200, 131, 4.0)):
        k[0, kj] = q[0, qi] * gain / 2.0            # score ~ gain * |q|^2 / 8 / 2: far above the rest of the row
    qkv = r16(torch.cat([q.reshape(B, N, C), k.reshape(B, N, C), v.reshape(B, N, C)], dim=-1))
    go = r16(torch.randn(B, N, C, generator=g))
    x = qkv.clone().requires_grad_(True)
    qq, kk, vv = [t.view(B, N, heads, 64).transpose(1, 2) for t in x.split(C, dim=-1)]
    sc = (qq @ kk.transpose(-1, -2)) * 0.125
    assert float(sc.detach().max()) > 60.0 and float(sc.detach().std()) > 3.0     # the range that makes the branch matter
    oref = (torch.softmax(sc, -1) @ vv).transpose(1, 2).reshape(B, N, C)
    oref.backward(go)
    xd = qkv.to(dev(), BF).requires_grad_(True)
    o = ops.attention(xd.clone(), None, heads, 0.125)
    assert torch.isfinite(o.float()).all()
    assert rel(o, oref) < 1e-2, rel(o, oref)
    o.backward(go.to(dev(), BF))
    gq = xd.grad.cpu().float()
    for i, nm in enumerate("qkv"):
        assert rel(gq[..., i * C:(i + 1) * C], x.grad[..., i * C:(i + 1) * C]) < 2e-2, f"d{nm}"


# ---- register epilogues (igemm_common.h: EF_*) against the generic LDS loop ---------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [(0, 0, 0, 0), (256, 256, 0, 0), (256, 192, 0, 0), (128, 128, 0, 0)])
def test_register_epilogue_forms_are_bit_identical_to_the_lds_loop(cfg):
    """Every compact register form (plain, GELU / SiLU, with saved derivative, residual add, derivative multiply,
    (acc + residual) x derivative, RoPE) must round exactly like the generic LDS loop it replaces, on every tile shape --
    a batch has to equal its images run alone bit for bit and the tile shape depends on the batch
    (test_large_f16d32_256_full_size_properties_and_oracle caught an fma contraction that differed between two
    instantiations).  Shapes: ragged M (tail rows masked), N = 384 (96-wide slabs on the 192 tiles: odd line alignment,
    a block without a line partner), linear layers and a 3x3 convolution through the halo kernel."""
    from transvae.hip import _lib as L, ops
    lib = L.load()
    g = torch.Generator(device=dev()).manual_seed(5)
    bf = torch.bfloat16
    M, K, N = 2 * 1024 + 40, 192, 384
    x = torch.randn(M, K, device=dev(), generator=g).to(bf)
    w = torch.randn(N, K, device=dev(), generator=g) * K ** -0.5
    b = torch.randn(N, device=dev(), generator=g) * 0.1
    res = torch.randn(M, N, device=dev(), generator=g).to(bf)
    gz = torch.randn(M, N, device=dev(), generator=g).to(bf)
    der = torch.rand(M, K, device=dev(), generator=g).to(bf)
    gres = torch.randn(M, K, device=dev(), generator=g).to(bf)
    tab = torch.randn(1024, 4, 32, device=dev(), generator=g)
    xc = torch.randn(3, 32, 32, 192, device=dev(), generator=g).to(bf)
    wc = torch.randn(192, 3, 3, 192, device=dev(), generator=g) * (9 * 192) ** -0.5
    bc = torch.randn(192, device=dev(), generator=g) * 0.1
    geo = ops._Geo("linear", x, w)
    xr = x[: 2 * 1024].contiguous()

    def cases():
        out = {}
        out["plain"] = ops.conv_forward(x, w, b, None, "linear", L.ACT_NONE, False)[0]
        out["gelu"] = ops.conv_forward(x, w, b, None, "linear", L.ACT_GELU, False)[0]
        out["silu"] = ops.conv_forward(x, w, b, None, "linear", L.ACT_SILU, False)[0]
        y, d = ops.conv_forward(x, w, b, None, "linear", L.ACT_GELU, "deriv")[:2]
        out["gelu+deriv"], out["gelu+deriv:saved"] = y, d
        y, d = ops.conv_forward(x, w, b, None, "linear", L.ACT_SILU, "deriv")[:2]
        out["silu+deriv"], out["silu+deriv:saved"] = y, d
        out["residual"] = ops.conv_forward(x, w, b, res, "linear", L.ACT_NONE, False)[0]
        out["dgrad"] = ops.conv_dgrad(geo, w, gz, x.shape)
        out["dgrad*deriv"] = ops.conv_dgrad(geo, w, gz, x.shape, aux=der, aux_act=L.ACT_DERIV)
        out["(dgrad+res)*deriv"] = ops.conv_dgrad(geo, w, gz, x.shape, residual=gres, aux=der, aux_act=L.ACT_DERIV)
        out["rope"] = ops.conv_forward(xr, w, b, None, "linear", L.ACT_NONE, False, rope=(tab, 1024, 256))[0]
        out["conv3x3 gelu+deriv"] = ops.conv_forward(xc, wc, bc, None, "c3s1", L.ACT_GELU, "deriv")[0]
        out["conv3x3 residual"] = ops.conv_forward(xc, wc, bc, xc, "c3s1", L.ACT_NONE, False)[0]
        return out
    try:
        lib.tv_set_igemm_config(*cfg)
        lib.tv_set_igemm_epilogue(1)
        reg = cases()
        lib.tv_set_igemm_epilogue(0)
        lds = cases()
    finally:
        lib.tv_set_igemm_epilogue(1)
        lib.tv_set_igemm_config(0, 0, 0, 0)
    for k in reg:
        assert torch.isfinite(reg[k].float()).all(), k
        assert torch.equal(reg[k], lds[k]), (k, cfg, float((reg[k].float() - lds[k].float()).abs().max()))


# ---- persistent column loop of the generic 256 x 256 tile (igemm_nt.hip) ----------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("M,K,N", [(1024, 384, 1536), (768 + 40, 512, 768), (512, 1536, 1024), (4096, 256, 512), (16384, 256, 2048), (2304, 512, 1280)])
def test_persistent_column_loop_is_bit_identical_to_one_tile_per_block(M, K, N):
    """A block of the 256 x 256 tile walks several column tiles of its row tile as one K-loop (the next tile's first stage
    lands under the epilogue, opened by a counted vmcnt): every register epilogue form must give the bits of the
    one-tile-per-block launch, on full and ragged row tiles, for the heuristic and for every forced walk length."""
    from transvae.hip import _lib as L, ops
    lib = L.load()
    g = torch.Generator(device=dev()).manual_seed(7)
    bf = torch.bfloat16
    x = torch.randn(M, K, device=dev(), generator=g).to(bf)
    w = torch.randn(N, K, device=dev(), generator=g) * K ** -0.5
    b = torch.randn(N, device=dev(), generator=g) * 0.1
    res = torch.randn(M, N, device=dev(), generator=g).to(bf)
    gz = torch.randn(M, N, device=dev(), generator=g).to(bf)
    der = torch.rand(M, K, device=dev(), generator=g).to(bf)
    gres = torch.randn(M, K, device=dev(), generator=g).to(bf)
    geo = ops._Geo("linear", x, w)

    def cases():
        out = {}
        out["plain"] = ops.conv_forward(x, w, b, None, "linear", L.ACT_NONE, False)[0]
        y, d = ops.conv_forward(x, w, b, None, "linear", L.ACT_GELU, "deriv")[:2]
        out["gelu+deriv"], out["gelu+deriv:saved"] = y, d
        out["silu"] = ops.conv_forward(x, w, b, None, "linear", L.ACT_SILU, False)[0]
        out["residual"] = ops.conv_forward(x, w, b, res, "linear", L.ACT_NONE, False)[0]
        out["dgrad"] = ops.conv_dgrad(geo, w, gz, x.shape)                     # [M, N] -> [M, K]: K / 256 column tiles
        out["dgrad*deriv"] = ops.conv_dgrad(geo, w, gz, x.shape, aux=der, aux_act=L.ACT_DERIV)
        out["(dgrad+res)*deriv"] = ops.conv_dgrad(geo, w, gz, x.shape, residual=gres, aux=der, aux_act=L.ACT_DERIV)
        return out
    try:
        lib.tv_set_igemm_config(256, 256, 0, 0)
        lib.tv_set_igemm_persist(0)
        ref = cases()
        got = {}
        for walk in (1, 2, 3, 4, 6):
            lib.tv_set_igemm_persist(walk)
            got[walk] = cases()
    finally:
        lib.tv_set_igemm_persist(0)      # (the library's default: the walk is off)
        lib.tv_set_igemm_config(0, 0, 0, 0)
    for walk, res_w in got.items():
        for k in ref:
            assert torch.isfinite(res_w[k].float()).all(), (k, walk)
            assert torch.equal(res_w[k], ref[k]), (k, walk, float((res_w[k].float() - ref[k].float()).abs().max()))


# ---- eight-phase ping-pong loop of the 256 x 256 tile (igemm_nt.hip, single-tap layers) ------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("bn", [256, 192])
@pytest.mark.parametrize("M,K,N", [(1024, 64, 768), (1024, 128, 1536), (768 + 40, 192, 768), (512, 1536, 768), (4096, 320, 1536),
                                   (16384, 384, 1536), (1000, 448, 384)])
def test_eight_phase_loop_is_bit_identical_to_the_plain_loop(M, K, N, bn):
    """The 256 x 256 tile runs 1x1 / linear layers through a four-phases-per-K-step wave-group ping-pong (regions of a stage
    buffer re-staged right behind their last read, counted vmcnt once per K-step): every epilogue form must give the bits of
    the plain two-stage loop -- K of one, two, an odd number and many steps, full and ragged row tiles."""
    from transvae.hip import _lib as L, ops
    lib = L.load()
    g = torch.Generator(device=dev()).manual_seed(11)
    bf = torch.bfloat16
    x = torch.randn(M, K, device=dev(), generator=g).to(bf)
    w = torch.randn(N, K, device=dev(), generator=g) * K ** -0.5
    b = torch.randn(N, device=dev(), generator=g) * 0.1
    res = torch.randn(M, N, device=dev(), generator=g).to(bf)

    def cases():
        out = {}
        out["plain"] = ops.conv_forward(x, w, b, None, "linear", L.ACT_NONE, False)[0]
        y, d = ops.conv_forward(x, w, b, None, "linear", L.ACT_GELU, "deriv")[:2]
        out["gelu+deriv"], out["gelu+deriv:saved"] = y, d
        out["residual"] = ops.conv_forward(x, w, b, res, "linear", L.ACT_NONE, False)[0]
        return out
    try:
        lib.tv_set_igemm_config(256, bn, 0, 0)   # (256 x 256 and 256 x 192 tiles: N is a multiple of 192 here, and of 256 but once)
        lib.tv_set_igemm_persist(-1)
        ref = cases()
        lib.tv_set_igemm_persist(-2)
        got = [cases() for _ in range(3)]     # (repeated: a race in the hand-placed waits would come and go)
    finally:
        lib.tv_set_igemm_persist(-2)
        lib.tv_set_igemm_config(0, 0, 0, 0)
    for r in got:
        for k in ref:
            assert torch.isfinite(r[k].float()).all(), k
            assert torch.equal(r[k], ref[k]), (k, float((r[k].float() - ref[k].float()).abs().max()))


# ---- register epilogue forms with shuffled stores (pixel-shuffle / polyphase) against the generic LDS loop -----------------------
@pytest.mark.gpu
@pytest.mark.parametrize("B,H", [(4, 32), (3, 24), (8, 64)])
def test_shuffled_store_register_forms_are_bit_identical_to_the_lds_loop(B, H):
    """DC paths (1x1 + pixel-shuffle store, the data gradient of the 2x2-stride-2 gather), the parity data gradient of a
    stride-2 3x3 convolution (x saved derivative) and the polyphase upsampling convolution (SiLU + saved derivative, rim
    phases clipped): the register forms compute each 16-byte piece's address like the LDS loop does and must give its bits
    -- power-of-two and other grids, full and ragged row tiles."""
    from transvae.hip import _lib as L, ops
    lib = L.load()
    g = torch.Generator(device=dev()).manual_seed(29)
    bf = torch.bfloat16
    C = 192

    def rnd(*shape):
        return torch.randn(*shape, device=dev(), generator=g)
    x = rnd(B, H, H, C).to(bf)
    xl = rnd(B, H // 2, H // 2, C).to(bf)                    # low-resolution input of the up path
    w_shuf = rnd(4 * C, 1, 1, C) * C ** -0.5                 # 1x1 -> 4C, stored pixel-shuffled to [B, H, H, C]
    b_shuf = rnd(4 * C) * 0.1
    res_hi = rnd(B, H, H, C).to(bf)
    w_un = rnd(C, 2, 2, C) * (4 * C) ** -0.5                 # 2x2 / stride-2 gather conv  [B,H,H,C] -> [B,H/2,H/2,C]
    w_s2 = rnd(C, 3, 3, C) * (9 * C) ** -0.5
    w_up = rnd(C, 3, 3, C) * (9 * C) ** -0.5
    b_up = rnd(C) * 0.1
    gz_lo = rnd(B, H // 2, H // 2, C).to(bf)
    der_hi = torch.rand(B, H, H, C, device=dev(), generator=g).to(bf)

    def cases():
        out = {}
        out["shuf"] = ops.conv_forward(xl, w_shuf, b_shuf, None, "shuf", L.ACT_NONE, False)[0]
        out["shuf+res"] = ops.conv_forward(xl, w_shuf, b_shuf, res_hi, "shuf", L.ACT_NONE, False)[0]
        g_un = ops._Geo("unshuf", x, w_un)
        out["unshuf dgrad"] = ops.conv_dgrad(g_un, w_un, gz_lo, x.shape)
        out["unshuf dgrad+res"] = ops.conv_dgrad(g_un, w_un, gz_lo, x.shape, residual=res_hi)
        g_s2 = ops._Geo("c3s2", x, w_s2)
        out["c3s2 dgrad*deriv"] = ops.conv_dgrad(g_s2, w_s2, gz_lo, x.shape, aux=der_hi, aux_act=L.ACT_DERIV)
        out["c3s2 dgrad"] = ops.conv_dgrad(g_s2, w_s2, gz_lo, x.shape)
        y, d = ops.conv_forward(xl, w_up, b_up, None, "c3up", L.ACT_SILU, "deriv")[:2]
        out["c3up silu+deriv"], out["c3up silu+deriv:saved"] = y, d
        out["c3up"] = ops.conv_forward(xl, w_up, b_up, None, "c3up", L.ACT_NONE, False)[0]
        return out
    try:
        lib.tv_set_igemm_epilogue(1)
        reg = cases()
        lib.tv_set_igemm_epilogue(0)
        lds = cases()
    finally:
        lib.tv_set_igemm_epilogue(1)
    for k in reg:
        assert torch.isfinite(reg[k].float()).all(), k
        assert torch.equal(reg[k], lds[k]), (k, float((reg[k].float() - lds[k].float()).abs().max()))


# ---- operands derived from a 3x3 weight in one launch (tv_conv3x3_derived) against the slice algebra they replace -----------------
@pytest.mark.gpu
@pytest.mark.parametrize("Cout,Cin", [(64, 32), (192, 384), (40, 72), (768, 1536)])
def test_derived_3x3_operands_equal_the_slice_algebra(Cout, Cin):
    """Polyphase forward operand, its 4x4 adjoint operand, the fold of the adjoint's weight gradient back onto the nine taps
    (plain and accumulating) and the parity operand of the stride-2 data gradient: bit-identical to the chain of slice adds /
    transposes / casts of torch that transvae/hip/ops.py ran before round 4 (~20-60 launches each)."""
    from transvae.hip import ops
    UP_SETS = (((0,), (1, 2)), ((0, 1), (2,)))
    UP_ADJ = ((2,), (1, 2), (0, 1), (0,))
    g = torch.Generator(device=dev()).manual_seed(Cout * 7 + Cin)
    w = torch.randn(Cout, 3, 3, Cin, device=dev(), generator=g) * (9 * Cin) ** -0.5

    def tapsum(sets_y, sets_x, src=w):
        acc = None
        for ky in sets_y:
            for kx in sets_x:
                acc = src[:, ky, kx, :] if acc is None else acc + src[:, ky, kx, :]
        return acc
    # forward operand
    wf = torch.empty((4, Cout, 2, 2, Cin), dtype=torch.float32, device=dev())
    for py in (0, 1):
        for px in (0, 1):
            for ty in (0, 1):
                for tx in (0, 1):
                    wf[2 * py + px, :, ty, tx, :] = tapsum(UP_SETS[py][ty], UP_SETS[px][tx])
    assert torch.equal(ops._up_fwd_weight(w, Cout, Cin), wf.to(BF).view(4 * Cout, 2, 2, Cin))
    # adjoint operand
    wd = torch.empty((Cin, 4, 4, Cout), dtype=torch.float32, device=dev())
    for ty in range(4):
        for tx in range(4):
            wd[:, ty, tx, :] = tapsum(UP_ADJ[ty], UP_ADJ[tx]).t()
    assert torch.equal(ops._up_dgrad_weight(w, Cout, Cin), wd.to(BF))
    # fold of the adjoint's weight gradient
    d16 = torch.randn(Cin, 4, 4, Cout, device=dev(), generator=g)
    taps = tuple(tuple(t for t in range(4) if k in UP_ADJ[t]) for k in range(3))
    dw = torch.empty((Cout, 3, 3, Cin), dtype=torch.float32, device=dev())
    for ky in range(3):
        for kx in range(3):
            dw[:, ky, kx, :] = tapsum(taps[ky], taps[kx], d16).t()
    assert torch.equal(ops._up_fold_wgrad(d16, Cout, Cin), dw)
    acc0 = torch.randn(Cout, 3, 3, Cin, device=dev(), generator=g)
    acc = acc0.clone()
    ops._up_fold_wgrad(d16, Cout, Cin, out=acc, accumulate=True)
    assert torch.equal(acc, acc0 + dw)
    # parity operand of the stride-2 data gradient
    wp = torch.zeros((4, Cin, 2, 2, Cout), dtype=torch.float32, device=dev())
    for py in (0, 1):
        for px in (0, 1):
            for ty in range(py + 1):
                for tx in range(px + 1):
                    ky = 1 if py == 0 else (2 if ty == 0 else 0)
                    kx = 1 if px == 0 else (2 if tx == 0 else 0)
                    wp[2 * py + px, :, ty, tx, :] = w[:, ky, kx, :].t()
    assert torch.equal(ops._s2_parity_weight(w, Cout, Cin), wp.to(BF).view(4 * Cin, 2, 2, Cout))


# ---- K-concatenated rows from two tensors (tv_igemm_nt_cat2) against the GEMM over the materialised concatenation -----------------
@pytest.mark.gpu
@pytest.mark.parametrize("T,K1,K2,N", [(4096, 384, 384, 1536), (2048 + 72, 1536, 384, 768), (1024, 64, 128, 768), (16384, 1536, 1536, 6144)])
def test_two_source_rows_equal_the_concatenated_gemm(T, K1, K2, N):
    """[x1 | x2] w^T with the eight-phase loop reading K-steps beyond K1 from the second tensor: the bits of the same GEMM on
    torch.cat([x1, x2], 1) -- forward form (bias + residual), data-gradient form (saved-derivative multiply), plain; ragged
    row tiles; K1 of one and of many K-steps; and None (the caller's fallback) where the tile has no two-source loop."""
    from transvae.hip import _lib as L, ops
    g = torch.Generator(device=dev()).manual_seed(T + K1 + N)
    bf = torch.bfloat16
    x1 = torch.randn(T, K1, device=dev(), generator=g).to(bf)
    x2 = torch.randn(T, K2, device=dev(), generator=g).to(bf)
    w = (torch.randn(N, K1 + K2, device=dev(), generator=g) * (K1 + K2) ** -0.5).to(bf)
    b = torch.randn(N, device=dev(), generator=g) * 0.1
    res = torch.randn(T, N, device=dev(), generator=g).to(bf)
    der = torch.rand(T, N, device=dev(), generator=g).to(bf)
    cat = torch.cat([x1, x2], 1)
    lib = L.load()
    try:
        for bn in (256, 192):
            lib.tv_set_igemm_config(256, bn, 0, 0)       # (the 256-row tiles whatever the heuristic says for this M: N is a multiple of 192 and, but once, of 256)
            for kw in (dict(), dict(bias=b, residual=res), dict(aux=der, aux_act=L.ACT_DERIV), dict(residual=res, aux=der, aux_act=L.ACT_ADD)):
                ref = ops.gemm_rows(cat, w, N, **kw)
                for _ in range(2):
                    got = ops.gemm_rows2(x1, x2, w, N, **kw)
                    assert got is not None, "the 256-row tiles have the two-source loop"
                    assert torch.equal(got, ref), (bn, sorted(kw), float((got.float() - ref.float()).abs().max()))
    finally:
        lib.tv_set_igemm_config(0, 0, 0, 0)
    small = ops.gemm_rows2(x1[:64].contiguous(), x2[:64].contiguous(), w, N)      # (a 128-row tile: no two-source loop)
    assert small is None or torch.equal(small, ops.gemm_rows(cat[:64].contiguous(), w, N))
