"""GPU parity at the shapes BASELINE.json's configs 3-5 launch (round-1 VERDICT "configs untested"):

  * DiT-B/2 (D 768, 12 heads) and DiT-XL/2 (D 1152, 16 heads, head dim 72): exact width / heads, depth cut to 2, fp32
    mode <= 1e-3 and bf16 mode vs ``oracle/dit.py`` at a token count where the host heuristics send EVERY Linear of
    the block to the one-workgroup-per-CU 256x256 / 192x384 kernels (a path DiT-S/2 only takes for two Linears);
  * DiT-S/2 at the bench's own per-GPU batch (768 images, M = 196608 token rows) vs the CPU oracle, so the kernels the
    benchmark times are compared with the oracle and not only with their sibling kernels;
  * SDXL-width UNet (block_out_channels 320/640/1280, heads 5/10/20, 77 x 2048 context, text_time 2816; transformer
    depth cut to 1/2/2): forward + every parameter gradient vs ``oracle/unet.py`` at 4x32x32, and one 4x128x128
    forward/backward (T = 4096 self-attention inside the model) checked through size-independent properties.

The oracle runs on the host cores in micro-batches (the gradient of a sum over samples is additive), which bounds its
autograd memory.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item(), ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()


def dit_pair(cfg, dtype, B, micro, seed=0):
    from oracle.dit import DiTOracle
    from uwudiff_amd.dit import DiT, DiTConfig

    torch.manual_seed(seed)
    ora = DiTOracle(**cfg)
    with torch.no_grad():
        for p in ora.parameters():  # non-zero gates / modulation: every branch carries signal
            p.copy_(torch.randn_like(p) * (0.03 if p.dim() > 1 else 0.02))
    model = DiT(DiTConfig(compute_dtype=dtype, **cfg), init="dit").cuda()
    model.load_state_dict(ora.state_dict())
    S, C = cfg["sample_size"], cfg["in_channels"]
    g = torch.Generator().manual_seed(seed + 1)
    x = torch.randn(B, C, S, S, generator=g)
    t = torch.randint(0, 1000, (B,), generator=g)
    pooled = torch.randn(B, cfg["cond_dim"], generator=g)
    dout = torch.randn(B, cfg["out_channels"], S, S, generator=g) / (S * S)
    ys = []
    for lo in range(0, B, micro):
        sl = slice(lo, lo + micro)
        yo = ora(x[sl], t[sl], added_cond_kwargs={"text_embeds": pooled[sl]})[0]
        yo.backward(dout[sl])
        ys.append(yo.detach())
    yo = torch.cat(ys)
    y = model(x.cuda(), t.cuda(), added_cond_kwargs={"text_embeds": pooled.cuda()})[0]
    y.backward(dout.cuda())
    torch.cuda.synchronize()
    og = dict(ora.named_parameters())
    grads = {name: (model.grad_view(name), og[name].grad) for name, _ in model.named_tensors()}
    return y, yo, grads


DIT_B2 = dict(depth=2, hidden=768, heads=12, patch=2, sample_size=32, in_channels=4, out_channels=4, cond_dim=1280)
DIT_XL2 = dict(depth=2, hidden=1152, heads=16, patch=2, sample_size=32, in_channels=4, out_channels=4, cond_dim=1280)
DIT_S2 = dict(depth=12, hidden=384, heads=6, patch=2, sample_size=32, in_channels=4, out_channels=4, cond_dim=1280)


@pytest.mark.parametrize("cfg", [DIT_B2, DIT_XL2], ids=["dit_b2_w768_h12", "dit_xl2_w1152_h16"])
def test_wide_dit_fp32_matches_oracle(cfg):
    y, yo, grads = dit_pair(cfg, "fp32", B=3, micro=3)
    l2, mx = rel(y, yo)
    assert l2 < 1e-3 and mx < 1e-3, (l2, mx)
    bad = {n: rel(g, go) for n, (g, go) in grads.items() if max(rel(g, go)) >= 1e-3}
    assert not bad, bad


@pytest.mark.parametrize("cfg,B", [(DIT_B2, 96), (DIT_XL2, 64)], ids=["dit_b2_M24576", "dit_xl2_M16384"])
def test_wide_dit_bf16_big_kernels_match_oracle(cfg, B):
    """M = B x 256 token rows is large enough that pick_r3() returns the 256-row ring for every Linear, i.e. the
    256x256 kernel (N % 256 == 0: every DiT-B/2 Linear; fc1 of DiT-XL/2) or the 192x384 kernel (N = 1152 / 3456) runs."""
    y, yo, grads = dit_pair(cfg, "bf16", B=B, micro=32, seed=5)
    l2, _ = rel(y, yo)
    assert l2 < 3e-2, l2
    bad = {n: rel(g, go)[0] for n, (g, go) in grads.items() if rel(g, go)[0] >= 6e-2}
    assert not bad, bad


def test_dit_s2_bench_batch_768_bf16_matches_oracle():
    """The bench's launch shapes (per-GPU batch 768, M = 196608): gemm_wide_kernel / gemm_big_kernel / the streaming
    weight-gradient kernel at their real tile and K-slice counts, end to end against the fp32 CPU oracle.  Depth 2 of the 12
    blocks: every block launch has the bench's shape, and the CPU oracle's 768-image pass takes ~35 s instead of 190 (the suite's
    budget: VERDICT r3 asks for <= 600 s); the full depth runs in test_full_depth_dit_properties below and in the loss-curve tests."""
    y, yo, grads = dit_pair(dict(DIT_S2, depth=2), "bf16", B=768, micro=96, seed=7)
    l2, _ = rel(y, yo)
    assert l2 < 3e-2, l2
    bad = {n: rel(g, go)[0] for n, (g, go) in grads.items() if rel(g, go)[0] >= 6e-2}
    assert not bad, bad


# ---------------------------------------------------------------------------------------------------- SDXL widths
SDXL_CUT = dict(in_channels=4, out_channels=4, block_out_channels=(320, 640, 1280), layers_per_block=2,
                down_block_types=("DownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D"),
                up_block_types=("CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "UpBlock2D"),
                transformer_layers_per_block=(1, 2, 2), attention_head_dim=(5, 10, 20), cross_attention_dim=2048,
                addition_embed_type="text_time", addition_time_embed_dim=256, projection_class_embeddings_input_dim=2816,
                norm_num_groups=32)


_ORA_INIT = {}  # seed -> the oracle's initial parameters (unet_models is called four times with two seeds)


def unet_models(dtype, seed=0, with_oracle=True):
    from oracle.unet import UNetOracle
    from uwudiff_amd.unet import UNet2DConditionModel

    torch.manual_seed(seed)
    # (every parameter is overwritten just below: the modules' own initialisers -- 40 s of the suite for 0.9 B parameters -- are
    # switched off while the oracle is built)
    import torch.nn.init as I

    saved = {n: getattr(I, n) for n in ("kaiming_uniform_", "uniform_", "normal_", "xavier_uniform_", "trunc_normal_")}
    try:
        for n in saved:
            setattr(I, n, lambda t, *a, **k: t)
        ora = UNetOracle(**SDXL_CUT)
    finally:
        for n, f in saved.items():
            setattr(I, n, f)
    with torch.no_grad():  # away from the near-zero init so every branch carries signal
        if seed in _ORA_INIT:  # (the same 0.9 G draws again: copied instead)
            for n, p in ora.named_parameters():
                p.copy_(_ORA_INIT[seed][n])
        else:
            for n, p in ora.named_parameters():
                if p.dim() > 1:
                    p.copy_(torch.randn_like(p) * (0.5 / p[0].numel() ** 0.5))
                elif n.endswith("bias"):
                    p.copy_(torch.randn_like(p) * 0.05)
                else:
                    p.copy_(1 + torch.randn_like(p) * 0.1)
            _ORA_INIT[seed] = {n: p.detach().clone() for n, p in ora.named_parameters()}
    model = UNet2DConditionModel(SDXL_CUT, compute_dtype=dtype, init_weights=False).cuda()  # (every parameter is loaded next)
    model.load_state_dict(ora.state_dict())
    return ora, model


def unet_inputs(B, S, seed):
    g = torch.Generator().manual_seed(seed)
    return dict(x=torch.randn(B, 4, S, S, generator=g), t=torch.randint(0, 1000, (B,), generator=g),
                ctx=torch.randn(B, 77, 2048, generator=g), pooled=torch.randn(B, 1280, generator=g),
                ids=torch.tensor([[1024.0, 1024, 0, 0, 1024, 1024]] * B),
                dout=torch.randn(B, 4, S, S, generator=g) / (S * S))


def unet_run(m, i, dev):
    mv = lambda v: v.to(dev)  # noqa: E731
    y = m(mv(i["x"]), mv(i["t"]), encoder_hidden_states=mv(i["ctx"]),
          added_cond_kwargs={"text_embeds": mv(i["pooled"]), "time_ids": mv(i["ids"])})[0]
    y.backward(mv(i["dout"]))
    return y


_ORACLE_RUN = {}  # the CPU oracle's forward + backward (a minute of the suite) is the same for both compute modes: run it once


@pytest.mark.parametrize("dtype,ybar,gbar", [("fp32", 1e-3, 2e-3), ("bf16", 4e-2, 0.12)])
def test_sdxl_width_unet_matches_oracle(dtype, ybar, gbar):
    ora, model = unet_models(dtype)  # (seeded: the same oracle weights for every parametrisation)
    i = unet_inputs(2, 32, seed=1)
    if "y" not in _ORACLE_RUN:
        _ORACLE_RUN["y"] = unet_run(ora, i, "cpu").detach()
        _ORACLE_RUN["g"] = {n: p.grad.clone() for n, p in ora.named_parameters()}
    yo, og = _ORACLE_RUN["y"], _ORACLE_RUN["g"]
    y = unet_run(model, i, "cuda")
    torch.cuda.synchronize()
    l2, mx = rel(y, yo)
    assert l2 < ybar, (l2, mx)
    bad = {}
    for n in model.P.registry:
        e = rel(model.grad_tensor(n), og[n])[0]
        if e > gbar:
            bad[n] = e
    assert not bad, bad


def test_sdxl_width_unet_128_latents_properties():
    """4 x 128 x 128 latents: T = 4096 / 1024 self-attention, stride-2 convolutions and the 77-token cross-attention at
    the widths of BASELINE config 4.  Properties that do not need a CPU run of this size: finite outputs, gradients
    additive over samples, and the per-sample output independent of its batch neighbours."""
    _, model = unet_models("bf16", seed=3)
    i = unet_inputs(2, 128, seed=4)
    model.flat.grad = torch.zeros_like(model.flat.data)
    y = unet_run(model, i, "cuda")
    full = model.flat.grad.clone()
    assert torch.isfinite(y).all() and torch.isfinite(full).all() and float(full.abs().max()) > 0
    model.flat.grad.zero_()
    ys = []
    for b in range(2):
        ib = {k: v[b:b + 1] for k, v in i.items()}
        ys.append(unet_run(model, ib, "cuda"))
    l2, mx = rel(torch.cat(ys), y)
    assert l2 < 2e-2, (l2, mx)  # bf16: split-K slice counts differ between the two batch sizes
    l2, mx = rel(model.flat.grad, full)
    assert l2 < 2e-2, (l2, mx)


# ---------------------------------------------------------------------------------------------------- full depth
def _full_depth_properties(model, make_inputs, B, steps=3, lr=2e-4, add_tol=2e-2):
    """Size-independent properties at the FULL depth of a BASELINE config (the oracle comparisons above
    cut the depth): finite loss and gradients, gradients additive over two half batches, and the eps-MSE loss decreasing
    over `steps` AdamW steps on a fixed batch."""
    from uwudiff_amd.optim import FusedAdamW

    inp = make_inputs(B)

    def loss_of(sl):
        out = model(inp["x"][sl], inp["t"][sl], **{k: (v[sl] if torch.is_tensor(v) else {kk: vv[sl] for kk, vv in v.items()})
                                                   for k, v in inp["kw"].items()})[0]
        n = inp["x"][sl].shape[0]
        return ((out - inp["eps"][sl]) ** 2).flatten(1).mean(1).sum() / B, n

    model.flat.grad = torch.zeros_like(model.flat.data)
    l_full, _ = loss_of(slice(0, B))
    l_full.backward()
    full = model.flat.grad.clone()
    assert torch.isfinite(l_full) and torch.isfinite(full).all() and float(full.abs().max()) > 0
    model.flat.grad.zero_()
    for sl in (slice(0, B // 2), slice(B // 2, B)):
        loss_of(sl)[0].backward()
    l2, mx = rel(model.flat.grad, full)
    assert l2 < add_tol, (l2, mx)
    opt = FusedAdamW(model.parameters(), lr=lr, weight_decay=0.0)
    losses = []
    for _ in range(steps + 1):
        model.flat.grad.zero_()
        l, _ = loss_of(slice(0, B))
        losses.append(float(l.detach()))
        l.backward()
        opt.step()
    assert all(torch.isfinite(torch.tensor(losses))) and losses[-1] < losses[0], losses
    return losses


def _dit_inputs(B, seed=0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    r = lambda *s: torch.randn(*s, device="cuda", generator=g)  # noqa: E731
    return dict(x=r(B, 4, 32, 32), t=torch.randint(0, 1000, (B,), device="cuda", generator=g).float(), eps=r(B, 4, 32, 32),
                kw={"added_cond_kwargs": {"text_embeds": r(B, 1280)}})


@pytest.mark.parametrize("preset,dtype,B", [("DiT-B/2", "bf16", 32), ("DiT-XL/2", "fp8", 32)], ids=["dit_b2_L12", "dit_xl2_L28_fp8"])
def test_full_depth_dit_properties(preset, dtype, B):
    from uwudiff_amd.dit import DiT

    torch.manual_seed(17)
    model = DiT.from_config(preset, cond_dim=1280, init="random", compute_dtype=dtype, fp8_scaling="jit").cuda()
    assert model.cfg.depth == {"DiT-B/2": 12, "DiT-XL/2": 28}[preset]
    # fp8: per-tensor scales are taken per call, so half batches quantise on slightly different grids
    _full_depth_properties(model, _dit_inputs, B, add_tol=2e-2 if dtype == "bf16" else 8e-2)


def test_full_depth_sdxl_unet_properties():
    """The whole SDXL stack (transformer depth 1 / 2 / 10, 2.57 B parameters, 77 x 2048 context) at 4x32x32, batch 2."""
    from uwudiff_amd.unet import UNet2DConditionModel

    torch.manual_seed(19)
    model = UNet2DConditionModel.from_config("sdxl", compute_dtype="bf16", device="cuda")  # (initial weights drawn on the GPU)
    assert sum(v.numel() for _, v in model.state_dict().items()) == 2_567_463_684

    def inputs(B):
        g = torch.Generator(device="cuda").manual_seed(3)
        r = lambda *s: torch.randn(*s, device="cuda", generator=g)  # noqa: E731
        ids = torch.tensor([[1024.0, 1024, 0, 0, 1024, 1024]] * B, device="cuda")
        return dict(x=r(B, 4, 32, 32), t=torch.randint(0, 1000, (B,), device="cuda", generator=g), eps=r(B, 4, 32, 32),
                    kw={"encoder_hidden_states": r(B, 77, 2048), "added_cond_kwargs": {"text_embeds": r(B, 1280), "time_ids": ids}})

    _full_depth_properties(model, inputs, 2, steps=3, lr=1e-5, add_tol=3e-2)
