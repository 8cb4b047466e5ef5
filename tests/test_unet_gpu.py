"""GPU parity of the UNet2DConditionModel-shape denoiser (HIP kernels) vs the fp32 CPU oracle (oracle/unet.py).

fp32 compute mode: <= 1e-3 relative (north-star bar).  bf16 mode: own tolerance.  The oracle itself is structurally
pinned by the public SDXL parameter count (tests/test_unet_cpu.py)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

TINY = dict(in_channels=4, out_channels=4, block_out_channels=(32, 64), layers_per_block=1,
            down_block_types=("DownBlock2D", "CrossAttnDownBlock2D"), up_block_types=("CrossAttnUpBlock2D", "UpBlock2D"),
            transformer_layers_per_block=(1, 2), attention_head_dim=(1, 1), cross_attention_dim=32,
            addition_embed_type="text_time", addition_time_embed_dim=8, projection_class_embeddings_input_dim=16 + 48,
            norm_num_groups=8)
PIX3 = dict(TINY, in_channels=3, out_channels=3, block_out_channels=(64, 128), attention_head_dim=(1, 2),
            transformer_layers_per_block=(1, 1))


def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item(), ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()


def run_pair(cfg, dtype, B=2, S=16, Tk=7, seed=0, mask=None, W=None):
    from oracle.unet import UNetOracle
    from uwudiff_amd.unet import UNet2DConditionModel

    torch.manual_seed(seed)
    ora = UNetOracle(**cfg)
    with torch.no_grad():  # away from the near-zero init so every branch carries signal
        for n, p in ora.named_parameters():
            if p.dim() > 1:
                p.copy_(torch.randn_like(p) * (0.5 / p[0].numel() ** 0.5))
            elif n.endswith("bias"):
                p.copy_(torch.randn_like(p) * 0.05)
            else:
                p.copy_(1 + torch.randn_like(p) * 0.1)
    model = UNet2DConditionModel(cfg, compute_dtype=dtype).cuda()
    model.load_state_dict(ora.state_dict())
    g = torch.Generator().manual_seed(seed + 1)
    W = W or S
    x = torch.randn(B, cfg["in_channels"], S, W, generator=g)
    t = torch.randint(0, 1000, (B,), generator=g)
    ctx = torch.randn(B, Tk, cfg["cross_attention_dim"], generator=g)
    pooled = torch.randn(B, 16, generator=g)
    ids = torch.tensor([[1024.0, 1024, 0, 0, 1024, 1024]] * B)
    dout = torch.randn(B, cfg["out_channels"], S, W, generator=g) / (S * W)
    yo = ora(x, t, encoder_hidden_states=ctx, encoder_attention_mask=mask,
             added_cond_kwargs={"text_embeds": pooled, "time_ids": ids})[0]
    yo.backward(dout)
    y = model(x.cuda(), t.cuda(), encoder_hidden_states=ctx.cuda(),
              encoder_attention_mask=None if mask is None else mask.cuda(),
              added_cond_kwargs={"text_embeds": pooled.cuda(), "time_ids": ids.cuda()})[0]
    y.backward(dout.cuda())
    torch.cuda.synchronize()
    og = dict(ora.named_parameters())
    grads = {n: (model.grad_tensor(n), og[n].grad) for n in model.P.registry}
    return y, yo, grads, model, ora


@pytest.mark.parametrize("cfg", [TINY, PIX3], ids=["tiny_latent", "tiny_pixel3"])
def test_unet_fp32_matches_oracle(cfg):
    y, yo, grads, model, ora = run_pair(cfg, "fp32")
    l2, mx = rel(y, yo)
    assert l2 < 1e-3 and mx < 1e-3, (l2, mx)
    for name, (g, go) in grads.items():
        l2, mx = rel(g, go)
        assert l2 < 2e-3, (name, l2, mx)
    # state_dict round trip in diffusers layout
    sd = model.state_dict()
    for k, v in ora.state_dict().items():
        torch.testing.assert_close(sd[k].cpu(), v, rtol=0, atol=0)


@pytest.mark.parametrize("dtype,bar", [("fp32", 1e-3), ("bf16", 4e-2)])
def test_unet_encoder_attention_mask(dtype, bar):
    """encoder_attention_mask [B,S] (1 = keep) reaches the cross-attention keys as the additive bias
    (1 - m) * -10000 (reference rope_unet.py:448-453); masked tokens must stop influencing the output."""
    mask = torch.tensor([[1, 1, 1, 1, 0, 0, 0], [1, 0, 1, 1, 1, 1, 0]])
    y, yo, grads, model, ora = run_pair(TINY, dtype, mask=mask)
    l2, mx = rel(y, yo)
    assert l2 < bar, (l2, mx)
    gbar = 2e-3 if dtype == "fp32" else 0.12
    bad = {n: rel(g, go)[0] for n, (g, go) in grads.items() if rel(g, go)[0] > gbar}
    assert not bad, bad
    y_nomask, yo_nomask, *_ = run_pair(TINY, dtype)
    assert rel(yo, yo_nomask)[0] > 5e-2 and rel(y, y_nomask)[0] > 5e-2  # the mask matters on this input
    with pytest.raises(ValueError):
        model(torch.zeros(2, 4, 16, 16, device="cuda"), torch.zeros(2, device="cuda"),
              encoder_hidden_states=torch.zeros(2, 7, 32, device="cuda"), encoder_attention_mask=torch.ones(2, 5).cuda(),
              added_cond_kwargs={"text_embeds": torch.zeros(2, 16).cuda(), "time_ids": torch.zeros(2, 6).cuda()})


def test_unet_bf16_close_to_oracle():
    y, yo, grads, _, _ = run_pair(TINY, "bf16", B=3)
    l2, _ = rel(y, yo)
    assert l2 < 4e-2, l2
    bad = {n: rel(g, go)[0] for n, (g, go) in grads.items() if rel(g, go)[0] > 0.12}
    assert not bad, bad


def test_unet_in_diffusion_loss_step():
    """The UNet plugs into the objective + optimizer exactly like the DiT (denoiser slot contract)."""
    from uwudiff_amd.objective import DiffusionLoss
    from uwudiff_amd.optim import FusedAdamW
    from uwudiff_amd.scheduler import EulerDiscreteScheduler
    from uwudiff_amd.unet import UNet2DConditionModel

    torch.manual_seed(0)
    m = UNet2DConditionModel(TINY, compute_dtype="bf16").cuda()
    opt = FusedAdamW(m.parameters(), lr=1e-4)
    lf = DiffusionLoss(EulerDiscreteScheduler.from_pretrained("sdxl"))
    x = torch.randn(4, 4, 16, 16, device="cuda")
    kw = dict(encoder_hidden_states=torch.randn(4, 7, 32, device="cuda"),
              added_cond_kwargs={"text_embeds": torch.randn(4, 16, device="cuda"),
                                 "time_ids": torch.tensor([[1024.0, 1024, 0, 0, 1024, 1024]] * 4, device="cuda")})
    p0 = m.flat.data.clone()
    loss, aux = lf(x, m, **kw)
    loss.backward()
    opt.step()
    assert torch.isfinite(loss) and aux.pred.shape == x.shape
    assert (m.flat.data - p0).abs().max().item() > 0


def test_unet_gradient_checkpointing_same_gradients_less_memory():
    """enable_gradient_checkpointing() (reference test_scripts/test_train.py:38-39): resnets and Transformer2D stacks are
    recomputed in the backward.  Output and parameter gradients agree with the plain path to within its own run-to-run
    noise (fp32 atomics in the GroupNorm statistics and bias gradients), with a fraction of the saved activations."""
    from tests.test_configs_gpu import unet_inputs, unet_models

    _, model = unet_models("bf16", seed=5)
    i = unet_inputs(2, 64, seed=6)
    mv = lambda v: v.cuda()  # noqa: E731

    def run(ckpt):
        model.enable_gradient_checkpointing(ckpt)
        model.flat.grad = torch.zeros_like(model.flat.data)
        torch.cuda.synchronize()
        torch.cuda.reset_peak_memory_stats()
        base = torch.cuda.memory_allocated()
        y = model(mv(i["x"]), mv(i["t"]), encoder_hidden_states=mv(i["ctx"]),
                  added_cond_kwargs={"text_embeds": mv(i["pooled"]), "time_ids": mv(i["ids"])})[0]
        held = torch.cuda.memory_allocated() - base  # what the graph keeps alive for the backward
        y.backward(mv(i["dout"]))
        torch.cuda.synchronize()
        return y.detach().clone(), model.flat.grad.clone(), held

    y0, g0, held0 = run(False)
    ya, ga, _ = run(False)  # run-to-run noise of the plain path (GroupNorm statistics are summed with fp32 atomics)
    y1, g1, held1 = run(True)
    model.enable_gradient_checkpointing(False)
    noise_y, noise_g = rel(ya, y0)[0], ((ga - g0).norm() / g0.norm()).item()
    assert rel(y1, y0)[0] <= max(3 * noise_y, 5e-3), (rel(y1, y0), noise_y)  # (bf16: one ulp is 4e-3)
    err = ((g1 - g0).norm() / g0.norm()).item()
    assert err <= max(3 * noise_g, 5e-3), (err, noise_g)
    assert held1 < 0.25 * held0, (held0, held1)


@pytest.mark.parametrize("cfg,W", [(dict(TINY, rope=True), None), (dict(PIX3, rope=True), 24)], ids=["tiny_latent", "pixel3_16x24"])
def test_rope_unet_fp32_matches_oracle(cfg, W):
    """RoPEUNet2DConditionModel (reference rope_unet.py:589-608): axial RoPE in every attention of every transformer block
    (q always, k in self-attention only; positions = make_axial_pos of the feature map, also non-square), learnable
    per-head log-frequencies with gradients -- forward and every gradient against the oracle restatement."""
    y, yo, grads, model, ora = run_pair(cfg, "fp32", W=W)
    l2, mx = rel(y, yo)
    assert l2 < 1e-3 and mx < 1e-3, (l2, mx)
    assert any("axial_rope" in n for n in grads)
    for name, (g, go) in grads.items():
        l2, mx = rel(g, go)
        assert l2 < 2e-3, (name, l2, mx)
    y0, *_ = run_pair({k: v for k, v in cfg.items() if k != "rope"}, "fp32", W=W)
    assert rel(y, y0)[0] > 1e-2  # the rotation matters on this input


def test_rope_unet_bf16_close_to_oracle():
    y, yo, grads, _, _ = run_pair(dict(TINY, rope=True), "bf16", B=3)
    l2, _ = rel(y, yo)
    assert l2 < 4e-2, l2
    bad = {n: rel(g, go)[0] for n, (g, go) in grads.items() if rel(g, go)[0] > 0.12}
    assert not bad, bad


def test_hd_unet_zero_init_is_exact_and_an_identity_free_start():
    """HDUNet2DConditionModel (rope_unet.py:562-580): residual-branch outputs start at EXACTLY zero, so the freshly built
    network outputs conv_out's bias = 0 everywhere, and only conv_out.weight receives a gradient."""
    from duwu.modules.rope_unet import HDUNet2DConditionModel, RoPEUNet2DConditionModel

    torch.manual_seed(0)
    for cls in (HDUNet2DConditionModel, RoPEUNet2DConditionModel):
        m = cls.from_config(dict(TINY), compute_dtype="fp32").cuda()
        for n in m.P.registry:
            if n.endswith((".conv2.weight", "attn1.to_out.0.weight", "attn2.to_out.0.weight", "ff.net.2.weight")) or n == "conv_out.weight":
                assert float(m.P.w32(n).abs().max()) == 0.0, n
        assert m.cfg_dict["rope"] == (cls is RoPEUNet2DConditionModel)
        x = torch.randn(2, 4, 16, 16, device="cuda")
        y = m(x, torch.tensor([3, 500], device="cuda"), encoder_hidden_states=torch.randn(2, 7, 32, device="cuda"),
              added_cond_kwargs={"text_embeds": torch.randn(2, 16, device="cuda"), "time_ids": torch.zeros(2, 6, device="cuda")})[0]
        assert float(y.detach().abs().max()) == 0.0
        (y * torch.randn_like(y)).sum().backward()
        nz = [n for n in m.P.registry if float(m.grad_tensor(n).abs().max()) > 0]
        assert nz and all(n.startswith("conv_out") for n in nz), nz
