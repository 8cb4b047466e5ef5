"""CPU: structural pins of the UNet restatement and host-side layout conversion."""
import torch


def test_oracle_sdxl_parameter_count_pin():
    # public SDXL UNet size: 2.567 B parameters
    from oracle.unet import SDXL_UNET_CONFIG, UNetOracle

    with torch.device("meta"):
        m = UNetOracle(**SDXL_UNET_CONFIG)
    n = sum(p.numel() for p in m.parameters())
    assert n == 2_567_463_684


def test_product_registry_matches_oracle_names_and_sizes():
    from oracle.unet import UNetOracle
    from uwudiff_amd.unet import TINY_UNET_CONFIG, UNet2DConditionModel

    cfg = {k: v for k, v in TINY_UNET_CONFIG.items() if k != "sample_size"}
    o = UNetOracle(**cfg)
    m = UNet2DConditionModel(cfg, compute_dtype="fp32")
    sd_o, sd_m = o.state_dict(), m.state_dict()
    assert set(sd_o) == set(sd_m)
    for k in sd_o:
        assert sd_o[k].shape == sd_m[k].shape, k
    m.load_state_dict(sd_o)
    for k, v in m.state_dict().items():
        assert torch.equal(v, sd_o[k]), k
    # init rule of reference unet_patch.py:34-45: residual-branch out layers start near zero
    m2 = UNet2DConditionModel(cfg, compute_dtype="fp32")
    sd = m2.state_dict()
    assert sd["down_blocks.0.resnets.0.conv2.weight"].abs().max() < 1e-3
    assert sd["mid_block.attentions.0.transformer_blocks.0.attn1.to_out.0.weight"].abs().max() < 1e-3
    assert sd["conv_out.weight"].abs().max() < 1e-3
    assert sd["down_blocks.0.resnets.0.conv1.weight"].abs().max() > 1e-2


def test_oracle_unet_rope_helpers_are_pinned_by_the_reference_fixture():
    """oracle/unet.py's axial_rope / make_axial_pos / AxialRoPE init (used by the RoPE-UNet oracle) against the fixture the
    REFERENCE's own src/duwu/modules/rope.py produced (tests/golden/axial_rope.npz, oracle/make_golden.py)."""
    import torch

    from oracle.unet import UNetOracle, _AxialRoPE, axial_rope, make_axial_pos
    from tests.golden_util import load

    _, d = load("axial_rope")
    y = axial_rope(d["x"][0], d["pos"][0], d["freqs_h"], d["freqs_w"])
    torch.testing.assert_close(y, d["y"][0], rtol=2e-6, atol=2e-6)
    torch.testing.assert_close(make_axial_pos(8, 8), d["pos"][0], rtol=0, atol=1e-7)
    torch.testing.assert_close(_AxialRoPE(64, 4).freqs_h.detach(), d["freqs_h"], rtol=1e-6, atol=1e-6)
    # non-square maps: the longer side spans [-1, 1], the shorter one [-ar, ar] (rope.py:10-27), (y, x) order
    p = make_axial_pos(2, 4)
    assert p.shape == (8, 2) and abs(float(p[:, 1].max()) - 0.75) < 1e-6 and abs(float(p[:, 0].max()) - 0.25) < 1e-6
    # the RoPE variant adds exactly 2 x [heads, head_dim / 4] parameters per attention; zero init is exact
    kw = dict(block_out_channels=(32, 64), layers_per_block=1, down_block_types=("DownBlock2D", "CrossAttnDownBlock2D"),
              up_block_types=("CrossAttnUpBlock2D", "UpBlock2D"), transformer_layers_per_block=(1, 2), attention_head_dim=(1, 1),
              cross_attention_dim=32, projection_class_embeddings_input_dim=64, addition_time_embed_dim=8, norm_num_groups=8)
    a, b = UNetOracle(**kw), UNetOracle(rope=True, **kw)
    n_attn = sum(1 for n, _ in b.named_parameters() if n.endswith("axial_rope.freqs_h"))
    assert n_attn == 2 * (2 + 2 + 4) and sum(p.numel() for p in b.parameters()) - sum(p.numel() for p in a.parameters()) == n_attn * 2 * 16
    b.init_weight_zero()
    assert float(b.conv_out.weight.abs().max()) == 0.0
