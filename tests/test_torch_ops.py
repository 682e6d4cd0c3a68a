"""`torch.ops.anystereo.*`: registration and the no-CPU-kernel rule on the host; on the GPU the registered
operators against the reference's golden vectors / the oracle, and gradients through the dispatcher."""
import pytest
import torch


def test_torch_ops_registered_and_cuda_only():
    import anystereo  # noqa: F401  (registers the namespace)
    from anystereo import torch_ops
    for name in torch_ops.OPS:
        op = getattr(torch.ops.anystereo, name)
        assert str(op.default._schema).startswith(f"anystereo::{name}(")
        assert torch._C._dispatch_has_kernel_for_dispatch_key(f"anystereo::{name}", "CUDA")
        assert not torch._C._dispatch_has_kernel_for_dispatch_key(f"anystereo::{name}", "CPU")
    z = torch.zeros(1, 96, 4, 8)
    with pytest.raises(NotImplementedError, match="CPU"):
        torch.ops.anystereo.gwc_volume(z, z, 4, 8)
    with pytest.raises(NotImplementedError, match="CPU"):
        torch.ops.anystereo.geo_corr_lookup([], [torch.zeros(1, 2, 8, 8)], torch.zeros(1, 1, 2, 8), 4)
    with pytest.raises(NotImplementedError, match="CPU"):
        torch.ops.anystereo.corr_sampler_forward(torch.zeros(1, 2, 8, 8), torch.zeros(1, 2, 2, 8), 4)


DEV = "cuda:0"


def _close(a, b, tol, what):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = (a - b).abs().max().item()
    lim = tol * (1.0 + b.abs().max().item())
    assert torch.isfinite(a).all() and err <= lim, f"{what}: max abs err {err:.3e} > {lim:.3e}"


@pytest.mark.gpu
def test_torch_ops_volumes_and_lookup_golden(golden):
    """corr_build_pyramid -> geo_pyramid -> geo_corr_lookup through the dispatcher == the reference's lookup (G3)."""
    import anystereo  # noqa: F401
    T = torch.ops.anystereo
    for name, L in (("lookup_igev_even", 2), ("lookup_igev_odd", 2), ("lookup_raft", 4)):
        g = golden(name)
        corr = T.corr_build_pyramid(g["f1"].to(DEV), g["f2"].to(DEV), L)
        geo = T.geo_pyramid(g["gev"].to(DEV), L) if "gev" in g else []
        assert len(corr) == L and len(geo) in (0, L)
        for i in range(L):
            _close(corr[i], g[f"corr{i}"], 3e-5, f"{name}/corr{i}")
        _close(T.geo_corr_lookup(geo, corr, g["disp"].to(DEV), 4), g["out"], 3e-5, f"{name}/lookup")


@pytest.mark.gpu
def test_torch_ops_gwc_regression_sampler():
    import anystereo  # noqa: F401
    from anystereo.harness.synthetic import det_uniform as U
    from oracle import ops as O
    T = torch.ops.anystereo
    fl, fr = U((1, 96, 5, 60), 1), U((1, 96, 5, 60), 2)
    _close(T.gwc_volume(fl.to(DEV), fr.to(DEV), 48, 8), O.gwc_volume(fl.double(), fr.double(), 48, 8), 2e-5, "gwc")
    cost = U((2, 48, 5, 7), 3, -4.0, 4.0)
    _close(T.disparity_regression(cost.to(DEV), True), O.disparity_regression(torch.softmax(cost.double(), 1), 48), 2e-5, "dispreg")
    vol, co = U((1, 4, 9, 9), 4), U((1, 2, 4, 9), 5, -2.0, 11.0)
    from anystereo import corr_sampler
    assert torch.equal(T.corr_sampler_forward(vol.to(DEV), co.to(DEV), 3), corr_sampler.forward(vol.to(DEV), co.to(DEV), 3)[0])
    x = U((1, 24, 6, 9), 6)
    _close(T.structure_feature(x.to(DEV)), O.structure_feature_v2isu(x.double()), 2e-5, "structure feature")


@pytest.mark.gpu
def test_torch_ops_update_block_golden(golden):
    """motion_encoder / convgru_step / disp_head with the weights passed as tensors == the reference's modules (G5)."""
    import anystereo.nn.update  # noqa: F401
    from anystereo.harness.synthetic import fill_module_deterministic
    from anystereo.models.base import default_args
    from anystereo.nn.update import BasicMultiUpdateBlock
    T = torch.ops.anystereo
    g = golden("update_igev")
    args = default_args("continuous_IGEVStereo")
    ub = BasicMultiUpdateBlock(args, hidden_dims=args.hidden_dims, geo_channels=8).eval()
    fill_module_deterministic(ub, base_seed=5)
    ub = ub.to(DEV)

    def wb(*mods):
        return [m.weight.detach() for m in mods], [m.bias.detach() for m in mods]
    e = ub.encoder
    with torch.no_grad():
        motion = T.motion_encoder(g["disp"].to(DEV), g["corr"].to(DEV), *wb(e.convc1, e.convc2, e.convd1, e.convd2, e.conv))
        _close(motion, g["motion"], 3e-5, "motion_encoder")
        _close(T.disp_head(g["net0"].to(DEV), *wb(ub.disp_head.conv1, ub.disp_head.conv2)), g["head"], 3e-5, "disp_head")
        net = [g[f"net{i}"].to(DEV) for i in range(3)]
        cz, cr, cq = g["ctx2"].to(DEV).split(128, dim=1)
        h16 = T.convgru_step(net[2], cz, cr, cq, [anystereo.ops.pool2x(net[1])], *wb(ub.gru16.convz, ub.gru16.convr, ub.gru16.convq))
        assert torch.equal(h16, ub.gru16(net[2], cz, cr, cq, anystereo.nn.update.pool2x(net[1])))
        _close(h16, g["gru16"], 3e-5, "convgru_step")


@pytest.mark.gpu
def test_torch_ops_liif_and_upsample_golden(golden):
    import anystereo  # noqa: F401
    from anystereo.models import __models__, default_args
    from anystereo.harness.synthetic import det_uniform as U
    T = torch.ops.anystereo
    model = __models__["continuous_IGEVStereo"](default_args("continuous_IGEVStereo")).to(DEV).eval()
    up = model.liif_up
    lin = [l for l in up.imnet.layers if isinstance(l, torch.nn.Linear)]
    enc = [sf_c for sf_c in (176, 32)]
    feats = [U((1, enc[0], 8, 12), 1).to(DEV), U((1, enc[1], 16, 24), 2).to(DEV)]
    coord = U((1, 500, 2), 3).to(DEV)
    with torch.no_grad():
        ref = up(feats, coord)
        out = T.liif_upsample(feats, coord, [l.weight.detach() for l in lin], [l.bias.detach() for l in lin])
    assert torch.equal(out, ref)
    disp = U((1, 1, 8, 12), 4, 0.0, 30.0).to(DEV)
    from anystereo import ops
    assert torch.equal(T.convex_upsample(disp, out, coord, None, True), ops.convex_upsample(disp, out, coord, mask_is_logits=True))


@pytest.mark.gpu
def test_torch_ops_autograd_through_dispatcher():
    """Gradients of the registered operators (AutogradCUDA kernel -> HIP backward) == autograd of the fp64 oracle."""
    import anystereo  # noqa: F401
    from anystereo.harness.synthetic import det_uniform as U
    from oracle import ops as O
    T = torch.ops.anystereo
    fl, fr = U((1, 96, 3, 40), 11), U((1, 96, 3, 40), 12)
    a, b = fl.to(DEV).requires_grad_(), fr.to(DEV).requires_grad_()
    wgt = U((1, 8, 48, 3, 40), 13)
    (T.gwc_volume(a, b, 48, 8) * wgt.to(DEV)).sum().backward()
    ar, br = fl.double().requires_grad_(), fr.double().requires_grad_()
    (O.gwc_volume(ar, br, 48, 8) * wgt.double()).sum().backward()
    _close(a.grad, ar.grad, 3e-5, "d gwc / d fl")
    _close(b.grad, br.grad, 3e-5, "d gwc / d fr")
    # lookup: gradient to the pyramid levels, none to disp
    f1, f2 = U((1, 32, 2, 24), 14).to(DEV).requires_grad_(), U((1, 32, 2, 24), 15).to(DEV).requires_grad_()
    disp = U((1, 1, 2, 24), 16, -3.0, 27.0).to(DEV)
    out = T.geo_corr_lookup([], T.corr_build_pyramid(f1, f2, 2), disp, 4)
    wl = U(tuple(out.shape), 17).to(DEV)
    (out * wl).sum().backward()
    g1, g2 = f1.detach().double().cpu().requires_grad_(), f2.detach().double().cpu().requires_grad_()
    pyr = O.corr_pyramid(O.all_pairs_corr(g1, g2), 2)
    (O.geo_corr_lookup(None, pyr, disp.double().cpu(), 4) * wl.double().cpu()).sum().backward()
    _close(f1.grad, g1.grad, 5e-5, "d lookup / d fmap1")
    _close(f2.grad, g2.grad, 5e-5, "d lookup / d fmap2")
    # weights passed as tensors receive gradients (functional_call)
    w = [U((8, 16, 3, 3), 20).mul(0.1).to(DEV).requires_grad_(), U((1, 8, 3, 3), 21).mul(0.1).to(DEV).requires_grad_()]
    bs = [U((8,), 22).to(DEV).requires_grad_(), U((1,), 23).to(DEV).requires_grad_()]
    x = U((1, 16, 6, 10), 24).to(DEV)
    T.disp_head(x, w, bs).sum().backward()
    wr = [t.detach().double().cpu().requires_grad_() for t in w]
    br_ = [t.detach().double().cpu().requires_grad_() for t in bs]
    F = torch.nn.functional
    F.conv2d(F.relu(F.conv2d(x.double().cpu(), wr[0], br_[0], padding=1)), wr[1], br_[1], padding=1).sum().backward()
    _close(w[0].grad, wr[0].grad, 1e-4, "d head / d conv1.weight")
    _close(bs[1].grad, br_[1].grad, 1e-4, "d head / d conv2.bias")
