"""CPU: pin the oracle (oracle/ops.py, oracle/model.py) against vectors captured from the imported
reference (tests/golden/make_golden.py).  Tolerances are fp32 rounding-order tolerances; the lookup
additionally absorbs grid_sample's normalise/denormalise round trip (utils.py:64, SURVEY.md A3:
~3e-6 relative to the coordinate, i.e. ~1e-5 of the local volume slope)."""
import os

import torch

from oracle import ops as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def close(a, b, rtol=1e-5, atol=1e-5):
    a, b = a.double(), b.double()
    assert a.shape == b.shape, (a.shape, b.shape)
    err = (a - b).abs().max().item()
    lim = atol + rtol * b.abs().max().item()
    assert err <= lim, f"max abs err {err:.3e} > {lim:.3e}"


def test_corr_and_pyramids_igev(golden):
    for tag in ("even", "odd"):
        g = golden(f"lookup_igev_{tag}")
        corr = O.corr_pyramid(O.all_pairs_corr(g["f1"], g["f2"]), 2)
        close(corr[0], g["corr0"])
        close(corr[1], g["corr1"])
        geo = O.geo_pyramid(g["gev"], 2)
        # reference layout [B,h,w,G,D]
        close(geo[0], g["geo0"], 0, 0)
        close(geo[1], g["geo1"], 0, 1e-7)


def test_lookup_igev(golden):
    for tag in ("even", "odd"):
        g = golden(f"lookup_igev_{tag}")
        corr = O.corr_pyramid(O.all_pairs_corr(g["f1"], g["f2"]), 2)
        geo = O.geo_pyramid(g["gev"], 2)
        out = O.geo_corr_lookup(geo, corr, g["disp"], 4)
        close(out, g["out"], rtol=2e-5, atol=1e-5)


def test_lookup_raft(golden):
    g = golden("lookup_raft")
    corr = O.corr_pyramid(O.all_pairs_corr(g["f1"], g["f2"]), 4)
    for i in range(4):
        close(corr[i], g[f"corr{i}"])
    out = O.geo_corr_lookup(None, corr, g["disp"], 4)
    close(out, g["out"], rtol=2e-5, atol=1e-5)


def test_corr_sampler_equals_lookup(golden):
    """sampler_kernel.cu:19-60 is the same function as the Python lookup of level 0."""
    g = golden("lookup_raft")
    vol = g["corr0"]
    b, h, w, w2 = vol.shape
    x = torch.arange(w).float().view(1, 1, w).expand(b, h, w)
    coords = torch.stack([x - g["disp"][:, 0], torch.zeros_like(x)], dim=1)
    out = O.corr_sampler_forward(vol, coords, 4)
    close(out, g["out"][:, :9], rtol=2e-5, atol=1e-5)
    # backward is the transpose of forward
    gr = torch.randn(out.shape, generator=torch.Generator().manual_seed(0))
    v = vol.clone().double().requires_grad_(True)
    O.corr_sampler_forward(v, coords, 4).backward(gr.double())
    close(O.corr_sampler_backward(vol, coords, gr, 4), v.grad.float(), 1e-5, 1e-5)


def test_gwc_and_dispreg(golden):
    g = golden("gwc_dispreg")
    close(O.gwc_volume(g["fl"], g["fr"], 48, 8), g["vol"], 1e-6, 1e-6)
    close(O.disparity_regression(torch.softmax(g["cost"], 1), 48), g["init_disp"], 1e-6, 1e-6)


def _ub(tag):
    from anystereo.harness.synthetic import fill_module_deterministic
    from anystereo.models.base import default_args
    from anystereo.nn.update import BasicMultiUpdateBlock
    args = default_args("continuous_IGEVStereo" if tag == "igev" else "continuous_RAFTStereo")
    ub = BasicMultiUpdateBlock(args, hidden_dims=args.hidden_dims, geo_channels=8 if tag == "igev" else 0).eval()
    fill_module_deterministic(ub, base_seed=5)
    return ub


def test_update_block(golden):
    for tag in ("igev", "raft"):
        g = golden(f"update_{tag}")
        ub = _ub(tag)
        net = [g["net0"], g["net1"], g["net2"]]
        inp = [list(g[f"ctx{i}"].split(128, dim=1)) for i in range(3)]
        with torch.no_grad():
            close(O.pool2x(net[0]), g["pool"], 1e-6, 1e-6)
            close(O.interp_to(net[2], net[1].shape[2], net[1].shape[3]), g["interp"], 1e-6, 1e-6)
            close(O.motion_encoder(ub.encoder, g["disp"], g["corr"]), g["motion"], 1e-5, 1e-5)
            close(O.conv_gru(ub.gru16, net[2], *inp[2], O.pool2x(net[1])), g["gru16"], 1e-5, 1e-5)
            close(O.disp_head(ub.disp_head, net[0]), g["head"], 1e-5, 1e-5)
            out, delta = O.update_block(ub, net, inp, g["corr"], g["disp"])
            for i in range(3):
                close(out[i], g[f"out{i}"], 1e-5, 2e-5)
            close(delta, g["delta"], 1e-5, 2e-5)
            lo = O.update_block(ub, net, inp, iter16=True, iter08=True, iter04=False, update=False)
            close(lo[1], g["lo1"], 1e-5, 2e-5)
            close(lo[2], g["lo2"], 1e-5, 2e-5)


FLAG_COMBOS = (("f16", dict(iter16=True, iter08=False, iter04=False, update=False)),
               ("f08_04", dict(iter16=False, iter08=True, iter04=True, update=True)),
               ("f04", dict(iter16=False, iter08=False, iter04=True, update=True)),
               ("fall_noup", dict(iter16=True, iter08=True, iter04=True, update=False)))


def test_update_block_flag_combinations(golden):
    """G5: every (iter16, iter08, iter04, update) pattern of BasicMultiUpdateBlock.forward (update.py:116-136; the slow-fast
    pre-updates of continuous_IGEVstereo.py:288-291 and the remaining ones) against the reference's outputs."""
    for tag in ("igev", "raft"):
        g, gf = golden(f"update_{tag}"), golden(f"update_flags_{tag}")
        ub = _ub(tag)
        inp = [list(g[f"ctx{i}"].split(128, dim=1)) for i in range(3)]
        for name, kw in FLAG_COMBOS:
            net = [g["net0"].clone(), g["net1"].clone(), g["net2"].clone()]
            with torch.no_grad():
                res = O.update_block(ub, net, inp, g["corr"] if kw["iter04"] else None, g["disp"] if kw["iter04"] else None, **kw)
            nets = res[0] if kw["update"] else res
            for i in range(3):
                close(nets[i], gf[f"{name}_net{i}"], 1e-5, 2e-5)
            if kw["update"]:
                close(res[1], gf[f"{name}_delta"], 1e-5, 2e-5)


def test_liif_pieces(golden):
    from anystereo.harness.synthetic import fill_module_deterministic
    from anystereo.nn.liif import liif_out_multi_scale_Training
    g = golden("liif")
    close(O.affinity(g["feat"]), g["aff"], 1e-5, 1e-6)
    for key in ("1p0", "1p5", "2p0", "2p95"):
        rel, qf = O.liif_query(g["feat"], g[f"coord_{key}"])
        close(qf, g[f"qfeat_{key}"], 0, 0)
        close(rel, g[f"rel_{key}"], 0, 2e-6)
    up = liif_out_multi_scale_Training(encoder_dim=208, mlphidden_list=[128, 64, 64], pos_dim=0, unfold="with_v2ISU",
                                       affinity_settings={"win_w": 3, "win_h": 3, "dilation": [1, 2, 4, 8]},
                                       number_input=2, chanels=[176, 32]).eval()
    fill_module_deterministic(up, base_seed=7, gain=2.0)
    with torch.no_grad():
        mask = O.liif_up_mask(up, [g["x4"], g["x2"]], g["coord"])
    close(mask, g["mask"], 1e-5, 1e-5)
    cu = O.convex_upsample(g["dlow"] * 4.0 * 1.5, torch.softmax(g["mask"], 1), g["coord"])
    close(cu, g["convex"], 1e-5, 1e-5)


def _whole(name):
    from anystereo.harness.synthetic import fill_module_deterministic, synthetic_pair
    from anystereo.models.base import default_args
    from oracle.model import OracleIGEV, OracleRAFT
    args = default_args("continuous_IGEVStereo" if name == "igev" else "continuous_RAFTStereo")
    model = (OracleIGEV if name == "igev" else OracleRAFT)(args).eval()
    fill_module_deterministic(model, base_seed=1)
    return model, synthetic_pair


def test_whole_model_vs_reference(golden):
    """EPE between the oracle model and the imported reference on identical weights/inputs.
    north_star bar: EPE delta < 1e-3."""
    for name in ("igev", "raft"):
        g = golden(f"model_{name}")
        H, W = int(g["H"]), int(g["W"])
        model, synthetic_pair = _whole(name)
        img1, img2 = synthetic_pair(1, H, W, shift=6, seed=99)
        with torch.no_grad():
            for s, key in ((1.0, "1p0"), (1.5, "1p5")):
                coord = O.make_coord([round(H * s), round(W * s)]).view(1, -1, 2)
                up = model(img1, img2, iters=3, test_mode=True, hr_coord=coord.clone(), scale=torch.tensor([[s]]))
                epe = (up - g[f"test_{key}"]).abs().mean().item()
                assert up.shape == g[f"test_{key}"].shape
                assert epe < 1e-3, f"{name} scale {s}: EPE vs reference {epe:.3e}"
            coord = O.make_coord([H, W]).view(1, -1, 2)
            res = model(img1, img2, iters=3, test_mode=False, hr_coord=coord.clone(), scale=torch.tensor([[1.0]]))
            preds = res[1] if name == "igev" else res
            if name == "igev":
                close(res[0], g["init_disp"], 1e-4, 1e-4)
            for i, p in enumerate(preds):
                assert (p - g[f"pred_{i}"]).abs().mean().item() < 1e-3


def test_preloop_tensors_vs_reference(golden):
    """The tensors the GRU loop starts from — matching features, geometry encoding volume, initial disparity, hidden states, context
    terms (continuous_IGEVstereo.py:245-276) — of the oracle model against the imported reference at 64x128 AND at 96x160 (a
    non-square size whose 1/32 map is 3 x 5): pins the backbone WIRING the oracle shares with the product (oracle/model.py) at more
    than one shape (round-5 review item 7)."""
    from _preloop import capture_preloop, check_preloop
    g = golden("preloop_igev")
    model, synthetic_pair = _whole("igev")
    for H, W in ((int(a), int(b)) for a, b in g["sizes"]):
        img1, img2 = synthetic_pair(1, H, W, shift=6, seed=99)
        coord = O.make_coord([H, W]).view(1, -1, 2)
        cap = capture_preloop(model, img1, img2, coord, torch.tensor([[1.0]]))
        worst = check_preloop(cap, g, f"{H}x{W}", 2e-4, "oracle")
        print(f"[preloop oracle {H}x{W}] " + ", ".join(f"{k} {v:.1e}" for k, v in worst.items()))


def test_model_options_vs_reference(golden):
    """The forward() branches evaluation.py can reach beyond the main G7 fixture (tests/golden/model_opts.npz, generated from the
    imported reference): `slow_fast_gru = True` (continuous_IGEVstereo.py:288-291, prune_raft_stereo.py:280-283), a `flow_init`
    argument (never read by either reference forward), RAFT's `output_raw=True` tuple (prune_raft_stereo.py:293-296) and IGEV's
    ignored `output_raw` (continuous_IGEVstereo.py:303-305) — oracle model vs reference, EPE bar 1e-3."""
    g = golden("model_opts")
    for name in ("igev", "raft"):
        H, W = (int(v) for v in g[f"{name}_HW"])
        model, synthetic_pair = _whole(name)
        img1, img2 = synthetic_pair(1, H, W, shift=6, seed=99)
        coord = O.make_coord([round(H * 1.5), round(W * 1.5)]).view(1, -1, 2)
        sc = torch.tensor([[1.5]])
        with torch.no_grad():
            base = model(img1, img2, iters=3, test_mode=True, hr_coord=coord.clone(), scale=sc)
            assert (base - g[f"{name}_base"]).abs().mean().item() < 1e-3
            with_fi = model(img1, img2, iters=3, flow_init=g[f"{name}_flow_init"], test_mode=True, hr_coord=coord.clone(), scale=sc)
            assert torch.equal(with_fi, base), "flow_init must not change the result (the reference never reads it)"
            raw = model(img1, img2, iters=3, test_mode=True, hr_coord=coord.clone(), scale=sc, output_raw=True)
            if name == "raft":
                assert isinstance(raw, tuple) and len(raw) == 2
                assert raw[0].shape == g["raft_raw_disp"].shape and (raw[0] - g["raft_raw_disp"]).abs().mean().item() < 1e-3
                assert (raw[1] - g["raft_raw_up"]).abs().mean().item() < 1e-3
            else:
                assert torch.is_tensor(raw) and (raw - g["igev_output_raw"]).abs().mean().item() < 1e-3
            model.args.slow_fast_gru = True
            sf = model(img1, img2, iters=3, test_mode=True, hr_coord=coord.clone(), scale=sc)
            epe = (sf - g[f"{name}_slowfast"]).abs().mean().item()
            assert epe < 1e-3, f"{name} slow_fast_gru: EPE vs reference {epe:.3e}"
            # the branch is really taken: its result differs from the default schedule's
            assert (g[f"{name}_slowfast"] - g[f"{name}_base"]).abs().mean().item() > 1e-3


def test_training_step_vs_reference(golden):
    """G8: loss and parameter gradients of one training forward/backward (train mode, frozen BatchNorm2d, LIIF every
    iteration, sequence_loss_multiscale) of the oracle model vs the imported reference."""
    import numpy as np
    from anystereo.harness.metrics import sequence_loss_multiscale
    from anystereo.harness.synthetic import tiny_train_case
    for name in ("igev", "raft"):
        z = np.load(f"{GOLDEN}/train_{name}.npz")
        model, _ = _whole(name)
        model.train()
        model.freeze_bn()
        h, w, img1, img2, coord, gt, scale = tiny_train_case(name)
        res = model(img1, img2, iters=3, hr_coord=coord.clone(), scale=scale)
        preds = res[1] if name == "igev" else res
        loss, _ = sequence_loss_multiscale(preds, gt, ((gt < 512) & (gt > 0)).float(), max_disp=model.args.max_disp)
        loss.backward()
        assert abs(loss.item() - float(z["loss"])) < 1e-3 * abs(float(z["loss"]))
        assert (preds[-1].detach() - torch.from_numpy(z["last_pred"])).abs().mean().item() < 1e-3
        named = dict(model.named_parameters())
        names = [str(n) for n in z["names"]]
        assert sorted(n for n, p in named.items() if p.grad is not None) == names
        norms = np.array([float(named[n].grad.double().norm()) for n in names])
        rel = np.abs(norms - z["norms"]) / (z["norms"] + 1e-6 * z["norms"].max())
        assert rel.max() < 2e-2, f"{name}: grad-norm mismatch {rel.max():.3e} at {names[int(rel.argmax())]}"
        for i, n in enumerate(str(x) for x in z["full_names"]):
            close(named[n].grad, torch.from_numpy(z[f"g{i}"]), rtol=2e-3, atol=1e-6)
