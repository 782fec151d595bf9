"""gs2m_optim.Adam (one fused HIP launch, include/gs2m_optim.h) against torch.optim.Adam as the reference builds it
(scene/gaussian_model.py:230-245: nine groups, per-group lr, eps=1e-15) -- same parameters, moments and state layout."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu

GROUPS = (("xyz", (3,), 1.6e-4), ("f_dc", (1, 3), 2.5e-3), ("f_rest", (15, 3), 2.5e-3 / 20), ("opacity", (1,), 0.05),
          ("scaling", (3,), 5e-3), ("rotation", (4,), 1e-3), ("albedo", (3,), 0.05), ("roughness", (1,), 0.05), ("metallic", (1,), 0.05))


def _model(P, seed, dev="cuda"):
    gen = torch.Generator().manual_seed(seed)
    return [torch.nn.Parameter(torch.randn((P,) + shp, generator=gen).to(dev)) for _, shp, _ in GROUPS]


def _groups(params):
    return [{"params": [p], "lr": lr, "name": name} for p, (name, _, lr) in zip(params, GROUPS)]


def _set_grads(params, seed, scale=1.0, skip=()):
    gen = torch.Generator().manual_seed(seed)
    for k, p in enumerate(params):
        g = (torch.randn(p.shape, generator=gen) * scale).to(p.device)
        p.grad = None if k in skip else g


@pytest.mark.parametrize("P", [1, 341, 4096, 20011])
def test_adam_matches_torch_bitwise(P):
    assert torch.cuda.is_available()
    import gs2m_optim
    pa, pb = _model(P, 0), _model(P, 0)
    ref = torch.optim.Adam(_groups(pa), lr=0.0, eps=1e-15)
    opt = gs2m_optim.Adam(_groups(pb), lr=0.0, eps=1e-15)
    for it in range(1, 8):
        for o in (ref, opt):
            o.param_groups[0]["lr"] = 1.6e-4 * 0.97 ** it  # update_learning_rate, GM:251-258
        skip = (3,) if it == 4 else ()                      # a group without a gradient is left alone, step not advanced
        _set_grads(pa, 100 + it, scale=10.0 ** (it - 4), skip=skip)
        _set_grads(pb, 100 + it, scale=10.0 ** (it - 4), skip=skip)
        ref.step()
        opt.step()
        for k, (a, b) in enumerate(zip(pa, pb)):
            name = GROUPS[k][0]
            sa, sb = ref.state[a], opt.state[b]
            assert float(sa["step"]) == float(sb["step"]), name
            assert torch.equal(sa["exp_avg"], sb["exp_avg"]), f"{name} exp_avg, step {it}"
            assert torch.equal(sa["exp_avg_sq"], sb["exp_avg_sq"]), f"{name} exp_avg_sq, step {it}"
            assert torch.equal(a, b), f"{name} param, step {it}: max diff {(a - b).abs().max().item():.3e}"


def test_state_dict_is_interchangeable_with_torch_adam():
    assert torch.cuda.is_available()
    import gs2m_optim
    pa, pb, pc = _model(257, 1), _model(257, 1), _model(257, 1)
    ref = torch.optim.Adam(_groups(pa), lr=0.0, eps=1e-15)
    opt = gs2m_optim.Adam(_groups(pb), lr=0.0, eps=1e-15)
    for it in range(3):
        _set_grads(pa, it)
        _set_grads(pb, it)
        ref.step()
        opt.step()
    sd_t, sd_f = ref.state_dict(), opt.state_dict()
    assert sd_t["param_groups"] == sd_f["param_groups"]
    assert sd_t["state"].keys() == sd_f["state"].keys()
    assert all(sd_t["state"][k].keys() == sd_f["state"][k].keys() for k in sd_t["state"])
    # resume a torch.optim.Adam checkpoint in the fused optimizer and the other way round (train.py:55-63)
    for p, q in zip(pc, pa):
        p.data.copy_(q.data)
    opt2 = gs2m_optim.Adam(_groups(pc), lr=0.0, eps=1e-15)
    opt2.load_state_dict(copy.deepcopy(sd_t))
    _set_grads(pa, 9)
    _set_grads(pc, 9)
    ref.step()
    opt2.step()
    for a, c in zip(pa, pc):
        assert torch.equal(a, c)
    ref2 = torch.optim.Adam(_groups(pb), lr=0.0, eps=1e-15)
    ref2.load_state_dict(copy.deepcopy(sd_f))
    _set_grads(pb, 9)
    ref2.step()
    for a, b in zip(pa, pb):
        assert torch.equal(a, b)


def test_densification_style_state_surgery():
    """cat_tensors_to_optimizer (GM:426-455) and _prune_optimizer (GM:388-407) edit exp_avg / exp_avg_sq and swap the
    parameter object; the fused step must pick the new tensors up."""
    assert torch.cuda.is_available()
    import gs2m_optim

    def run(make):
        params = _model(300, 2)
        opt = make(_groups(params))
        _set_grads(params, 0)
        opt.step()
        keep = torch.arange(300, device="cuda") % 3 != 0
        for group in opt.param_groups:
            old = group["params"][0]
            st = opt.state.pop(old)
            ext = torch.zeros((50,) + old.shape[1:], device="cuda")
            st["exp_avg"] = torch.cat((st["exp_avg"][keep], torch.zeros_like(ext)), dim=0)
            st["exp_avg_sq"] = torch.cat((st["exp_avg_sq"][keep], torch.zeros_like(ext)), dim=0)
            new = torch.nn.Parameter(torch.cat((old.data[keep], ext + 0.25), dim=0).requires_grad_(True))
            group["params"][0] = new
            opt.state[new] = st
        params = [g["params"][0] for g in opt.param_groups]
        _set_grads(params, 1)
        opt.step()
        return params

    import functools
    a = run(functools.partial(torch.optim.Adam, lr=0.0, eps=1e-15))
    b = run(functools.partial(gs2m_optim.Adam, lr=0.0, eps=1e-15))
    for x, y in zip(a, b):
        assert x.shape == y.shape and torch.equal(x, y)


def test_more_tensors_than_one_launch_and_unaligned_views():
    assert torch.cuda.is_available()
    import gs2m_optim
    gen = torch.Generator().manual_seed(3)
    base = [torch.randn(5000 + 17 * k, generator=gen).cuda() for k in range(37)]
    pa = [torch.nn.Parameter(t.clone()) for t in base]
    # odd offsets: 4-byte aligned storage views, scalar path
    pb = [torch.nn.Parameter(torch.cat((torch.zeros(1, device="cuda"), t))[1:]) for t in base]
    assert any(p.data_ptr() % 16 for p in pb)
    ref = torch.optim.Adam(pa, lr=1e-2)
    opt = gs2m_optim.Adam(pb, lr=1e-2)
    for it in range(3):
        _set_grads(pa, it)
        _set_grads(pb, it)
        ref.step()
        opt.step()
    for a, b in zip(pa, pb):
        assert torch.equal(a, b)
