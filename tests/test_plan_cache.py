"""CPU tests of the launch-plan cache (sipmask_amd/plan_cache.py): a plan snapshots the weights, so it must be
rebuilt after ANY in-place weight update -- optimizer.step, HipSGD's raw-pointer update (version bumped by hand),
Module.load_state_dict and mmcv-style loading through _load_from_state_dict -- and an LRU keeps a few geometries.
The engine build itself needs a GPU; here the builder is a counter."""
import torch
import torch.nn as nn

from sipmask_amd.plan_cache import PlanCache, module_tensors


class _Net(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv = nn.Conv2d(3, 4, 1)
        self.bn = nn.BatchNorm2d(4)


def _cached(cache, net, key, builds):
    def build():
        builds.append(key)
        return {"key": key, "w0": float(net.conv.weight.flatten()[0])}
    return cache.get(key, module_tensors(net), build)


def test_same_weights_same_plan_and_lru():
    net, cache, builds = _Net(), PlanCache(capacity=2), []
    a = _cached(cache, net, "A", builds)
    assert _cached(cache, net, "A", builds) is a and builds == ["A"]
    _cached(cache, net, "B", builds)
    _cached(cache, net, "A", builds)              # A is now the most recent
    _cached(cache, net, "C", builds)              # evicts B
    assert builds == ["A", "B", "C"] and len(cache) == 2
    _cached(cache, net, "A", builds)
    assert builds == ["A", "B", "C"]
    _cached(cache, net, "B", builds)
    assert builds == ["A", "B", "C", "B"]


def test_optimizer_step_invalidates():
    net, cache, builds = _Net(), PlanCache(), []
    p0 = _cached(cache, net, "A", builds)
    opt = torch.optim.SGD(net.parameters(), lr=0.1)
    net.conv(torch.randn(1, 3, 2, 2)).sum().backward()
    opt.step()
    p1 = _cached(cache, net, "A", builds)
    assert builds == ["A", "A"] and p1 is not p0 and p1["w0"] != p0["w0"]
    # a raw-pointer update (HipSGD writes p.data through the C ABI) followed by the explicit version bump
    with torch.no_grad():
        net.conv.weight.data.mul_(2.0)            # .data: does NOT bump the parameter's version counter
    torch.autograd.graph.increment_version(net.conv.weight)
    p2 = _cached(cache, net, "A", builds)
    assert len(builds) == 3 and p2["w0"] == 2 * p1["w0"]


def test_state_dict_loading_invalidates():
    net, cache, builds = _Net(), PlanCache(), []
    _cached(cache, net, "A", builds)
    sd = {k: v.clone() + 1 for k, v in net.state_dict().items()}
    net.load_state_dict(sd)                        # Module.load_state_dict: param.copy_ in place
    _cached(cache, net, "A", builds)
    assert len(builds) == 2
    # the mmcv load_checkpoint route: module._load_from_state_dict per module, never Module.load_state_dict
    sd2 = {k: v.clone() + 1 for k, v in net.state_dict().items()}
    for name, m in net.named_modules():
        if name:
            m._load_from_state_dict(sd2, name + ".", {}, True, [], [], [])
    _cached(cache, net, "A", builds)
    assert len(builds) == 3
    # running statistics are buffers: a BN update in train mode must invalidate too (they are folded into the convs)
    net.train()
    net.bn(torch.randn(2, 4, 3, 3))
    _cached(cache, net, "A", builds)
    assert len(builds) == 4


def test_detector_and_head_use_the_cache():
    from sipmask_amd.synthetic import build_synthetic_detector
    det = build_synthetic_detector(50, seed=0)
    assert isinstance(det._engines, PlanCache) and isinstance(det.bbox_head._engines, PlanCache)


def test_writes_through_dot_data_need_an_explicit_invalidate_and_load_state_dict_drops_plans():
    """`p.data.copy_` keeps data_ptr AND _version, so the fingerprint cannot see it -- documented; the two
    answers are the load_state_dict post-hook (attach_invalidation) and the public invalidate()."""
    net, builds = _Net(), []
    cache = PlanCache().attach_invalidation(net)

    def get():
        return cache.get("k", module_tensors(net), lambda: builds.append(1) or float(net.conv.weight.flatten()[0]))
    w0 = get()
    net.conv.weight.data.mul_(2.0)                       # invisible to the fingerprint ...
    assert get() == w0 and len(builds) == 1              # ... (the documented caveat)
    cache.invalidate()                                   # the caller's obligation after such a write
    assert get() == 2 * w0 and len(builds) == 2
    # load_state_dict drops the plans even when the loaded tensors are written through .data-like paths
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    sd["conv.weight"] = sd["conv.weight"] * 0 + 5.0
    net.load_state_dict(sd)
    assert get() == 5.0 and len(builds) == 3
    # the detector exposes the same thing
    from sipmask_amd.synthetic import build_synthetic_detector
    det = build_synthetic_detector(50, seed=0)
    det._engines._plans["x"] = object()
    det.invalidate_plans()
    assert len(det._engines) == 0
