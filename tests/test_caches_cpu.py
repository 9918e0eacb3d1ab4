"""Host-side caches that must not go stale: the Flux modulation-table dependency list and the pipeline's cached
timestep / guidance tensors. CPU only: no kernel is called."""
import pytest
import torch


def _tiny_flux():
    from omgsr_amd.diffusers_api.transformer_flux import FluxTransformer2DModel
    return FluxTransformer2DModel(num_layers=1, num_single_layers=1, attention_head_dim=128, num_attention_heads=1, joint_attention_dim=64,
                                  pooled_projection_dim=32, in_channels=16)


def test_flux_modulation_deps_follow_replaced_parameters():
    m = _tiny_flux()
    d1 = m._mod_deps()
    assert m._mod_deps() is d1                                   # steady state: the cached list, no module walk
    want = [p for mm in (m.time_text_embed, m.norm_out) for p in mm.parameters()]
    for b in list(m.transformer_blocks) + list(m.single_transformer_blocks):
        for nm in ("norm1", "norm1_context", "norm"):
            if hasattr(b, nm):
                want += list(getattr(b, nm).parameters())
    assert len(want) == len(d1) and all(a is b for a, b in zip(want, d1))
    m.load_state_dict({k: v.clone() for k, v in m.state_dict().items()})          # in place: same objects, versions bump
    assert m._mod_deps() is d1
    m.load_state_dict({k: v.clone() for k, v in m.state_dict().items()}, assign=True)      # objects REPLACED
    d2 = m._mod_deps()
    assert d2 is not d1 and len(d2) == len(d1) and all(a is not b for a, b in zip(d1, d2))
    assert all(p is q for p, q in zip(d2, [p for mm in (m.time_text_embed, m.norm_out) for p in mm.parameters()]))


def test_range_fallback_state_machine_without_gpu(monkeypatch):
    """RangeFallback bookkeeping (sticky after the first overflow, reset restores the saved policy) with the device calls stubbed."""
    from omgsr_amd import ops, precision
    from omgsr_amd.nn import Conv2d, Linear
    net = torch.nn.Sequential(Conv2d(32, 32, 3, padding=1), Linear(32, 32))
    net[0].op_split, net[0].w_split, net[1].op_split = 3, 2, 2
    state = dict(act=torch.float16, precise=True, ovf=[True, False])
    monkeypatch.setattr(ops, "set_compute_dtype", lambda dt, operand_dtype=None: state.update(act=operand_dtype or torch.float16, precise=dt == torch.float32))
    monkeypatch.setattr(ops, "precise", lambda: state["precise"])
    monkeypatch.setattr(ops, "act_dtype", lambda: state["act"])
    monkeypatch.setattr(ops, "overflow_seen", lambda reset=True: state["ovf"].pop(0))
    monkeypatch.setattr(ops, "mx_saturation_seen", lambda reset=True: False)
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a, **k: None)
    rf = precision.RangeFallback(net)
    calls = []
    import warnings
    with warnings.catch_warnings(record=True) as wl:
        warnings.simplefilter("always")
        rf.run(lambda: calls.append(state["act"]), "test")
    assert calls == [torch.float16, torch.bfloat16] and rf.count == 1 and rf.sticky and len(wl) == 1
    assert (net[0].op_split, net[0].w_split, net[1].op_split, net[1].w_split) == (2, 2, 2, 2)
    state.update(act=torch.bfloat16, precise=False)              # someone else switched the process-wide tier
    rf.run(lambda: calls.append((state["act"], state["precise"])), "test")
    assert calls[-1] == (torch.bfloat16, True) and rf.count == 1 and len(calls) == 3      # one pass, re-asserted, guard word not read
    rf.reset()
    assert not rf.sticky and state["act"] == torch.float16
    assert (net[0].op_split, net[0].w_split, net[1].op_split, net[1].w_split) == (3, 2, 2, 1)


def test_two_pipelines_one_sticky_reassert_their_own_mode(monkeypatch):
    """the operand dtype is process-wide. Pipeline A goes sticky (fp32 stream, bf16 operands); a second accurate pipeline B
    and a bf16 pipeline C in the same process re-assert THEIR tier at the top of every run instead of inheriting A's."""
    from omgsr_amd import ops, precision
    from omgsr_amd.nn import Conv2d
    state = dict(act=torch.float16, precise=True)
    monkeypatch.setattr(ops, "set_compute_dtype", lambda dt, operand_dtype=None: state.update(
        act=torch.bfloat16 if dt == torch.bfloat16 else (operand_dtype or torch.float16), precise=dt == torch.float32))
    monkeypatch.setattr(ops, "precise", lambda: state["precise"])
    monkeypatch.setattr(ops, "act_dtype", lambda: state["act"])
    monkeypatch.setattr(ops, "overflow_seen", lambda reset=True: False)
    sat = [True, True, False, False]
    monkeypatch.setattr(ops, "mx_saturation_seen", lambda reset=True: sat.pop(0) if sat else False)
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a, **k: None)
    a = precision.RangeFallback(torch.nn.Sequential(Conv2d(32, 32, 3)), weight_dtype=torch.float32)
    b = precision.RangeFallback(torch.nn.Sequential(Conv2d(32, 32, 3)), weight_dtype=torch.float32)
    c = precision.RangeFallback(torch.nn.Sequential(Conv2d(32, 32, 3)), weight_dtype=torch.bfloat16)
    a.enter()
    assert state == dict(act=torch.bfloat16, precise=True)
    seen = []
    b.run(lambda: seen.append((state["act"], state["precise"])), "B")
    c.run(lambda: seen.append((state["act"], state["precise"])), "C")
    a.run(lambda: seen.append((state["act"], state["precise"])), "A")
    b.run(lambda: seen.append((state["act"], state["precise"])), "B")
    assert seen == [(torch.float16, True), (torch.bfloat16, False), (torch.bfloat16, True), (torch.float16, True)]
    assert a.sticky and not b.sticky and not c.sticky
    # the MX saturation diagnostic is counted per pipeline in the accurate tier (B saw it once, A once), never raised as an overflow
    assert b.mx_saturation_count == 1 and a.mx_saturation_count == 1 and c.mx_saturation_count == 0 and b.count == 0


def test_graph_cache_drops_entries_captured_under_an_older_cache_epoch():
    """a captured graph reads the one-slot caches' values by address; any rebuild since the capture makes it stale."""
    from omgsr_amd import nn as onn
    from omgsr_amd.nn import InputCache
    e0 = onn.cache_epoch()
    c = InputCache()
    t1, t2 = torch.zeros(1), torch.zeros(1)
    c.get((t1,), (), lambda: 1)
    assert onn.cache_epoch() == e0 + 1
    c.get((t1,), (), lambda: 1)
    assert onn.cache_epoch() == e0 + 1                         # a hit rebuilds nothing
    c.get((t2,), (), lambda: 2)
    c.get((t1,), (), lambda: 1)                                # A -> B -> A: two rebuilds
    assert onn.cache_epoch() == e0 + 3
    from omgsr_amd import precision
    p0 = precision.policy_epoch()
    precision.set_operand_split(torch.nn.Sequential(onn.Conv2d(32, 32, 3)), [r"."])
    assert precision.policy_epoch() > p0


def test_mx_saturation_demotes_the_fixed_scale_layers_and_reset_restores_them(monkeypatch):
    """when the MX saturation bit fires, the layers in the fixed-scale fp8 form (op_split 3) fall back to fp16 correction
    segments (op_split 2, weight split kept), the call is recomputed once, the pipeline stays in that form; fp6 layers (op_split 4: a scale per
    block) are untouched; reset() returns to the fp8 form - also when a range fallback was entered after the demotion."""
    from omgsr_amd import nn as onn, ops, precision
    from omgsr_amd.nn import Conv2d, Linear
    state = dict(act=torch.float16, precise=True)
    monkeypatch.setattr(ops, "set_compute_dtype", lambda dt, operand_dtype=None: state.update(
        act=torch.bfloat16 if dt == torch.bfloat16 else (operand_dtype or torch.float16), precise=dt == torch.float32))
    monkeypatch.setattr(ops, "precise", lambda: state["precise"])
    monkeypatch.setattr(ops, "act_dtype", lambda: state["act"])
    monkeypatch.setattr(ops, "overflow_seen", lambda reset=True: False)
    sat = [True, False, False]
    monkeypatch.setattr(ops, "mx_saturation_seen", lambda reset=True: sat.pop(0) if sat else False)
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a, **k: None)
    net = torch.nn.Sequential(Linear(64, 64), Conv2d(64, 128, 3, padding=1), Conv2d(64, 128, 1), Linear(64, 64))
    for m, sp in zip(net, (3, 4, 3, 1)):
        m.op_split, m.w_split = sp, (2 if sp > 1 else 1)
    rf = precision.RangeFallback(net, weight_dtype=torch.float32)
    dropped = []
    rf.on_mode_change = lambda: dropped.append(1)
    p0, calls = precision.policy_epoch(), []
    with pytest.warns(UserWarning, match="recomputed with fp16 correction segments"):
        rf.run(lambda: calls.append([m.op_split for m in net]), "T")
    assert calls == [[3, 4, 3, 1], [2, 4, 2, 1]]                     # one recompute, in the demoted form
    assert rf.mx_demoted and rf.mx_saturation_count == 1 and not rf.sticky and rf.count == 0
    assert precision.policy_epoch() > p0 and dropped                  # packed weights / captured graphs of the fp8 form are not reused
    assert [m.w_split for m in net] == [2, 2, 2, 1]
    rf.run(lambda: calls.append([m.op_split for m in net]), "T")     # sticky: no second recompute
    assert len(calls) == 3 and calls[-1] == [2, 4, 2, 1]
    rf.enter()                                                        # a later fp16 overflow: every layer split, bf16 operands
    assert [m.op_split for m in net] == [2, 2, 2, 2] and rf.sticky
    rf.reset()
    assert [m.op_split for m in net] == [3, 4, 3, 1] and not rf.sticky and not rf.mx_demoted
    # OMGSR_MX_SAT_FALLBACK=0: round 5's behaviour (diagnostic only)
    monkeypatch.setenv("OMGSR_MX_SAT_FALLBACK", "0")
    sat[:] = [True]
    rf2 = precision.RangeFallback(net, weight_dtype=torch.float32)
    n = []
    with pytest.warns(UserWarning, match="OMGSR_MX_LINEAR=0"):
        rf2.run(lambda: n.append(1), "T")
    assert len(n) == 1 and not rf2.mx_demoted and [m.op_split for m in net] == [3, 4, 3, 1]
