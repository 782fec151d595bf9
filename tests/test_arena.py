"""gs2m_arena (gradient arenas with an explicit registry) on the CPU: what the data-parallel reducer relies on."""
import torch

import gs2m_arena

gs2m_arena.enable()  # what gs2m_dp.GradReducer does: without a reducer nothing is registered


def test_lookup_finds_exact_entries_only():
    a = gs2m_arena.GradArena("cpu", [("m", (5, 3)), ("o", (5, 1)), ("sh", (5, 16, 3))], key="t1")
    m, o, sh = a["m"], a["o"], a["sh"]
    assert m.shape == (5, 3) and sh.shape == (5, 16, 3) and m.is_contiguous()
    assert m.untyped_storage().data_ptr() == sh.untyped_storage().data_ptr() == a.flat.untyped_storage().data_ptr()
    for name, t in (("m", m), ("o", o), ("sh", sh)):
        arena, off, n = gs2m_arena.lookup(t)
        assert arena is a and n == t.numel() and off % 4 == 0
    assert gs2m_arena.lookup(sh[:, :4]) is None, "a slice of an entry is not an entry"
    assert gs2m_arena.lookup(m[1:]) is None
    assert gs2m_arena.lookup(torch.zeros(5, 3)) is None and gs2m_arena.lookup(None) is None
    assert gs2m_arena.lookup(m.double()) is None
    # a detached alias (what autograd keeps as a leaf's .grad) is still the entry
    assert gs2m_arena.lookup(m.detach())[0] is a


def test_views_are_not_kept_by_the_registry():
    """autograd takes a gradient over without a copy only while nothing else references the tensor"""
    a = gs2m_arena.GradArena("cpu", [("x", (8, 4))], key="t2")
    v = a["x"]
    assert v._use_count() == 1, "the arena must not hold its views"  # test-only use of the private counter
    assert a["x"] is not v and a["x"].data_ptr() == v.data_ptr()


def test_contiguous_range_refuses_foreign_entries():
    a = gs2m_arena.GradArena("cpu", [("a", (6, 3)), ("mid", (6, 2)), ("b", (6, 4)), ("c", (6, 1))], key="t3")
    lay = {name: (off, n) for name, off, n, _ in a.layout}
    assert gs2m_arena.contiguous_range(a, [lay["a"], lay["mid"]]) == (lay["a"][0], lay["mid"][0] + lay["mid"][1])
    assert gs2m_arena.contiguous_range(a, [lay["a"], lay["b"]]) is None, "`mid` lies between and was not passed"
    assert gs2m_arena.contiguous_range(a, [lay["b"], lay["c"]]) == (lay["b"][0], lay["c"][0] + lay["c"][1])
    assert gs2m_arena.contiguous_range(a, [lay["a"], lay["mid"], lay["b"], lay["c"]]) == (0, lay["c"][0] + lay["c"][1])


def test_registry_keeps_the_latest_arenas_of_a_producer_and_releases():
    first = gs2m_arena.GradArena("cpu", [("x", (4,))], key="t4")
    x0 = first["x"]
    second = gs2m_arena.GradArena("cpu", [("x", (4,))], key="t4")
    assert gs2m_arena.lookup(x0) is not None and gs2m_arena.lookup(second["x"]) is not None
    third = gs2m_arena.GradArena("cpu", [("x", (4,))], key="t4")
    assert gs2m_arena.lookup(x0) is None, "only the two latest arenas of a producer stay registered"
    assert gs2m_arena.lookup(third["x"])[0] is third
    gs2m_arena.release()
    assert gs2m_arena.lookup(third["x"]) is None


def test_nothing_is_registered_without_a_reducer():
    gs2m_arena.enable(False)
    try:
        a = gs2m_arena.GradArena("cpu", [("x", (4,))], key="t5")
        assert gs2m_arena.lookup(a["x"]) is None, "single-GPU training must not keep arenas alive"
    finally:
        gs2m_arena.enable()
