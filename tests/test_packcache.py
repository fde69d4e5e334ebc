"""The weight-pack / affine cache must never serve an entry of a dead owner (round-2 driver failure: `id()`, device
address, `_version` and shape of a freed nn.Parameter all reappeared on a new one -> the previous conv's pack was used).
CPU tests of the cache object itself; the GPU regression over the real ops is in test_gpu_parity.py."""
import gc

import torch
import torch.nn as nn

from orientedreppoints_amd import _packcache


def test_recycled_owner_never_hits():
    cache = _packcache.OwnerCache("t")
    recycled = 0
    seen_ids = set()
    for i in range(300):
        p = nn.Parameter(torch.full((18, 256, 1, 1), float(i)))
        state = _packcache.tensor_state(p.detach())
        recycled += id(p) in seen_ids
        seen_ids.add(id(p))
        assert cache.get(p, state) is None, "entry of a dead parameter served to a new one (iteration %d)" % i
        cache.put(p, state, i)
        assert cache.get(p, state) == i
        del p
    assert recycled > 0, "the test did not exercise id() recycling"
    gc.collect()
    assert len(cache) == 0                                   # every entry left with its owner


def test_entry_follows_versions_and_identity():
    cache = _packcache.OwnerCache("t")
    a, b = nn.Parameter(torch.zeros(4)), nn.Parameter(torch.zeros(4))
    sa = _packcache.tensor_state(a.detach())
    cache.put(a, sa, "A")
    assert cache.get(a, sa) == "A"
    assert cache.get(b, _packcache.tensor_state(b.detach())) is None
    with torch.no_grad():
        a.add_(1.0)                                          # in-place update bumps the version -> miss
    assert cache.get(a, _packcache.tensor_state(a.detach())) is None
    # a forged entry under b's id whose weakref points at a (what a recycled id looks like) must not hit for b
    cache._entries[id(b)] = cache._entries[id(a)]
    assert cache.get(b, sa) is None
    del cache._entries[id(b)]
    del a
    gc.collect()
    assert len(cache) == 0


def test_dead_owner_callback_does_not_remove_a_newer_entry():
    cache = _packcache.OwnerCache("t")
    a = nn.Parameter(torch.zeros(4))
    cache.put(a, 0, "A")
    ref_a = cache._entries[id(a)][0]
    b = nn.Parameter(torch.ones(4))
    cache._entries[id(a)] = (__import__("weakref").ref(b), 1, "B")      # id(a) recycled by b before a's callback ran
    key = id(a)
    del a
    gc.collect()
    assert cache._entries[key][2] == "B"


def test_invalidate_all_clears_every_registered_cache():
    import importlib
    deform_conv = importlib.import_module('orientedreppoints_amd.mmdet_ops.deform_conv')   # the package exports a function of that name
    fused_norm = importlib.import_module('orientedreppoints_amd.mmdet_ops.fused_norm')
    owners = [nn.Parameter(torch.zeros(2)) for _ in range(5)]
    caches = [deform_conv._packed_cache, deform_conv._packed_cache_h, deform_conv._packed_heads,
              fused_norm._packed_1x1, fused_norm._affine_cache]
    for c, o in zip(caches, owners):
        c.put(o, 0, "x")
    assert all(len(c) >= 1 for c in caches)
    deform_conv.invalidate_packed_weights()
    assert all(len(c) == 0 for c in caches)
