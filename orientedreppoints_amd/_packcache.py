"""Identity-safe cache of device-side re-layouts of module parameters (DeformConv / 3x3 / 1x1 weight packs, folded
eval-mode BatchNorm affines).

An entry belongs to ONE live Python object (an nn.Parameter or an nn.Module):

* the entry holds a weak reference to its owner and a hit requires `entry.owner() is obj` -- CPython recycles `id()`s
  and the caching allocator recycles device addresses, so `(id, data_ptr, _version, shape)` of a freed parameter can
  all reappear on a new one (build a model, drop it, build another of the same shape);
* the weak reference's callback removes exactly that entry when the owner dies; nothing is ever cleared wholesale and
  there is no size cap to thrash against (the cache is as large as the set of live packed parameters);
* `state` (storage address, version counters, dtype, device) catches in-place updates of a live owner.  Writes through
  `.data` bump no version counter: training-mode callers pass `cache=False` (see deform_conv.py), everything else
  calls `invalidate()`.
"""
import weakref


class OwnerCache:
    def __init__(self, name):
        self.name = name
        self._entries = {}          # id(owner) -> (weakref(owner), state, payload)
        self.hits = self.misses = 0

    def get(self, owner, state):
        e = self._entries.get(id(owner))
        if e is not None and e[0]() is owner and e[1] == state:
            self.hits += 1
            return e[2]
        self.misses += 1
        return None

    def put(self, owner, state, payload):
        key = id(owner)
        entries = self._entries

        def _gone(ref, key=key, entries=entries):
            e = entries.get(key)
            if e is not None and e[0] is ref:        # a recycled id may already carry a newer owner's entry
                del entries[key]
        entries[key] = (weakref.ref(owner, _gone), state, payload)
        return payload

    def invalidate(self):
        self._entries.clear()

    def __len__(self):
        return len(self._entries)


_all = []


def new_cache(name):
    c = OwnerCache(name)
    _all.append(c)
    return c


def invalidate_all():
    """Drop every cached pack / affine of every cache (after loading a checkpoint through `.data`, EMA swaps, ...)."""
    for c in _all:
        c.invalidate()


def stats():
    return {c.name: {"entries": len(c), "hits": c.hits, "misses": c.misses} for c in _all}


def tensor_state(t):
    return (t.data_ptr(), t._version, tuple(t.shape), t.dtype, t.device.index)
