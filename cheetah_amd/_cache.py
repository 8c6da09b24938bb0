"""Identity keys for host-side caches.

A cache keyed on `id(tensor)` can be fooled: CPython recycles the id of a freed tensor, so a NEW tensor assigned to the
same attribute may compare equal to the key of one that no longer exists (observed: `Cavity.is_active` stale in 49 of 200
trials). `TensorKey` keeps the keyed tensors referenced and compares with `is`, plus `_version` (in-place edits) and
`requires_grad` (a graph-free cached product must not be served to a caller that needs gradients). This is the rule of
the reference's `cache_transfer_map` (cheetah/utils/cache.py:29-52), made safe against id reuse.
"""

from __future__ import annotations


class TensorKey:
    __slots__ = ("tensors", "versions", "flags")

    def __init__(self, tensors):
        self.tensors = tuple(tensors)
        self.versions = tuple([t._version for t in self.tensors])
        self.flags = tuple([t.requires_grad for t in self.tensors])

    def matches(self, tensors) -> bool:
        mine = self.tensors
        if len(mine) != len(tensors):
            return False
        for a, b, v, f in zip(mine, tensors, self.versions, self.flags):
            if a is not b or b._version != v or b.requires_grad != f:
                return False
        return True
