"""Mirror of cheetah/utils/vector.py."""
from __future__ import annotations


def squash_index_for_unavailable_dims(index: tuple, shape: tuple) -> tuple:
    """Index a tensor of `shape` with the trailing part of a longer vector index; size-1 dims take index 0
    (utils/vector.py)."""
    if len(shape) == 0:
        return ()
    tail = index[-len(shape):]
    return tuple(0 if size == 1 else i for i, size in zip(tail, shape))
