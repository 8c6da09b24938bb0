"""Mirror of cheetah/utils/names.py."""
from __future__ import annotations

from ..accelerator.element import merge_element_names  # noqa: F401


class UniqueNameGenerator:
    """Callable producing `prefix_0`, `prefix_1`, ... (utils/names.py:4-14)."""

    def __init__(self, prefix: str):
        self._prefix, self._counter = prefix, 0

    def __call__(self) -> str:
        name = f"{self._prefix}_{self._counter}"
        self._counter += 1
        return name
