"""`evaluate_expression` for Bmad-style infix arithmetic (mirror of cheetah/converters/utils/infix.py:22-54)."""
from __future__ import annotations

from typing import Any

from ..lattice_text import _Infix


def evaluate_expression(expression: str, context: dict | None = None) -> Any:
    """Value of an infix expression over numbers, the names in `context` and the lattice-file functions; SyntaxError if
    it does not parse or names something unknown."""
    return _Infix(expression, context or {}).parse()
