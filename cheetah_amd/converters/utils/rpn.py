"""`evaluate_expression` for Elegant's reverse-Polish arithmetic (mirror of cheetah/converters/utils/rpn.py:6-...)."""
from __future__ import annotations

from typing import Any

from ..lattice_text import _evaluate_rpn


def evaluate_expression(expression: str, context: dict | None = None) -> Any:
    """Value of an RPN expression (optionally in double quotes) over numbers and the names in `context`; SyntaxError if
    the stack does not reduce to one value or a token is neither a number, an operator nor a known name."""
    return _evaluate_rpn(expression, context or {})
