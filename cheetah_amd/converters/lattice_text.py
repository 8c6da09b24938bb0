"""Front end for the text lattice formats of Bmad (`.bmad`) and Elegant (`.lte`).

Behavioural mirror of cheetah/converters/utils/{fortran_namelist,infix,rpn}.py — same statements understood, same
namespace rules, same warnings — written as a small tokenizer + recursive-descent evaluator instead of regular
expressions over whole lines:

  statement            meaning                                                     reference
  -------------------  ----------------------------------------------------------  ---------------------------
  call, file = f       splice another file (relative to the calling file, `$VAR`   fortran_namelist.py:57-74
                       path parts from the environment)
  obj[prop] = expr     set a property; `type::pat*` wildcards                      :195-222
  name = expr          variable                                                    :225-241
  n: line = (a, -b)    beam line (a leading '-' marks a reversed sub-line)         :287-310
  n: overlay = {..}..  Bmad overlay, kept as raw text                              :313-354
  n: type, k = v, ...  element; `type` may name an earlier element to inherit from :244-284
  use, n               line to build                                               :357-371

Expressions: infix with + - * / ^, unary minus, parentheses, `obj[prop]` look-ups and the functions below; Elegant's
reverse-Polish strings ("1 2 +") are tried when the infix reading fails; anything else becomes a string with a
`PhysicsWarning`. Variables and functions live in separate namespaces (`abs = -0.6` followed by `abs(abs)` is legal
Bmad, tests/resources/bmad_tutorial_lattice.bmad).
"""

from __future__ import annotations

import math
import os
import re
import warnings
from copy import deepcopy
from pathlib import Path
from typing import Any

from ..warnings import NotUnderstoodPropertyWarning, PhysicsWarning

_ELECTRON_MASS_MEV = 0.51099895069  # scipy.constants "electron mass energy equivalent in MeV" (CODATA 2022)

FUNCTIONS = {
    "sqrt": math.sqrt, "sin": math.sin, "asin": math.asin, "cos": math.cos, "acos": math.acos, "tan": math.tan,
    "atan": math.atan, "abs": abs, "log": math.log,
}
#: bare words that stay strings (fortran_namelist.py:148-149)
KEYWORDS = {"open", "electron", "t", "f", "traveling_wave", "full"}


def initial_context() -> dict:
    """Constants every lattice file may use (fortran_namelist.py:383-397)."""
    return {
        "pi": math.pi, "twopi": 2 * math.pi, "c_light": 299792458.0, "emass": _ELECTRON_MASS_MEV * 1e-3,
        "m_electron": _ELECTRON_MASS_MEV * 1e6, "raddeg": math.pi / 180,
    }


# ---------------------------------------------------------------------------------------------------
# reading: comments, includes, continuation lines
def _resolve_env(path: Path) -> Path:
    return Path(*[os.environ[part[1:]] if part.startswith("$") else part for part in path.parts])


def read_statements(path: Path) -> list[str]:
    """Lower-cased logical statements of `path` with all `call, file = ...` includes spliced in."""
    physical = _read_physical_lines(Path(path))
    logical: list[str] = []
    pending = ""
    for line in physical:
        line = line.split("#")[0].strip()  # '#' starts a comment in Elegant files
        if not line and not pending:
            continue
        joined = (pending + " " + line).strip() if pending else line
        if joined.endswith("&"):
            pending = joined[:-1].rstrip()
            continue
        # a statement goes on while it ends in ',' or has an open bracket
        if joined.endswith(",") or joined.count("{") > joined.count("}") or joined.count("(") > joined.count(")"):
            pending = joined
            continue
        pending = ""
        logical.append(joined)
    if pending:
        logical.append(pending)
    statements = []
    for line in logical:
        statements += [part.strip() for part in _split_outside_quotes(line, ";") if part.strip()]
    return statements


def _read_physical_lines(path: Path) -> list[str]:
    lines = []
    with open(path) as f:
        for raw in f:
            line = raw.split("!")[0].strip()
            if not line:
                continue
            m = re.match(r"call\s*,\s*file\s*=\s*(.+)$", line, flags=re.IGNORECASE)
            if m:
                included = _resolve_env(Path(m.group(1).strip().strip('"')))
                if not included.is_absolute():
                    included = path.parent / included
                lines += _read_physical_lines(included)
            else:
                lines.append(line.lower())  # lower-cased late: environment variables are case sensitive
    return lines


def _split_outside_quotes(text: str, sep: str) -> list[str]:
    parts, depth, quoted, start = [], 0, False, 0
    for i, ch in enumerate(text):
        if ch == '"':
            quoted = not quoted
        elif not quoted and ch in "({[":
            depth += 1
        elif not quoted and ch in ")}]":
            depth -= 1
        elif not quoted and depth == 0 and ch == sep:
            parts.append(text[start:i])
            start = i + 1
    parts.append(text[start:])
    return parts


# ---------------------------------------------------------------------------------------------------
# expressions
_TOKEN = re.compile(r"\s*(?:(\d+\.?\d*(?:[ed][+-]?\d+)?|\.\d+(?:[ed][+-]?\d+)?)|([a-z_][a-z0-9_\.]*)|(.))")


class _Infix:
    """Recursive descent over: expr := term (('+'|'-') term)* ; term := unary (('*'|'/') unary)* ;
    unary := '-' unary | power ; power := atom ('^' unary)? ; atom := number | name | name '(' expr ')' |
    name '[' name ']' | '(' expr ')'."""

    def __init__(self, text: str, context: dict):
        self.tokens = []
        pos = 0
        while pos < len(text):
            m = _TOKEN.match(text, pos)
            if not m or m.end() == pos:
                break
            pos = m.end()
            if m.group(1) is not None:
                self.tokens.append(("num", float(m.group(1).replace("d", "e"))))
            elif m.group(2) is not None:
                self.tokens.append(("name", m.group(2)))
            elif m.group(3).strip():
                self.tokens.append(("op", m.group(3)))
        self.i = 0
        self.context = context

    def peek(self):
        return self.tokens[self.i] if self.i < len(self.tokens) else (None, None)

    def take(self, kind=None, value=None):
        tok = self.peek()
        if tok[0] is None or (kind and tok[0] != kind) or (value and tok[1] != value):
            raise SyntaxError(f"unexpected token {tok[1]!r}")
        self.i += 1
        return tok

    def parse(self):
        value = self.expr()
        if self.peek()[0] is not None:
            raise SyntaxError(f"trailing token {self.peek()[1]!r}")
        return value

    def expr(self):
        value = self.term()
        while self.peek() in (("op", "+"), ("op", "-")):
            op = self.take()[1]
            rhs = self.term()
            value = value + rhs if op == "+" else value - rhs
        return value

    def term(self):
        value = self.unary()
        while self.peek() in (("op", "*"), ("op", "/")):
            op = self.take()[1]
            rhs = self.unary()
            value = value * rhs if op == "*" else value / rhs
        return value

    def unary(self):
        if self.peek() == ("op", "-"):
            self.take()
            return -self.unary()
        if self.peek() == ("op", "+"):
            self.take()
            return self.unary()
        return self.power()

    def power(self):
        base = self.atom()
        if self.peek() == ("op", "^"):      # binds tighter than unary minus, right associative
            self.take()
            return base ** self.unary()
        return base

    def atom(self):
        kind, value = self.peek()
        if kind == "num":
            self.take()
            return value
        if kind == "op" and value == "(":
            self.take()
            inner = self.expr()
            self.take("op", ")")
            return inner
        if kind == "name":
            self.take()
            nxt = self.peek()
            if nxt == ("op", "(") and value in FUNCTIONS:       # functions have their own namespace
                self.take()
                arg = self.expr()
                self.take("op", ")")
                return FUNCTIONS[value](arg)
            if nxt == ("op", "["):
                self.take()
                key = self.take("name")[1]
                self.take("op", "]")
                try:
                    return self.context[value][key]
                except (KeyError, TypeError):
                    raise SyntaxError(f"unknown property {value}[{key}]")
            if value in self.context and isinstance(self.context[value], (int, float)):
                return self.context[value]
            raise SyntaxError(f"unknown name {value!r}")
        raise SyntaxError(f"unexpected token {value!r}")


def _evaluate_rpn(text: str, context: dict):
    """Elegant's reverse-Polish expressions (converters/utils/rpn.py)."""
    stack: list[float] = []
    binary = {"+": lambda a, b: a + b, "-": lambda a, b: a - b, "*": lambda a, b: a * b, "/": lambda a, b: a / b,
              "^": lambda a, b: a**b}
    unary = {k: FUNCTIONS[k] for k in ("sqrt", "sin", "cos", "tan", "asin")}
    text = text.split("#", 1)[0]   # trailing comment
    for token in [t for t in re.split(r"(\+|\-|\*|/|\^)|\s", text.strip().strip('"')) if t]:
        if token in binary:
            if len(stack) < 2:
                raise SyntaxError(f"need two values before {token}")
            b, a = stack.pop(), stack.pop()
            stack.append(binary[token](a, b))
        elif token in unary:
            if not stack:
                raise SyntaxError(f"need one value before {token}")
            stack.append(unary[token](stack.pop()))
        else:
            try:
                stack.append(float(token))
                continue
            except ValueError:
                pass
            m = re.fullmatch(r"([a-z0-9_]+)\[([a-z0-9_]+)\]", token)
            if m and isinstance(context.get(m.group(1)), dict) and m.group(2) in context[m.group(1)]:
                stack.append(context[m.group(1)][m.group(2)])
            elif isinstance(context.get(token), (int, float)):
                stack.append(context[token])
            else:
                raise SyntaxError(f"{token} is not a number or a variable")
    if len(stack) != 1:
        raise SyntaxError("stack not empty after evaluation")
    return stack[0]


def evaluate(expression: str, context: dict) -> Any:
    """Value of a right-hand side (fortran_namelist.py:124-168): int, float, keyword, known name, infix, RPN, string."""
    expression = expression.strip()
    try:
        return int(expression)
    except ValueError:
        pass
    try:
        return float(expression)
    except ValueError:
        pass
    if expression in KEYWORDS:
        return expression
    if expression in context:
        return context[expression]
    try:
        return _Infix(expression, context).parse()
    except (SyntaxError, ZeroDivisionError, ValueError, TypeError, OverflowError):
        pass
    try:
        return _evaluate_rpn(expression, context)
    except (SyntaxError, ZeroDivisionError, ValueError, TypeError, OverflowError):
        warnings.warn(f"Could not evaluate expression '{expression}'. It will now be treated as a string. This may lead "
                      "to unexpected behaviour.", category=PhysicsWarning, stacklevel=2)
        return expression.strip('"')


# ---------------------------------------------------------------------------------------------------
# statements
_NAME = r'(?:[a-z0-9_\-\.]+|"[a-z0-9_\-\.\:]+")'


def parse(path: Path | str) -> dict:
    """Execute a lattice file into a context: variables (numbers / strings), elements (dicts with `element_type`),
    beam lines (lists of names) and `__use__`."""
    context = initial_context()
    for statement in read_statements(Path(path)):
        _execute(statement, context)
    return context


def _execute(statement: str, context: dict) -> None:
    if statement == "return":
        return
    m = re.fullmatch(r'use\s*,\s*(' + _NAME + r')', statement)
    if m:
        context["__use__"] = m.group(1).strip('" ')
        return
    m = re.fullmatch(r"([a-z0-9_\*:%]+)\[([a-z0-9_%]+)\]\s*=(.*)", statement)
    if m:
        target, prop, value = m.group(1), m.group(2), evaluate(m.group(3), context)
        for name in _match_objects(target, context):
            context.setdefault(name, {})
            context[name][prop] = value
        return
    m = re.fullmatch(r"\s*(" + _NAME + r")\s*:\s*(.*)", statement)
    if m:
        _define(m.group(1).strip('" '), m.group(2).strip(), context, statement)
        return
    m = re.fullmatch(r"([a-z0-9_]+)\s*=(.*)", statement)
    if m:
        context[m.group(1)] = evaluate(m.group(2), context)
        return
    raise ValueError(f"Line '{statement}' not understood. Please check the syntax and try again.")


def _match_objects(target: str, context: dict) -> list[str]:
    if "*" not in target and "%" not in target:
        return [target]
    kind, _, pattern = target.partition("::")  # e.g. quadrupole::q* (fortran_namelist.py:171-192)
    regex = pattern.replace("*", ".*").replace("%", ".")
    return [k for k, v in context.items()
            if re.fullmatch(regex, k) and isinstance(v, dict) and v.get("element_type") == kind]


def _define(name: str, body: str, context: dict, statement: str) -> None:
    m = re.fullmatch(r"line\s*=\s*\((.*)\)", body)
    if m:
        context[name] = [member.strip().strip('"') for member in m.group(1).split(",") if member.strip()]
        return
    if re.match(r"overlay\s*=", body):
        context[name] = {"overlay_definition": body}
        return
    kind, _, rest = body.partition(",")
    kind = kind.strip()
    if not re.fullmatch(r"[a-z0-9_]+", kind):
        raise ValueError(f"Line '{statement}' not understood. Please check the syntax and try again.")
    properties = deepcopy(context[kind]) if isinstance(context.get(kind), dict) else {"element_type": kind}
    for assignment in _split_outside_quotes(rest, ","):
        if not assignment.strip():
            continue
        key, eq, value = assignment.partition("=")
        if not eq:
            continue
        properties[key.strip()] = evaluate(value.strip(), context)
    context[name] = properties


def check_understood(understood: list[str], properties: dict) -> None:
    """Warn about every property that is neither used nor knowingly ignored (fortran_namelist.py:432-451)."""
    for prop, value in properties.items():
        if not any(re.fullmatch(pattern, prop) for pattern in understood):
            warnings.warn(f"Property {prop} with value {value} for element type {properties['element_type']} is "
                          "currently not understood.", category=NotUnderstoodPropertyWarning, stacklevel=3)
