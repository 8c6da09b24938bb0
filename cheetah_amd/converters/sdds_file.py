"""Reader for SDDS ("Self Describing Data Sets") files — the container Elegant writes particle distributions in.

The reference delegates this to the third-party `sdds` package (`cheetah/converters/elegant.py:467-480`, `sdds.load`); that
package is optional here: `converters.elegant.convert_beam` uses it when it is installed and this module otherwise, so that
`ParticleBeam.from_elegant` works without it. Written from the published format description (SDDS protocol versions 1-5):

    SDDS<version>                       first line
    !# little-endian | !# big-endian    optional, byte order of binary data
    ! ...                               comment lines anywhere in the header and in ASCII data
    &description ... &end               namelist commands, possibly spanning several lines; values may be "quoted"
    &parameter name=..., type=..., [fixed_value=...] &end
    &column name=..., type=... &end
    &data mode=ascii|binary, [lines_per_row=..], [no_row_counts=1], [additional_header_lines=..],
          [column_major_order=1], [endian=little|big] &end
    <pages>

    &array name=..., type=..., [dimensions=N] &end

ASCII page: one line per parameter that has no fixed value, then each array (a line with its `dimensions` sizes, then its
elements over as many lines as they take), then (if there are columns) the row count and the rows.
Binary page: int32 row count, the parameters (strings as int32 length + bytes), the arrays (int32 size per dimension, then
the elements), then the table row by row, or column by column with `column_major_order=1`.

Supported: parameters, arrays and columns of every type (short, ushort, long, ulong, long64, ulong64, float, double,
longdouble, character, string), ASCII and binary, both byte orders, any number of pages. `longdouble` in binary data is read
as the x86 80-bit extended format in 16 bytes (what SDDS writes on the machines Elegant runs on) and rounded to a Python
float; in big-endian binary data its layout depends on the writing machine and it raises NotImplementedError, as does
`&include`.

`load(path)` returns an `SddsData` with the attributes the converter reads from the `sdds` package's object:
`parameterName`, `columnName`, `columnData[column][page][row]`, `parameterData[parameter][page]`,
`getParameterValueList(name)`, `getColumnValueLists(name)`; arrays are `arrayName`, `arrayDimensions[array][page]` and
`arrayData[array][page]` (the elements in file order, last dimension fastest).
"""

from __future__ import annotations

import math
import re
import struct
from pathlib import Path

_NUMERIC = {  # SDDS type -> (struct code, size)
    "short": ("h", 2), "ushort": ("H", 2), "long": ("i", 4), "ulong": ("I", 4), "long64": ("q", 8), "ulong64": ("Q", 8),
    "float": ("f", 4), "double": ("d", 8),
}
_INTEGER = {"short", "ushort", "long", "ulong", "long64", "ulong64"}
_LONGDOUBLE_BYTES = 16


def _extended_to_float(chunk: bytes) -> float:
    """x86 extended precision, little endian: 64-bit significand with an explicit leading bit, then sign + 15-bit exponent."""
    significand, = struct.unpack_from("<Q", chunk, 0)
    tail, = struct.unpack_from("<H", chunk, 8)
    sign = -1.0 if tail & 0x8000 else 1.0
    exponent = tail & 0x7FFF
    if exponent == 0x7FFF:
        return sign * math.inf if significand << 1 & (2**64 - 1) == 0 else math.nan
    if exponent == 0:
        exponent = 1            # subnormals share the smallest exponent
    try:
        return sign * math.ldexp(float(significand), exponent - 16383 - 63)
    except OverflowError:
        return sign * math.inf


class SddsData:
    """The parsed file: names in file order, data indexed [name][page] (parameters) and [name][page][row] (columns)."""

    def __init__(self) -> None:
        self.description: dict[str, str] = {}
        self.parameterName: list[str] = []
        self.parameterDefinition: list[dict[str, str]] = []
        self.parameterData: list[list] = []
        self.columnName: list[str] = []
        self.columnDefinition: list[dict[str, str]] = []
        self.columnData: list[list[list]] = []
        self.arrayName: list[str] = []
        self.arrayDefinition: list[dict[str, str]] = []
        self.arrayDimensions: list[list[list[int]]] = []
        self.arrayData: list[list[list]] = []
        self.mode = "ascii"

    @property
    def loaded_pages(self) -> int:
        if self.columnData:
            return len(self.columnData[0])
        if self.parameterData:
            return len(self.parameterData[0])
        return len(self.arrayData[0]) if self.arrayData else 0

    def getParameterValueList(self, name: str) -> list:
        return self.parameterData[self.parameterName.index(name)]

    def getColumnValueLists(self, name: str) -> list[list]:
        return self.columnData[self.columnName.index(name)]


# ---- header ---------------------------------------------------------------------------------------------------------------
_FIELD = re.compile(r"""\s*([A-Za-z_][A-Za-z_0-9]*)\s*=\s*("(?:[^"\\]|\\.)*"|[^,\s]*)\s*,?""")


def _parse_fields(body: str) -> dict[str, str]:
    fields, pos = {}, 0
    body = body.strip()
    while pos < len(body):
        m = _FIELD.match(body, pos)
        if m is None:
            raise ValueError(f"SDDS header: cannot parse namelist fields at {body[pos:pos + 40]!r}")
        value = m.group(2)
        if value.startswith('"'):
            value = re.sub(r"\\(.)", r"\1", value[1:-1])
        fields[m.group(1)] = value
        pos = m.end()
    return fields


def _read_header(raw: bytes):
    """-> (version, commands [(name, fields)], byte order or None, offset of the first data byte)."""
    pos = raw.find(b"\n")
    first = raw[: pos if pos >= 0 else len(raw)].decode("latin-1").strip()
    m = re.fullmatch(r"SDDS(\d+)", first)
    if m is None:
        raise ValueError(f"not an SDDS file: the first line is {first[:20]!r}")
    version = int(m.group(1))
    if not 1 <= version <= 5:
        raise ValueError(f"SDDS protocol version {version} is not supported")
    pos = pos + 1 if pos >= 0 else len(raw)
    commands, endian = [], None
    while True:
        if pos >= len(raw):
            raise ValueError("SDDS header ends without a &data command")
        end = raw.find(b"\n", pos)
        end = len(raw) if end < 0 else end
        line = raw[pos:end].decode("latin-1")
        stripped = line.strip()
        if not stripped:
            pos = end + 1
            continue
        if stripped.startswith("!"):
            if stripped.startswith("!#"):
                word = stripped[2:].strip().lower()
                if word in ("little-endian", "big-endian"):
                    endian = "<" if word.startswith("little") else ">"
            pos = end + 1
            continue
        if not stripped.startswith("&"):
            raise ValueError(f"SDDS header: expected a namelist command, found {stripped[:40]!r}")
        # a command runs up to its &end, which may sit on a later line; quoted values may contain '&'
        text, scan, in_quote = "", pos, False
        while True:
            if scan >= len(raw):
                raise ValueError("SDDS header: namelist command without &end")
            ch = chr(raw[scan])
            if ch == "\\" and in_quote:
                text += ch + chr(raw[scan + 1])
                scan += 2
                continue
            if ch == '"':
                in_quote = not in_quote
            if not in_quote and raw[scan:scan + 4].lower() == b"&end":
                scan += 4
                break
            text += ch
            scan += 1
        nl = raw.find(b"\n", scan)
        pos = len(raw) if nl < 0 else nl + 1
        # comment lines inside a multi-line command are dropped
        text = "\n".join(part for part in text.split("\n") if not part.strip().startswith("!"))
        m = re.match(r"\s*&([A-Za-z_]+)(.*)", text, re.S)
        name = m.group(1).lower()
        commands.append((name, _parse_fields(m.group(2).replace("\n", " "))))
        if name == "data":
            return version, commands, endian, pos


# ---- ASCII data -----------------------------------------------------------------------------------------------------------
def _convert(token: str, sdds_type: str):
    if sdds_type in _INTEGER:
        return int(token)
    if sdds_type in ("float", "double", "longdouble"):
        return float(token)
    if sdds_type == "character":
        return token[:1] if not token.startswith("\\") else chr(int(token[1:], 8))
    return token


_TOKEN = re.compile(r'"((?:[^"\\]|\\.)*)"|(\S+)')


def _tokens(line: str) -> list[str]:
    return [re.sub(r"\\(.)", r"\1", m.group(1)) if m.group(1) is not None else m.group(2) for m in _TOKEN.finditer(line)]


def _read_ascii(data: SddsData, text: str, fixed: dict, opts: dict) -> None:
    lines = [ln for ln in text.split("\n") if not ln.lstrip().startswith("!")]
    pos = int(opts.get("additional_header_lines", 0) or 0)
    lines_per_row = int(opts.get("lines_per_row", 1) or 1)
    no_row_counts = bool(int(opts.get("no_row_counts", 0) or 0))
    ptypes = [d["type"] for d in data.parameterDefinition]
    ctypes = [d["type"] for d in data.columnDefinition]

    def next_line(skip_blank: bool):
        nonlocal pos
        while pos < len(lines):
            ln = lines[pos]
            pos += 1
            if ln.strip() or not skip_blank:
                return ln
        return None

    def at_end() -> bool:
        nonlocal pos
        while pos < len(lines) and not lines[pos].strip():
            pos += 1
        return pos >= len(lines)

    atypes = [d["type"] for d in data.arrayDefinition]
    adims = [int(d.get("dimensions", 1) or 1) for d in data.arrayDefinition]
    if not data.parameterName and not ctypes and not atypes:
        return
    while not at_end():      # a page starts with its first non-blank line
        page_params = []
        for i, (name, t) in enumerate(zip(data.parameterName, ptypes)):
            if name in fixed:
                page_params.append(_convert(fixed[name], t) if t != "string" else fixed[name])
                continue
            ln = next_line(t != "string")
            if ln is None:
                raise ValueError(f"SDDS ASCII data ends inside the parameters of page {data.loaded_pages + 1}")
            if t == "string":
                tok = _tokens(ln)
                page_params.append(tok[0] if len(tok) == 1 and ln.strip().startswith('"') else ln.strip())
            else:
                page_params.append(_convert(_tokens(ln)[0], t))
        page_arrays = []
        for name, t, nd in zip(data.arrayName, atypes, adims):
            ln = next_line(True)
            if ln is None:
                raise ValueError(f"SDDS ASCII data ends before the dimensions of array {name}")
            sizes = [int(tok) for tok in _tokens(ln)[:nd]]
            if len(sizes) != nd or min(sizes) < 0:
                raise ValueError(f"SDDS ASCII array {name}: expected {nd} sizes, found {ln.strip()[:40]!r}")
            total, values = math.prod(sizes), []
            while len(values) < total:
                ln = next_line(True)
                if ln is None:
                    raise ValueError(f"SDDS ASCII data ends after {len(values)} of {total} elements of array {name}")
                values += [_convert(tok, t) for tok in _tokens(ln)]
            page_arrays.append((sizes, values[:total]))
        rows_of = [[] for _ in ctypes]
        if ctypes:
            if no_row_counts:
                count = None
            else:
                ln = next_line(True)
                if ln is None:
                    raise ValueError("SDDS ASCII data ends before the row count")
                count = int(_tokens(ln)[0])
            done = 0
            while count is None or done < count:
                toks: list[str] = []
                blank = False
                for _ in range(lines_per_row):
                    ln = next_line(count is not None)      # without row counts a blank line ends the page
                    if ln is None or (count is None and not ln.strip()):
                        blank = True
                        break
                    toks += _tokens(ln)
                if blank:
                    if count is None:
                        break
                    raise ValueError(f"SDDS ASCII data ends after {done} of {count} rows")
                if len(toks) < len(ctypes):
                    raise ValueError(f"SDDS ASCII row {done + 1} has {len(toks)} values for {len(ctypes)} columns")
                for c, t in enumerate(ctypes):
                    rows_of[c].append(_convert(toks[c], t))
                done += 1
        for i, v in enumerate(page_params):
            data.parameterData[i].append(v)
        for i, (sizes, values) in enumerate(page_arrays):
            data.arrayDimensions[i].append(sizes)
            data.arrayData[i].append(values)
        for c, col in enumerate(rows_of):
            data.columnData[c].append(col)


# ---- binary data ----------------------------------------------------------------------------------------------------------
class _Cursor:
    def __init__(self, raw: bytes, pos: int, endian: str) -> None:
        self.raw, self.pos, self.endian = raw, pos, endian

    def take(self, code: str, size: int, count: int = 1):
        end = self.pos + size * count
        if end > len(self.raw):
            raise ValueError("SDDS binary data ends inside a page")
        values = struct.unpack_from(f"{self.endian}{count}{code}", self.raw, self.pos)
        self.pos = end
        return values

    def value(self, sdds_type: str):
        if sdds_type == "string":
            (n,) = self.take("i", 4)
            if n < 0 or self.pos + n > len(self.raw):
                raise ValueError("SDDS binary data: bad string length")
            s = self.raw[self.pos:self.pos + n].decode("latin-1")
            self.pos += n
            return s
        if sdds_type == "character":
            return self.take("c", 1)[0].decode("latin-1")
        return self.values(sdds_type, 1)[0]

    def values(self, sdds_type: str, count: int) -> list:
        if sdds_type == "longdouble":
            if self.endian != "<":
                raise NotImplementedError("SDDS type longdouble in big-endian binary data is not supported")
            end = self.pos + _LONGDOUBLE_BYTES * count
            if end > len(self.raw):
                raise ValueError("SDDS binary data ends inside a page")
            out = [_extended_to_float(self.raw[at:at + 10]) for at in range(self.pos, end, _LONGDOUBLE_BYTES)]
            self.pos = end
            return out
        if sdds_type in _NUMERIC:
            return list(self.take(*_NUMERIC[sdds_type], count))
        return [self.value(sdds_type) for _ in range(count)]


def _read_binary(data: SddsData, raw: bytes, pos: int, endian: str, fixed: dict, opts: dict) -> None:
    cur = _Cursor(raw, pos, endian)
    ptypes = [d["type"] for d in data.parameterDefinition]
    ctypes = [d["type"] for d in data.columnDefinition]
    column_major = bool(int(opts.get("column_major_order", 0) or 0))
    all_numeric = all(t in _NUMERIC for t in ctypes)
    while cur.pos < len(raw):
        (count,) = cur.take("i", 4)
        if count == -(2**31):            # rows counted in 64 bits
            (count,) = cur.take("q", 8)
        if count < 0:
            raise ValueError(f"SDDS binary data: negative row count {count}")
        for i, (name, t) in enumerate(zip(data.parameterName, ptypes)):
            if name in fixed:
                data.parameterData[i].append(_convert(fixed[name], t) if t != "string" else fixed[name])
            else:
                data.parameterData[i].append(cur.value(t))
        for i, definition in enumerate(data.arrayDefinition):
            sizes = list(cur.take("i", 4, int(definition.get("dimensions", 1) or 1)))
            if min(sizes) < 0:
                raise ValueError(f"SDDS binary data: array {definition['name']} has sizes {sizes}")
            data.arrayDimensions[i].append(sizes)
            data.arrayData[i].append(cur.values(definition["type"], math.prod(sizes)))
        cols = [[] for _ in ctypes]
        if ctypes and count:
            if column_major:
                for c, t in enumerate(ctypes):
                    cols[c] = cur.values(t, count)
            elif all_numeric:
                fmt = endian + "".join(_NUMERIC[t][0] for t in ctypes)
                size = struct.calcsize(fmt)
                if cur.pos + size * count > len(raw):
                    raise ValueError("SDDS binary data ends inside a page")
                rows = list(struct.iter_unpack(fmt, raw[cur.pos:cur.pos + size * count]))
                cur.pos += size * count
                cols = [list(col) for col in zip(*rows)]
            else:
                for _ in range(count):
                    for c, t in enumerate(ctypes):
                        cols[c].append(cur.value(t))
        for c, col in enumerate(cols):
            data.columnData[c].append(col)


def load(path) -> SddsData:
    """Read every page of an SDDS file."""
    raw = Path(path).read_bytes()
    version, commands, endian, pos = _read_header(raw)
    data = SddsData()
    fixed: dict[str, str] = {}
    opts: dict[str, str] = {}
    for name, fields in commands:
        if name == "description":
            data.description = fields
        elif name in ("parameter", "column", "array"):
            if "name" not in fields:
                raise ValueError(f"SDDS header: &{name} without a name")
            t = fields.get("type", "").lower()
            if t not in _NUMERIC and t not in ("string", "character", "longdouble"):
                raise ValueError(f"SDDS header: &{name} {fields['name']} has unknown type {t!r}")
            fields["type"] = t
            if name == "parameter":
                data.parameterName.append(fields["name"])
                data.parameterDefinition.append(fields)
                data.parameterData.append([])
                if "fixed_value" in fields:
                    fixed[fields["name"]] = fields["fixed_value"]
            elif name == "column":
                data.columnName.append(fields["name"])
                data.columnDefinition.append(fields)
                data.columnData.append([])
            else:
                data.arrayName.append(fields["name"])
                data.arrayDefinition.append(fields)
                data.arrayDimensions.append([])
                data.arrayData.append([])
        elif name == "include":
            raise NotImplementedError("SDDS &include commands are not supported")
        elif name == "data":
            opts = fields
        # &associate and unknown commands carry no data: ignored
    data.mode = opts.get("mode", "binary").lower()
    if data.mode == "ascii":
        _read_ascii(data, raw[pos:].decode("latin-1"), fixed, opts)
    elif data.mode == "binary":
        declared = opts.get("endian", "").lower()
        if declared in ("little", "big"):
            endian = "<" if declared == "little" else ">"
        _read_binary(data, raw, pos, endian or "<", fixed, opts)
    else:
        raise ValueError(f"SDDS &data mode {data.mode!r} is neither ascii nor binary")
    return data
