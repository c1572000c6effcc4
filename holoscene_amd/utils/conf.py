"""Minimal HOCON reader + pyhocon-style accessor.

The reference parses its ``confs/**/*.conf`` with pyhocon (training/holoscene_train.py:48) and the
model reads them through ``get_int/get_float/get_bool/get_list/get_string/get_config``
(model/network.py:757-770).  pyhocon is a pure convenience dependency, so this build carries a
small reader for the subset those files use: ``key = value``, nested ``key { ... }`` blocks,
``[a, b]`` lists, ``#`` / ``//`` comments, numbers, booleans, bare and quoted strings.
A real pyhocon ``ConfigTree`` works too -- the model only needs the accessor methods.
"""
import re

_MISSING = object()


class Conf(dict):
    def _lookup(self, key, default):
        cur = self
        for part in key.split("."):
            if not isinstance(cur, dict) or part not in cur:
                if default is _MISSING:
                    raise KeyError(f"No configuration setting found for key {key}")
                return default
            cur = cur[part]
        return cur

    def get(self, key, default=None):
        return self._lookup(key, default)

    def get_int(self, key, default=_MISSING):
        v = self._lookup(key, default)
        return v if v is None else int(v)

    def get_float(self, key, default=_MISSING):
        v = self._lookup(key, default)
        return v if v is None else float(v)

    def get_bool(self, key, default=_MISSING):
        v = self._lookup(key, default)
        if isinstance(v, str):
            return v.lower() in ("true", "yes", "on")
        return v if v is None else bool(v)

    def get_string(self, key, default=_MISSING):
        v = self._lookup(key, default)
        return v if v is None else str(v)

    def get_list(self, key, default=_MISSING):
        v = self._lookup(key, default)
        return v if v is None else list(v)

    def get_config(self, key, default=_MISSING):
        v = self._lookup(key, default)
        return Conf(v) if isinstance(v, dict) and not isinstance(v, Conf) else v


_TOKEN = re.compile(r"""\s*(?:(?P<punct>[{}\[\],=:])|"(?P<qstr>(?:[^"\\]|\\.)*)"|(?P<bare>[^\s{}\[\],=:"]+))""")


def _scalar(tok):
    low = tok.lower()
    if low in ("true", "false"):
        return low == "true"
    if low == "null":
        return None
    try:
        return int(tok)
    except ValueError:
        pass
    try:
        return float(tok)
    except ValueError:
        return tok


def _tokens(text):
    lines = []
    for line in text.splitlines():
        out, in_q = [], False
        i = 0
        while i < len(line):
            ch = line[i]
            if ch == '"':
                in_q = not in_q
            if not in_q and (ch == "#" or line.startswith("//", i)):
                break
            out.append(ch)
            i += 1
        lines.append("".join(out))
    text = "\n".join(lines)
    pos, toks = 0, []
    while pos < len(text):
        m = _TOKEN.match(text, pos)
        if not m:
            if text[pos:].strip() == "":
                break
            raise ValueError(f"cannot tokenise config near {text[pos:pos + 30]!r}")
        pos = m.end()
        if m.group("punct"):
            toks.append(("p", m.group("punct")))
        elif m.group("qstr") is not None:
            toks.append(("s", m.group("qstr")))
        else:
            toks.append(("b", m.group("bare")))
    return toks


def _parse_value(toks, i):
    kind, val = toks[i]
    if kind == "p" and val == "{":
        return _parse_object(toks, i + 1)
    if kind == "p" and val == "[":
        items, i = [], i + 1
        while toks[i] != ("p", "]"):
            if toks[i] == ("p", ","):
                i += 1
                continue
            v, i = _parse_value(toks, i)
            items.append(v)
        return items, i + 1
    if kind == "s":
        return val, i + 1
    return _scalar(val), i + 1


def _parse_object(toks, i):
    obj = Conf()
    while i < len(toks):
        kind, val = toks[i]
        if kind == "p" and val == "}":
            return obj, i + 1
        if kind == "p" and val == ",":
            i += 1
            continue
        key = val
        i += 1
        if toks[i] in (("p", "="), ("p", ":")):
            i += 1
        v, i = _parse_value(toks, i)
        cur = obj
        parts = key.split(".")
        for part in parts[:-1]:
            cur = cur.setdefault(part, Conf())
        if isinstance(v, dict) and isinstance(cur.get(parts[-1]), dict):
            cur[parts[-1]].update(v)
        else:
            cur[parts[-1]] = v
    return obj, i


def parse_string(text):
    obj, _ = _parse_object(_tokens(text), 0)
    return obj


def parse_file(path):
    with open(path) as f:
        return parse_string(f.read())


def get_class(kls):
    """Dotted-name class lookup, as the reference's utils.general.get_class (utils/general.py:188-194)."""
    parts = kls.split(".")
    m = __import__(".".join(parts[:-1]))
    for comp in parts[1:]:
        m = getattr(m, comp)
    return m
