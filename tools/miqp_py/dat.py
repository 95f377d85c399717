"""Reader for the OPL ``.dat`` subset used by the reference fixtures.

Grammar (fixtures: cplexmodel/cplexmodel_testcase.dat,
cplexmodel/test_sos.dat, cplexmodel/cplexmodel.dat):

    file   := { name '=' value ';' }
    value  := number | '[' { value } ']' | '{' { tuple } '}'
    tuple  := '<' number ... '>'

Elements are separated by blanks, newlines and/or commas; ``/* ... */`` and
``// ...`` comments are skipped.  The result is a plain dict name -> python
value (float/int, nested lists, list of tuples for sets).
"""
import re

_TOKEN = re.compile(r"\s*(?:(/\*.*?\*/)|(//[^\n]*)|([\[\]{}<>;=,])|"
                    r"([-+]?(?:\d+\.?\d*(?:[eE][-+]?\d+)?|\.\d+(?:[eE][-+]?\d+)?))|"
                    r"([A-Za-z_][A-Za-z_0-9]*))", re.S)


def _tokens(text):
    pos = 0
    n = len(text)
    while pos < n:
        m = _TOKEN.match(text, pos)
        if not m:
            if text[pos:].strip() == "":
                return
            raise ValueError("bad .dat syntax near %r" % text[pos:pos + 40])
        pos = m.end()
        if m.group(1) or m.group(2):
            continue
        if m.group(3):
            if m.group(3) != ",":
                yield ("p", m.group(3))
        elif m.group(4):
            s = m.group(4)
            if re.fullmatch(r"[-+]?\d+", s):
                yield ("n", int(s))
            else:
                yield ("n", float(s))
        else:
            yield ("id", m.group(5))


def parse_dat(text):
    toks = list(_tokens(text))
    i = 0

    def value():
        nonlocal i
        k, v = toks[i]
        if k == "n":
            i += 1
            return v
        if (k, v) == ("p", "["):
            i += 1
            out = []
            while toks[i] != ("p", "]"):
                out.append(value())
            i += 1
            return out
        if (k, v) == ("p", "{"):
            i += 1
            out = []
            while toks[i] != ("p", "}"):
                out.append(value())
            i += 1
            return out
        if (k, v) == ("p", "<"):
            i += 1
            out = []
            while toks[i] != ("p", ">"):
                out.append(value())
            i += 1
            return tuple(out)
        raise ValueError("unexpected token %r" % (toks[i],))

    res = {}
    while i < len(toks):
        k, name = toks[i]
        if k != "id":
            raise ValueError("expected a name, got %r" % (toks[i],))
        if toks[i + 1] != ("p", "="):
            raise ValueError("expected '=' after %s" % name)
        i += 2
        res[name] = value()
        if toks[i] != ("p", ";"):
            raise ValueError("expected ';' after %s" % name)
        i += 1
    return res


def load_dat(path):
    with open(path) as f:
        return parse_dat(f.read())
