#!/usr/bin/env python3
"""Build-container only: reads /root/reference/py/multivariate_py.cpp AS TEXT and writes
tests/golden/class_surface.json -- for every pybind11 class of the module its Python name,
C++ class, base class, the keyword list of its py::init (order, default or null = required),
and the methods/properties def()'d on MultivariateSearch / MultivariateSolution.

    python scripts/gen_class_surface.py [--src /root/reference/py/multivariate_py.cpp]

The fixture is data (names and default values); tests/test_abi.py compares the classes of
bboptpy_amd.multivariate and bbo_params_default() with it, so no expected signature is typed
by hand.  Commented-out bindings (// ...) of the reference are skipped like the compiler does.
"""
import argparse
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def strip_comments(text):
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return "\n".join(line.split("//", 1)[0] for line in text.splitlines())


def split_top(s):
    """split at commas that are outside (), <>, {} and string literals"""
    out, depth, cur, in_str = [], 0, "", False
    for ch in s:
        if ch == ">" and cur.endswith("-"):        # '->' of a lambda's return type
            cur += ch
            continue
        if in_str:
            cur += ch
            if ch == '"':
                in_str = False
            continue
        if ch == '"':
            in_str = True
            cur += ch
        elif ch in "(<{":
            depth += 1
            cur += ch
        elif ch in ")>}":
            depth -= 1
            cur += ch
        elif ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def balanced(text, start):
    """text[start] == '(' -> index one past its matching ')'"""
    depth, i, in_str = 0, start, False
    while i < len(text):
        ch = text[i]
        if in_str:
            in_str = ch != '"'
        elif ch == '"':
            in_str = True
        elif ch == "(":
            depth += 1
        elif ch == ")":
            depth -= 1
            if depth == 0:
                return i + 1
        i += 1
    raise ValueError("unbalanced parenthesis")


def literal(tok):
    """a C++ default -> a JSON value (bool / int / float / None for nullptr / the token itself
    for an enum)"""
    t = tok.strip()
    if t == "true":
        return True
    if t == "false":
        return False
    if t == "nullptr":
        return None
    try:
        return int(t)
    except ValueError:
        pass
    try:
        return float(t.rstrip("fF"))
    except ValueError:
        return {"cxx": t}


def parse(text):
    text = strip_comments(text)
    classes = {}
    # py::class_<Cxx[, Base]> var(m, "Name");
    for m in re.finditer(r'py::class_<\s*([\w:]+)\s*(?:,\s*([\w:]+)\s*)?>\s*(\w+)\s*\(\s*m\s*,\s*"(\w+)"\s*\)',
                         text):
        cxx, base, var, name = m.groups()
        # the function body this declaration lives in: up to the next "\nvoid " / end
        end = text.find("\nvoid ", m.end())
        body = text[m.end(): end if end > 0 else len(text)]
        entry = {"cxx": cxx, "base_cxx": base, "init": None, "defs": [], "properties": []}
        for d in re.finditer(r"\b%s\s*\.\s*(def|def_property_readonly)\s*\(" % re.escape(var), body):
            stop = balanced(body, d.end() - 1)
            args = split_top(body[d.end(): stop - 1])
            if d.group(1) == "def_property_readonly":
                entry["properties"].append(json.loads(args[0]))
            elif args[0].startswith("py::init"):
                kws = []
                for a in args[1:]:
                    k = re.match(r'"(\w+)"_a\s*(?:=\s*(.+))?$', a, flags=re.S)
                    if not k:
                        continue
                    kws.append({"name": k.group(1), "required": k.group(2) is None,
                                "default": None if k.group(2) is None else literal(k.group(2))})
                types = re.match(r"py::init<(.*)>\s*\(\s*\)$", args[0], flags=re.S)
                entry["init"] = {"keywords": kws,
                                 "cxx_types": [t.strip() for t in split_top(types.group(1))]
                                 if types else None}
            else:
                nm = json.loads(args[0])
                kws = [re.match(r'"(\w+)"_a', a).group(1) for a in args[1:]
                       if re.match(r'"(\w+)"_a', a)]
                entry["defs"].append({"name": nm, "keywords": kws})
        classes[name] = entry
    by_cxx = {v["cxx"]: k for k, v in classes.items()}
    for v in classes.values():
        v["base"] = by_cxx.get(v["base_cxx"])
    return classes


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--src", default="/root/reference/py/multivariate_py.cpp")
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden", "class_surface.json"))
    a = ap.parse_args()
    with open(a.src) as fh:
        classes = parse(fh.read())
    doc = {"source": "py/multivariate_py.cpp (parsed as text by scripts/gen_class_surface.py)",
           "classes": classes}
    with open(a.out, "w") as fh:
        json.dump(doc, fh, indent=1, sort_keys=True)
        fh.write("\n")
    print("%d classes -> %s" % (len(classes), a.out), file=sys.stderr)


if __name__ == "__main__":
    main()
