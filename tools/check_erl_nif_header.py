#!/usr/bin/env python
"""First-contact check of the dirty-NIF shim against a REAL Erlang/OTP.

nif/nxsig_nif.c has only ever been compiled against tests/stub/erl_nif.h, a header written from the erl_nif documentation
(this image has no BEAM).  Where an OTP installation exists this tool compares, for every enif_* function the shim calls,
the prototype the stub declares with the one OTP's own headers declare (erl_nif.h + erl_nif_api_funcs.h: the
ERL_NIF_API_FUNC_DECL(ret, name, (args)) table, plain prototypes, static inline definitions and function-like macros) and
fails on any difference in return type, parameter count or parameter types.

    python tools/check_erl_nif_header.py --list                 # (re)write nif/enif_functions_used.txt from the shim
    python tools/check_erl_nif_header.py <dir with erl_nif.h>   # exit 0: every prototype agrees; 1: mismatches; 2: header missing
    make -C nif check-header                                    # the same against the installed OTP (skips without `erl`)

tests/test_nif_header_check.py runs it here against an OTP-style rendering of the stub (must pass) and against copies with
a changed return type / parameter / missing function (must fail), and keeps nif/enif_functions_used.txt up to date."""
from __future__ import annotations

import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "nif", "nxsig_nif.c")
STUB = os.path.join(ROOT, "tests", "stub", "erl_nif.h")
USED = os.path.join(ROOT, "nif", "enif_functions_used.txt")

# types that differ in spelling only: OTP's enums travel as int at the C ABI, and OTP spells some integer typedefs differently
ALIASES = {
    "ErlNifCharEncoding": "int",
    "ErlNifResourceFlags": "ErlNifResourceFlags",
    "ErlNifSInt64": "ErlNifSInt64",
    "unsigned int": "unsigned",
}


def strip_comments(text: str) -> str:
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    return re.sub(r"//[^\n]*", " ", text)


def used_functions(shim_src: str) -> list[str]:
    return sorted(set(re.findall(r"\b(enif_[a-z0-9_]+)\s*\(", strip_comments(shim_src))))


def norm_type(t: str) -> str:
    """a parameter or return type without its name, qualifiers in canonical order, one space between tokens"""
    t = re.sub(r"\bERL_NIF_INLINE\b|\bstatic\b|\bextern\b|\binline\b|\bERL_NAPI_ATTR_[A-Z_]+\b", " ", t)
    t = re.sub(r"\s+", " ", t).strip()
    t = re.sub(r"\s*\*\s*", "* ", t).strip()
    # array parameters decay: `const T arr[]` == `const T* arr`
    m = re.match(r"(.*?)(\w+)\s*\[\s*\]$", t)
    if m:
        t = m.group(1).strip() + "* " + m.group(2)
    toks = t.split(" ")
    # drop a trailing parameter name: the last token when it is an identifier and something precedes it that already names a type
    if len(toks) >= 2 and re.fullmatch(r"[A-Za-z_]\w*", toks[-1]) and toks[-1] not in ("int", "unsigned", "char", "long", "double", "void", "size_t"):
        if not (len(toks) == 2 and toks[0] in ("const", "unsigned", "struct", "enum")):
            toks = toks[:-1]
    t = " ".join(toks)
    t = t.replace("unsigned int", "unsigned")
    for a, b in ALIASES.items():
        t = re.sub(r"\b%s\b" % re.escape(a), b, t)
    t = re.sub(r"\s*\*", "*", t)
    t = re.sub(r"\bconst (\w+)\b", r"\1 const", t)   # west const -> east const
    return re.sub(r"\s+", " ", t).strip()


def split_params(args: str) -> list[str]:
    args = args.strip()
    if args in ("", "void"):
        return []
    out, depth, cur = [], 0, ""
    for ch in args:
        if ch == "(":
            depth += 1
        elif ch == ")":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur)
            cur = ""
        else:
            cur += ch
    out.append(cur)
    return [norm_type(p) for p in out]


def parse_header(text: str) -> dict:
    """name -> ("proto", ret, [param types]) | ("macro", n_params)"""
    text = strip_comments(text)
    protos: dict = {}
    for m in re.finditer(r"ERL_NIF_API_FUNC_DECL\s*\(\s*([^,]+?)\s*,\s*(enif_\w+)\s*,\s*\((.*?)\)\s*\)\s*;", text, flags=re.S):
        protos[m.group(2)] = ("proto", _ret(m.group(1)), split_params(m.group(3)))
    for m in re.finditer(r"^[ \t]*#[ \t]*define[ \t]+(enif_\w+)\(([^)]*)\)", text, flags=re.M):
        protos.setdefault(m.group(1), ("macro", len([a for a in m.group(2).split(",") if a.strip()])))
    # plain prototypes and static inline definitions: <ret> enif_name(<args>) ; | {
    for m in re.finditer(r"(?:^|[;}\n])\s*((?:[A-Za-z_][\w \t\*]*?)?)\b(enif_\w+)\s*\(([^;{}]*?)\)\s*(?:;|\{)", text, flags=re.S):
        ret = m.group(1).strip()
        if not ret or ret.startswith("#") or ret.endswith("return"):
            continue
        protos.setdefault(m.group(2), ("proto", _ret(ret), split_params(m.group(3))))
    return protos


def _ret(r: str) -> str:
    r = re.sub(r"\bERL_NIF_INLINE\b|\bstatic\b|\bextern\b|\binline\b|\bERL_NAPI_ATTR_[A-Z_]+\b", " ", r)
    r = re.sub(r"\s+", " ", r).strip()
    r = re.sub(r"\s*\*", "*", r)
    return r.replace("unsigned int", "unsigned")


def read_otp(dirpath: str) -> str | None:
    main = os.path.join(dirpath, "erl_nif.h")
    if not os.path.exists(main):
        return None
    text = open(main, errors="replace").read()
    api = os.path.join(dirpath, "erl_nif_api_funcs.h")
    if os.path.exists(api):
        text += "\n" + open(api, errors="replace").read()
    return text


def compare(used: list[str], stub: dict, real: dict) -> list[str]:
    bad = []
    for f in used:
        if f not in stub:
            bad.append(f"{f}: used by the shim but not declared in tests/stub/erl_nif.h")
            continue
        if f not in real:
            bad.append(f"{f}: not declared by this OTP's erl_nif.h / erl_nif_api_funcs.h")
            continue
        s, r = stub[f], real[f]
        if r[0] == "macro" or s[0] == "macro":
            ns = len(s[2]) if s[0] == "proto" else s[1]
            nr = len(r[2]) if r[0] == "proto" else r[1]
            if ns != nr:
                bad.append(f"{f}: {ns} parameters in the stub, {nr} in OTP (function-like macro)")
            continue
        if s[1] != r[1]:
            bad.append(f"{f}: returns `{s[1]}` in the stub, `{r[1]}` in OTP")
        if len(s[2]) != len(r[2]):
            bad.append(f"{f}: {len(s[2])} parameters in the stub, {len(r[2])} in OTP")
        else:
            for i, (a, b) in enumerate(zip(s[2], r[2])):
                if a != b:
                    bad.append(f"{f}: parameter {i + 1} is `{a}` in the stub, `{b}` in OTP")
    return bad


def main(argv) -> int:
    used = used_functions(open(SHIM).read())
    if len(argv) > 1 and argv[1] == "--list":
        with open(USED, "w") as fh:
            fh.write("# enif_* functions nif/nxsig_nif.c calls (tools/check_erl_nif_header.py --list; kept current by tests/test_nif_header_check.py)\n")
            fh.write("\n".join(used) + "\n")
        print(f"{len(used)} functions -> {os.path.relpath(USED, ROOT)}")
        return 0
    if len(argv) < 2:
        print(__doc__)
        return 2
    text = read_otp(argv[1])
    if text is None:
        print(f"check-header: no erl_nif.h under {argv[1]}", file=sys.stderr)
        return 2
    stub = parse_header(open(STUB).read())
    real = parse_header(text)
    bad = compare(used, stub, real)
    for line in bad:
        print("MISMATCH", line)
    print(f"check-header: {len(used)} enif_* functions used by the shim, {len(used) - len({b.split(':')[0] for b in bad})} agree with {argv[1]}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
