"""tools/check_erl_nif_header.py — the first-contact check of nif/nxsig_nif.c against a REAL OTP's erl_nif.h.
No BEAM here, so the tool is exercised against an OTP-STYLE rendering of the stub written by this test (the API table as
ERL_NIF_API_FUNC_DECL lines with parameter names, an enum where OTP has one, static inline tuple helpers): it must pass, and it must
fail on a changed return type, a changed parameter, a dropped parameter and a missing function.  Against an installed OTP:
`make -C nif check-header`."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import check_erl_nif_header as chk  # noqa: E402


def otp_style(stub_text: str) -> tuple[str, str]:
    """(erl_nif.h, erl_nif_api_funcs.h) in the shape OTP ships them, generated from the stub's prototypes"""
    text = chk.strip_comments(stub_text)
    api, inline = [], []
    for m in re.finditer(r"^([A-Za-z_][\w \t\*]*?)\b(enif_\w+)\s*\(([^;{}]*?)\)\s*;", text, flags=re.M | re.S):
        ret, name, args = m.group(1).strip(), m.group(2), " ".join(m.group(3).split())
        # give every unnamed parameter a name, the way OTP's table spells them
        params = []
        for i, a in enumerate([x.strip() for x in args.split(",")] if args.strip() else []):
            if re.fullmatch(r"(const )?[A-Za-z_]\w*\*?", a) or a.endswith("*"):
                a = f"{a} p{i}"
            params.append(a)
        args2 = ", ".join(params).replace("int encoding", "ErlNifCharEncoding encoding")
        if name.startswith("enif_make_tuple") and name[-1].isdigit():
            inline.append(f"static ERL_NIF_INLINE {ret} {name}({args2})\n{{ ERL_NIF_TERM a[] = {{0}}; return enif_make_tuple_from_array(p0, a, 0); }}")
        else:
            api.append(f"ERL_NIF_API_FUNC_DECL({ret},{name},({args2}));")
    main = ("#ifndef __ERL_NIF_H__\n#define __ERL_NIF_H__\ntypedef enum { ERL_NIF_LATIN1 = 1, ERL_NIF_UTF8 = 2 } ErlNifCharEncoding;\n"
            "#define ERL_NIF_API_FUNC_DECL(RET_TYPE, NAME, ARGS) extern RET_TYPE NAME ARGS\n#include \"erl_nif_api_funcs.h\"\n"
            + "\n".join(inline) + "\n#endif\n")
    return main, "\n".join(api) + "\n"


def run(dirpath):
    return subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_erl_nif_header.py"), str(dirpath)], capture_output=True, text=True)


def write(tmp, main, api):
    tmp.mkdir(exist_ok=True)
    (tmp / "erl_nif.h").write_text(main)
    (tmp / "erl_nif_api_funcs.h").write_text(api)
    return tmp


def test_the_list_of_enif_functions_the_shim_uses_is_committed_and_current():
    used = chk.used_functions(open(chk.SHIM).read())
    listed = [ln.strip() for ln in open(chk.USED) if ln.strip() and not ln.startswith("#")]
    assert listed == used, "run: python tools/check_erl_nif_header.py --list"
    stub = chk.parse_header(open(chk.STUB).read())
    assert not [f for f in used if f not in stub], "the shim calls a function the stub does not declare"
    assert 20 <= len(used) <= 60


def test_an_otp_style_header_that_agrees_passes_and_every_kind_of_drift_fails(tmp_path):
    main, api = otp_style(open(chk.STUB).read())
    r = run(write(tmp_path / "same", main, api))
    assert r.returncode == 0, r.stdout + r.stderr
    assert "26 enif_* functions" in r.stdout or "functions used by the shim" in r.stdout
    cases = {
        "ret": api.replace("ERL_NIF_API_FUNC_DECL(void,enif_keep_resource", "ERL_NIF_API_FUNC_DECL(int,enif_keep_resource"),
        "param": api.replace("ErlNifSInt64* ip", "long* ip"),
        "count": re.sub(r"\(enif_alloc_resource\),?", "", api).replace("enif_alloc_resource,(ErlNifResourceType* type, size_t size)",
                                                                     "enif_alloc_resource,(ErlNifResourceType* type)"),
        "missing": "\n".join(ln for ln in api.splitlines() if "enif_inspect_binary" not in ln) + "\n",
    }
    for name, mutated in cases.items():
        assert mutated != api, name
        r = run(write(tmp_path / name, main, mutated))
        assert r.returncode == 1 and "MISMATCH" in r.stdout, (name, r.stdout, r.stderr)
    assert run(tmp_path / "nowhere").returncode == 2


def test_makefile_has_the_check_header_target():
    mk = open(os.path.join(ROOT, "nif", "Makefile")).read()
    assert "check-header" in mk and "check_erl_nif_header.py" in mk
