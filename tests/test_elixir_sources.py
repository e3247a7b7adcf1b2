"""The Elixir sources cannot be compiled in this image (no BEAM).  Next to the NIF name / arity cross-check of test_nif_shim.py,
a structural check: every file's do/fn blocks are closed and its brackets balance (strings, heredocs and comments stripped)."""
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_elixir_blocks_and_brackets_balance():
    files = sorted(glob.glob(os.path.join(ROOT, "elixir", "lib", "**", "*.ex"), recursive=True)
                   + glob.glob(os.path.join(ROOT, "elixir", "test", "*.exs")) + [os.path.join(ROOT, "elixir", "mix.exs")])
    assert len(files) >= 10
    for f in files:
        t = open(f).read()
        t = re.sub(r'"""(.|\n)*?"""', '""', t)
        t = re.sub(r'"(\\.|[^"\\])*"', '""', t)
        t = re.sub(r"#.*", "", t)
        opens = len(re.findall(r"\bdo\b(?!:)", t)) + len(re.findall(r"\bfn\b", t))
        assert opens == len(re.findall(r"\bend\b", t)), f
        for a, b in ("()", "[]", "{}"):
            assert t.count(a) == t.count(b), (f, a)
        # clauses of one function stay together (the compiler warns otherwise): a name does not reappear after another def
        seen, last = set(), None
        for m in re.finditer(r"^\s*defp?\s+(\w+[?!]?)[\s(]", t, re.M):
            name = m.group(1)
            if name != last:
                assert name not in seen, (f, name)
                seen.add(name)
            last = name


def test_generated_golden_exs_is_in_sync_with_the_json():
    """elixir/test/golden_vectors_test.exs is generated from tests/golden/reference_vectors.json (tools/gen_elixir_golden_test.py):
    the committed file must be what the generator writes today, and must cover every family of vectors the JSON holds"""
    import json
    import subprocess
    import sys

    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_elixir_golden_test.py"), "--check"])
    assert r.returncode == 0, "run python tools/gen_elixir_golden_test.py"
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_vectors.json")))
    text = open(os.path.join(ROOT, "elixir", "test", "golden_vectors_test.exs")).read()
    n_tests = len(re.findall(r'^  test "', text, re.M))
    families = [k for k in g if isinstance(g[k], list) and k != "fft_rows"]   # fft_rows pins Nx.fft itself (not an NxSignal function)
    assert n_tests == sum(len(g[k]) for k in families), (n_tests, {k: len(g[k]) for k in families})
    for v in g["windows"]:
        assert v["src"] in text
