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
