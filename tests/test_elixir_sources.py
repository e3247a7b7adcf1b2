"""The Elixir sources cannot be compiled in this image (no BEAM).  Next to the NIF name / arity cross-check of test_nif_shim.py,
a structural check: every file's do/fn blocks are closed and its brackets balance (strings, heredocs and comments stripped)."""
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_elixir_blocks_and_brackets_balance():
    files = sorted(glob.glob(os.path.join(ROOT, "elixir", "lib", "**", "*.ex"), recursive=True)
                   + glob.glob(os.path.join(ROOT, "elixir", "test", "*.exs")) + [os.path.join(ROOT, "elixir", "mix.exs"), os.path.join(ROOT, "elixir", "smoke.exs")])
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


def test_smoke_exs_calls_only_what_the_host_modules_define():
    """elixir/smoke.exs (round 6: `elixir elixir/smoke.exs` is the one-command first contact with a BEAM): every NxSignalAMD function it
    calls is defined in elixir/lib with that arity range, the NIF loader honours NXSIG_NIF_PATH, and it ends in one PASS / FAIL line"""
    t = open(os.path.join(ROOT, "elixir", "smoke.exs")).read()
    lib = {os.path.relpath(f, os.path.join(ROOT, "elixir", "lib")): open(f).read()
           for f in glob.glob(os.path.join(ROOT, "elixir", "lib", "**", "*.ex"), recursive=True)}
    main = lib["nx_signal_amd.ex"]
    for fn in ("context", "stft", "istft", "last_dispatch"):
        assert re.search(r"^\s*def %s\(" % fn, main, re.M), fn
        assert "sig.%s(" % fn in t, fn
    assert re.search(r"def rectangular\(", lib["nx_signal_amd/windows.ex"]) and re.search(r"for kind <- \[[^\]]*:hann", lib["nx_signal_amd/windows.ex"])
    assert re.search(r"def to_device\(", lib["nx_signal_amd/device_tensor.ex"]) and re.search(r"def from_device\(", lib["nx_signal_amd/device_tensor.ex"])
    assert "NXSIG_NIF_PATH" in lib["nx_signal_amd/nif.ex"] and "NXSIG_NIF_PATH" in t
    assert 'Mix.install([{:nx, "~> 0.11"}])' in t
    assert t.count("nxsig smoke PASS") >= 1 and t.count("nxsig smoke FAIL") >= 1 and "System.halt(1)" in t
    # the doctest literals are the reference's (lib/nx_signal.ex:46-65), as held in tests/golden/reference_vectors.json
    assert "[1.0, -1.0, 3.0, -1.0, 5.0, -1.0]" in t and "[0.0, 200.0]" in t and "[0.0025, 0.005, 0.0075]" in t
