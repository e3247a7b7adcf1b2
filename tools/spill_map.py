"""Where do a kernel's scratch (spill) instructions sit relative to its loops?

    python tools/spill_map.py file.s <mangled-name-substring>

For every function of the assembly file whose name contains the substring: the line offsets of scratch_load / scratch_store
instructions with their basic-block label, and the loop back edges (label, from-line, to-line), so that one can see at a glance
whether the spills of a register-capped kernel landed in the streaming loop or in a cold block.
"""
import re
import sys


def main():
    path, key = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    starts = [i for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)]
    for si, s in enumerate(starts):
        name = lines[s].split(":")[0]
        if key not in name:
            continue
        e = starts[si + 1] if si + 1 < len(starts) else len(lines)
        body = lines[s:e]
        labs, lab = {}, None
        spills = []
        for i, l in enumerate(body):
            m = re.match(r"^(\.LBB\d+_\d+):", l)
            if m:
                lab = m.group(1)
                labs[lab] = i
            if "scratch_" in l and not l.strip().startswith(";"):
                spills.append((i, lab, l.strip().split(";")[0][:60]))
        edges = []
        for i, l in enumerate(body):
            m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
            if m and m.group(1) in labs and labs[m.group(1)] < i:
                edges.append((labs[m.group(1)], i, m.group(1)))
        print(name, "lines", len(body))
        for a, b, lb in edges:
            n = sum(1 for i, _, _ in spills if a <= i <= b)
            print(f"  loop {lb}: lines {a}..{b} ({b - a} long), scratch ops inside: {n}")
        print(f"  scratch ops total: {len(spills)}")


if __name__ == "__main__":
    main()
