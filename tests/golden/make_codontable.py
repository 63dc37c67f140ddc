"""Extracts DATA from the reference source for the protein arm's pin (run in the build container,
where /root/reference exists; the output is committed, this script never runs on the GPU box):

  * the 64 (codon -> residue) pairs of the CODONTABLE literal, reference src/lib.rs:691-777
    (every `m.insert("XYZ", b'R');` line);
  * the order in which add_sequence walks the six reading frames, reference src/lib.rs:280-300:
    for i in 0..3 { forward skip(i) ; reverse-complement skip(i) }, read off the loop body by
    checking which of `sequence` / `rc` is translated first inside `for i in 0..3`.

Writes tests/golden/codontable.json = {"table": {"TTT": "F", ...}, "frames": [[strand, skip], ...],
"source": {...}}.  Only data goes into the fixture -- no source text.
"""
import json
import os
import re
import sys

REF = "/root/reference/src/lib.rs"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "codontable.json")


def main():
    lines = open(REF).read().split("\n")
    # locate the literal: from `static ref CODONTABLE` to the closing `};`
    start = next(i for i, l in enumerate(lines) if "static ref CODONTABLE" in l)
    end = next(i for i in range(start, len(lines)) if lines[i].strip() == "};")
    pat = re.compile(r'm\.insert\("([A-Z]{3})",\s*b\'(.)\'\);')
    table = {}
    for l in lines[start:end]:
        mm = pat.search(l)
        if mm:
            assert mm.group(1) not in table, "duplicate codon in the literal"
            table[mm.group(1)] = mm.group(2)
    assert len(table) == 64, len(table)
    assert set("".join(table)) == set("ACGT")

    # frame order of the protein arm: inside `for i in 0..3`, which buffer is translated first
    loop = next(i for i in range(len(lines)) if "for i in 0..3" in lines[i])
    body = "\n".join(lines[loop:loop + 25])
    first_fwd = body.index("sequence\n") if "sequence\n" in body else body.index("sequence")
    first_rc = body.index("rc.iter()")
    assert ".skip(i)" in body
    order = ["forward", "revcomp"] if first_fwd < first_rc else ["revcomp", "forward"]
    frames = [[strand, i] for i in range(3) for strand in order]

    doc = {
        "table": dict(sorted(table.items())),
        "frames": frames,
        "aa_ksize": "ksize / 3 (integer division), reference src/lib.rs:278",
        "source": {"table": "src/lib.rs:%d-%d" % (start + 1, end + 1), "frames": "src/lib.rs:%d-%d" % (loop + 1, loop + 21)},
    }
    with open(OUT, "w") as fh:
        json.dump(doc, fh, indent=1, sort_keys=True)
        fh.write("\n")
    print("wrote", OUT, len(table), "codons; frames", frames)


if __name__ == "__main__":
    sys.exit(main())
