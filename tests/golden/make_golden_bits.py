#!/usr/bin/env python3
"""Transcribe the reference's known-answer tests for bit-vector distances
(test/sql/bit.sql with test/expected/bit.out, the psql transcript pg_regress compares
against) into tests/golden/bit_known_answers.json.  Runs only where /root/reference is
mounted; the JSON is committed so the tests never read the reference tree."""
import json
import os
import re
import sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bit_known_answers.json")
BIT = r"'([01]*)'(?:::varbit\((\d+)\))?"


def main():
    lines = open(os.path.join(REF, "test", "expected", "bit.out")).read().split("\n")
    cases = []
    for i, line in enumerate(lines):
        m = re.match(r"^SELECT (hamming_distance|jaccard_distance)\(" + BIT + ", " + BIT + r"\);$", line)
        if not m:
            m2 = re.match(r"^SELECT " + BIT + r" (<~>|<%>) " + BIT + ";$", line)
            if not m2:
                continue
            func = {"<~>": "hamming_distance", "<%>": "jaccard_distance"}[m2.group(3)]
            a, b = m2.group(1), m2.group(4)
        else:
            func, a, b = m.group(1), m.group(2), m.group(4)
        nxt = lines[i + 1]
        if nxt.startswith("ERROR:"):
            cases.append({"func": func, "a": a, "b": b, "error": nxt[len("ERROR:"):].strip(), "line": i + 1})
        else:
            cases.append({"func": func, "a": a, "b": b, "value": float(lines[i + 3].strip()), "line": i + 1})
    json.dump({"source": "test/expected/bit.out", "cases": cases}, open(OUT, "w"), indent=1)
    print("%d cases -> %s" % (len(cases), OUT))


if __name__ == "__main__":
    main()
