#!/usr/bin/env python3
"""Transcribe the reference's known-answer tests for the distance path into
tests/golden/distance_known_answers.json.

Runs only where /root/reference is mounted (the build container); the JSON it
writes is committed so the tests never read the reference tree.  Source files:
test/expected/vector_type.out and test/expected/halfvec.out (the psql
transcripts pg_regress compares against, i.e. the reference's own outputs).
"""
import json
import os
import re
import sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "distance_known_answers.json")

FUNCS = ("l2_distance", "inner_product", "cosine_distance", "l1_distance", "vector_norm", "l2_norm",
         "l2_normalize")
OPS = {"<->": "l2_distance", "<#>": "negative_inner_product", "<=>": "cosine_distance", "<+>": "l1_distance"}
VEC = r"'(\[[^']*\])'(?:::(\w+))?"


def parse_vec(text):
    return [float(x) for x in text.strip("[]").split(",") if x.strip() != ""]


def parse_value(line):
    v = line.strip()
    if v.startswith("["):
        return parse_vec(v)
    return {"Infinity": "inf", "-Infinity": "-inf", "NaN": "nan"}.get(v, None) or float(v)


def cases_of(path, default_type):
    lines = open(path).read().split("\n")
    out = []
    for i, line in enumerate(lines):
        if not line.startswith("SELECT "):
            continue
        stmt = line[len("SELECT "):].rstrip(";")
        wrap = None
        m = re.match(r"^round\((.*)::numeric, (\d+)\)$", stmt)
        if m:
            stmt, wrap = m.group(1), ("round", int(m.group(2)))
        m = re.match(r"^(.*)::real$", stmt)
        if m:
            stmt, wrap = m.group(1), ("real",)
        func = args = None
        m = re.match(r"^(\w+)\(" + VEC + r"(?:, " + VEC + r")?\)$", stmt)
        if m and m.group(1) in FUNCS:
            func = m.group(1)
            args = [m.group(2)] + ([m.group(4)] if m.group(4) else [])
            typ = m.group(3) or m.group(5) or default_type
        else:
            m = re.match(r"^" + VEC + r" (<->|<#>|<=>|<\+>) " + VEC + r"$", stmt)
            if m:
                func = OPS[m.group(3)]
                args = [m.group(1), m.group(4)]
                typ = m.group(2) or m.group(5) or default_type
        if func is None:
            continue
        nxt = lines[i + 1]
        case = {"source": "%s:%d" % (os.path.relpath(path, REF), i + 1), "type": typ, "func": func,
                "args": [parse_vec(a) for a in args]}
        if wrap:
            case["wrap"] = list(wrap)
        if nxt.startswith("ERROR:"):
            case["error"] = nxt[len("ERROR:"):].strip()
        else:
            case["expect"] = parse_value(lines[i + 3])
        out.append(case)
    return out


def main():
    cases = cases_of(os.path.join(REF, "test/expected/vector_type.out"), "vector")
    cases += cases_of(os.path.join(REF, "test/expected/halfvec.out"), "halfvec")
    cases = [c for c in cases if c["type"] in ("vector", "halfvec")]
    json.dump({"generated_by": "tests/golden/make_golden.py", "reference": "pgvector v0.8.6 test/expected",
               "cases": cases}, open(OUT, "w"), indent=1)
    print("wrote %d cases to %s" % (len(cases), OUT))


if __name__ == "__main__":
    main()
