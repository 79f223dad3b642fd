#!/usr/bin/env python3
"""Lifts the golden vectors of the reference's own unit tests into a JSON fixture.

Source: /root/reference/src/edge264_check.c
  test_intra_decoding  :169-278  fixed border, expected 4x4 (14 modes), 8x8 (31 modes
                                 exercised: loop runs 0..I8x8_HU_8), 16x16 (7), chroma (7)
  test_inter_decoding  :282-359  src[i*21+j] = (i*21+j)*37, 48 luma cases + 3 chroma
Mode names are resolved to numbers through the enums of
/root/reference/src/edge264_internal.h (:564-634 intra, and the INTER_* enum).

Run in the build container (where /root/reference exists):
    python tests/golden/extract_check_vectors.py
writes tests/golden/check_vectors.json (committed; the GPU box has no /root/reference).
"""
import json
import os
import re
import sys

REF = os.environ.get("EDGE264_REF", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))


def enum_values(text: str, name: str) -> dict:
    m = re.search(r"enum\s+" + name + r"\s*\{(.*?)\}", text, re.S)
    out, v = {}, 0
    for item in m.group(1).split(","):
        item = re.sub(r"//.*", "", item).strip()
        if not item:
            continue
        if "=" in item:
            k, e = [s.strip() for s in item.split("=")]
            v = int(e, 0)
        else:
            k = item
        out[k] = v
        v += 1
    return out


def array_rows(body: str) -> list:
    """Top-level {...} rows of a C 2-D initializer."""
    rows, depth, cur = [], 0, ""
    for ch in body:
        if ch == "{":
            depth += 1
            if depth == 1:
                cur = ""
                continue
        if ch == "}":
            depth -= 1
            if depth == 0:
                rows.append(cur)
                continue
        if depth >= 1:
            cur += ch
    return rows


def grab(text: str, decl_regex: str) -> str:
    m = re.search(decl_regex + r"\s*=\s*\{", text)
    i = m.end() - 1
    depth = 0
    for j in range(i, len(text)):
        depth += text[j] == "{"
        depth -= text[j] == "}"
        if depth == 0:
            return text[i + 1:j]
    raise ValueError(decl_regex)


def main() -> int:
    check = open(os.path.join(REF, "src/edge264_check.c")).read()
    internal = open(os.path.join(REF, "src/edge264_internal.h")).read()
    inter_c = open(os.path.join(REF, "src/edge264_inter.c")).read()
    out = {"source": "edge264_check.c:169-359", "intra_border": {
        "comment": "p[x-stride]=194+4x, p[x-2*stride]=198+4x for x=-1..15, p[y*stride-1]=186-4y (check.c:173-180)"}}

    def ints(row):
        return [int(t, 0) for t in re.findall(r"-?\d+", row)]
    for key, enum, arr in (("intra4x4", "Intra4x4Modes", r"I4x4_expect\[\]\[16\]"),
                           ("intra8x8", "Intra8x8Modes", r"I8x8_expect\[\]\[64\]"),
                           ("intra16x16", "Intra16x16Modes", r"I16x16_expect\[\]\[256\]"),
                           ("intra_chroma", "IntraChromaModes", r"IC8x8_expect\[\]\[128\]")):
        rows = [ints(r) for r in array_rows(grab(check, arr))]
        names = sorted(enum_values(internal, enum).items(), key=lambda kv: kv[1])
        out[key] = [{"mode": names[i][1], "name": names[i][0], "expect": rows[i]} for i in range(len(rows))]
    # inter luma: first element of each row is the mode enumerator name
    src = internal + inter_c
    m = re.search(r"enum\s*\w*\s*\{[^}]*INTER_4xH_QPEL_00[^}]*\}", src, re.S)
    names, v = {}, 0
    for item in re.search(r"\{(.*)\}", m.group(0), re.S).group(1).split(","):
        item = re.sub(r"//.*", "", item).strip()
        if item:
            names[item] = v
            v += 1
    luma = []
    for r in array_rows(grab(check, r"luma_expect\[\]\[257\]")):
        nm = re.match(r"\s*(\w+)", r).group(1)
        vals = ints(r.split(",", 1)[1])
        luma.append({"name": nm, "mode": names[nm], "expect": vals})
    out["inter_luma"] = luma
    out["inter_src"] = "src[i*21+j] = ((i*21+j)*37) & 255 for i,j in 0..20; block origin src+44 (check.c:286-290, 344)"
    chroma = []
    for w, h, cw in ((16, 16, 8), (8, 16, 4), (4, 8, 2)):
        m = re.search(r'assert_block\("INTER_CHROMA_%dx%d".*?\(uint8_t\[\]\)\{(.*?)\}\);' % (w, h), check, re.S)
        chroma.append({"w": w, "h": h, "cols": cw, "rows": h, "ABCD": [3, 21, 5, 35], "expect": ints(m.group(1))})
    out["inter_chroma"] = chroma
    path = os.path.join(HERE, "check_vectors.json")
    with open(path, "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print("wrote", path, {k: len(v) for k, v in out.items() if isinstance(v, list)})
    return 0


if __name__ == "__main__":
    sys.exit(main())
