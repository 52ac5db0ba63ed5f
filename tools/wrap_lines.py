#!/usr/bin/env python3
"""Wraps over-long lines of C++ sources at 160 columns without touching tokens: comment-only lines are re-flowed, a trailing `// comment` moves onto its own line above the
code, and code is broken after `; `, `, ` or before ` && ` / ` || ` / ` ? ` at parenthesis depth <= the shallowest possible (never inside string or character literals).
Preprocessor lines, lines ending in a backslash and anything it cannot break safely are left alone.

  python tools/wrap_lines.py gatling_amd/csrc/gi_render.cpp [...]"""
import re
import sys

LIMIT = 160


def split_code_comment(line):
    """(code, comment) with comment starting at the first // outside literals, or (line, None)."""
    i, n, q = 0, len(line), None
    while i < n:
        c = line[i]
        if q:
            if c == "\\": i += 2; continue
            if c == q: q = None
        elif c in "\"'": q = c
        elif c == "/" and i + 1 < n and line[i + 1] == "/": return line[:i].rstrip(), line[i:]
        elif c == "/" and i + 1 < n and line[i + 1] == "*":
            j = line.find("*/", i + 2)
            if j < 0: return line, None
            i = j + 2; continue
        i += 1
    return line, None


def reflow_comment(indent, text):
    words, out, cur = text.split(" "), [], indent + "//"
    for w in words:
        if w == "" and cur.endswith("//"): cur += " "; continue
        if len(cur) + 1 + len(w) > LIMIT and cur.strip() != "//": out.append(cur.rstrip()); cur = indent + "// " + w
        else: cur += ("" if cur.endswith(" ") else " ") + w
    out.append(cur.rstrip())
    return out


def break_points(code):
    """[(position after which to break, depth)] outside literals."""
    pts, depth, q, i, n = [], 0, None, 0, len(code)
    while i < n:
        c = code[i]
        if q:
            if c == "\\": i += 2; continue
            if c == q: q = None
        elif c in "\"'": q = c
        elif c in "([{": depth += 1
        elif c in ")]}": depth -= 1
        elif c == "/" and code[i:i + 2] == "/*":
            j = code.find("*/", i + 2); i = (j + 2) if j >= 0 else n; continue
        elif code[i:i + 2] in ("; ", ", ") : pts.append((i + 1, depth, code[i]))
        elif code[i:i + 4] in (" && ", " || ") or code[i:i + 3] == " ? ": pts.append((i, depth, "o"))
        i += 1
    return pts


def wrap_code(indent, code):
    lines, rest, first = [], code, True
    while len(rest) > LIMIT:
        lead = indent if first else indent + "    "
        body = rest if first else rest
        pts = [(p, d, k) for p, d, k in break_points(body) if len(indent) + 4 < p <= LIMIT - (0 if first else 0)]
        if not pts: return None
        dmin = min(d for _, d, _ in pts)
        cands = [p for p, d, k in pts if d == dmin and (k == ";" or True)]
        semis = [p for p, d, k in pts if d == dmin and k == ";"]
        p = max(semis) if semis and max(semis) > LIMIT * 0.45 else max(cands)
        lines.append(body[:p].rstrip())
        rest = indent + ("  " if dmin == 0 and body[:p].rstrip().endswith(";") and False else "    ") + body[p:].lstrip()
        first = False
    lines.append(rest)
    return lines


def process(path):
    out, changed = [], 0
    for line in open(path).read().split("\n"):
        if len(line) <= LIMIT or line.lstrip().startswith("#") or line.rstrip().endswith("\\"):
            out.append(line); continue
        indent = line[:len(line) - len(line.lstrip())]
        code, comment = split_code_comment(line)
        if code.strip() == "" and comment is not None:
            out += reflow_comment(indent, comment[2:].strip()); changed += 1; continue
        new = []
        if comment is not None:
            new += reflow_comment(indent, comment[2:].strip())
        if len(code) > LIMIT:
            w = wrap_code(indent, code)
            if w is None:
                out.append(line); continue
            new += w
        else:
            new.append(code)
        out += new; changed += 1
    open(path, "w").write("\n".join(out))
    return changed


def reflow_paragraphs(path):
    """Second pass: a run of prose comment lines that holds a wrap remainder (a line of at most three words that is not the run's last) is re-flowed as one paragraph."""
    lines, out, i, fixed = open(path).read().split("\n"), [], 0, 0
    def prose(l):
        t = l.lstrip()
        return t.startswith("// ") and not t.startswith("//  ") and "   " not in t[3:] and not re.match(r"// -{8,}", t)
    while i < len(lines):
        if not prose(lines[i]):
            out.append(lines[i]); i += 1; continue
        indent = lines[i][:len(lines[i]) - len(lines[i].lstrip())]
        j = i
        while j < len(lines) and prose(lines[j]) and lines[j].startswith(indent + "// "): j += 1
        run = lines[i:j]
        if any(len(l.split()) <= 4 for l in run[:-1]) and len(run) > 1:   # ("//" + three words)
            out += reflow_comment(indent, " ".join(l.lstrip()[3:].strip() for l in run)); fixed += 1
        else:
            out += run
        i = j
    open(path, "w").write("\n".join(out))
    return fixed


if __name__ == "__main__":
    args = sys.argv[1:]
    if args and args[0] == "--reflow":
        for p in args[1:]:
            print(p, reflow_paragraphs(p), "paragraphs re-flowed")
    else:
        for p in args:
            print(p, process(p), "lines wrapped")
