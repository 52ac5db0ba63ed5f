#!/usr/bin/env python3
"""Wraps over-long lines of C++ sources at 160 columns without touching tokens: comment-only lines are re-flowed, a trailing `// comment` moves onto its own line above the
code, and code is broken after `; `, `, ` or before ` && ` / ` || ` / ` ? ` at parenthesis depth <= the shallowest possible (never inside string or character literals).
Preprocessor lines, the lines of a macro and anything it cannot break safely are left alone.

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


def tokens(text):
    """[(gap, word)]: the words of a comment with the number of spaces in front of each (sentences here end in two spaces; that is kept)."""
    out, gap = [], 1
    for w in text.strip().split(" "):
        if w == "": gap += 1; continue
        out.append((gap, w)); gap = 1
    return out


def reflow_comment(indent, text, limit=None):
    limit = limit or LIMIT
    out, cur = [], None
    for gap, w in tokens(text):
        if cur is None: cur = indent + "// " + w
        elif len(cur) + gap + len(w) > limit: out.append(cur); cur = indent + "// " + w
        else: cur += " " * gap + w
    out.append(cur if cur is not None else indent + "//")
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
    m = re.match(r"^(\s*(?:template <[^{}]*> )?[\w:<>,*&\s\[\]()\"=.+-]*?\)(?: const)?) (\{ .* \})$", code)
    if m and not re.match(r"^\s*(if|for|while|else|switch|do)\b", code) and len(m.group(1)) <= LIMIT and len(indent) + len(m.group(2)) <= LIMIT:
        pts = break_points(m.group(1))
        if all(d > 0 for _, d, _ in pts):  # the signature ends at depth 0 right before the body
            return [m.group(1), indent + m.group(2)]
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


def is_comment_only(line):
    return line.lstrip().startswith("//")


def indent_of(line):
    return len(line) - len(line.lstrip())


def split_at_limit(indent, text):
    """(first line, remainder text or '') of a comment whose words are `text`."""
    lines = reflow_comment(indent, text)
    first = lines[0]
    rest = text.strip()[len(first) - len(indent) - 3:].strip()
    return first, rest


def balanced_comment(indent, text):
    """The comment in as few lines as LIMIT allows, those lines about equally long (no two-word remainder)."""
    n = len(reflow_comment(indent, text))
    lo, hi = len(indent) + 20, LIMIT
    while lo < hi:
        mid = (lo + hi) // 2
        if len(reflow_comment(indent, text, mid)) <= n: hi = mid
        else: lo = mid + 1
    return reflow_comment(indent, text, lo)


def process(path):
    """One pass.  (a) an over-long comment-only line keeps what fits; the overflow moves to the front of the next line when that is a prose comment line of the same
    indentation (cascading), else becomes a line of its own; (b) an over-long code line loses its trailing comment to the line(s) above -- together with the
    comment-only continuation lines aligned under it -- and is then broken at `) {` of a one-line function, or after `; ` / `, ` / before ` && ` ..."""
    src, out, changed, i = open(path).read().split("\n"), [], 0, 0
    while i < len(src):
        line = src[i]; i += 1
        in_macro = i >= 2 and src[i - 2].rstrip().endswith("\\")  # (the last line of a macro ends without a backslash)
        if len(line) <= LIMIT or line.lstrip().startswith("#") or line.rstrip().endswith("\\") or in_macro:
            out.append(line); continue
        indent = line[:indent_of(line)]
        code, comment = split_code_comment(line)
        if code.strip() == "" and comment is not None:
            first, carry = split_at_limit(indent, comment[2:])
            out.append(first); changed += 1
            while carry:
                nxt = src[i] if i < len(src) else None
                # the next line continues the sentence: the line that overflowed stopped mid-sentence and a prose comment line of the same indentation follows
                prose = nxt is not None and is_comment_only(nxt) and indent_of(nxt) == len(indent) and nxt.lstrip().startswith("// ") \
                    and not nxt.lstrip().startswith("//  ") and "   " not in nxt.lstrip()[3:] and not re.match(r"// -{8,}", nxt.lstrip()) \
                    and not re.search(r"[.:;)]$", carry.strip())
                if not prose:
                    if len(carry) < 40:  # a stub: balance it with the line it fell off instead
                        out.pop(); out += balanced_comment(indent, first[len(indent) + 3:] + " " + carry)
                    else: out += reflow_comment(indent, carry)
                    break
                i += 1
                first, carry = split_at_limit(indent, carry + " " + nxt.lstrip()[3:].rstrip())
                out.append(first)
            continue
        new = []
        if comment is not None:
            text = comment[2:].strip()
            while i < len(src) and is_comment_only(src[i]) and indent_of(src[i]) >= max(len(indent) + 8, 16) and src[i].strip() != "//":
                text += " " + src[i].strip()[2:].strip(); i += 1  # continuation lines aligned under the trailing comment
            new += balanced_comment(indent, text)
        if len(code) > LIMIT:
            w = wrap_code(indent, code)
            if w is None:
                new.append(code)
            else:
                new += w
        else:
            new.append(code)
        out += new; changed += 1
    open(path, "w").write("\n".join(out))
    return changed


if __name__ == "__main__":
    for p in sys.argv[1:]:
        print(p, process(p), "lines wrapped")
