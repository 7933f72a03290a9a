#!/usr/bin/env python3
"""One-off formatter (round 6): wrap a markdown file at 160 columns.  Paragraph and list lines are re-wrapped with
their indentation kept; a table with a row longer than the limit becomes a bullet list (one bullet per row, one sub-bullet per further
column, labelled with the column's header); code fences and short tables are left alone.  usage: wrap_md.py FILE [WIDTH]"""
import re
import sys
import textwrap

path = sys.argv[1]
W = int(sys.argv[2]) if len(sys.argv) > 2 else 160
lines = open(path).read().split("\n")
out, i, fence = [], 0, False


def wrap(text, first, rest):
    return textwrap.wrap(text, width=W, initial_indent=first, subsequent_indent=rest, break_long_words=False, break_on_hyphens=False) or [first.rstrip()]


def cells(row):
    row = row.strip()
    if row.startswith("|"):
        row = row[1:]
    if row.endswith("|"):
        row = row[:-1]
    return [c.strip() for c in re.split(r"(?<!\\)\|", row)]


while i < len(lines):
    ln = lines[i]
    if ln.lstrip().startswith("```"):
        fence = not fence
        out.append(ln)
        i += 1
        continue
    if fence:
        out.append(ln)
        i += 1
        continue
    if ln.startswith("|"):
        j = i
        while j < len(lines) and lines[j].startswith("|"):
            j += 1
        block = lines[i:j]
        if max(len(b) for b in block) <= W or len(block) < 3:
            out.extend(block)
        else:
            head = cells(block[0])
            for row in block[2:]:
                cs = cells(row)
                out.extend(wrap(f"**{cs[0]}**" if not cs[0].startswith("**") else cs[0], "* ", "  "))
                for name, c in zip(head[1:], cs[1:]):
                    if c:
                        out.extend(wrap(f"*{name}:* {c}", "  - ", "    "))
        i = j
        continue
    if len(ln) <= W:
        out.append(ln)
        i += 1
        continue
    m = re.match(r"^(\s*)((?:[*\-+]|\d+[.)])\s+)?", ln)
    indent, bullet = m.group(1), m.group(2) or ""
    out.extend(wrap(ln[len(indent) + len(bullet):], indent + bullet, indent + " " * len(bullet)))
    i += 1
open(path, "w").write("\n".join(out))
