#!/usr/bin/env python3
"""Re-wrap the prose of a markdown file to <= WIDTH columns (default 120) without changing what it renders to: paragraphs and list
items are re-flowed (continuation lines of an item keep its text indent), tables, headings, code fences and blank lines are left
alone.    tools/md_wrap.py DESIGN.md [width]"""
import re
import sys
import textwrap

path = sys.argv[1]
W = int(sys.argv[2]) if len(sys.argv) > 2 else 120
lines = open(path).read().split("\n")
out, para, indent0, indent1 = [], [], "", ""
item = re.compile(r"^(\s*)([-*+]|\d+[.)])\s+")


def flush():
    global para
    if para:
        text = " ".join(s.strip() for s in para)
        out.extend(textwrap.wrap(text, W, initial_indent=indent0, subsequent_indent=indent1, break_long_words=False, break_on_hyphens=False))
        para = []


fence = False
for ln in lines:
    if ln.lstrip().startswith("```"):
        flush(); fence = not fence; out.append(ln); continue
    if fence or ln.lstrip().startswith("|") or ln.startswith("#") or not ln.strip() or ln.lstrip().startswith("<!--"):
        flush(); out.append(ln); continue
    m = item.match(ln)
    if m:
        flush()
        indent0 = ln[:m.end()]
        indent1 = " " * len(indent0)
        para = [ln[m.end():]]
        indent0 = ln[:m.end()]
        continue
    if not para:
        lead = len(ln) - len(ln.lstrip())
        indent0 = indent1 = " " * lead
    para.append(ln)
flush()
open(path, "w").write("\n".join(out))
