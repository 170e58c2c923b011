#!/usr/bin/env python
"""Re-wrap the paragraphs and list items of a Markdown file to 118 columns (tables, headings and code fences untouched):
    python tools/reflow_md.py DESIGN.md"""
import re, sys, textwrap


def reflow(text, width=118):
    out = []
    for para in text.split("\n\n"):
        lines = para.split("\n")
        if any(l.startswith("|") or l.startswith("```") or l.startswith("#") for l in lines):
            out.append(para)
            continue
        items, cur = [], None
        for l in lines:
            m = re.match(r"^(\s*)(\*|-|\d+\.)\s+(.*)$", l)
            if m and not l.startswith("**"):
                if cur is not None:
                    items.append(cur)
                cur = [m.group(1) + m.group(2) + " ", m.group(3)]
            elif cur is None:
                cur = ["", l.strip()]
            else:
                cur[1] += " " + l.strip()
        if cur is not None:
            items.append(cur)
        out.append("\n".join(textwrap.fill(body, width=width, initial_indent=head, subsequent_indent=" " * len(head),
                                          break_long_words=False, break_on_hyphens=False) for head, body in items))
    return "\n\n".join(out)


if __name__ == "__main__":
    s = open(sys.argv[1]).read()
    parts = re.split(r"(```.*?```)", s, flags=re.S)
    open(sys.argv[1], "w").write("".join(p if p.startswith("```") else reflow(p) for p in parts).rstrip() + "\n")
