#!/usr/bin/env python3
"""One-off (round 5): splits the rounds-1..4 lab notebook that DESIGN.md had become into
docs/history/round{1..4}.md.  A block = the text from a bold round marker ("**Round 3 additions.**", "**Round 4 kernels.**",
"... round 4**", "## 11. Round 4 against ...") up to the next marker or section heading; it goes to that round's file under its
section heading.  Text without a marker is the round-1 base (edited in place by later rounds) and goes to round1.md.  Prose
lines are re-wrapped at 140 characters; tables and code blocks are left alone."""
import os
import re
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MARK = re.compile(r"^(\*\*[^*]*\b[Rr]ound[ -]?(\d)\b[^*]*\*\*|## \d+\. Round (\d) against)")


def wrap(lines):
    out, in_code = [], False
    for ln in lines:
        if ln.startswith("```"):
            in_code = not in_code
        if in_code or ln.startswith("|") or len(ln) <= 140 or ln.startswith("    "):
            out.append(ln)
            continue
        m = re.match(r"^(\s*(?:[*-]|\d+\.)\s+)?", ln)
        lead = m.group(1) or ""
        body = ln[len(lead):]
        out.extend(textwrap.wrap(body, width=140, initial_indent=lead, subsequent_indent=" " * len(lead),
                                 break_long_words=False, break_on_hyphens=False))
    return out


def main(src):
    lines = open(src).read().split("\n")
    files = {r: [] for r in (1, 2, 3, 4)}
    section, cur, emitted = "(preamble)", 1, {r: None for r in files}
    for ln in lines:
        if ln.startswith("## "):
            section = ln
            m = MARK.match(ln)
            cur = int(m.group(3)) if m and m.group(3) else 1
        else:
            m = MARK.match(ln)
            if m:
                cur = int(m.group(2) or m.group(3))
            elif ln.startswith("# "):
                continue
        if cur not in files:
            cur = 4
        if emitted[cur] != section:
            files[cur] += ["", section if section.startswith("## ") else "## " + section, ""]
            emitted[cur] = section
        if not ln.startswith("## "):
            files[cur].append(ln)
    os.makedirs(os.path.join(ROOT, "docs", "history"), exist_ok=True)
    for r, body in files.items():
        head = [f"# Design history - round {r}", "",
                f"What DESIGN.md said about round {r} when round 4 ended (split out in round 5 by tools/split_design_history.py; the",
                "text is unchanged apart from re-wrapping).  Measurements quoted here are that round's; the CURRENT state of every",
                "kernel is in /DESIGN.md." + ("  Unmarked text of the old file - the round-1 base that later rounds edited in place - is here."
                                               if r == 1 else ""), ""]
        with open(os.path.join(ROOT, "docs", "history", f"round{r}.md"), "w") as f:
            f.write("\n".join(wrap(head + body)).rstrip() + "\n")
        print(r, len(body))


if __name__ == "__main__":
    main(sys.argv[1])
