#!/usr/bin/env python3
"""One-off (round 4): cut DESIGN.md into a reviewable main document and three companions, every prose line <= 120
characters.  docs/RESULTS.md <- section 6 (numbers), docs/EXPERIMENTS.md <- 6.1, 6.2, 9a, 9b (what was tried, round by round),
docs/MULTI_GPU.md <- section 7.  Tables whose cells are paragraphs become headed paragraphs.  Usage: python tools/split_design.py"""
import os
import re
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W = 120


def wrap_block(lines):
    """re-flow one paragraph / list item; keeps the leading bullet and indents continuation lines"""
    text = " ".join(l.strip() for l in lines)
    m = re.match(r"^(\s*)([*-] |\d+\. )?", lines[0])
    indent = m.group(1) or ""
    bullet = m.group(2) or ""
    body = text[len(bullet):] if text.startswith(bullet) else text
    return textwrap.wrap(body, W, initial_indent=indent + bullet, subsequent_indent=indent + " " * len(bullet), break_long_words=False,
                         break_on_hyphens=False)


def table_to_paragraphs(rows):
    """a markdown table with paragraph-sized cells -> one headed paragraph per row"""
    cells = [[c.strip() for c in r.strip().strip("|").split("|")] for r in rows]
    head, body = cells[0], [c for c in cells[2:]]
    out = []
    for c in body:
        out.append("")
        out += textwrap.wrap("**%s**" % c[0], W, break_long_words=False, break_on_hyphens=False)
        for h, v in zip(head[1:], c[1:]):
            if v:
                out += textwrap.wrap("*%s:* %s" % (h, v), W, initial_indent="  ", subsequent_indent="  ", break_long_words=False, break_on_hyphens=False)
    return out


def reflow(md):
    out, para, table, code = [], [], [], False

    def flush():
        nonlocal para, table
        if para:
            out.extend(wrap_block(para))
            para = []
        if table:
            if max(len(r) for r in table) > 200:
                out.extend(table_to_paragraphs(table))
            else:
                out.extend(table)
            table = []

    for line in md.splitlines():
        if line.startswith("```"):
            flush()
            code = not code
            out.append(line)
        elif code:
            out.append(line)
        elif line.startswith("|"):
            if para:
                flush()
            table.append(line)
        elif not line.strip():
            flush()
            out.append("")
        elif line.startswith("#"):
            flush()
            out.append(line)
        elif re.match(r"^\s*([*-] |\d+\. )", line):
            flush()
            para = [line]
        else:
            if table:
                flush()
            para.append(line)
    flush()
    return "\n".join(out) + "\n"


def main():
    src = open(os.path.join(ROOT, "DESIGN.md")).read()
    parts = re.split(r"(?m)^(?=## )", src)
    by = {}
    head = parts[0]
    for p in parts[1:]:
        key = re.match(r"## (\S+)", p).group(1).rstrip(".")
        by[key] = p
    main_keys = ["1", "2", "3", "4", "5", "8", "9", "10"]
    docs = os.path.join(ROOT, "docs")
    os.makedirs(docs, exist_ok=True)
    six = by["6"]
    cut = six.index("### 6.1")
    results, experiments = six[:cut], six[cut:]
    pointer = ("## 6. Results\n\nNumbers, per round and per kernel: [docs/RESULTS.md](docs/RESULTS.md).  What was tried to raise them, lever by lever, with the "
               "measurement that decided each: [docs/EXPERIMENTS.md](docs/EXPERIMENTS.md).\n\n"
               "## 7. Multi-GPU\n\nThe stripe / tile decomposition, the exchange plan, the overlap and what has and has not met hardware: "
               "[docs/MULTI_GPU.md](docs/MULTI_GPU.md).\n\n")
    body = head + "".join(by[k] for k in ["1", "2", "3", "4", "5"]) + pointer + "".join(by[k] for k in ["8", "9", "10"])
    open(os.path.join(ROOT, "DESIGN.md"), "w").write(reflow(body))
    open(os.path.join(docs, "RESULTS.md"), "w").write(reflow("# Results (DESIGN.md section 6)\n\n" + results.replace("## 6. Results", "## Results")))
    open(os.path.join(docs, "EXPERIMENTS.md"), "w").write(reflow("# Experiments, round by round (DESIGN.md sections 6.1, 6.2, 9a, 9b)\n\n" + experiments + by.get("9a", "") + by.get("9b", "")))
    open(os.path.join(docs, "MULTI_GPU.md"), "w").write(reflow("# Multi-GPU (DESIGN.md section 7)\n\n" + by["7"]))
    for f in ("DESIGN.md", "docs/RESULTS.md", "docs/EXPERIMENTS.md", "docs/MULTI_GPU.md"):
        t = open(os.path.join(ROOT, f)).read().splitlines()
        print(f, len(t), "lines; longest", max(len(l) for l in t))


if __name__ == "__main__":
    main()
