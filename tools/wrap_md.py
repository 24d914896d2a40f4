"""Re-flows the prose of a Markdown file to lines of at most WIDTH characters (default 120) so that it can be read and diffed: paragraphs and list items are wrapped
(continuation lines of a list item are indented under its text), tables, fenced code blocks, headings and HTML are left alone.  python tools/wrap_md.py FILE [WIDTH]"""
import re, sys, textwrap
path = sys.argv[1]; W = int(sys.argv[2]) if len(sys.argv) > 2 else 120
out, para, in_code = [], [], False


def flush():
    global para
    if not para:
        return
    first = para[0]
    m = re.match(r"^(\s*)([-*+]|\d+\.)\s+", first)
    if m:
        ind = " " * len(m.group(0)); head = m.group(0)
        text = " ".join([first[len(m.group(0)):].strip()] + [l.strip() for l in para[1:]])
        lines = textwrap.wrap(text, W, initial_indent=head, subsequent_indent=ind, break_long_words=False, break_on_hyphens=False)
    else:
        ind = re.match(r"^\s*", first).group(0)
        text = " ".join(l.strip() for l in para)
        lines = textwrap.wrap(text, W, initial_indent=ind, subsequent_indent=ind, break_long_words=False, break_on_hyphens=False)
    out.extend(lines or [""]); para = []


for line in open(path).read().split("\n"):
    s = line.rstrip()
    if s.lstrip().startswith("```"):
        flush(); out.append(s); in_code = not in_code; continue
    if in_code or s.lstrip().startswith("|") or s.startswith("#") or s.lstrip().startswith("<") or s.strip() in ("---", "***"):
        flush(); out.append(s); continue
    if not s.strip():
        flush(); out.append(""); continue
    if re.match(r"^\s*([-*+]|\d+\.)\s+", s) and para:      # a new list item ends the previous one
        flush()
    para.append(s)
flush()
open(path, "w").write("\n".join(out).rstrip("\n") + "\n")
