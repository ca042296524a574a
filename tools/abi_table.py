#!/usr/bin/env python3
"""Markdown table of every export of include/surfd_hip.h: the reference interface named in the header comment above it (or in
its section banner) and the Python wrapper(s) of surfd_amd/ that bind it.  `python tools/abi_table.py` prints the table that
INTEGRATION.md carries as its appendix; tests/test_abi_cpu.py checks that the appendix lists every export."""
import glob, os, re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CITE = re.compile(r"[A-Za-z_][\w/]*\.(?:pyx|py)\b(?::[\d\-,\s]+\d)?")


def exports():
    """[(name, citations of its section banner, citations of the comment right above it)] in header order"""
    text = open(os.path.join(ROOT, "include", "surfd_hip.h")).read()
    out, section, comment, in_banner, banner = [], [], [], False, []
    pos = 0
    tok = re.compile(r"/\*(.*?)\*/|\b(?:int64_t|int|void|const char \*|long long)\s*\*?\s*(surfd_[a-z0-9_]+)\s*\(|;", re.S)
    for m in tok.finditer(text):
        if m.group(1) is not None:
            c = m.group(1)
            if re.search(r"-{20,}", c):                # a dashed line opens or closes a section banner
                if in_banner:
                    section = banner; in_banner = False
                else:
                    in_banner, banner = True, []
                comment = []
            elif in_banner:
                banner += CITE.findall(c)
            else:
                comment = CITE.findall(c)
        elif m.group(2) is not None:
            out.append((m.group(2), list(section), list(comment)))
        else:
            comment = [] if False else comment          # a declaration's own comment also covers the overloads that follow it
    return out


def wrappers():
    use = {}
    for f in sorted(glob.glob(os.path.join(ROOT, "surfd_amd", "*.py"))):
        src = open(f).read()
        for n in set(re.findall(r"\b(surfd_[a-z0-9_]+)\b", src)):
            if os.path.basename(f) != "_native.py":
                use.setdefault(n, []).append(os.path.basename(f))
    return use


def table():
    use = wrappers()
    rows = ["| export | reference interface it stands for (header comment) | bound by |", "|---|---|---|"]
    for name, sec, own in exports():
        cites = [c for c in (own or sec) if not c.startswith(("tests/", "bench.py", "tools/"))] or ([] if own else sec)
        seen, cs = set(), []
        for c in cites:
            if c not in seen:
                seen.add(c); cs.append(c)
        rows.append(f"| `{name}` | {', '.join('`%s`' % c for c in cs[:4]) or 'no counterpart (library housekeeping / measurement)'} | "
                    f"{', '.join('`surfd_amd/%s`' % w for w in use.get(name, [])) or '`surfd_amd/_native.py` (signature only)'} |")
    return "\n".join(rows)


if __name__ == "__main__":
    print(table())
