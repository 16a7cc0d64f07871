#!/usr/bin/env python3
"""Wraps source lines longer than 140 columns (host C++ of libiqgpu): a trailing comment moves to a line of its own above
the code; what is still too long breaks behind the last `; `, `, `, ` && `, ` || ` or ` ? ` in front of column 138 that is not
inside a string literal.  usage: tools/wrap140.py file..."""
import re
import sys

LIM = 140


def outside_string(s, pos):
    q = False
    i = 0
    while i < pos:
        if s[i] == '\\':
            i += 2
            continue
        if s[i] == '"':
            q = not q
        i += 1
    return not q


def split_comment(line):
    i = 0
    while True:
        j = line.find('//', i)
        if j < 0:
            return line, None
        if outside_string(line, j):
            return line[:j].rstrip(), line[j:]
        i = j + 2


def wrap(line):
    if len(line) <= LIM or line.lstrip().startswith(('#', '//')):
        return [line]
    ind = len(line) - len(line.lstrip())
    code, com = split_comment(line)
    out = []
    if com is not None and code.strip():
        out.append(' ' * ind + com)
        line = code
    while len(line) > LIM:
        best = -1
        for tok in ('; ', ', ', ' && ', ' || ', ' ? ', ' : ', ' = ', ' << '):
            k = line.rfind(tok, ind + 20, LIM - 2)
            while k >= 0 and not outside_string(line, k):
                k = line.rfind(tok, ind + 20, k)
            if k >= 0:
                k += len(tok.rstrip()) if tok.startswith((';', ',')) else 0
                best = max(best, k)
        if best < 0:
            break
        out.append(line[:best].rstrip())
        line = ' ' * (ind + 4) + line[best:].lstrip()
    out.append(line)
    return out


for path in sys.argv[1:]:
    src = open(path).read().split('\n')
    res = []
    for l in src:
        res.extend(wrap(l))
    open(path, 'w').write('\n'.join(res))
    left = [i + 1 for i, l in enumerate(res) if len(l) > LIM]
    print(path, "lines now", len(res), "still long:", left)
