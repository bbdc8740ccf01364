#!/usr/bin/env python3
"""unifdef_lite - resolve #if / #ifdef blocks whose condition is built ONLY from the given macros, drop the
`#ifndef X / #define X v / #endif` default blocks of those macros, and replace their remaining uses by the value.
Used once in round 6 to prune the experiment switches that lost.
    tools/unifdef_lite.py FILE NAME=VALUE [NAME=VALUE ...]        (rewrites FILE in place)"""
import re
import sys


def evaluate(expr, defs):
    e = re.sub(r"//.*$", "", expr).strip()
    e = re.sub(r"defined\s*\(\s*(\w+)\s*\)", lambda m: "1" if m.group(1) in defs else "UNKNOWN_" + m.group(1), e)
    ids = set(re.findall(r"[A-Za-z_]\w*", e))
    if not ids <= set(defs):
        return None
    for k, v in defs.items():
        e = re.sub(r"\b%s\b" % k, str(v), e)
    e = e.replace("&&", " and ").replace("||", " or ")
    e = re.sub(r"!(?!=)", " not ", e)
    return bool(eval(e))


def process(lines, defs):
    out = []
    stack = []          # entries: dict(known, taken, emitting_before, any_taken)
    i = 0
    emitting = True
    while i < len(lines):
        ln = lines[i]
        s = ln.strip()
        m = re.match(r"#\s*(if|ifdef|ifndef|elif|else|endif)\b(.*)", s)
        if not m:
            if emitting:
                out.append(ln)
            i += 1
            continue
        kind, rest = m.group(1), m.group(2)
        if kind in ("if", "ifdef", "ifndef"):
            # the default block of a pruned macro: #ifndef X \n #define X ... \n #endif
            if kind == "ifndef" and rest.split()[0] in defs and i + 2 < len(lines) and \
                    re.match(r"\s*#\s*define\s+%s\b" % rest.split()[0], lines[i + 1]) and lines[i + 2].strip().startswith("#endif"):
                i += 3
                continue
            if kind == "if":
                val = evaluate(rest, defs)
            else:
                name = rest.split()[0]
                val = None if name not in defs else (kind == "ifdef")
            stack.append({"known": val is not None, "outer": emitting, "any": bool(val), "cur": bool(val)})
            if val is None:
                if emitting:
                    out.append(ln)
            else:
                emitting = emitting and val
        elif kind == "elif":
            top = stack[-1]
            if not top["known"]:
                if top["outer"]:
                    out.append(ln)
            else:
                val = evaluate(rest, defs)
                if val is None:
                    raise SystemExit("mixed #elif at line %d" % (i + 1))
                take = val and not top["any"]
                top["any"] = top["any"] or take
                emitting = top["outer"] and take
        elif kind == "else":
            top = stack[-1]
            if not top["known"]:
                if top["outer"]:
                    out.append(ln)
            else:
                emitting = top["outer"] and not top["any"]
                top["any"] = True
        else:
            top = stack.pop()
            if not top["known"]:
                if top["outer"]:
                    out.append(ln)
            emitting = top["outer"]
        i += 1
    return out


def main():
    path = sys.argv[1]
    defs = {}
    for a in sys.argv[2:]:
        k, v = a.split("=")
        defs[k] = int(v)
    lines = open(path).read().split("\n")
    out = process(lines, defs)
    text = "\n".join(out)
    for k, v in defs.items():
        text = re.sub(r"\b%s\b(?!\w)" % k, "%%%%%s%%%%" % k, text)      # mark the remaining uses: resolved by hand
    open(path, "w").write(text)
    n = sum(text.count("%%" + k + "%%") for k in defs)
    print(path, "lines", len(lines), "->", len(out), "remaining uses marked:", n)


if __name__ == "__main__":
    main()
