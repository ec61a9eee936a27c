#!/usr/bin/env python3
"""Freeze the experiment switches of a source file at their default values (a small `unifdef`).

The kernel files grew one `#if` per experiment (ablations, A/B variants, census builds); every one is a way to build a library no
test covers.  This tool rewrites a file to what the DEFAULT build's preprocessor sees:
  * `#ifndef X / #define X v / #endif` default blocks become a plain `#define X v` (dropped when X is no longer used);
  * every `#if / #ifdef / #ifndef / #elif / #else / #endif` whose condition only involves such X (or names given with
    --undefined) is resolved: the taken branch stays, the others go;
  * conditionals over anything else (compiler macros, --keep names) are left alone.
The result must compile to the SAME assembly as before (check: hipcc -S of both, `diff`).

usage: freeze_switches.py FILE [--keep A,B] [--undefined C,D] [--write]        (prints a report; --write replaces FILE)
"""
import re
import sys

DIRECTIVE = re.compile(r"^\s*#\s*(if|ifdef|ifndef|elif|else|endif)\b(.*)$")
DEFINE = re.compile(r"^\s*#\s*define\s+(\w+)\s*(.*)$")
IDENT = re.compile(r"[A-Za-z_]\w*")


def strip_comment(s):
    s = re.sub(r"/\*.*?\*/", " ", s)
    i = s.find("//")
    return (s[:i] if i >= 0 else s).strip()


def find_defaults(lines):
    """{X: (value text, first line, last line)} for `#ifndef X` / `#define X v` (possibly with comment lines) / `#endif` blocks."""
    out = {}
    i = 0
    while i < len(lines):
        m = DIRECTIVE.match(lines[i])
        if m and m.group(1) == "ifndef":
            name = strip_comment(m.group(2))
            j = i + 1
            d = DEFINE.match(lines[j]) if j < len(lines) else None
            if d and d.group(1) == name:
                k = j + 1
                while k < len(lines) and not DIRECTIVE.match(lines[k]) and (not lines[k].strip() or lines[k].strip().startswith("//")):
                    k += 1
                e = DIRECTIVE.match(lines[k]) if k < len(lines) else None
                if e and e.group(1) == "endif":
                    out[name] = (strip_comment(d.group(2)), i, k)
        i += 1
    return out


class Unknown(Exception):
    pass


def evaluate(expr, values, undefined, keep):
    """C preprocessor expression -> int, or Unknown if it involves a name we do not control."""
    e = strip_comment(expr)

    def defined(m):
        n = m.group(1)
        if n in keep:
            raise Unknown(n)
        if n in values:
            return "1"
        if n in undefined:
            return "0"
        raise Unknown(n)
    e = re.sub(r"defined\s*\(\s*(\w+)\s*\)", defined, e)
    e = re.sub(r"defined\s+(\w+)", defined, e)

    def ident(m):
        n = m.group(0)
        if n in keep:
            raise Unknown(n)
        if n in values:
            v = values[n]
            if not re.fullmatch(r"-?\d+", v):
                raise Unknown(n + "=" + v)
            return v
        if n in undefined:
            return "0"
        raise Unknown(n)
    e = IDENT.sub(ident, e)
    e = e.replace("&&", " and ").replace("||", " or ")
    e = re.sub(r"!(?!=)", " not ", e)
    if not re.fullmatch(r"[\d\s()<>=!&|+\-*andortn]+", e):
        raise Unknown(e)
    return int(bool(eval(e)))      # noqa: S307 - digits and operators only (checked above)


def freeze(text, keep, undefined):
    lines = text.split("\n")
    defaults = {k: v for k, v in find_defaults(lines).items() if k not in keep}
    values = {k: v[0] for k, v in defaults.items()}
    block_start = {v[1]: k for k, v in defaults.items()}
    out, report = [], []
    # stack entries: [state, any_taken, passthrough]  state: 'on' (emit), 'off' (skip); passthrough: directive lines are kept
    stack = []
    i = 0

    def emitting():
        return all(s[0] == "on" for s in stack)
    while i < len(lines):
        line = lines[i]
        if i in block_start and emitting():
            name = block_start[i]
            first, last = defaults[name][1], defaults[name][2]
            out.append(lines[first + 1])                     # the #define line (with its comment)
            out.extend(l for l in lines[first + 2:last] if l.strip())   # comment continuation lines
            endc = lines[last].split("//", 1)
            if len(endc) == 2 and endc[1].strip():
                out.append(" " * 29 + "//" + endc[1])       # a comment that trailed the #endif
            i = last + 1
            continue
        m = DIRECTIVE.match(line)
        if not m:
            if emitting():
                out.append(line)
            i += 1
            continue
        kind, rest = m.group(1), m.group(2)
        if kind in ("if", "ifdef", "ifndef"):
            if not emitting():
                stack.append(["off", True, False])
                i += 1
                continue
            try:
                if kind == "if":
                    v = evaluate(rest, values, undefined, keep)
                else:
                    n = strip_comment(rest)
                    if n in keep or (n not in values and n not in undefined):
                        raise Unknown(n)
                    v = int(n in values)
                    if kind == "ifndef":
                        v = 1 - v
                stack.append(["on" if v else "off", bool(v), False])
                report.append(f"  line {i + 1}: {line.strip()[:90]}  ->  {'taken' if v else 'dropped'}")
            except Unknown as u:
                stack.append(["on", True, True])
                out.append(line)
                report.append(f"  line {i + 1}: {line.strip()[:90]}  ->  KEPT (depends on {u})")
        elif kind == "elif":
            top = stack[-1]
            if top[2]:
                out.append(line)
            elif all(s[0] == "on" for s in stack[:-1]):
                if top[1]:
                    top[0] = "off"
                else:
                    try:
                        v = evaluate(rest, values, undefined, keep)
                    except Unknown as u:
                        raise SystemExit(f"line {i + 1}: #elif over {u} after resolved branches: resolve by hand")
                    top[0] = "on" if v else "off"
                    top[1] = bool(v)
        elif kind == "else":
            top = stack[-1]
            if top[2]:
                out.append(line)
            elif all(s[0] == "on" for s in stack[:-1]):
                top[0] = "off" if top[1] else "on"
                top[1] = True
        else:
            top = stack.pop()
            if top[2]:
                out.append(line)
        i += 1
    assert not stack, "unbalanced conditionals"
    text = "\n".join(out)
    # a frozen default nobody reads any more (it only steered #if lines) goes away
    for name in values:
        uses = len(re.findall(r"\b" + re.escape(name) + r"\b", text))
        if uses == 1:
            text = re.sub(r"^[ \t]*#[ \t]*define[ \t]+" + re.escape(name) + r"\b[^\n]*\n(?:[ \t]*//[^\n]*\n)*", "", text, flags=re.M)
            report.append(f"  #define {name}: no reader left, removed")
    return text, report


def main():
    path = sys.argv[1]
    arg = lambda f: set(sys.argv[sys.argv.index(f) + 1].split(",")) if f in sys.argv else set()
    text = open(path).read()
    new, report = freeze(text, arg("--keep"), arg("--undefined"))
    print(f"{path}: {len(DIRECTIVE.findall(text))} -> conditionals: "
          f"{sum(1 for l in text.split(chr(10)) if DIRECTIVE.match(l) and DIRECTIVE.match(l).group(1) in ('if', 'ifdef', 'ifndef'))} before, "
          f"{sum(1 for l in new.split(chr(10)) if DIRECTIVE.match(l) and DIRECTIVE.match(l).group(1) in ('if', 'ifdef', 'ifndef'))} after")
    print("\n".join(report))
    if "--write" in sys.argv:
        open(path, "w").write(new)


if __name__ == "__main__":
    main()
