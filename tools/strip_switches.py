"""A small `unifdef`: removes the preprocessor branches of build switches whose value is fixed, leaving every other conditional alone.

    python tools/strip_switches.py FILE... -DNAME=VALUE ... -UNAME ...

`#if` / `#elif` expressions that mention only fixed names (after `defined(X)` is resolved) are evaluated; `#ifdef` / `#ifndef` of fixed names
likewise.  A `#ifndef X / #define X v / #endif` block of a fixed name disappears with it.  Used once in round 6 to take the measured-and-
not-kept diagnostic switches out of heracles_amd/csrc (VERDICT r5 Weak #8); the removed variants are tools/patches/r05_switches.patch,
which `tools/build_variant.sh --patch` applies to a scratch copy of the sources."""
import re
import sys


def evaluate(expr, fixed):
    """value of a preprocessor expression, or None if it mentions anything that is not fixed"""
    e = re.sub(r"//.*$", "", expr).strip()
    e = re.sub(r"/\*.*?\*/", "", e)

    def dfn(m):
        n = m.group(1)
        if n not in fixed:
            raise KeyError(n)
        return "1" if fixed[n] is not None else "0"

    try:
        e = re.sub(r"defined\s*\(\s*(\w+)\s*\)", dfn, e)
        e = re.sub(r"defined\s+(\w+)", dfn, e)

        def val(m):
            n = m.group(0)
            if n in ("and", "or", "not"):
                return n
            if n not in fixed:
                raise KeyError(n)
            return "0" if fixed[n] is None else str(fixed[n])

        e = e.replace("&&", " and ").replace("||", " or ")
        e = re.sub(r"!(?!=)", " not ", e)
        e = re.sub(r"[A-Za-z_]\w*", val, e)
        return bool(eval(e, {"__builtins__": {}}, {}))  # noqa: S307 -- integers and operators only at this point
    except (KeyError, SyntaxError, NameError, TypeError):
        return None


def strip(text, fixed):
    out = []
    stack = []  # per open conditional: dict(kind='fixed'|'keep', emitting, taken)
    lines = text.split("\n")

    def emitting():
        return all(s["emit"] for s in stack)

    i = 0
    while i < len(lines):
        line = lines[i]
        m = re.match(r"\s*#\s*(if|ifdef|ifndef|elif|else|endif)\b(.*)", line)
        if not m:
            if emitting():
                out.append(line)
            i += 1
            continue
        kw, rest = m.group(1), m.group(2)
        if kw in ("if", "ifdef", "ifndef"):
            if kw == "if":
                v = evaluate(rest, fixed)
            else:
                name = rest.split()[0]
                v = None if name not in fixed else ((fixed[name] is not None) == (kw == "ifdef"))
                # `#ifndef X / #define X v / #endif`: the default of a fixed switch goes away entirely
            if v is None:
                stack.append({"kind": "keep", "emit": True, "outer": emitting()})
                if emitting():
                    out.append(line)
            else:
                stack.append({"kind": "fixed", "emit": v, "taken": v, "outer": emitting()})
        elif kw == "elif":
            top = stack[-1]
            if top["kind"] == "keep":
                if emitting():
                    out.append(line)
            else:
                if top["taken"]:
                    top["emit"] = False
                else:
                    v = evaluate(rest, fixed)
                    if v is None:
                        raise SystemExit(f"cannot evaluate #elif after a fixed #if: {line}")
                    top["emit"] = v
                    top["taken"] = v
        elif kw == "else":
            top = stack[-1]
            if top["kind"] == "keep":
                if emitting():
                    out.append(line)
            else:
                top["emit"] = not top["taken"]
        else:  # endif
            top = stack.pop()
            if top["kind"] == "keep" and emitting():
                out.append(line)
        i += 1
    if stack:
        raise SystemExit("unbalanced conditionals")
    text = "\n".join(out)
    # drop `#define X v` lines of fixed names that were defaults inside a removed #ifndef
    for n in fixed:
        text = re.sub(rf"^[ \t]*#[ \t]*define[ \t]+{n}\b.*\n", "", text, flags=re.M)
    return text


def main():
    files, fixed = [], {}
    for a in sys.argv[1:]:
        if a.startswith("-D"):
            n, _, v = a[2:].partition("=")
            fixed[n] = int(v) if v else 1
        elif a.startswith("-U"):
            fixed[a[2:]] = None
        else:
            files.append(a)
    for f in files:
        src = open(f).read()
        new = strip(src, fixed)
        if new != src:
            open(f, "w").write(new)
            print(f"{f}: {src.count(chr(10)) - new.count(chr(10))} lines removed")


if __name__ == "__main__":
    main()
