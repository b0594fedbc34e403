"""Removes the timing-ablation / probe switches from the shipped kernel sources: every preprocessor
conditional whose condition mentions only LAB macros is resolved with those macros UNDEFINED, everything else is left alone.

    python tools/ablate/strip_lab.py viquae_amd/csrc/knn.hip ...      rewrites the files in place

The lab build is kept as a patch: `tools/ablate/lab_switches.patch` (apply with `patch -p1 < tools/ablate/lab_switches.patch`,
then tools/ab_build.sh <name> "-DMQ_ABL_..." as before); the two rejected probe kernels live next to it."""
import re
import sys

LAB = re.compile(r"^(MQ_ABL_\w+|MQ_ABLATE_\w+|MQ_PROBE\w*|MQ_SCREEN_TOP_BARRIER|MQ_SCREEN_F16|MQ_X_DMA_NT|MQ_GEMM_ABL_\w+|MQ_SYNTH_EPI)$")


def lab_only(expr):
    names = re.findall(r"[A-Za-z_]\w*", expr)
    names = [n for n in names if n != "defined"]
    return bool(names) and all(LAB.match(n) for n in names)


def evaluate(kind, expr):
    """Truth of the directive with every lab macro undefined (an undefined macro is 0 in #if arithmetic)."""
    if kind == "ifdef":
        return False
    if kind == "ifndef":
        return True
    e = re.sub(r"defined\s*\(\s*\w+\s*\)|defined\s+\w+", "0", expr)
    e = re.sub(r"[A-Za-z_]\w*", "0", e).replace("||", " or ").replace("&&", " and ").replace("!", " not ")
    return bool(eval(e, {}, {}))  # noqa: S307 - digits and operators only


def strip(lines):
    out = []
    stack = []  # entries: None (foreign conditional) or dict(state="live"|"dead"|"done")
    for line in lines:
        m = re.match(r"^\s*#\s*(ifdef|ifndef|if|elif|else|endif)\b(.*)$", line)
        dead = any(s is not None and s["state"] != "live" for s in stack)
        if not m:
            if not dead:
                out.append(line)
            continue
        kind, rest = m.group(1), m.group(2).split("//")[0].strip()
        if kind in ("ifdef", "ifndef", "if"):
            if lab_only(rest):
                stack.append({"state": "live" if evaluate(kind, rest) else "dead"})
            else:
                stack.append(None)
                if not dead:
                    out.append(line)
        elif kind == "elif":
            top = stack[-1]
            if top is None:
                if not dead:
                    out.append(line)
            elif top["state"] == "live":
                top["state"] = "done"
            elif top["state"] == "dead":
                if not lab_only(rest):
                    raise SystemExit(f"cannot resolve mixed #elif: {line.strip()}")
                top["state"] = "live" if evaluate("if", rest) else "dead"
        elif kind == "else":
            top = stack[-1]
            if top is None:
                if not dead:
                    out.append(line)
            else:
                top["state"] = "live" if top["state"] == "dead" else "done"
        else:  # endif
            top = stack.pop()
            if top is None and not any(s is not None and s["state"] != "live" for s in stack):
                out.append(line)
    if stack:
        raise SystemExit("unbalanced conditionals")
    return out


if __name__ == "__main__":
    for path in sys.argv[1:]:
        with open(path) as f:
            src = f.readlines()
        new = strip(src)
        if new != src:
            with open(path, "w") as f:
                f.writelines(new)
            print(f"{path}: {len(src)} -> {len(new)} lines")
