#!/usr/bin/env python
"""tests/emu/chevrons.py -- TEST INFRASTRUCTURE: g++ does not parse `kernel<<<grid, block, shmem, stream>>>(args)`; this rewrites the launches
of a .hip source into the equivalent hipLaunchKernelGGL(...) calls (which the product also uses) for the CPU model's build.  The other thing g++ cannot take is
gfx950 inline assembly: a statement  asm("v_xxx %0, %1, ..." : "=v"(d) : "v"(a), ...);  becomes a call  emu_asm::v_xxx(d, a, ...)  of the
instruction's functional model in hip/hip_runtime.h, operands in the order of the template (so the PRODUCT text carries no test conditional).
Nothing else in the text changes.   chevrons.py in.hip out.cpp"""
import re
import sys


def split_top(s):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "([{<":
            depth += 1
        elif ch in ")]}>":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    out.append(cur.strip())
    return out


def rewrite(text):
    out = []
    pos = 0
    while True:
        i = text.find("<<<", pos)
        if i < 0:
            out.append(text[pos:])
            break
        # kernel expression: identifier (with namespaces) and an optional template argument list, backwards from i
        j = i
        while j > 0 and text[j - 1].isspace():
            j -= 1
        if text[j - 1] == ">":
            depth, k = 0, j - 1
            while True:
                if text[k] == ">":
                    depth += 1
                elif text[k] == "<":
                    depth -= 1
                    if depth == 0:
                        break
                k -= 1
            j = k
        k = j
        while k > 0 and (text[k - 1].isalnum() or text[k - 1] in "_:"):
            k -= 1
        kernel = text[k:i].strip()
        e = text.find(">>>", i)
        cfg = split_top(text[i + 3:e])
        while len(cfg) < 4:
            cfg.append("0" if len(cfg) == 2 else "nullptr")
        a = text.find("(", e)
        depth, b = 0, a
        while True:
            if text[b] == "(":
                depth += 1
            elif text[b] == ")":
                depth -= 1
                if depth == 0:
                    break
            b += 1
        args = text[a + 1:b].strip()
        out.append(text[pos:k])
        out.append("hipLaunchKernelGGL((%s), dim3(%s), dim3(%s), %s, %s%s)" % (kernel, cfg[0], cfg[1], cfg[2], cfg[3], (", " + args) if args else ""))
        pos = b + 1
    return "".join(out)


ASM_RE = re.compile(r'asm\(\s*"(v_[a-z0-9_]+)\s+([^"]*)"\s*:([^;]*?)\)\s*;')


def rewrite_asm(text):
    """single-instruction VALU asm statements -> emu_asm::<mnemonic>(operands in template order); the first operand is the destination"""
    def one(m):
        mnem, templ, rest = m.group(1), m.group(2), m.group(3)
        exprs = []
        for part in rest.split(":"):                      # outputs, then inputs; each a list of "constraint"(expr)
            for c in split_top(part):
                c = c.strip()
                if not c:
                    continue
                mm = re.match(r'"[^"]*"\s*\((.*)\)$', c, re.S)
                if not mm:
                    raise SystemExit("chevrons.py: cannot read asm operand %r" % c)
                exprs.append(mm.group(1).strip())
        ops = []
        for tok in templ.split(","):                      # operands in template order: %N, or an inline constant passed through
            tok = tok.strip()
            mm = re.match(r"%(\d+)$", tok)
            ops.append(exprs[int(mm.group(1))] if mm else tok)
        return "emu_asm::%s(%s);" % (mnem, ", ".join(ops))
    out = ASM_RE.sub(one, text)
    # an asm statement with an EMPTY template emits no instruction (a liveness hint to the device compiler: csrc keep_whole): dropped
    out = re.sub(r'asm\s*(volatile)?\s*\(\s*""\s*:[^;]*?\)\s*;', "", out)
    if re.search(r"\basm\s*(volatile)?\s*\(", out):
        raise SystemExit("chevrons.py: an asm statement was not understood")
    return out


if __name__ == "__main__":
    src = open(sys.argv[1]).read()
    open(sys.argv[2], "w").write('#line 1 "%s"\n' % sys.argv[1] + rewrite_asm(rewrite(src)))
