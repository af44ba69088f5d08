"""TEST INFRASTRUCTURE.  Source-to-source step of the CPU emulation build (tests/emu/build_emu.py): the two HIP constructs a plain
C++ compiler cannot parse are rewritten, nothing else is touched --

    kernel<T, 7><<<grid, block, lds, stream>>>(a, b)   ->   HIPEMU_LAUNCH((kernel<T, 7>), (grid), (block), (lds), (stream), a, b)
    extern __shared__ [attr] u64 tile[];                ->   u64 *tile = (u64 *)hipemu::dyn_lds();

so that what runs on the CPU is the library's own kernels and host code (tests/emu/hipemu/hip/hip_runtime.h supplies the rest)."""
import re
import sys

_EXTERN_SHARED = re.compile(r"extern\s+__shared__\s+(?:__attribute__\(\(aligned\(\d+\)\)\)\s+)?([A-Za-z_][\w:]*)\s+(\w+)\s*\[\s*\]\s*;")


def _match_forward(s, i, open_ch, close_ch):
    """s[i] == open_ch: index just past its partner."""
    depth = 0
    while i < len(s):
        c = s[i]
        if c == open_ch:
            depth += 1
        elif c == close_ch:
            depth -= 1
            if depth == 0:
                return i + 1
        elif c == '"':                                   # string literal
            i += 1
            while s[i] != '"':
                i += 2 if s[i] == "\\" else 1
        elif c == "'" and i + 2 < len(s) and (s[i + 2] == "'" or (s[i + 1] == "\\" and s[i + 3] == "'")):
            i += 3 if s[i + 2] == "'" else 4
            continue
        i += 1
    raise ValueError("unbalanced %s" % open_ch)


def _split_top(s):
    parts, depth, cur = [], 0, ""
    for c in s:
        if c in "([{":
            depth += 1
        elif c in ")]}":
            depth -= 1
        if c == "," and depth == 0:
            parts.append(cur.strip())
            cur = ""
        else:
            cur += c
    parts.append(cur.strip())
    return parts


def _kernel_start(s, end):
    """s[:end] ends with the kernel expression (identifier, optional template arguments): its first index."""
    i = end
    while i > 0 and s[i - 1].isspace():
        i -= 1
    if i > 0 and s[i - 1] == ">":                        # template arguments: back to the matching '<'
        depth = 0
        while i > 0:
            i -= 1
            if s[i] == ">":
                depth += 1
            elif s[i] == "<":
                depth -= 1
                if depth == 0:
                    break
        while i > 0 and s[i - 1].isspace():
            i -= 1
    j = i
    while j > 0 and (s[j - 1].isalnum() or s[j - 1] in "_:"):
        j -= 1
    if j == i:
        raise ValueError("no kernel name before <<< near: %r" % s[max(0, end - 80):end])
    return j


def translate(src: str) -> str:
    out, pos = [], 0
    while True:
        at = src.find("<<<", pos)
        if at < 0:
            break
        line_start = src.rfind("\n", 0, at) + 1
        if "//" in src[line_start:at]:                   # inside a line comment
            out.append(src[pos:at + 3])
            pos = at + 3
            continue
        k0 = _kernel_start(src, at)
        close = src.find(">>>", at)
        cfg = _split_top(src[at + 3:close])
        if not 2 <= len(cfg) <= 4:
            raise ValueError("launch configuration with %d arguments: %r" % (len(cfg), src[at:close + 3]))
        cfg += ["0"] * (4 - len(cfg))
        p = close + 3
        while src[p].isspace():
            p += 1
        if src[p] != "(":
            raise ValueError("no argument list after >>> near %r" % src[at:p + 20])
        end = _match_forward(src, p, "(", ")")
        args = src[p + 1:end - 1].strip()
        out.append(src[pos:k0])
        out.append("HIPEMU_LAUNCH((%s), (%s), (%s), (%s), (%s)%s)" % (src[k0:at].strip(), cfg[0], cfg[1], cfg[2], cfg[3], ", " + args if args else ""))
        pos = end
    out.append(src[pos:])
    s = "".join(out)
    return _EXTERN_SHARED.sub(lambda m: "%s *%s = (%s *)hipemu::dyn_lds();" % (m.group(1), m.group(2), m.group(1)), s)


if __name__ == "__main__":
    sys.stdout.write(translate(open(sys.argv[1]).read()))
