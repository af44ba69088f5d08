#!/usr/bin/env python3
"""Splits sanitizer log files (ASAN/TSAN/UBSAN `log_path` output) into reports and says whose they are: a report with a frame in
libzkstark_emu* is the library's (or the emulator's); one whose frames lie only in liboracle.so / libgomp / numpy / torch is the test
infrastructure's (the CPU oracle's OpenMP loops are not instrumented, so TSan cannot see libgomp's barriers: known false positives).
    python tools/sanitizer_report_summary.py <log files...>"""
import re
import sys


def main(paths):
    text = ""
    for p in paths:
        try:
            text += open(p, errors="replace").read() + "\n"
        except OSError:
            pass
    blocks = re.split(r"(?m)^(?=WARNING: |==\d+==ERROR: |\S+:\d+:\d+: runtime error:)", text)
    blocks = [b for b in blocks if b.startswith(("WARNING: ", "==")) or "runtime error:" in b.split("\n", 1)[0]]
    ours = [b for b in blocks if "libzkstark_emu" in b or "zk_evm_amd/csrc" in b]
    print("# sanitizer reports: %d, of them with a frame in the library or the emulator: %d; the others lie wholly in liboracle.so / libgomp"
          " (the CPU oracle's uninstrumented OpenMP loops) or the Python runtime" % (len(blocks), len(ours)))
    for b in ours:
        print("\n".join(ln[:260] for ln in b.splitlines()[:40]))
    kinds = {}
    for b in blocks:
        if b in ours:
            continue
        m = re.search(r"#1 (\S+)", b)
        kinds[m.group(1) if m else "?"] = kinds.get(m.group(1) if m else "?", 0) + 1
    for k, n in sorted(kinds.items(), key=lambda kv: -kv[1])[:12]:
        print("#   not ours: %4d x %s" % (n, k))
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
