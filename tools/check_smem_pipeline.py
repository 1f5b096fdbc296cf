"""Safety check for the hand-pipelined scalar loads of k_me16b / k_me8b (me_bits.hpp: splanes_issue / splanes_wait).

The kernel requests a frame's planes with s_load_dwordx16 / x8 from an inline-asm statement and collects
them with a hand-placed `s_waitcnt lgkmcnt(0)` a frame later.  The compiler does not know the registers are
in flight in between: a copy or a spill of them there would read garbage.  This script walks hipcc's
assembly of the kernel (straight-line order, which is the order of the unrolled frame loop) and fails if any
instruction between such a request and the next `s_waitcnt lgkmcnt(0)` reads or writes one of the requested
registers, or if a requested tuple is never waited for before the block ends in something other than the
loop's own control flow.

usage: python tools/check_smem_pipeline.py file.s kernel-name-pattern [...]     (exit status 1 on a finding)"""
import re
import sys


def sregs(tok):
    """SGPR numbers named by an operand token: s5, s[4:7]"""
    m = re.fullmatch(r"s\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"s(\d+)", tok)
    return {int(m.group(1))} if m else set()


def check(lines, pat):
    start = next((i for i, l in enumerate(lines) if re.match(r"^_Z\w*%s\w*:" % pat, l)), None)
    if start is None:
        return ["kernel %s not found" % pat], 0
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    inflight = {}      # sgpr -> line number of the request
    problems, requests = [], 0
    in_asm = False
    for i in range(start + 1, end):
        s = lines[i].strip()
        if "#ASMSTART" in s:
            in_asm = True
            continue
        if "#ASMEND" in s:
            in_asm = False
            continue
        if not s or s.startswith(";") or s.startswith("."):
            continue
        op, _, rest = s.partition(" ")
        rest = rest.split(";")[0]
        toks = [t.strip() for t in re.split(r",\s*(?![^\[]*\])", rest) if t.strip()]
        if in_asm and op.startswith("s_load_dword"):
            dst = sregs(toks[0])
            clash = dst & set(inflight)
            if clash:
                problems.append("line %d: request overwrites registers still in flight: %s" % (i + 1, s))
            for t in toks[1:]:
                if sregs(t) & set(inflight):
                    problems.append("line %d: request reads a register in flight: %s" % (i + 1, s))
            for r in dst:
                inflight[r] = i + 1
            requests += 1
            continue
        if op == "s_waitcnt" and "lgkmcnt(0)" in rest:
            inflight.clear()
            continue
        if inflight:
            used = set()
            for t in toks:
                used |= sregs(t)
            hit = used & set(inflight)
            if hit:
                problems.append("line %d: %s touches s%s requested at line %d and not yet waited for"
                                % (i + 1, s, sorted(hit), inflight[min(hit)]))
    return problems, requests


def main():
    lines = open(sys.argv[1]).read().split("\n")
    bad = 0
    for pat in sys.argv[2:]:
        problems, n = check(lines, pat)
        for p in problems:
            print("%s: %s" % (pat, p))
        print("%s: %d hand-issued scalar loads checked, %d problems" % (pat, n, len(problems)))
        bad += len(problems) + (n == 0)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
