"""Safety check for the hand-pipelined scalar loads of k_me16b (me_bits.hpp: splanes_issue / splanes_wait).

The kernel requests a frame's planes with s_load_dwordx16 / x8 from an inline-asm statement and collects
them with a hand-placed `s_waitcnt lgkmcnt(0)` a frame later.  The compiler does not know the registers are
in flight in between: a copy or a spill of them there would read garbage (it knows they are LIVE, so it
never hands them to another value -- what it may do is move them).  This script reads hipcc's assembly of
the kernel, builds the control-flow graph (labels, s_branch / s_cbranch_*), propagates the set of SGPRs with a
hand-issued load outstanding forward through it (union over predecessors; any `s_waitcnt lgkmcnt(0)` empties
it) and fails if an instruction touches a register of that set.

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


def parse(lines, start, end):
    """-> blocks: list of dicts {label, insts: [(lineno, op, toks, in_asm)], succ: [labels], fall: bool}"""
    blocks = [{"label": "entry", "insts": [], "succ": [], "fall": True}]
    in_asm = False
    for i in range(start + 1, end):
        raw = lines[i]
        m = re.match(r"^(\.LBB\d+_\d+):", raw)
        if m:
            blocks.append({"label": m.group(1), "insts": [], "succ": [], "fall": True})
            continue
        s = raw.strip()
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
        b = blocks[-1]
        if b.get("closed"):     # code behind a branch without a label of its own: a new block
            blocks.append({"label": "anon%d" % i, "insts": [], "succ": [], "fall": True})
            b = blocks[-1]
        b["insts"].append((i + 1, op, toks, in_asm))
        if op == "s_branch":
            b["succ"].append(toks[0])
            b["fall"] = False
            b["closed"] = True
        elif op.startswith("s_cbranch"):    # ends the block: the target and the fall-through both follow
            b["succ"].append(toks[-1])
            b["closed"] = True
        elif op == "s_endpgm":
            b["fall"] = False
            b["closed"] = True
    return blocks


def check(lines, pat):
    start = next((i for i, l in enumerate(lines) if re.match(r"^_Z\w*%s\w*:" % pat, l)), None)
    if start is None:
        return ["kernel %s not found" % pat], 0
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    blocks = parse(lines, start, end)
    index = {b["label"]: k for k, b in enumerate(blocks)}
    succ = []
    for k, b in enumerate(blocks):
        s = [index[l] for l in b["succ"] if l in index]
        if b["fall"] and k + 1 < len(blocks):
            s.append(k + 1)
        succ.append(s)
    entry_set = [dict() for _ in blocks]       # sgpr -> line of the request, at block entry
    problems, requests = {}, 0

    def run(k, state, report):
        nonlocal requests
        st = dict(state)
        for (ln, op, toks, in_asm) in blocks[k]["insts"]:
            if in_asm and op.startswith("s_load_dword"):
                dst = sregs(toks[0])
                used = set()
                for t in toks[1:]:
                    used |= sregs(t)
                if report and (dst | used) & set(st):
                    problems[ln] = "line %d: request touches registers still in flight: %s %s" % (ln, op, ", ".join(toks))
                for r in dst:
                    st[r] = ln
                if report:
                    requests += 1
                continue
            if op == "s_waitcnt" and any("lgkmcnt(0)" in t for t in toks):
                st.clear()
                continue
            if st:
                used = set()
                for t in toks:
                    used |= sregs(t)
                hit = used & set(st)
                if hit and report:
                    problems[ln] = ("line %d: %s %s touches s%s requested at line %d and not yet waited for"
                                    % (ln, op, ", ".join(toks), sorted(hit), st[min(hit)]))
        return st

    work = [0]
    seen = {0}
    while work:                                 # forward dataflow to a fixed point (sets only grow)
        k = work.pop()
        out = run(k, entry_set[k], False)
        for j in succ[k]:
            grown = False
            for r, ln in out.items():
                if r not in entry_set[j]:
                    entry_set[j][r] = ln
                    grown = True
            if grown or j not in seen:
                seen.add(j)
                work.append(j)
    for k in sorted(seen):
        run(k, entry_set[k], True)
    return [problems[k] for k in sorted(problems)], requests


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
