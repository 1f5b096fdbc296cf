"""Per-basic-block instruction counts of one kernel in hipcc's assembly (-S --cuda-device-only):
   python tools/isa_blocks.py file.s k_me16b   -> block, total, VALU, v_bitop3, lane spills, waits, scratch"""
import re
import sys

path, pat = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*%s\w*:" % pat, l))
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
lab, order, st = "entry", ["entry"], {}
for l in lines[start + 1:end + 1]:
    m = re.match(r"^(\.LBB\d+_\d+):\s*(;.*)?", l)
    if m:
        lab = m.group(1)
        order.append(lab)
        st.setdefault(lab, {})["hdr"] = (m.group(2) or "").strip()
        continue
    s = l.strip()
    if not re.match(r"^(v_|s_|ds_|global_|buffer_|flat_|scratch_)", s):
        continue
    d = st.setdefault(lab, {})
    d["n"] = d.get("n", 0) + 1
    if s.startswith("v_"): d["valu"] = d.get("valu", 0) + 1
    if "v_bitop3" in s: d["bitop"] = d.get("bitop", 0) + 1
    if "lane_b32" in s: d["lane"] = d.get("lane", 0) + 1
    if s.startswith("s_waitcnt"): d["wait"] = d.get("wait", 0) + 1
    if s.startswith("scratch_"): d["scr"] = d.get("scr", 0) + 1
tot = {}
for lb in order:
    d = st.get(lb, {})
    if d.get("n", 0) >= int(sys.argv[3]) if len(sys.argv) > 3 else d.get("n", 0) >= 6:
        print("%-12s n=%3d valu=%3d bitop=%2d lane=%2d wait=%d scr=%d  %s" % (lb, d["n"], d.get("valu", 0), d.get("bitop", 0), d.get("lane", 0), d.get("wait", 0), d.get("scr", 0), d.get("hdr", "")))
    for k in ("n", "valu", "bitop", "lane", "wait", "scr"):
        tot[k] = tot.get(k, 0) + d.get(k, 0)
print("static totals:", tot)
