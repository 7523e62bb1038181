"""Print the sequence of memory / wait / matrix instructions of one kernel from a `hipcc -S --cuda-device-only` listing.

usage: isa_events.py LISTING.s SYMBOL_PREFIX
One line per basic block: L = buffer load, V<n> = s_waitcnt vmcnt(n), g<n> = lgkmcnt(n), w/r = LDS write/read,
M = MFMA, | = barrier, ^ = branch.  A wait in front of every load means the loads leave one round trip at a time.
"""
import re
import sys


def events(listing, prefix):
    i = listing.find("\n" + prefix)
    if i < 0:
        raise SystemExit("no symbol starting with " + prefix)
    body = listing[i:listing.find("s_endpgm", i)]
    lines, cur = [], ""
    for ln in body.split("\n"):
        t = ln.strip()
        if t.startswith(".LBB"):
            if cur.split(": ")[-1]:
                lines.append(cur)
            cur = t.split(":")[0] + ": "
        elif t.startswith("buffer_load") or t.startswith("global_load"):
            cur += "L"
        elif t.startswith("s_waitcnt"):
            m = re.search(r"vmcnt\((\d+)\)", t)
            if m:
                cur += "V" + m.group(1)
            m = re.search(r"lgkmcnt\((\d+)\)", t)
            if m:
                cur += "g" + m.group(1)
        elif t.startswith("ds_write"):
            cur += "w"
        elif t.startswith("ds_read"):
            cur += "r"
        elif "mfma" in t:
            cur += "M"
        elif t.startswith("s_barrier"):
            cur += "|"
        elif t.startswith("s_cbranch") or t.startswith("s_branch"):
            cur += "^"
    lines.append(cur)
    return lines


if __name__ == "__main__":
    print("\n".join(events(open(sys.argv[1]).read(), sys.argv[2])))
