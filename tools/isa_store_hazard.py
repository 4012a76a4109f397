"""Dev tool: scan a hipcc -S dump for `buffer_store_dwordx3/x4 ... <SGPR soffset>` whose data VGPRs are written by the very next
instruction (the sequence hipcc's hazard recognizer does not pad, and that corrupted stores on gfx950: tools/hazard_probe.hip).
    python tools/isa_store_hazard.py /tmp/enc.s"""
import re, sys

cur, prev, n = None, None, 0
for line in open(sys.argv[1]):
    t = line.strip()
    if re.match(r"^_Z\w+:", t):
        cur = t.split(":")[0]
        prev = None
        continue
    if not t or t.startswith(";") or t.startswith("."):
        continue
    if prev:
        m = re.match(r"buffer_store_dwordx[34] v\[(\d+):(\d+)\], \w+, s\[\d+:\d+\], (s\d+|m0)\b", prev)
        if m and re.match(r"v_", t):
            lo, hi = int(m.group(1)), int(m.group(2))
            d = re.match(r"v_\w+ v(\d+)|v_\w+ v\[(\d+):(\d+)\]", t)
            if d:
                regs = range(int(d.group(1)), int(d.group(1)) + 1) if d.group(1) else range(int(d.group(2)), int(d.group(3)) + 1)
                if any(lo <= r <= hi for r in regs):
                    n += 1
                    print(cur[:70], "|", prev, "->", t)
    prev = t
print(n, "unpadded store-data overwrites")
