"""Dev tool: outline of a kernel's ISA (from `hipcc -S --cuda-device-only`): runs of MFMA / LDS-DMA / scratch / barrier / store / wait / branch
instructions in program order, so that spills inside a k-loop or a drained pipeline show at a glance.
    python tools/isa_outline.py /tmp/enc.s gemm_pt_kernel F16Li2
"""
import re
import sys


def outline(path, *needles):
    s = open(path).read()
    funcs = re.split(r"\n(?=_Z\w+:)", s)
    for f in funcs:
        name = f.split(":", 1)[0]
        if not all(nd in name for nd in needles):
            continue
        lines = f.split("\n")
        ev = []
        for i, l in enumerate(lines):
            l = l.strip()
            if l.startswith("v_mfma"): ev.append("M")
            elif l.startswith("scratch_store"): ev.append("S")
            elif l.startswith("scratch_load"): ev.append("L")
            elif l.startswith("s_barrier"): ev.append("B")
            elif l.startswith("buffer_load") and " lds" in l: ev.append("D")
            elif l.startswith("global_load_lds"): ev.append("D")
            elif l.startswith("buffer_store") or l.startswith("global_store"): ev.append("W")
            elif l.startswith("buffer_load") or l.startswith("global_load"): ev.append("G")
            elif l.startswith("s_waitcnt") and "vmcnt" in l: ev.append("v(%s)" % re.search(r"vmcnt\((\d+)\)", l).group(1))
            elif l.startswith("s_cbranch") or l.startswith("s_branch"): ev.append("j")
            elif l.startswith("ds_read") or l.startswith("ds_load"): ev.append("r")
            elif l.startswith("ds_write") or l.startswith("ds_store"): ev.append("w")
            elif re.match(r"\.LBB\d+_\d+:", l): ev.append("|" + l.split(":")[0][4:])
        out, prev, cnt = [], None, 0
        for t in ev:
            if t == prev:
                cnt += 1
            else:
                if prev:
                    out.append(prev + (str(cnt) if cnt > 1 else ""))
                prev, cnt = t, 1
        out.append(prev + (str(cnt) if cnt > 1 else ""))
        m = re.search(r"\.vgpr_count:\s+(\d+)", f)
        print(name[:60], "lines", len(lines))
        print(" ".join(out))
        print()


if __name__ == "__main__":
    outline(sys.argv[1], *sys.argv[2:])
