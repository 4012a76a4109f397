"""Dependent VALU chains in the ISA of the library's kernels (dev tool, CPU: hipcc cross-compiles): runs of consecutive vector instructions in which each
one reads the destination the previous one wrote -- an instruction-level parallelism of one wherever hipcc scheduled independent work depth first
(round 5: FFN-1's GELU was 7 back-to-back dependent v_pk_fma_f32 per pair, the GEMM-shaped scan's column maxima 16 dependent v_max3 per column).
    python tools/isa_chains.py [min_len, default 6] [kernel substring ...]"""
import os, re, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import test_kernel_isa as t

min_len = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 6
subs = [a for a in sys.argv[1:] if not a.isdigit()]
reg = re.compile(r"v\[(\d+):(\d+)\]|v(\d+)")


def regs(tok):
    out = set()
    for m in reg.finditer(tok):
        if m.group(3) is not None:
            out.add(int(m.group(3)))
        else:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


for unit in ("atlas_hip", "encoder"):
    for name, body in t._functions(t._asm(unit)).items():
        if subs and not any(s in name for s in subs):
            continue
        ins = [l.strip() for l in body.split("\n") if l.strip().startswith("v_") and not l.strip().startswith("v_mfma")]
        # (consecutive VECTOR instructions only: scalar fillers between them do not break a chain)
        best, cur, prev_dst, start = [], 1, None, 0
        chains = []
        for i, l in enumerate(ins):
            parts = l.split(None, 1)
            ops = parts[1].split(",") if len(parts) > 1 else []
            dst = regs(ops[0]) if ops else set()
            src = set().union(*[regs(o) for o in ops[1:]]) if len(ops) > 1 else set()
            if prev_dst and (prev_dst & src):
                cur += 1
            else:
                if cur >= min_len:
                    chains.append((cur, ins[start].split()[0], ins[i - 1].split()[0]))
                cur, start = 1, i
            prev_dst = dst
        if chains:
            chains.sort(reverse=True)
            total = sum(c for c, _, _ in chains)
            print(f"{name[:70]:70s} {len(chains):4d} chains >= {min_len}, {total:5d} instructions in them; longest: " + ", ".join(f"{c} x {a}..{b}" for c, a, b in chains[:4]))
