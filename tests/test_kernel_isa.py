"""What the compiled hot kernels look like (CPU: hipcc cross-compiles gfx950 here; ~1 minute): properties of the ISA that measurements
depend on and that a compiler or source change can silently break.

  * no scratch (register spills) in the steady state: the persistent refresh GEMMs use none at all; nor do the scan kernels, at the 128-VGPR cap of their
    16-wave workgroups (a scratch access counts in vmcnt and would drain the ring of slab loads; and a private segment alone costs
    ~12 us per launch);
  * no 16-byte buffer store whose data registers the very next instruction overwrites while its soffset is an SGPR: hipcc's hazard
    recognizer does not pad that sequence and gfx950 corrupts the store (measured: tools/hazard_probe.hip, profiles/r03/hazard_probe.txt).
"""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "atlas_amd", "csrc")


def _asm(name):
    """device ISA of csrc/<name>.hip with the product build's flags (atlas_amd/build.py), cached by source mtime"""
    from atlas_amd import build

    src = os.path.join(CSRC, name + ".hip")
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(ROOT, "include", "atlas_hip.h")]
    out = os.path.join("/tmp", f"atlas_isa_{name}_{int(max(os.path.getmtime(d) for d in deps))}.s")
    if not os.path.exists(out):
        subprocess.check_call([build._hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-w",
                               "-mllvm", "-amdgpu-mfma-vgpr-form=1", src, "-o", out])
    return open(out).read()


def _functions(asm):
    return {f.split(":", 1)[0]: f for f in re.split(r"\n(?=_Z\w+:)", asm) if f.startswith("_Z")}


def _scratch_bytes(body):
    return int(re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", body).group(1))


def test_persistent_gemm_has_no_scratch_and_a_clean_loop():
    fns = {k: v for k, v in _functions(_asm("encoder")).items() if "gemm_pt_kernel" in k}
    assert len(fns) == 6, sorted(fns)                        # fp16 / bf16 x EPI 1, 2, 3 (EPI 4, the V^T epilogue of rounds 3-4, lives on in the tuning build: cfg 10)
    for name, body in fns.items():
        assert _scratch_bytes(body) == 0, f"{name} spills {_scratch_bytes(body)} bytes"
        assert "scratch_" not in body, name
        # Round 5 (ATLAS_PT_RSPLIT): three flavours of a k-tile -- a tile's first, middle and last iteration -- of 64 MFMAs each. The k-step-0 half
        # (32 MFMAs) is one uninterrupted run in every flavour and the last iteration's 64 are; in the other two the k-step-1 half carries the
        # twelve fragment reads of the NEXT k-tile (one W read in front of each row of four MFMAs) and exactly one wait on the vector-memory
        # counter: group A's vmcnt(4) in front of its four activation reads. Nothing else may drain the LDS-DMA pipeline mid-tile.
        assert body.count("v_mfma") == 3 * 64, (name, body.count("v_mfma"))
        lines = [l.strip() for l in body.split("\n") if l.strip() and not l.strip().startswith((";", "."))]
        mf = [i for i, l in enumerate(lines) if l.startswith("v_mfma")]
        inside = [lines[a + 1: b] for a, b in zip(mf, mf[1:]) if 0 < b - a - 1 <= 8]          # what sits BETWEEN the MFMAs of a multiply phase
        flat = [l for g in inside for l in g]
        nread = sum(l.startswith("ds_read_b128") for l in flat)                                # two flavours x twelve reads ahead (hipcc rotates the
        assert 20 <= nread <= 24, (name, nread, flat)                                          #  loop: a read or two may open the block behind the MFMAs)
        assert set(l for l in flat if l.startswith("s_waitcnt")) == {"s_waitcnt vmcnt(4)"}, (name, flat)
        assert not [l for l in flat if l.startswith(("buffer_", "global_", "scratch_", "s_barrier"))], (name, flat)


def test_scan_has_no_scratch_and_a_clean_loop():
    fns = {k: v for k, v in _functions(_asm("atlas_hip")).items() if "scan_kernelILi16ELi1ELi8E" in k}
    assert len(fns) == 4, sorted(fns)                        # 64 / 96 queries per pass x the certifying twin and the one that trusts pmax
    for name, body in fns.items():
        nmf = 48 if "ELi6EEE" in name else 32                # MFMAs of a ring revolution: 8 k-steps x 4 (6) query fragments
        if "ELi8ELi0ELi" in name:                            # + the certifying twin's Gram MFMA per k-step (row norms = its diagonal)
            nmf += 8
        lines = [l.strip() for l in body.split("\n")]
        mf = [i for i, l in enumerate(lines) if l.startswith("v_mfma")]
        first = next(i for i in mf if sum(1 for j in mf if i <= j < i + 600) >= nmf)
        last = [j for j in mf if first <= j < first + 600][nmf - 1]
        loop = lines[first:last + 1]
        assert sum(l.startswith("v_mfma") for l in loop) == nmf
        assert not any(re.match(r"s_waitcnt.*vmcnt\(0\)", l) for l in loop), f"{name}: the ring is drained inside a revolution"
        # no private segment at all: a kernel with one pays ~12 us per launch for it (1M-row step 0.309 -> 0.297 ms, tools/lib_ab.py,
        # profiles/r03/scan_builds_noscratch.txt); the lane-dependent cold-path addresses are formed where they are used
        assert _scratch_bytes(body) == 0 and "scratch_" not in body, (name, _scratch_bytes(body))


def test_dma_staged_scan_has_no_scratch_and_counted_waits_only():
    """dscan_kernel (round 6: the 64-query pass, slab through LDS-DMA, queries in 96 VGPRs): both twins without a private segment; a tile's
    twelve stages are 192 MFMAs (+ 48 Gram MFMAs in the certifying twin) and 48 nt LDS-DMA pieces; between the FIRST and the LAST MFMA of a
    tile nothing waits for vmcnt(0) -- the pipeline of three stages in flight is only ever waited on by count (vmcnt(8) in front of a stage's
    barrier) -- and no LDS-DMA descriptor lives in VGPRs (no readfirstlane loop around a piece)."""
    fns = {k: v for k, v in _functions(_asm("atlas_hip")).items() if "dscan_kernelILi" in k}
    assert len(fns) == 2, sorted(fns)
    for name, body in fns.items():
        cert = "ILi2E" in name
        assert _scratch_bytes(body) == 0 and "scratch_" not in body, (name, _scratch_bytes(body))
        lines = [l.strip() for l in body.split("\n") if l.strip() and not l.strip().startswith((";", "."))]
        mf = [i for i, l in enumerate(lines) if l.startswith("v_mfma")]
        assert len(mf) == 192 + (48 if cert else 0), (name, len(mf))
        dma = [l for l in lines if l.startswith("buffer_load_dwordx4") and " lds" in l]
        assert len(dma) == 48 + 12 and all(l.endswith(" nt lds") for l in dma), (name, len(dma))       # a tile's stages + the prologue's three
        # the steady-state STAGES: the code between two consecutive barriers that holds a stage's four DMA pieces, its sixteen fragment reads and
        # (hipcc pipelines the second k-step across the barrier) sixteen slab MFMAs. None of them waits for vmcnt(0) -- the pipeline of three stages in
        # flight is only ever waited on by count -- nor touches scratch or writes a scalar into a VGPR lane. (The flush / exchange paths, which do drain
        # the pipeline, sit between other barriers; where hipcc lays them out in the text is its business.)
        bars = [i for i, l in enumerate(lines) if l == "s_barrier"]
        segs = [lines[a + 1:b] for a, b in zip(bars, bars[1:])]
        stages = [g for g in segs if sum(l.startswith("buffer_load_dwordx4") and " lds" in l for l in g) == 4 and sum(l.startswith("ds_read_b128") for l in g) >= 16
                  and sum(l.startswith("v_mfma") for l in g) >= 16 and len(g) < 140]
        assert len(stages) >= 8, (name, len(stages))
        for g in stages:
            assert sum(l == "s_waitcnt vmcnt(8)" for l in g) == 1, (name, g)
            assert not any(re.match(r"s_waitcnt.*vmcnt\(0\)", l) for l in g), f"{name}: the DMA pipeline is drained inside a stage"
            assert not any(l.startswith(("scratch_", "v_writelane", "global_", "flat_")) for l in g), (name, g)
        # (hipcc parks ~100 scalars of the tile-boundary code in VGPR lanes; a stage may fetch one or two of them back -- a v_readlane is one VALU slot)
        assert sum(l.startswith("v_readlane") for g in stages for l in g) <= 12, name
        assert body.count("v_readfirstlane_b32") <= 12, (name, body.count("v_readfirstlane_b32"))


def test_gemm_shaped_scan_has_a_clean_k_loop():
    """gscan_kernel (batches above 96 queries): at the 256-register cap of its 8-wave workgroup (128 accumulators + 96 fragment registers); no
    instantiation carries a private segment (the fragment registers are filled ASYNCHRONOUSLY by single ds_reads between the LDS-DMA pieces: a
    spill or a copy of one in front of the phase's wait would read garbage); the MFMAs of a k-tile are free of vector-memory waits and every
    LDS-DMA descriptor lives in SGPRs (no readfirstlane loop around a DMA)"""
    fns = {k: v for k, v in _functions(_asm("atlas_hip")).items() if "gscan_kernelILi" in k}
    assert len(fns) == 15, sorted(fns)                       # scan, sample, certifying scan x the 256-, 192- and 128-query column tile + (round 6) the two scans' nt twins
    for name, body in fns.items():
        fb = int(re.search(r"gscan_kernelILi\dELi(\d)E", name).group(1))
        assert _scratch_bytes(body) == 0 and "scratch_" not in body, (name, _scratch_bytes(body))
        lines = [l.strip() for l in body.split("\n")]
        mf = [i for i, l in enumerate(lines) if l.startswith("v_mfma")]
        assert len(mf) == 24 * fb, (name, len(mf))           # the k-step of a tile's first k-tile (C = 0) and the two of every k-tile
        # the last 16 x fb MFMAs = the accumulate k-step pair of the steady loop: nothing but MFMAs (and the certifying twin's v_dot2) in between
        loop = lines[mf[8 * fb]:mf[-1] + 1]
        # (a lone `s_waitcnt lgkmcnt(0)` between the k-steps is hipcc's wait for a scalar load: the phase's ds_reads were waited for in front of the barrier)
        assert not any(l.startswith(("scratch_", "buffer_", "global_", "ds_")) or (l.startswith("s_waitcnt") and "vmcnt" in l) for l in loop), name
        dma = [i for i, l in enumerate(lines) if l.startswith("buffer_load_dwordx4") and " lds" in l]
        assert len(dma) >= 8 * fb, (name, len(dma))
        assert body.count("v_readfirstlane_b32") <= 6, (name, body.count("v_readfirstlane_b32"))   # the wave index and the clamped query rows, not descriptors


def test_no_unpadded_overwrite_of_store_data():
    for unit in ("encoder", "atlas_hip"):
        cur = prev = None
        bad = []
        for line in _asm(unit).split("\n"):
            t = line.strip()
            if re.match(r"^_Z\w+:", t):
                cur, prev = t.split(":")[0], None
                continue
            if not t or t.startswith(";") or t.startswith("."):
                continue
            if prev:
                m = re.match(r"buffer_store_dwordx[34] v\[(\d+):(\d+)\], \w+, s\[\d+:\d+\], (s\d+|m0)\b", prev)
                d = re.match(r"v_\w+ v(\d+)\b|v_\w+ v\[(\d+):(\d+)\]", t)
                if m and d:
                    lo, hi = int(m.group(1)), int(m.group(2))
                    regs = range(int(d.group(1)), int(d.group(1)) + 1) if d.group(1) else range(int(d.group(2)), int(d.group(3)) + 1)
                    if any(lo <= r <= hi for r in regs):
                        bad.append((cur, prev, t))
            prev = t
        assert not bad, bad[:3]


def test_certifying_gemm_shaped_scan_has_no_scratch_and_no_inline_asm_dots():
    """gscan_kernel<2, FB> (round 5): 0 B of scratch at the register cap; its v_dot2 are compiler-visible instructions, not inline asm. hipcc's
    hazard recognizer does not look inside inline asm, and a DOT's result read by a DIFFERENT VALU opcode needs 3 wait states (LLVM
    GCNHazardRecognizer: DotWriteDifferentVALURead): an asm-volatile version of the certifier lost one of a fragment's 96 dots per tile on the
    GPU (found by the fragment-coverage test) -- and, pinned between the MFMA groups, was slower than leaving the placement to hipcc. The
    round-4 selects are gone: at most a handful more v_cndmask than the trusting twin has (they belong to the filter epilogue)."""
    fns = _functions(_asm("atlas_hip"))
    cert = {k: v for k, v in fns.items() if "gscan_kernelILi2E" in k}
    assert len(cert) == 6, sorted(cert)                      # x the nt twin of the slab DMA (passes of one column tile)
    for name, body in cert.items():
        assert _scratch_bytes(body) == 0, f"{name} spills {_scratch_bytes(body)} bytes"
        lines = [l.strip() for l in body.split("\n")]
        ins = [l for l in lines if l and not l.startswith((".", "//"))]
        dots = [i for i, l in enumerate(ins) if l.startswith("v_dot2")]
        assert len(dots) >= 16, (name, len(dots))
        for i in dots:
            assert not ins[i - 1].startswith(";;#ASMSTART"), f"{name}: an inline-asm v_dot2 (invisible to the hazard recognizer)"
        twin = fns[name.replace("gscan_kernelILi2E", "gscan_kernelILi0E")]
        assert body.count("v_cndmask") <= twin.count("v_cndmask") + 16, (name, body.count("v_cndmask"), twin.count("v_cndmask"))
