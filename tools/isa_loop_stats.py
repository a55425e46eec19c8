"""Instruction mix of the MFMA loops of a HIP source's gfx950 kernels, from the compiler's assembly (no GPU needed): per kernel the
innermost loop that holds MFMAs -- MFMA / scalar / vector / LDS / buffer instruction counts per trip and the non-MFMA : MFMA ratio.

A wave issues roughly one instruction per 4-5 cycles and a v_mfma_f32_32x32x16_f16 occupies the SIMD's matrix pipe for 32, so one
wave hides ~5 other instructions per MFMA and two waves per SIMD ~10; a loop above that is ISSUE-bound whatever its memory system
does (round 5: the haloed 3x3 kernel's runtime slab state machine, 16 per MFMA on its 64 x 128 tile).  Static counts: instructions
behind rarely-taken branches inside the loop are counted as if executed.

    python tools/isa_loop_stats.py action-detection_amd/csrc/conv_pl.hip [substring of the kernel name]
    python tools/isa_loop_stats.py --rev d2d64d6 action-detection_amd/csrc/conv_pl.hip conv_pl9      # the file as of a commit
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INC = os.path.join(ROOT, "action-detection_amd", "csrc")
NOT_SALU = ("s_waitcnt", "s_barrier", "s_nop", "s_cbranch", "s_branch", "s_endpgm")


def assembly(src, rev=None):
    with tempfile.TemporaryDirectory() as d:
        path = src
        if rev:      # the source as of `rev`, compiled beside today's headers of the same directory tree at that revision
            tree = os.path.join(d, "csrc")
            os.makedirs(tree)
            names = subprocess.run(["git", "-C", ROOT, "ls-tree", "--name-only", rev, "action-detection_amd/csrc/"], capture_output=True,
                                   text=True, check=True).stdout.split()
            for n in names:
                if n.endswith((".h", ".inc", ".hip")):
                    with open(os.path.join(tree, os.path.basename(n)), "wb") as f:
                        f.write(subprocess.run(["git", "-C", ROOT, "show", "%s:%s" % (rev, n)], capture_output=True, check=True).stdout)
            path, inc = os.path.join(tree, os.path.basename(src)), tree
        else:
            inc = INC
        out = os.path.join(d, "a.s")
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S", path, "-o", out, "-I", inc],
                       check=True, stderr=subprocess.DEVNULL)
        return open(out).read().split("\n")


def loops(lines, want):
    starts = [(i, l.split(":")[0]) for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)]
    for idx, (i, sym) in enumerate(starts):
        end = starts[idx + 1][0] if idx + 1 < len(starts) else len(lines)
        name = subprocess.run(["c++filt", sym], capture_output=True, text=True).stdout.strip()
        name = name.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
        if want and want not in name:
            continue
        body = lines[i:end]
        meta = {k: next((l.split(":")[1].strip() for l in body if l.startswith("; " + k + ":")), "?") for k in ("NumVgprs", "ScratchSize")}
        # Loop membership from the compiler's block annotations ("=>This Loop Header" on the header block, "in Loop: Header=BBx_y" on
        # the others -- also on fall-through blocks printed as "; %bb.N:").  (Scanning from the header for a branch back to its label
        # misses rotated loops, whose latch blocks are laid out IN FRONT of the header and fall through into it: round 6's epilogue
        # change made the compiler rotate the K loop of most conv_pl_kernel variants.)
        member = {}
        cur = None
        for l in body:
            m = re.match(r"^(\.LBB\d+_\d+):|^; %bb\.\d+:", l)
            if m:
                h = re.search(r"in Loop: Header=(BB\d+_\d+)", l)
                if "Loop Header" in l and m.group(1):
                    cur = m.group(1)[2:]              # ".LBB15_68" -> "BB15_68"
                elif h:
                    cur = h.group(1)
                else:
                    cur = None
                continue
            if cur is not None and l.startswith("\t") and not l.strip().startswith((";", ".")):
                member.setdefault(cur, []).append(l.split()[0])
        best = None
        for ops in member.values():
            nm = sum(o.startswith("v_mfma") for o in ops)
            if nm and (best is None or nm > best[0]):
                best = (nm, ops)
        if best is None:
            continue
        nm, ops = best
        salu = sum(o.startswith("s_") and not o.startswith(NOT_SALU) for o in ops)
        valu = sum(o.startswith("v_") and not o.startswith("v_mfma") for o in ops)
        lds = sum(o.startswith("ds_") for o in ops)
        vmem = sum(o.startswith(("buffer_", "global_", "scratch_")) for o in ops)
        yield name, nm, salu, valu, lds, vmem, len(ops), meta


def main():
    args = sys.argv[1:]
    rev = None
    if args and args[0] == "--rev":
        rev, args = args[1], args[2:]
    src = args[0]
    want = args[1] if len(args) > 1 else ""
    print("# %s%s: the MFMA loop of every kernel (per trip)" % (src, " @ " + rev if rev else ""))
    print("# %-44s %5s %5s %5s %4s %5s %6s %9s  %s" % ("kernel", "MFMA", "SALU", "VALU", "LDS", "VMEM", "total", "non:MFMA", "VGPRs / scratch"))
    for name, nm, salu, valu, lds, vmem, tot, meta in loops(assembly(os.path.join(ROOT, src) if not os.path.isabs(src) else src, rev), want):
        print("%-46s %5d %5d %5d %4d %5d %6d %9.1f  %s / %s" % (name[:46], nm, salu, valu, lds, vmem, tot, (tot - nm) / nm,
                                                             meta["NumVgprs"], meta["ScratchSize"]))


if __name__ == "__main__":
    main()
