#!/usr/bin/env python3
"""Build-time audit of the compiled GEMM kernels that carry the hand-scheduled K loop (wv_gemm_kloop.inc).

The accumulators live in AGPRs that the K-loop statements update in place while the compiler believes them unchanged
from zero_acc on (see gen_gemm_kloop.py) and only asm statements ever read them.  That is only sound if the compiler
itself never touches an AGPR in these kernels (and, for speed, never spills inside the K loop): this script reads the ISA hipcc produced (-save-temps .s) and
fails the build otherwise.  It also reports registers, scratch and the loop's instruction mix."""
import re
import sys


def kernels(text):
    for m in re.finditer(r"^(_ZN2wv\S+):\s*; @\S+\n(.*?)\.end_amdhsa_kernel", text, re.S | re.M):
        yield m.group(1), m.group(2)


def audit(path):
    text = open(path).read()
    bad = 0
    for name, body in kernels(text):
        if "WVACC" not in body:   # only the kernels that carry the hand-scheduled loop
            continue
        lines = body.split("\n")
        inasm, blocks, cur = False, [], None
        for k, l in enumerate(lines):
            if "ASMSTART" in l:
                inasm, cur = True, [k, None, []]
            elif "ASMEND" in l:
                inasm = False
                cur[1] = k
                blocks.append(cur)
            elif inasm:
                cur[2].append(l.strip())
        zero = next((b for b in blocks if sum(x.startswith("v_accvgpr_write_b32") for x in b[2]) == 128), None)
        loop_stmts = [b for b in blocks if any(x.startswith("v_mfma") for x in b[2])]
        if zero is None or not loop_stmts:
            print(f"AUDIT {name}: zero_acc / K-loop statements not found")
            bad += 1
            continue
        fence = [len(lines), len(lines), []]   # the accumulators are live to the end of the kernel (read-out statements)
        asm_lines = set()
        for b in blocks:
            asm_lines.update(range(b[0], b[1] + 1))
        problems = []
        first_loop_stmt = min(b[0] for b in loop_stmts)
        last_loop_stmt = max(b[1] for b in loop_stmts)
        for k in range(zero[1] + 1, fence[0]):
            if k in asm_lines:
                continue
            code = lines[k].split(";")[0]
            if "scratch_" in code and first_loop_stmt < k < last_loop_stmt:   # (a spill in the K loop: not wrong, but not acceptable)
                problems.append((k, "scratch access", lines[k].strip()))
            if re.search(r"v_accvgpr|[ ,\[]a\d+\b|\ba\[\d+:\d+\]", code):
                problems.append((k, "compiler touches an AGPR", lines[k].strip()))
        # every statement that names the accumulators must have resolved c0..c3 to the SAME four AGPR tuples: the values
        # have one definition and no copies, so a statement that sees them elsewhere would accumulate into (or read out
        # of) registers the others never touch
        bases = {}
        for b in blocks:
            if not any(x.startswith(".set WVACC0,-1") for x in b[2]):
                continue
            got = []
            for x in b[2]:
                m = re.match(r"\.ifc\s+(a\[(\d+):\d+\]),(a\[\d+:\d+\])$", x)
                if m and m.group(1) == m.group(3):
                    got.append(int(m.group(2)))
            bases.setdefault(tuple(got), []).append(b[0])
        if len(bases) != 1 or len(next(iter(bases))) != 4 or len(set(next(iter(bases)))) != 4:
            problems.append((0, "asm statements disagree on the accumulator registers", str({k: len(v) for k, v in bases.items()})))
        nscratch = sum("scratch_" in l.split(";")[0] for l in lines)
        short = re.sub(r"^_ZN2wv15gemm_f64_kernelI|EEvNS_9GemmArgsTIdEE$", "", name)
        print(f"audit <{short}>: {len(blocks)} asm statements, scratch ops in the kernel {nscratch}, "
              f"inside the accumulator region {sum(p[1] == 'scratch access' for p in problems)}, "
              f"compiler AGPR references inside it {sum(p[1] == 'compiler touches an AGPR' for p in problems)}, "
              f"accumulators at a{list(next(iter(bases))) if bases else '?'} in all {sum(len(v) for v in bases.values())} statements that name them")
        for p in problems[:6]:
            print("   ", p)
        bad += len(problems) > 0
    return bad


if __name__ == "__main__":
    sys.exit(1 if audit(sys.argv[1]) else 0)
