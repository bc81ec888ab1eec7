#!/usr/bin/env python3
"""Writes wv_gemm_kloop.inc: the hand-scheduled gfx950 instruction streams of the float64 MFMA GEMM's K loop
(wv_gemm.hip, 128 x 128 block tile, four waves of 64 x 64, two workgroups per CU).

Why not hipcc's loop: its k-step issues the ten fragment reads and then waits for them, a wave
alone on its SIMD - the partner workgroup in its prologue / epilogue - runs at ~85 % of the MFMA rate, and every
HIP-level attempt to software-pipeline the reads ended in spills at the 256-register budget.  Here the register file
is laid out by hand:

  4 x a[32]   the 64 accumulators: four 1024-bit values in AGPRs (one per A fragment i; acc(i, j, r) = registers
              2 (4 j + r), +1 of value i), DEFINED by one statement before the loop (zero_acc, "=a" outputs), passed to
              every K-loop statement and to the read-out statements (stage_acc, read_acc8) as "a" INPUT operands.  The text names them through assembler symbols (a[WVACC0+6:WVACC0+7]) that a
              directive prelude sets from wherever the register allocator put each operand.  Two other forms were
              tried in round 4 and are wrong or unusable: (1) literal a[0:127] behind the compiler's back - hipcc then
              used a[0:3] as the data registers of an epilogue's ds_read2_b64 / global_store (AGPRs are ordinary load /
              store data registers on gfx950; found by tools/check_gemm_asm.py, M21 rows 64-79); (2) "+a" in-out
              operands on every statement - correct, but with eight statements on different branches of the loop the
              register allocator spilled all 128 accumulator registers to scratch around every one of them.  As
              inputs the four values have ONE definition and no copies: they stay where they are for the whole loop
              (the statements do modify them in place; the compiler never looks inside them - every read-out is
              a volatile asm statement too, which keeps it behind the loop).  `make` audits the ISA for this: no
              compiler reference to an AGPR anywhere in these kernels, no scratch access inside the K loop.
  v[48:63]    A fragments, double-buffered by k-step: A[buf][i] = v[48 + 8 buf + 2 i]
  v[64:95]    B ring: value n = 4 j + r of a k-step lives in v[64 + 2 n]; the value for the NEXT k-step is fetched into
              the same pair right after the last MFMA that reads this k-step's value has been issued (61 MFMAs ahead of
              its first use)
  v[96:127]   staging of the next chunk's operand tiles (global_load_dwordx4 -> ds_write_b128)
  s[84:89]    chunk base pointers, loop counter, LDS stage toggle

Two kinds of statement:

  run     n >= 1 full chunks (all 4 x 16 fragments live, every chunk stages its successor) in ONE statement with the
          loop inside.  The loop is rotated: the barrier sits twelve MFMAs into the LAST k-step of a chunk - by then
          every read of this chunk's LDS stage has landed in registers and every ds_write of the next stage has
          been issued - and the rest of that k-step fetches the first k-step of the next chunk.  256 MFMAs per
          chunk issue back to back; no LDS latency is exposed anywhere in the steady state.
  chunk   one chunk with fragment ranges [I0, I1) x [J0, J1) (the partial chunks next to a triangular operand's
          diagonal, and the last chunk of a tile, which stages nothing); pipelined inside the chunk, barrier at
          its end.

LDS operations complete in issue order, so every wait is a counted s_waitcnt lgkmcnt(N) computed here by simulating
the instruction stream (N = LGKM operations issued after the producing read, at most 14: see Stream.need).  Global
loads are consumed by ds_writes behind counted vmcnt waits.  Nothing inside a statement depends on the compiler's
scheduling; the compiler sees one opaque instruction with the literal registers as clobbers.
"""
import sys

VA0, VB0, VS0 = 48, 64, 96
S_PA, S_PB, S_N, S_D = 84, 86, 88, 89
STAGE_BYTES = 18432  # one LDS stage of one operand: 16 x 144 doubles (K-strided) = 128 x 18 doubles (K-contiguous)
WR_STRIDE = 4608     # LDS byte stride between a thread's four staging pieces (both layouts)


class Layout:
    """LDS image of one operand stage.  spread: a wave's fragments are every OTHER 16-row group of the 128-row panel
    (rows 16 w + 32 f; wv_gemm.hip, SPREAD) instead of four consecutive ones."""

    def __init__(self, kcontig, spread=False):
        self.kc = kcontig
        if kcontig:      # [rows][18]
            self.frag = 16 * 18 * 8   # next 16 rows
            self.kstep = 4 * 8        # next 4 k
        else:            # [16][rows + 16], rows = 128
            self.frag = 16 * 8
            self.kstep = 4 * 144 * 8
        if spread:
            self.frag *= 2


def acc(i, j, r):
    x = 2 * (4 * j + r)
    return f"a[WVACC{i}+{x}:WVACC{i}+{x + 1}]"


# assembler-time lookup of the four accumulator operands' base registers (no instructions)
def _prelude(sym, file, width, last):
    """Per operand k: reset the symbol (assembler symbols are global to the whole .s file - without the reset an operand
    that prints in an unforeseen way would silently keep the previous statement's value), the .ifc chain, and an
    assembler error when nothing matched."""
    out = []
    for k in range(4):
        out.append(f".set {sym}{k},-1")
        out += [f".ifc %[c{k}],{file}[{b}:{b + width - 1}]\\n\\t.set {sym}{k},{b}\\n\\t.endif" for b in range(0, last + 1)]
        out.append(f".if {sym}{k}<0\\n\\t.error \\\"gen_gemm_kloop: accumulator operand c{k} is not a {file}[b:b+{width - 1}] tuple\\\"\\n\\t.endif")
    return out


ACC_PRELUDE = _prelude("WVACC", "a", 32, 96)


def va(buf, i):
    x = VA0 + 8 * buf + 2 * i
    return f"v[{x}:{x + 1}]"


def vb(n):
    x = VB0 + 2 * n
    return f"v[{x}:{x + 1}]"


def vs(q):
    x = VS0 + 4 * q
    return f"v[{x}:{x + 3}]"


class Stream:
    """Instruction list + the LGKM / VM counters needed for counted waits."""

    def __init__(self, regs=None):
        self.regs = regs     # None: the 128 x 128 kernels' map (module functions va / vb / acc); else an object with those
        self.lines = []
        self.issued = 0      # LGKM operations issued so far
        self.known = 0       # every LGKM operation with index < known is complete
        self.ready = {}      # register name -> index of the read that fills it
        self.vm_issued = 0
        self.vm_known = 0
        self.vm_ready = {}
        self.n_mfma = 0

    def emit(self, s):
        self.lines.append(s)

    def room(self):
        """The LGKM counter has four bits.  A counted wait is only exact while the counter has never been asked to
        hold more than 15 (measured: a burst of 17 reads followed by lgkmcnt(14) let an MFMA read a fragment that had
        not landed), so no operation is issued while 15 may be outstanding: wait for the oldest four first."""
        if self.issued - self.known >= 15:
            self.emit("s_waitcnt lgkmcnt(11)")
            self.known = self.issued - 11

    def ds_read(self, dst, addr, off):
        self.room()
        self.emit(f"ds_read_b64 {dst}, {addr} offset:{off}")
        self.ready[dst] = self.issued
        self.issued += 1

    def ds_write(self, addr, src, off):
        self.need_vm(src)
        self.room()
        self.emit(f"ds_write_b128 {addr}, {src} offset:{off}")
        self.issued += 1

    def gload(self, dst, voff, sbase):
        self.emit(f"global_load_dwordx4 {dst}, {voff}, {sbase}")
        self.vm_ready[dst] = self.vm_issued
        self.vm_issued += 1

    def need(self, regs):
        p = max(self.ready.get(r, -1) for r in regs)
        if p >= self.known:
            n = self.issued - (p + 1)
            assert n <= 14, "room() keeps at most 15 operations outstanding"
            self.emit(f"s_waitcnt lgkmcnt({n})")
            self.known = self.issued - n

    def need_vm(self, reg):
        p = self.vm_ready.get(reg, -1)
        if p >= self.vm_known:
            n = self.vm_issued - (p + 1)
            self.emit(f"s_waitcnt vmcnt({n})")
            self.vm_known = self.vm_issued - n

    def drain_lgkm(self):
        self.emit("s_waitcnt lgkmcnt(0)")
        self.known = self.issued

    def mfma(self, i, j, r, buf):
        m = self.regs
        a, b = (m.va(buf, i), m.vb(4 * j + r)) if m else (va(buf, i), vb(4 * j + r))
        self.need([a, b])
        c = m.acc(i, j, r) if m else acc(i, j, r)
        self.emit(f"v_mfma_f64_4x4x4_4b_f64 {c}, {a}, {b}, {c}")
        self.n_mfma += 1


def kstep_stream(st, mfmas, side):
    """MFMAs in order; after MFMA m at most one side operation whose `earliest` <= m, in (earliest, sequence) order.
    `side` = list of (earliest, fn); fn(st) emits the operation.  What is left is flushed behind the last MFMA."""
    side = sorted(enumerate(side), key=lambda t: (t[1][0], t[0]))
    q = [s for _, s in side]
    qi = 0
    while qi < len(q) and q[qi][0] < 0:   # before the first MFMA
        q[qi][1](st)
        qi += 1
    for m, fn in enumerate(mfmas):
        fn(st)
        if qi < len(q) and q[qi][0] <= m:
            q[qi][1](st)
            qi += 1
    while qi < len(q):
        q[qi][1](st)
        qi += 1


def frag_reads(la, lb, I, J, s, buf, ard="%[ard]", brd="%[brd{}]"):
    """Side operations fetching k-step s: A fragments into buffer `buf`, B values into their ring pairs.
    Returns (a_ops, b_ops) as lists of (key, fn)."""
    a_ops, b_ops = [], []
    for i in range(*I):
        a_ops.append((i, (lambda st, i=i: st.ds_read(va(buf, i), ard, i * la.frag + s * la.kstep))))
    for j in range(*J):
        for r in range(4):
            n = 4 * j + r
            b_ops.append((n, (lambda st, j=j, r=r, n=n: st.ds_read(vb(n), brd.format(r), j * lb.frag + s * lb.kstep))))
    return a_ops, b_ops


def frag_rows(I, j, lower):
    """A fragments that meet B column set j: all of I, or - `lower`, a wave on the diagonal of a lower-only product -
    only the 16 x 16 accumulator blocks on and below the diagonal of its 64 x 64 block (i >= j)."""
    return [i for i in range(*I) if not lower or i >= j]


def mfma_list(I, J, buf, lower=False):
    out = []
    for j in range(*J):
        for r in range(4):
            for i in frag_rows(I, j, lower):
                out.append(lambda st, i=i, j=j, r=r: st.mfma(i, j, r, buf))
    return out


def last_use_index(I, J, lower=False):
    """index (within the k-step's MFMA list) of the last MFMA reading B value n"""
    idx = {}
    m = 0
    for j in range(*J):
        for r in range(4):
            m += len(frag_rows(I, j, lower))
            idx[4 * j + r] = m - 1
    return idx


def gen_run(la, lb, J=(0, 4), lower=False, oneop=False, bar_slot=12):
    """The rotated loop over n full chunks.  J / lower: the two wave roles on a DIAGONAL tile of a square lower-only
    product (wv_gemm.hip, TRIOPT): `lower` = the wave's own 64 x 64 block lies on the diagonal (40 MFMAs per k-step), J =
    (0, 2) = the wave owns 32 columns of the full block below (32 MFMAs).  oneop: both fragment kinds are read from ONE
    staged operand (A == B, same rows): only the A pieces are loaded and written."""
    I = (0, 4)
    lu = last_use_index(I, J, lower)
    nm = len(mfma_list(I, J, 0, lower))
    write_slot0 = nm // 2 + 2

    def body(st, record=None):
        # k-steps 0..2: fetch the next k-step of the same stage; stage the next chunk
        for s in range(3):
            buf = s & 1
            a_ops, b_ops = frag_reads(la, lb, I, J, s + 1, buf ^ 1)
            side = [(4 * i + 1, fn) for i, fn in a_ops] + [(lu[n], fn) for n, fn in b_ops]
            if s == 0:   # the next chunk's operand tiles: eight 16-byte pieces per thread
                for q in range(4):
                    side.append((4 * q + 2, lambda st, q=q: st.gload(vs(q), f"%[oa{q}]", f"s[{S_PA}:{S_PA + 1}]")))
                for q in range(0 if oneop else 4):
                    side.append((4 * (q + 4) + 2, lambda st, q=q: st.gload(vs(q + 4), f"%[ob{q}]", f"s[{S_PB}:{S_PB + 1}]")))
            if s == 1:   # chunk bases of the chunk after
                side.append((2, lambda st: st.emit(f"s_add_u32 s{S_PA}, s{S_PA}, %[sal]")))
                side.append((6, lambda st: st.emit(f"s_addc_u32 s{S_PA + 1}, s{S_PA + 1}, %[sah]")))
                if not oneop:
                    side.append((10, lambda st: st.emit(f"s_add_u32 s{S_PB}, s{S_PB}, %[sbl]")))
                    side.append((14, lambda st: st.emit(f"s_addc_u32 s{S_PB + 1}, s{S_PB + 1}, %[sbh]")))
            if s == 2:
                for q in range(4):
                    side.append((write_slot0 + 4 * q, lambda st, q=q: st.ds_write("%[awr]", vs(q), q * WR_STRIDE)))
                for q in range(0 if oneop else 4):
                    side.append((write_slot0 + 4 * (q + 4), lambda st, q=q: st.ds_write("%[bwr]", vs(q + 4), q * WR_STRIDE)))
            kstep_stream(st, mfma_list(I, J, buf, lower), side)
        # k-step 3: toggle the stage, barrier, fetch k-step 0 of the next chunk from the other stage
        buf = 1
        side = []
        tog = [f"v_add_u32 %[ard], s{S_D}, %[ard]"] + [f"v_add_u32 %[brd{r}], s{S_D}, %[brd{r}]" for r in range(4)] + \
              [f"v_subrev_u32 %[awr], s{S_D}, %[awr]"] + ([] if oneop else [f"v_subrev_u32 %[bwr], s{S_D}, %[bwr]"]) + \
              [f"s_sub_u32 s{S_D}, 0, s{S_D}", f"s_sub_u32 s{S_N}, s{S_N}, 1"]
        for t, line in enumerate(tog):
            side.append((t, lambda st, line=line: st.emit(line)))

        def barrier(st):
            st.drain_lgkm()
            st.emit("s_barrier")
            if record is not None:
                record.append(len(st.lines))
        side.append((bar_slot, barrier))
        a_ops, b_ops = frag_reads(la, lb, I, J, 0, 0)
        side.append((bar_slot + 1, b_ops[0][1]))          # what the next chunk's first MFMA needs comes first
        for i, fn in a_ops:
            side.append((bar_slot + 1, fn))
        for n, fn in b_ops[1:]:
            side.append((max(lu[n], bar_slot + 1), fn))
        kstep_stream(st, mfma_list(I, J, buf, lower), side)

    # order of the reads behind the barrier = order of the prologue's reads (the loop head must see the same LGKM
    # history whichever way it is entered)
    probe = Stream()
    mark = []
    body(probe, mark)
    pro_reads = [l for l in probe.lines[mark[0]:] if l.startswith("ds_read")]
    assert len(pro_reads) == (I[1] - I[0]) + 4 * (J[1] - J[0])
    pro = Stream()             # the prologue: the same reads from a drained counter
    pro.drain_lgkm()
    for l in pro_reads:
        dst, addr, off = l.split()[1].rstrip(","), l.split()[2], l.split("offset:")[1]
        pro.ds_read(dst, addr, off)

    def carry(src, since):     # the LGKM state `src` leaves, renumbered from operation `since`
        st = Stream()
        st.issued, st.known = src.issued - since, max(0, src.known - since)
        st.ready = {k: v - since for k, v in src.ready.items() if v >= since}
        return st

    first = carry(pro, 0)
    body(first)
    second = carry(first, first.issued - pro.issued)   # the state one iteration leaves must reproduce its stream
    assert (second.issued, second.known, second.ready) == (pro.issued, pro.known, pro.ready), "loop head state differs"
    body(second)
    assert first.lines == second.lines, "loop-carried wait state is not a fixed point"
    assert first.n_mfma == 4 * nm

    lines = [f"s_mov_b64 s[{S_PA}:{S_PA + 1}], %[pa]"] + ([] if oneop else [f"s_mov_b64 s[{S_PB}:{S_PB + 1}], %[pb]"]) + \
            [f"s_mov_b32 s{S_N}, %[n]", f"s_mov_b32 s{S_D}, %[d]"]
    lines += pro.lines
    lines.append("L_wv_kloop_%=:")
    lines += first.lines
    lines += [f"s_cmp_lg_u32 s{S_N}, 0", "s_cbranch_scc1 L_wv_kloop_%=", "s_waitcnt lgkmcnt(0)"]
    lines += ["s_nop 15"]   # the last MFMAs' results have left the pipe before compiler code may read an accumulator
    return lines


def gen_chunk(la, lb, I, J, stage, lower=False, oneop=False):
    """One chunk, fragment ranges I x J, pipelined inside the chunk, barrier at the end."""
    st = Stream()
    st.drain_lgkm()    # (whatever the compiler's code before the statement left in flight)
    lu = last_use_index(I, J, lower)
    nm = len(mfma_list(I, J, 0, lower))
    a_ops, b_ops = frag_reads(la, lb, I, J, 0, 0)
    # prologue order: what the first MFMAs need first
    for n, fn in b_ops[:1]:
        fn(st)
    for i, fn in a_ops:
        fn(st)
    for n, fn in b_ops[1:]:
        fn(st)
    for s in range(4):
        buf = s & 1
        side = []
        if s < 3:
            a_ops, b_ops = frag_reads(la, lb, I, J, s + 1, buf ^ 1)
            ni = I[1] - I[0]
            side += [(min(nm - 1, 1 + k * max(1, ni)), fn) for k, (i, fn) in enumerate(a_ops)]
            side += [(lu[n], fn) for n, fn in b_ops]
        if stage and s == 0:
            for q in range(4):
                side.append((-1, lambda st, q=q: st.gload(vs(q), f"%[oa{q}]", "%[pa]")))
            for q in range(0 if oneop else 4):
                side.append((-1, lambda st, q=q: st.gload(vs(q + 4), f"%[ob{q}]", "%[pb]")))
        if stage and s == 3:
            for q in range(4):
                side.append((nm // 2, lambda st, q=q: st.ds_write("%[awr]", vs(q), q * WR_STRIDE)))
            for q in range(0 if oneop else 4):
                side.append((nm // 2, lambda st, q=q: st.ds_write("%[bwr]", vs(q + 4), q * WR_STRIDE)))
        kstep_stream(st, mfma_list(I, J, buf, lower), side)
    assert st.n_mfma == 4 * nm
    st.drain_lgkm()
    st.emit("s_barrier")
    st.emit("s_nop 15")
    return st.lines



# ---------------------------------------------------------------------------------------------------------------
# The SMALL tile (64 x 64 block tile, the four waves side by side: 64 rows x 16 columns each - few fits in flight and
# the lowest recursion level): sixteen accumulators per wave, so they stay ordinary VGPR values of the compiler (four
# 256-bit "+v" operands, named through assembler symbols like the AGPR ones); one statement kind, the rotated run of
# n >= 1 full staged chunks - single chunks (a tile's last, partial ones) stay hipcc's.
# ---------------------------------------------------------------------------------------------------------------
class SmallRegs:
    VA0, VB0, VS0 = 128, 144, 152      # A fragments double-buffered (16), B ring (8), staging pieces (16): v[128:167]

    @staticmethod
    def va(buf, i):
        x = SmallRegs.VA0 + 8 * buf + 2 * i
        return f"v[{x}:{x + 1}]"

    @staticmethod
    def vb(n):
        x = SmallRegs.VB0 + 2 * n
        return f"v[{x}:{x + 1}]"

    @staticmethod
    def vs(q):
        x = SmallRegs.VS0 + 4 * q
        return f"v[{x}:{x + 3}]"

    @staticmethod
    def acc(i, j, r):
        return f"v[WVC{i}+{2 * r}:WVC{i}+{2 * r + 1}]"


SMALL_PRELUDE = _prelude("WVC", "v", 8, 248)
S_DB = 90   # B's stage toggle (the two operands' stages differ in size when A is K-contiguous)


class SmallLayout:
    def __init__(self, kcontig):
        self.kc = kcontig
        if kcontig:      # [64][18]
            self.frag, self.kstep, self.wr_stride = 16 * 18 * 8, 4 * 8, 32 * 18 * 8
        else:            # [16][64 + 16]
            self.frag, self.kstep, self.wr_stride = 16 * 8, 4 * 80 * 8, 8 * 80 * 8


def gen_small_run(la, lb, bar_slot=4):
    I, J = (0, 4), (0, 1)
    R = SmallRegs
    lu = last_use_index(I, J)          # B value r is free after MFMA 4 r + 3
    a_slots = [0, 1, 2, 5]             # where the next k-step's A fragments are fetched (one side operation per MFMA)

    def reads(s, buf):
        a_ops = [(i, (lambda st, i=i: st.ds_read(R.va(buf, i), "%[ard]", i * la.frag + s * la.kstep))) for i in range(4)]
        b_ops = [(r, (lambda st, r=r: st.ds_read(R.vb(r), f"%[brd{r}]", s * lb.kstep))) for r in range(4)]
        return a_ops, b_ops

    def body(st, record=None):
        for s in range(3):
            buf = s & 1
            a_ops, b_ops = reads(s + 1, buf ^ 1)
            side = [(a_slots[i], fn) for i, fn in a_ops] + [(lu[r], fn) for r, fn in b_ops]
            if s == 0:
                for q in range(2):
                    side.append((3 + q, lambda st, q=q: st.gload(R.vs(q), f"%[oa{q}]", f"s[{S_PA}:{S_PA + 1}]")))
                for q in range(2):
                    side.append((6 + q, lambda st, q=q: st.gload(R.vs(q + 2), f"%[ob{q}]", f"s[{S_PB}:{S_PB + 1}]")))
            if s == 1:
                side.append((4, lambda st: st.emit(f"s_add_u32 s{S_PA}, s{S_PA}, %[sal]")))
                side.append((6, lambda st: st.emit(f"s_addc_u32 s{S_PA + 1}, s{S_PA + 1}, %[sah]")))
                side.append((8, lambda st: st.emit(f"s_add_u32 s{S_PB}, s{S_PB}, %[sbl]")))
                side.append((10, lambda st: st.emit(f"s_addc_u32 s{S_PB + 1}, s{S_PB + 1}, %[sbh]")))
            kstep_stream(st, mfma_list(I, J, buf), side)
        # k-step 3: toggle the stages, barrier, fetch k-step 0 of the next chunk from the other stage
        buf = 1
        tog = [f"v_add_u32 %[ard], s{S_D}, %[ard]"] + [f"v_add_u32 %[brd{r}], s{S_DB}, %[brd{r}]" for r in range(4)] + \
              [f"v_subrev_u32 %[awr], s{S_D}, %[awr]", f"v_subrev_u32 %[bwr], s{S_DB}, %[bwr]",
               f"s_sub_u32 s{S_D}, 0, s{S_D}", f"s_sub_u32 s{S_DB}, 0, s{S_DB}", f"s_sub_u32 s{S_N}, s{S_N}, 1"]
        side = []
        # the next stage's pieces as late as the barrier allows (a chunk is 64 MFMAs - ~1 k cycles - and the global loads of
        # k-step 0 need most of it); the address toggles ride behind them (this k-step's own reads were issued in k-step 2)
        for q in range(2):
            side.append((q, lambda st, q=q: st.ds_write("%[awr]", R.vs(q), q * la.wr_stride)))
        for q in range(2):
            side.append((2 + q, lambda st, q=q: st.ds_write("%[bwr]", R.vs(q + 2), q * lb.wr_stride)))

        def barrier(st):
            for line in tog:
                st.emit(line)
            st.drain_lgkm()
            st.emit("s_barrier")
            if record is not None:
                record.append(len(st.lines))
        side.append((bar_slot, barrier))
        a_ops, b_ops = reads(0, 0)
        side.append((bar_slot + 1, b_ops[0][1]))
        for i, fn in a_ops:
            side.append((bar_slot + 1, fn))
        for r, fn in b_ops[1:]:
            side.append((max(lu[r], bar_slot + 1), fn))
        kstep_stream(st, mfma_list(I, J, buf), side)

    probe = Stream(R)
    mark = []
    body(probe, mark)
    pro_reads = [l for l in probe.lines[mark[0]:] if l.startswith("ds_read")]
    assert len(pro_reads) == 8
    pro = Stream(R)
    pro.drain_lgkm()
    for l in pro_reads:
        dst, addr, off = l.split()[1].rstrip(","), l.split()[2], l.split("offset:")[1]
        pro.ds_read(dst, addr, off)

    def carry(src, since):
        st = Stream(R)
        st.issued, st.known = src.issued - since, max(0, src.known - since)
        st.ready = {k: v - since for k, v in src.ready.items() if v >= since}
        return st

    first = carry(pro, 0)
    body(first)
    second = carry(first, first.issued - pro.issued)
    assert (second.issued, second.known, second.ready) == (pro.issued, pro.known, pro.ready), "loop head state differs"
    body(second)
    assert first.lines == second.lines, "loop-carried wait state is not a fixed point"
    assert first.n_mfma == 64
    lines = [f"s_mov_b64 s[{S_PA}:{S_PA + 1}], %[pa]", f"s_mov_b64 s[{S_PB}:{S_PB + 1}], %[pb]",
             f"s_mov_b32 s{S_N}, %[n]", f"s_mov_b32 s{S_D}, %[da]", f"s_mov_b32 s{S_DB}, %[db]"]
    lines += pro.lines
    lines.append("L_wv_kloop_%=:")
    lines += first.lines
    lines += [f"s_cmp_lg_u32 s{S_N}, 0", "s_cbranch_scc1 L_wv_kloop_%=", "s_waitcnt lgkmcnt(0)", "s_nop 15"]
    return lines


def emit_small_run(name, la, lb):
    lines = gen_small_run(la, lb)
    text = "      WV_SMALL_ACC_PRELUDE\n" + "\n".join(f'      "{l}\\n\\t"' for l in lines)
    clob = ", ".join(f'"v{k}"' for k in range(SmallRegs.VA0, SmallRegs.VS0 + 16))
    return f"""
// {len(lines)} instructions; loop body {sum(1 for l in lines if l.startswith('v_mfma'))} MFMAs
__device__ __forceinline__ void {name}(v8i& c0, v8i& c1, v8i& c2, v8i& c3, uint32_t ard, uint32_t brd0, uint32_t brd1, uint32_t brd2,
    uint32_t brd3, uint32_t awr, uint32_t bwr, uint32_t oa0, uint32_t oa1, uint32_t ob0, uint32_t ob1, const char* pa,
    const char* pb, uint32_t sal, uint32_t sah, uint32_t sbl, uint32_t sbh, int n, int da, int db) {{
  asm volatile(
{text}
      : [c0] "+v"(c0), [c1] "+v"(c1), [c2] "+v"(c2), [c3] "+v"(c3), [ard] "+v"(ard), [brd0] "+v"(brd0), [brd1] "+v"(brd1),
        [brd2] "+v"(brd2), [brd3] "+v"(brd3), [awr] "+v"(awr), [bwr] "+v"(bwr)
      : [oa0] "v"(oa0), [oa1] "v"(oa1), [ob0] "v"(ob0), [ob1] "v"(ob1), [pa] "s"(pa), [pb] "s"(pb), [sal] "s"(sal),
        [sah] "s"(sah), [sbl] "s"(sbl), [sbh] "s"(sbh), [n] "s"(n), [da] "s"(da), [db] "s"(db)
      : "memory", "scc", "s{S_PA}", "s{S_PA + 1}", "s{S_PB}", "s{S_PB + 1}", "s{S_N}", "s{S_D}", "s{S_DB}", {clob});
}}
"""


CLOBBER_V = ", ".join(f'"v{k}"' for k in range(VA0, 128))
ACC_ARGS = "const v32i& c0, const v32i& c1, const v32i& c2, const v32i& c3"
ACC_OPS = '[c0] "a"(c0), [c1] "a"(c1), [c2] "a"(c2), [c3] "a"(c3)'


def c_string(lines):
    return "      WV_KLOOP_ACC_PRELUDE\n" + "\n".join(f'      "{l}\\n\\t"' for l in lines)


def emit_run(name, la, lb, **kw):
    lines = gen_run(la, lb, **kw)
    return f"""
// {len(lines)} instructions; loop body {sum(1 for l in lines if l.startswith('v_mfma'))} MFMAs
__device__ __forceinline__ void {name}({ACC_ARGS}, uint32_t ard, uint32_t brd0, uint32_t brd1, uint32_t brd2,
    uint32_t brd3, uint32_t awr, uint32_t bwr, uint32_t oa0, uint32_t oa1, uint32_t oa2, uint32_t oa3, uint32_t ob0,
    uint32_t ob1, uint32_t ob2, uint32_t ob3, const char* pa, const char* pb, uint32_t sal, uint32_t sah, uint32_t sbl,
    uint32_t sbh, int n, int d) {{
  asm volatile(
{c_string(lines)}
      : [ard] "+v"(ard), [brd0] "+v"(brd0), [brd1] "+v"(brd1), [brd2] "+v"(brd2), [brd3] "+v"(brd3), [awr] "+v"(awr),
        [bwr] "+v"(bwr)
      : {ACC_OPS}, [oa0] "v"(oa0), [oa1] "v"(oa1), [oa2] "v"(oa2), [oa3] "v"(oa3), [ob0] "v"(ob0), [ob1] "v"(ob1), [ob2] "v"(ob2),
        [ob3] "v"(ob3), [pa] "s"(pa), [pb] "s"(pb), [sal] "s"(sal), [sah] "s"(sah), [sbl] "s"(sbl), [sbh] "s"(sbh),
        [n] "s"(n), [d] "s"(d)
      : "memory", "scc", "s{S_PA}", "s{S_PA + 1}", "s{S_PB}", "s{S_PB + 1}", "s{S_N}", "s{S_D}", WV_KLOOP_CLOBBERS);
}}
"""


def emit_chunk(name, la, lb, I, J, stage, **kw):
    lines = gen_chunk(la, lb, I, J, stage, **kw)
    if stage:
        return f"""
__device__ __forceinline__ void {name}({ACC_ARGS}, uint32_t ard, uint32_t brd0, uint32_t brd1, uint32_t brd2,
    uint32_t brd3, uint32_t awr, uint32_t bwr, uint32_t oa0, uint32_t oa1, uint32_t oa2, uint32_t oa3, uint32_t ob0,
    uint32_t ob1, uint32_t ob2, uint32_t ob3, const char* pa, const char* pb) {{
  asm volatile(
{c_string(lines)}
      :
      : {ACC_OPS}, [ard] "v"(ard), [brd0] "v"(brd0), [brd1] "v"(brd1), [brd2] "v"(brd2), [brd3] "v"(brd3), [awr] "v"(awr),
        [bwr] "v"(bwr), [oa0] "v"(oa0), [oa1] "v"(oa1), [oa2] "v"(oa2), [oa3] "v"(oa3), [ob0] "v"(ob0), [ob1] "v"(ob1),
        [ob2] "v"(ob2), [ob3] "v"(ob3), [pa] "s"(pa), [pb] "s"(pb)
      : "memory", WV_KLOOP_CLOBBERS);
}}
"""
    return f"""
__device__ __forceinline__ void {name}({ACC_ARGS}, uint32_t ard, uint32_t brd0, uint32_t brd1, uint32_t brd2,
    uint32_t brd3) {{
  asm volatile(
{c_string(lines)}
      :
      : {ACC_OPS}, [ard] "v"(ard), [brd0] "v"(brd0), [brd1] "v"(brd1), [brd2] "v"(brd2), [brd3] "v"(brd3)
      : "memory", WV_KLOOP_CLOBBERS);
}}
"""


def main(out_path):
    ks, kc = Layout(False, spread=True), Layout(True, spread=True)
    ks_block = Layout(False)   # the diagonal-tile roles keep the 64 x 64 corner per wave
    out = ["// GENERATED by gen_gemm_kloop.py - do not edit.  Hand-scheduled gfx950 ISA of the float64 GEMM K loop;",
           "// register layout, pipeline and wait discipline are described in the generator.",
           "#pragma once",
           f"#define WV_KLOOP_CLOBBERS {CLOBBER_V}",
           "// assembler-time lookup of the accumulator operands' base registers (directives only, no instructions)",
           "#define WV_KLOOP_ACC_PRELUDE \\\n" + " \\\n".join(f'  "{l}\\n\\t"' for l in ACC_PRELUDE),
           "#define WV_KLOOP_ACC_PRELUDE_OUT WV_KLOOP_ACC_PRELUDE",
           "#define WV_SMALL_ACC_PRELUDE \\\n" + " \\\n".join(f'  "{l}\\n\\t"' for l in SMALL_PRELUDE),
           "namespace wv { namespace kloop {",
           "typedef int v32i __attribute__((ext_vector_type(32)));  // sixteen float64 accumulators in 32 AGPRs",
           "typedef int v8i __attribute__((ext_vector_type(8)));    // four float64 accumulators of the small tile in 8 VGPRs",
           "// the accumulators come into being here (zeros), in registers of the allocator's choosing ...",
           "__device__ __forceinline__ void zero_acc(v32i& c0, v32i& c1, v32i& c2, v32i& c3) {",
           "  asm volatile(",
           "      WV_KLOOP_ACC_PRELUDE_OUT",
           "\n".join(f'      "v_accvgpr_write_b32 a[WVACC{g}+{q}], 0\\n\\t"' for g in range(4) for q in range(32)),
           '      : [c0] "=a"(c0), [c1] "=a"(c1), [c2] "=a"(c2), [c3] "=a"(c3));',
           "}",
           "// ... and are read out by asm statements only (stage_acc / read_acc8 below, volatile: they stay behind the K-loop",
           "// statements).  The compiler never extracts an element itself: it believes the four values unchanged since",
           "// zero_acc and would be free to read them in front of the loop.  (A \"+a\" hand-back statement behind the loop",
           "// was tried: with it hipcc started copying / spilling the 1024-bit tuples inside the K loop.)",
           "// Read-out for alpha = +-1 (every product of the pipeline): the 32 accumulators of A fragments 2 H, 2 H + 1 (32 rows",
           "// of the wave's 64 x 64 block) -> the wave's own LDS staging region (row stride 66 doubles) at b<r> + (16 (i & 1) 66",
           "// + 16 j) doubles, sign bit flipped for alpha = -1 (the same bits as the product with -1.0).  No float64 multiply -",
           "// the float64 vector unit is the datapath the partner workgroup's MFMAs are saturating - and no compiler scheduling:",
           "// hipcc read all 64 accumulators out first and spilled them (round 4).",
           "template <int H> __device__ __forceinline__ void stage_acc_half(const v32i& c0, const v32i& c1, const v32i& c2, const v32i& c3,",
           "    uint32_t b0, uint32_t b1, uint32_t b2, uint32_t b3, int sign);"]
    for H in range(2):
        out.append(f"template <> __device__ __forceinline__ void stage_acc_half<{H}>(const v32i& c0, const v32i& c1, const v32i& c2, const v32i& c3,")
        out.append("    uint32_t b0, uint32_t b1, uint32_t b2, uint32_t b3, int sign) {")
        out.append("  asm volatile(")
        out.append("      WV_KLOOP_ACC_PRELUDE")
        out.append("\n".join(
               f'      "v_accvgpr_read_b32 v{VA0 + 2 * (k % 8)}, a[WVACC{k >> 4}+{2 * (k & 15)}]\\n\\t"\n'
               f'      "v_accvgpr_read_b32 v{VA0 + 2 * (k % 8) + 1}, a[WVACC{k >> 4}+{2 * (k & 15) + 1}]\\n\\t"\n'
               f'      "v_xor_b32 v{VA0 + 2 * (k % 8) + 1}, %[sign], v{VA0 + 2 * (k % 8) + 1}\\n\\t"\n'
               f'      "ds_write_b64 %[b{k & 3}], v[{VA0 + 2 * (k % 8)}:{VA0 + 2 * (k % 8) + 1}] offset:{(((k >> 4) & 1) * 16 * 66 + ((k >> 2) & 3) * 16) * 8}\\n\\t"'
               for k in range(32 * H, 32 * H + 32)))
        out.append('      "s_waitcnt lgkmcnt(0)\\n\\t"')
        out.append('      :')
        out.append('      : [c0] "a"(c0), [c1] "a"(c1), [c2] "a"(c2), [c3] "a"(c3), [b0] "v"(b0), [b1] "v"(b1), [b2] "v"(b2), [b3] "v"(b3),')
        out.append('        [sign] "s"(sign)')
        out.append('      : "memory", ' + ", ".join(f'"v{VA0 + q}"' for q in range(16)) + ');')
        out.append("}")
    out += [
           "// General alpha (tests only): accumulators 8 G .. 8 G + 7 (A fragment G / 2, doubles 8 (G & 1) ..) as sixteen words",
           "// (low, high) for the caller to scale and stage.",
           "template <int G> __device__ __forceinline__ void read_acc8(const v32i& c0, const v32i& c1, const v32i& c2, const v32i& c3, uint32_t (&w)[16]);"]
    for G in range(8):
        out.append(f"template <> __device__ __forceinline__ void read_acc8<{G}>(const v32i& c0, const v32i& c1, const v32i& c2, const v32i& c3, uint32_t (&w)[16]) {{")
        out.append("  asm volatile(")
        out.append("      WV_KLOOP_ACC_PRELUDE")
        out.append("\n".join(f'      "v_accvgpr_read_b32 %[w{q}], a[WVACC{G >> 1}+{16 * (G & 1) + q}]\\n\\t"' for q in range(16)))
        out.append("      : " + ", ".join(f'[w{q}] "=v"(w[{q}])' for q in range(16)))
        out.append('      : [c0] "a"(c0), [c1] "a"(c1), [c2] "a"(c2), [c3] "a"(c3));')
        out.append("}")
    # SS: A and B K-strided (M'M, trailing update, T, M21); CS: A K-contiguous (S = A21 M11')
    for tag, la, lb in (("ss", ks, ks), ("cs", kc, ks)):
        out.append(emit_run(f"run_{tag}", la, lb))
        out.append(emit_chunk(f"chunk_{tag}_full_last", la, lb, (0, 4), (0, 4), False))
        fams = {"ss": ("ilo", "ihi", "jlo"), "cs": ("jhi",)}[tag]
        for fam in fams:
            for live in (1, 2, 3):
                I, J = {"ilo": ((0, live), (0, 4)), "ihi": ((4 - live, 4), (0, 4)),
                        "jlo": ((0, 4), (0, live)), "jhi": ((0, 4), (4 - live, 4))}[fam]
                for stage in (True, False):
                    out.append(emit_chunk(f"chunk_{tag}_{fam}{live}_{'stage' if stage else 'last'}", la, lb, I, J, stage))
    # the trailing update's off-diagonal tiles (uniform contraction range, no partial chunks): a 64 x 64 corner per wave
    out.append("// ---- 64 x 64 corner per wave (not spread): the trailing update's lower tiles")
    out.append(emit_run("run_ss_blk", ks_block, ks_block))
    out.append(emit_chunk("chunk_ss_full_last_blk", ks_block, ks_block, (0, 4), (0, 4), False))
    # diagonal tiles of a square lower-only product (TRIOPT + ONEOP in wv_gemm.hip): one staged operand, two wave roles
    out.append("// ---- diagonal tiles of a square lower-only product: ONE staged operand (the B pieces / pointers of the common")
    out.append("// signature are unused); low = a wave on the diagonal (accumulator blocks i >= j), half = 32 columns of the block below")
    for role, kw in (("low", dict(J=(0, 4), lower=True)), ("half", dict(J=(0, 2), lower=False))):
        out.append(emit_run(f"run_ss_{role}1", ks_block, ks_block, oneop=True, **kw))
        out.append(emit_chunk(f"chunk_ss_{role}1_last", ks_block, ks_block, (0, 4), kw["J"], False, lower=kw["lower"], oneop=True))
    out.append("// ---- the small 64 x 64 tile (four waves side by side): the rotated run of full staged chunks")
    out.append(emit_small_run("srun_ss", SmallLayout(False), SmallLayout(False)))
    out.append(emit_small_run("srun_cs", SmallLayout(True), SmallLayout(False)))
    out.append("} }  // namespace wv::kloop")
    with open(out_path, "w") as f:
        f.write("\n".join(out) + "\n")




# ---------------------------------------------------------------------------------------------------------------
# Verification of an emitted stream (run at generation time): replays the instructions against a model of the LGKM
# queue and checks that every MFMA reads fragments that (a) have landed - covered by a counted wait while never more
# than 15 LGKM operations were outstanding - and (b) are the ones its accumulator needs next: A fragment (i, k-step),
# B value (j, r, k-step) of the right chunk, every accumulator taking k-steps 0..3 of each chunk once and in order.
# ---------------------------------------------------------------------------------------------------------------
import re as _re


def verify(lines, la, lb, I, J, loop_iters=1, lower=False, small=False):
    head = lines.index("L_wv_kloop_%=:") if "L_wv_kloop_%=:" in lines else None
    if head is not None:
        tail = next(k for k, l in enumerate(lines) if l.startswith("s_cbranch_scc1"))
        seq = lines[:head] + lines[head + 1:tail] * loop_iters + lines[tail + 1:]
    else:
        seq = lines
    pending = []           # [(dst, tag)] in issue order; tag None for a write
    regs = {}              # register -> tag of the value it holds (landed)
    inflight = set()       # registers with a read in flight
    acc_steps = {}         # accumulator -> number of MFMAs applied
    toggles = {"ard": 0, "brd0": 0, "brd1": 0, "brd2": 0, "brd3": 0}   # stage (chunk parity) an address points at
    for l in seq:
        m = _re.match(r"ds_read_b64 (v\[\d+:\d+\]), %\[(\w+)\] offset:(\d+)", l)
        if m:
            dst, op, off = m.group(1), m.group(2), int(m.group(3))
            assert len(pending) < 15, "more than 15 LGKM operations outstanding"
            lay = la if op == "ard" else lb
            s, rem = divmod(off, lay.kstep) if not lay.kc else (off % lay.frag // lay.kstep, None)
            if lay.kc:
                f = off // lay.frag
            else:
                f = rem // lay.frag
                assert rem % lay.frag == 0
            assert 0 <= s < 4 and 0 <= f < 4, (l, s, f)
            tag = ("A", f, s, toggles[op]) if op == "ard" else ("B", f, int(op[3]), s, toggles[op])
            pending.append((dst, tag))
            inflight.add(dst)
            continue
        if l.startswith("ds_write_b128"):
            assert len(pending) < 15
            pending.append((None, None))
            continue
        m = _re.match(r"s_waitcnt lgkmcnt\((\d+)\)", l)
        if m:
            n = int(m.group(1))
            while len(pending) > n:
                dst, tag = pending.pop(0)
                if dst is not None:
                    regs[dst] = tag
                    if all(d != dst for d, _ in pending):
                        inflight.discard(dst)
            continue
        m = _re.match(r"v_add_u32 %\[(\w+)\], s\d+, %\[\w+\]", l)
        if m and m.group(1) in toggles:
            toggles[m.group(1)] ^= 1
            continue
        m = _re.match(r"v_mfma_f64_4x4x4_4b_f64 ([av]\[WV(?:ACC|C)(\d)\+(\d+):WV(?:ACC|C)\d\+\d+\]), (v\[\d+:\d+\]), (v\[\d+:\d+\]), ([av]\[\S+\])", l)
        if m:
            c, a, b = m.group(1), m.group(4), m.group(5)
            assert c == m.group(6)
            assert a not in inflight and b not in inflight, f"operand still in flight: {l}"
            i, jr = int(m.group(2)), int(m.group(3)) // 2
            j, r = jr >> 2, jr & 3
            n = acc_steps.get(c, 0)
            chunk, s = divmod(n, 4)
            assert regs.get(a) == ("A", i, s, chunk & 1), (l, regs.get(a), ("A", i, s, chunk & 1))
            assert regs.get(b) == ("B", j, r, s, chunk & 1), (l, regs.get(b), ("B", j, r, s, chunk & 1))
            acc_steps[c] = n + 1
    accfn = SmallRegs.acc if small else acc
    want = {accfn(i, j, r) for j in range(*J) for i in frag_rows(I, j, lower) for r in range(4)}
    assert set(acc_steps) == want and all(v == 4 * loop_iters for v in acc_steps.values()), "accumulators / k-steps"


def _self_check():
    ks, kc = Layout(False), Layout(True)
    ksx, kcx = Layout(False, spread=True), Layout(True, spread=True)
    for la, lb in ((ksx, ksx), (kcx, ksx), (ks, ks), (kc, ks)):
        for iters in (1, 2, 5):
            verify(gen_run(la, lb), la, lb, (0, 4), (0, 4), iters)
        for I, J in [((0, 4), (0, 4))] + [((0, v), (0, 4)) for v in (1, 2, 3)] + [((4 - v, 4), (0, 4)) for v in (1, 2, 3)] + \
                    [((0, 4), (0, v)) for v in (1, 2, 3)] + [((0, 4), (4 - v, 4)) for v in (1, 2, 3)]:
            for stage in (True, False):
                verify(gen_chunk(la, lb, I, J, stage), la, lb, I, J)
    for la in (SmallLayout(False), SmallLayout(True)):
        for iters in (1, 2, 5):
            verify(gen_small_run(la, SmallLayout(False)), la, SmallLayout(False), (0, 4), (0, 1), iters, small=True)
    for J, lower in (((0, 4), True), ((0, 2), False)):
        for iters in (1, 2, 5):
            verify(gen_run(ks, ks, J=J, lower=lower, oneop=True), ks, ks, (0, 4), J, iters, lower=lower)
        verify(gen_chunk(ks, ks, (0, 4), J, False, lower=lower, oneop=True), ks, ks, (0, 4), J, lower=lower)


if __name__ == "__main__":
    _self_check()
    main(sys.argv[1] if len(sys.argv) > 1 else "wv_gemm_kloop.inc")
