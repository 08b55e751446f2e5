"""Static scan for the ONE pattern round 3's fault model names (profiles/r03_fault_model.txt, r03_decoder_hazard.txt
sections 8-9): in a kernel that runs two waves per SIMD, a VALU instruction consumes the result of an LDS read in a
PROLOGUE -- code this wave reaches with no MFMA issued since its last s_barrier -- and MFMA code follows with no
s_barrier in between, so the SIMD partner (already past the prologue) can be in its LDS bursts and MFMAs while this
wave's VALU reads the just-returned registers.  That is where the failing round-2 build lost one fc_p weight (lanes 48-63
of a v_mov out of a ds_read_b128) -- with unequal wave priorities; the hardware cause is open, no s_setprio ships, and
the shipped decoder has a barrier there.  This tool checks the same shape in every matrix-core kernel: sibling of
tools/audit_mfma_war.py (same parser, control-flow graph and fixed-point data flow).

Per instruction two facts:
  prologue  (forward, MUST over all paths): no v_mfma since the last s_barrier / kernel entry;
  exposed   (backward, MAY over some path): a v_mfma is reached before any s_barrier.
A finding = first VALU (non-MFMA v_*) consumer of a ds_read destination with prologue AND exposed.  Findings are grouped
into REGIONS (the stretch between one s_barrier -- or the kernel entry -- and the first MFMA behind it, in layout order):
  * a region with >= LONG consumers is a PROLOGUE in the sense of the fault model (the decoder's tile prologue: fc_p on
    the VALU from LDS-loaded weights, ~100 consumers, v_mov / v_pk_fma out of ds_read_b128 results; the failing wave was
    ~250 instructions behind its partner).  It must end with a barrier: tests/test_isa_audit.py asserts there is none.
  * shorter regions are PHASE ENTRIES of the steady state (a slab's bias / conditioning rows read from LDS and applied on
    the VALU right behind the slab-end barrier, a dozen instructions before the phase's first MFMA).  Both waves of a SIMD
    leave the barrier together and the window is a few instructions; every kernel has them, they have run through every
    cold-process and parity test of three rounds, and a barrier behind each would cost the decoder ~40 barriers per
    tile.  Listed for the record, not treated as findings.

usage: audit_prologue_lds.py file.s kernel_symbol_substring"""
import sys

from audit_mfma_war import kernel_body, parse, regs

VALU_SKIP = ("v_mfma", "v_smfma", "v_nop", "v_readfirstlane", "v_readlane")
LONG = 32


def _cfg(ins, labels):
    n = len(ins)
    leaders = {0} | set(labels.values())
    for k, (_, mn, ops) in enumerate(ins):
        if mn.startswith(('s_cbranch', 's_branch', 's_endpgm', 's_setpc')) and k + 1 < n:
            leaders.add(k + 1)
    starts = sorted(x for x in leaders if x < n)
    block_of, blocks = {}, []
    for bi, st in enumerate(starts):
        en = starts[bi + 1] if bi + 1 < len(starts) else n
        blocks.append((st, en))
        block_of[st] = bi
    succ = []
    for (st, en) in blocks:
        _, mn, ops = ins[en - 1]
        out = []
        if mn.startswith(('s_cbranch', 's_branch')):
            tgt = ops.strip()
            if tgt in labels and labels[tgt] in block_of:
                out.append(block_of[labels[tgt]])
        if not mn.startswith(('s_branch', 's_endpgm', 's_setpc')) and en in block_of:
            out.append(block_of[en])
        succ.append(out)
    pred = [[] for _ in blocks]
    for b, ss in enumerate(succ):
        for s_ in ss:
            pred[s_].append(b)
    return blocks, succ, pred


def audit(path, sym):
    """-> (stats, findings); findings = [(asm line, consumer text, asm line of the ds_read, ds_read text)]"""
    start, body = kernel_body(path, sym)
    ins, labels = parse(body)
    blocks, succ, pred = _cfg(ins, labels)
    nb = len(blocks)
    is_mfma = lambda mn: mn.startswith(('v_mfma', 'v_smfma'))

    # ---- forward MUST: prologue at block entry
    pro_in = [True] * nb                      # optimistic start, entry block fixed True
    changed = True
    while changed:
        changed = False
        for b in range(nb):
            v = True if b == 0 else all(pro_out for pro_out in (_pro_out(ins, blocks[p], pro_in[p], is_mfma) for p in pred[b])) \
                if pred[b] else True
            if b == 0:
                v = v and True
            if v != pro_in[b]:
                pro_in[b] = v
                changed = True
    # ---- backward MAY: exposed at block exit
    exp_out = [False] * nb
    changed = True
    while changed:
        changed = False
        for b in reversed(range(nb)):
            v = any(_exp_in(ins, blocks[s_], exp_out[s_], is_mfma) for s_ in succ[b])
            if v != exp_out[b]:
                exp_out[b] = v
                changed = True
    # ---- pending LDS-read destinations (MAY, forward): reg -> instruction index of the ds_read
    pend_in = [dict() for _ in range(nb)]
    findings, stats = {}, {'mfma': 0, 'ds_reads': 0, 'valu_consumers': 0, 'prologue_consumers': 0}

    def transfer(b, state, final):
        pend = dict(state)
        st, en = blocks[b]
        pro = pro_in[b]
        # exposed per instruction needs a backward sweep of the block
        exposed = [False] * (en - st)
        e = exp_out[b]
        for k in range(en - 1, st - 1, -1):
            mn = ins[k][1]
            if mn == 's_barrier':
                e = False
            elif is_mfma(mn):
                e = True
            exposed[k - st] = e           # "an MFMA is reachable from here (this instruction included) before a barrier"
        for k in range(st, en):
            li, mn, ops = ins[k]
            if mn == 's_barrier':
                pro = True
                continue
            if is_mfma(mn):
                pro = False
                if final:
                    stats['mfma'] += 1
            o = [x.strip() for x in ops.split(',')]
            if mn.startswith(('ds_read', 'ds_load')):
                if final:
                    stats['ds_reads'] += 1
                for r in regs(o[0]):
                    pend[r] = k
                continue
            if mn.startswith('v_') and not mn.startswith(VALU_SKIP):
                src = set()
                for t in o[1:]:
                    src.update(regs(t))
                hit = [r for r in src if r in pend]
                if hit:
                    if final:
                        stats['valu_consumers'] += 1
                        if pro:
                            stats['prologue_consumers'] += 1
                        if pro and exposed[k - st]:
                            rd = pend[hit[0]]
                            findings[li] = (start + li + 1, mn + ' ' + ops, start + ins[rd][0] + 1,
                                            ins[rd][1] + ' ' + ins[rd][2])
                    for r in hit:
                        pend.pop(r, None)
            # any write kills a pending destination (first operand of v_* / loads)
            if mn.startswith(('v_', 'global_load', 'buffer_load', 'flat_load', 'scratch_load')) and o and not mn.endswith('lds'):
                for r in regs(o[0]):
                    pend.pop(r, None)
        return pend

    work = list(range(nb))
    rounds = 0
    while work and rounds < 50 * nb:
        rounds += 1
        b = work.pop(0)
        out = transfer(b, pend_in[b], False)
        for s_ in succ[b]:
            before = len(pend_in[s_])
            for r, k in out.items():
                pend_in[s_].setdefault(r, k)
            if len(pend_in[s_]) != before and s_ not in work:
                work.append(s_)
    for b in range(nb):
        transfer(b, pend_in[b], True)
    # ---- regions: findings keyed by the last s_barrier in front of them (layout order)
    barrier_lines = [li for li, mn, _ in ins if mn == 's_barrier']
    regions = {}
    for li in sorted(findings):
        prev = max([b for b in barrier_lines if b < li], default=-1)
        regions.setdefault(prev, []).append(findings[li])
    stats['regions'] = sorted((start + k + 1 if k >= 0 else 0, len(v)) for k, v in regions.items())
    long_ones = [f for v in regions.values() if len(v) >= LONG for f in v]
    stats['phase_entry_consumers'] = sum(len(v) for v in regions.values() if len(v) < LONG)
    return stats, sorted(long_ones)


def _pro_out(ins, blk, pro, is_mfma):
    for k in range(blk[0], blk[1]):
        mn = ins[k][1]
        if mn == 's_barrier':
            pro = True
        elif is_mfma(mn):
            pro = False
    return pro


def _exp_in(ins, blk, e, is_mfma):
    for k in range(blk[1] - 1, blk[0] - 1, -1):
        mn = ins[k][1]
        if mn == 's_barrier':
            e = False
        elif is_mfma(mn):
            e = True
    return e


if __name__ == '__main__':
    st, fs = audit(sys.argv[1], sys.argv[2])
    print(st)
    for f in fs[:40]:
        print("line %d: %s   <- LDS read at line %d: %s" % f)
    print("%d consumer(s) in prologue-class regions" % len(fs) if fs else "none")
    sys.exit(1 if fs else 0)
