"""Static audit of MFMA-source write-after-read distances in a gfx950 kernel's assembly.

For every instruction that writes VGPRs ASYNCHRONOUSLY (ds_read*, global/buffer/scratch loads that are not LDS-DMA:
their data lands whenever the memory pipe returns it, not in program order with the matrix pipe) find the most recent
v_mfma that READ one of those registers as SrcA / SrcB / SrcC and report the distance in issued MFMAs and in wait
states.  A conservative invariant of the shipped decoder's generated code (profiles/r03_decoder_hazard.txt): it was
designed against the round's first attribution of round 2's failure, which section 7 there refutes -- the failing build
violates it 213 times, but repairing the violations does not repair that build.  Rules:
  A/B: a load destination must not alias SrcA or SrcB of any of the last `min_mfma_gap` MFMAs, unless that MFMA is
       at least 32 wait states back (round 2's failing decoder reused the registers of the MFMAs issued 0-6 wait
       states earlier);
  C:   a load destination that aliases SrcC of a recent MFMA must be at least `min_c_states` wait states behind it
       (the distance hipcc's own hazard recognizer keeps; inline-asm loads are not padded by hipcc, so this is the
       check that an asm-issued ds_read did not land on a just-read accumulator).
An s_barrier resets the history (the partner wave has to arrive too, so nothing of the previous phase is still queued).
Basic-block data flow over the kernel's control-flow graph (per MFMA the smallest distance over all paths, to a fixed
point), so loop-carried and branch-joined distances are covered.
usage: audit_mfma_war.py file.s kernel_symbol_substring [min_mfma_gap=6] [min_c_states=6]"""
import re
import sys

AB_STATES_OK = 32     # ... or this many wait states behind the reading MFMA (round 2: 32 wait states in front of the
                      # prefetch made the failing build clean)
LOADS = ("ds_read", "ds_load", "global_load", "buffer_load", "flat_load", "scratch_load")


def regs(tok):
    out = []
    for m in re.finditer(r'\bv\[(\d+):(\d+)\]|\bv(\d+)\b', tok):
        if m.group(1) is not None:
            out += list(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.append(int(m.group(3)))
    return out


def kernel_body(path, sym):
    lines = open(path).read().splitlines()
    start = next(i for i, l in enumerate(lines) if re.match(r'^_Z\S*:', l) and sym in l)
    end = next(i for i in range(start, len(lines)) if 's_endpgm' in lines[i])
    return start, lines[start:end + 1]


def parse(body):
    """-> list of (line_index, mnemonic, operand string) of real instructions, label -> position in that list."""
    ins, labels = [], {}
    for i, l in enumerate(body):
        s = l.split(';')[0].strip() if not l.strip().startswith(';;#') else ''
        m = re.match(r'^(\.LBB\S+):', l)
        if m:
            labels[m.group(1)] = len(ins)
            continue
        if not s or s.endswith(':') or s.startswith('.'):
            continue
        parts = s.split(None, 1)
        ins.append((i, parts[0], parts[1] if len(parts) > 1 else ''))
    return ins, labels


def audit(path, sym, min_mfma_gap=6, min_c_states=6):
    """-> (stats, problems).  Basic-block data flow: the state at a block's entry is the union of its predecessors'
    exit states (per MFMA the smallest distance over all paths), iterated to a fixed point."""
    start, body = kernel_body(path, sym)
    ins, labels = parse(body)
    n = len(ins)
    horizon = max(min_mfma_gap, 8)
    # ---- basic blocks
    leaders = {0} | set(labels.values())
    for k, (_, mn, ops) in enumerate(ins):
        if mn.startswith(('s_cbranch', 's_branch', 's_endpgm', 's_setpc')) and k + 1 < n:
            leaders.add(k + 1)
    starts = sorted(x for x in leaders if x < n)
    block_of = {}
    blocks = []
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

    problems, stats = {}, {'mfma': 0, 'loads': 0, 'min_ab_gap': None, 'min_c_states': None}

    def transfer(bi, state, final):
        """state: {mfma line: [a, b, c, states, mfmas]} -> exit state"""
        hist = {k: list(v) for k, v in state.items()}
        st, en = blocks[bi]
        for k in range(st, en):
            li, mn, ops = ins[k]
            states = 1
            if mn == 's_barrier':
                # every wave of the workgroup (the SIMD partner included) has drained its phase: the MFMAs before a
                # barrier that follows an s_waitcnt on the weight stream are hundreds of cycles old when it releases
                hist = {}
                continue
            if mn == 's_nop':
                states = int(ops.strip() or 0) + 1
            if mn.startswith(LOADS) and not mn.endswith('lds') and ' lds' not in ops:
                dst = set(regs(ops.split(',')[0]))
                if final:
                    stats['loads'] += 1
                for ml, h in hist.items():
                    for kind, regset in (('a', h[0]), ('b', h[1]), ('c', h[2])):
                        if not (dst & regset):
                            continue
                        if kind == 'c':
                            if stats['min_c_states'] is None or h[3] < stats['min_c_states']:
                                stats['min_c_states'] = h[3]
                            if h[3] < min_c_states:
                                problems[(li, ml, 'SrcC')] = (start + li + 1, mn + ' ' + ops, start + ml + 1, 'SrcC', h[4], h[3])
                        else:
                            if stats['min_ab_gap'] is None or h[4] < stats['min_ab_gap']:
                                stats['min_ab_gap'] = h[4]
                            if h[4] < min_mfma_gap and h[3] < AB_STATES_OK:
                                problems[(li, ml, kind)] = (start + li + 1, mn + ' ' + ops, start + ml + 1,
                                                            'Src' + kind.upper(), h[4], h[3])
            for h in hist.values():
                h[3] += states
            if mn.startswith('v_mfma') or mn.startswith('v_smfma'):
                o = [x.strip() for x in ops.split(',')]
                if final:
                    stats['mfma'] += 1
                for h in hist.values():
                    h[4] += 1
                hist[li] = [frozenset(regs(o[1])), frozenset(regs(o[2])),
                            frozenset(regs(o[3])) if len(o) > 3 else frozenset(), 0, 0]
                hist = {ml: h for ml, h in hist.items() if h[4] <= horizon and (h[3] <= 64 or h[4] < min_mfma_gap)}
        return hist

    def merge(dst, src):
        changed = False
        for ml, h in src.items():
            d = dst.get(ml)
            if d is None:
                dst[ml] = list(h)
                changed = True
            else:
                if h[3] < d[3]:
                    d[3] = h[3]
                    changed = True
                if h[4] < d[4]:
                    d[4] = h[4]
                    changed = True
        return changed

    entry = [dict() for _ in blocks]
    work = list(range(len(blocks)))
    rounds = 0
    while work and rounds < 50 * len(blocks):
        rounds += 1
        bi = work.pop(0)
        out = transfer(bi, entry[bi], False)
        for sj in succ[bi]:
            if merge(entry[sj], out) and sj not in work:
                work.append(sj)
    problems.clear()
    stats.update({'mfma': 0, 'loads': 0, 'min_ab_gap': None, 'min_c_states': None})
    for bi in range(len(blocks)):
        transfer(bi, entry[bi], True)
    return stats, sorted(problems.values())


if __name__ == '__main__':
    gap = int(sys.argv[3]) if len(sys.argv) > 3 else 6
    cst = int(sys.argv[4]) if len(sys.argv) > 4 else 6
    st, pr = audit(sys.argv[1], sys.argv[2], gap, cst)
    print(st)
    for p in pr[:40]:
        print("line %d: %s  <- MFMA at line %d read it as %s, %d MFMAs / %d wait states earlier" % p)
    print("%d problem(s)" % len(pr))
    sys.exit(1 if pr else 0)
