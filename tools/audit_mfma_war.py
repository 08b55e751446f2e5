"""Static audit of MFMA-source write-after-read distances in a gfx950 kernel's assembly.

For every instruction that writes VGPRs ASYNCHRONOUSLY (ds_read*, global/buffer/scratch loads that are not LDS-DMA:
their data lands whenever the memory pipe returns it, not in program order with the matrix pipe) find the most recent
v_mfma that READ one of those registers as SrcA / SrcB / SrcC and report the distance in issued MFMAs and in wait
states.  Rules (profiles/r03_decoder_hazard.txt):
  A/B: a load destination must not alias SrcA or SrcB of any of the last `min_mfma_gap` MFMAs
       (round 2's failing decoder reused the registers of the MFMAs issued 3-6 wait states earlier);
  C:   a load destination that aliases SrcC of a recent MFMA must be at least `min_c_states` wait states behind it
       (the distance hipcc's own hazard recognizer keeps; inline-asm loads are not padded by hipcc, so this is the
       check that an asm-issued ds_read did not land on a just-read accumulator).
An s_barrier resets the history (the partner wave has to arrive too, so nothing of the previous phase is still queued).
Linear scan in program order; at every backward branch the target block is re-scanned once with the history at the
branch (loop-carried distances).
usage: audit_mfma_war.py file.s kernel_symbol_substring [min_mfma_gap=6] [min_c_states=6]"""
import re
import sys

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
    start, body = kernel_body(path, sym)
    ins, labels = parse(body)
    problems, stats = [], {'mfma': 0, 'loads': 0, 'min_ab_gap': None, 'min_c_states': None}

    def scan(lo, hi, hist, count):
        # hist: list of dicts {line, a, b, c, states_since, mfmas_since}, most recent last
        for k in range(lo, hi):
            li, mn, ops = ins[k]
            states = 1
            if mn == 's_barrier':
                # every wave of the workgroup (the SIMD partner included) has drained its phase: the MFMAs before a
                # barrier that follows an s_waitcnt on the weight stream are hundreds of cycles old when it releases
                hist[:] = []
                continue
            if mn == 's_nop':
                states = int(ops.strip() or 0) + 1
            if mn.startswith(LOADS) and not mn.endswith('lds') and ' lds' not in ops:
                dst = set(regs(ops.split(',')[0]))
                if count:
                    stats['loads'] += 1
                for h in hist:
                    for kind in ('a', 'b', 'c'):
                        if dst & h[kind]:
                            if kind == 'c':
                                if stats['min_c_states'] is None or h['states'] < stats['min_c_states']:
                                    stats['min_c_states'] = h['states']
                                if h['states'] < min_c_states:
                                    problems.append((start + li + 1, mn + ' ' + ops, start + h['line'] + 1, 'SrcC',
                                                     h['mfmas'], h['states']))
                            else:
                                if stats['min_ab_gap'] is None or h['mfmas'] < stats['min_ab_gap']:
                                    stats['min_ab_gap'] = h['mfmas']
                                if h['mfmas'] < min_mfma_gap:
                                    problems.append((start + li + 1, mn + ' ' + ops, start + h['line'] + 1,
                                                     'Src' + kind.upper(), h['mfmas'], h['states']))
            for h in hist:
                h['states'] += states
            if mn.startswith('v_mfma') or mn.startswith('v_smfma'):
                o = [x.strip() for x in ops.split(',')]
                if count:
                    stats['mfma'] += 1
                for h in hist:
                    h['mfmas'] += 1
                hist.append({'line': li, 'a': set(regs(o[1])), 'b': set(regs(o[2])),
                             'c': set(regs(o[3])) if len(o) > 3 else set(), 'states': 0, 'mfmas': 0})
                # far enough back on both scales: forget
                hist[:] = [h for h in hist if h['mfmas'] <= max(min_mfma_gap, 8) and h['states'] <= 64 or h['mfmas'] < min_mfma_gap]
            m = re.match(r'^s_cbranch\S*|^s_branch', mn)
            if m and count:
                tgt = ops.strip()
                if tgt in labels and labels[tgt] <= k:
                    # loop back edge: re-scan the head of the loop with the history here
                    scan(labels[tgt], min(labels[tgt] + 400, k), [dict(h) for h in hist], False)
        return hist

    scan(0, len(ins), [], True)
    # de-duplicate (the back-edge re-scan can repeat findings)
    seen, uniq = set(), []
    for p in problems:
        if p[:4] not in seen:
            seen.add(p[:4])
            uniq.append(p)
    return stats, uniq


if __name__ == '__main__':
    gap = int(sys.argv[3]) if len(sys.argv) > 3 else 6
    cst = int(sys.argv[4]) if len(sys.argv) > 4 else 6
    st, pr = audit(sys.argv[1], sys.argv[2], gap, cst)
    print(st)
    for p in pr[:40]:
        print("line %d: %s  <- MFMA at line %d read it as %s, %d MFMAs / %d wait states earlier" % p)
    print("%d problem(s)" % len(pr))
    sys.exit(1 if pr else 0)
