"""Static audit of hand-issued global loads in a gfx950 kernel's assembly: every VGPR written by a
`global_load_*` (not `... lds`) must not be touched again before an `s_waitcnt vmcnt(N)` has
retired that load (vector-memory operations retire in order).  Linear scan in program order;
a loop body is re-scanned once with the state at its back edge.
usage: audit_vmcnt.py file.s kernel_symbol_substring"""
import re, sys

def regs(tok):
    out = []
    for m in re.finditer(r'\bv\[(\d+):(\d+)\]|\bv(\d+)\b', tok):
        if m.group(1) is not None:
            out += list(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.append(int(m.group(3)))
    return out

def audit(path, sym):
    """-> (counts of what was scanned, list of (line, instruction, load line))."""
    lines = open(path).read().splitlines()
    start = next(i for i, l in enumerate(lines) if l.startswith('_Z') and sym in l and l.rstrip().endswith(':') or
                 (sym in l and re.match(r'^_Z\S+:\s', l)))
    end = next(i for i in range(start, len(lines)) if 's_endpgm' in lines[i])
    body = lines[start:end + 1]
    labels = {}
    for i, l in enumerate(body):
        m = re.match(r'^(\.LBB\S+):', l)
        if m:
            labels[m.group(1)] = i
    outstanding = []          # (line, [dst regs] or None for DMA / other)
    problems = []
    seen = {'loads': 0, 'dma_or_stores': 0, 'waits': 0}

    def step(i, l):
        s = l.split(';')[0].strip()
        if not s or s.endswith(':') or s.startswith('.'):
            return
        parts = s.split(None, 1)
        mn = parts[0]
        ops = parts[1] if len(parts) > 1 else ''
        if mn == 's_waitcnt':
            m = re.search(r'vmcnt\((\d+)\)', ops)
            if m:
                seen['waits'] += 1
                n = int(m.group(1))
                while len(outstanding) > n:
                    outstanding.pop(0)
            return
        touched = set(regs(ops))
        for (li, dst) in outstanding:
            if dst and touched & set(dst):
                problems.append((start + i + 1, s, start + li + 1))
        if mn.startswith(('global_load', 'buffer_load', 'flat_load', 'scratch_load')):
            if 'lds' in mn:
                seen['dma_or_stores'] += 1
                outstanding.append((i, None))
            else:
                seen['loads'] += 1
                outstanding.append((i, regs(ops.split(',')[0])))
        elif mn.startswith(('global_store', 'buffer_store', 'flat_store', 'scratch_store', 'global_atomic', 'buffer_atomic')):
            seen['dma_or_stores'] += 1
            outstanding.append((i, None))     # gfx9: stores count in vmcnt too

    i = 0
    rescanned = set()
    n = len(body)
    while i < n:
        l = body[i]
        step(i, l)
        m = re.search(r's_cbranch_\S+\s+(\.LBB\S+)|s_branch\s+(\.LBB\S+)', l)
        if m:
            tgt = labels.get(m.group(1) or m.group(2))
            if tgt is not None and tgt < i and (tgt, i) not in rescanned:
                rescanned.add((tgt, i))
                for j in range(tgt, i + 1):
                    step(j, body[j])
        i += 1
    seen['lines'] = n
    return seen, problems


def main(path, sym):
    seen, problems = audit(path, sym)
    print("%s: %d lines, %d register loads, %d LDS-DMA / stores, %d vmcnt waits scanned (loop bodies twice); "
          "%d potential use-before-landed" % (sym, seen['lines'], seen['loads'], seen['dma_or_stores'], seen['waits'],
                                              len(problems)))
    for p in problems[:20]:
        print("  line %d: %s   (load at line %d)" % p)

if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2])
