"""Static audit: packed-fp32 VALU instructions with op_sel in a gfx950 assembly file.

v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 whose `op_sel:[...]` has a bit set (a source taken from the HIGH register of its
pair for the LOW result) return wrong values in lanes 48-63 while another wave of the same SIMD executes v_mfma instructions
(tools/hazard/pk_f32_under_mfma.hip; op_sel_hi alone, v_pk_mov_b32 and the plain forms are not affected).
usage: audit_pk_f32.py file.s [file.s ...]  -> lists (kernel, line, instruction); exit status 1 if any."""
import re
import sys

FORMS = re.compile(r'\b(v_pk_(?:fma|mul|add)_f32)\b')
OP_SEL = re.compile(r'\bop_sel:\[([01,]+)\]')


def audit(path):
    """-> ([(kernel symbol, line number, instruction)], number of kernels seen)"""
    found, cur, kernels = [], None, set()
    for i, l in enumerate(open(path), 1):
        m = re.match(r'^(_Z\S+|[A-Za-z_]\w*):\s', l)
        if m and not l.startswith('.'):
            cur = m.group(1)
        if '.amdhsa_kernel' in l:
            kernels.add(l.split()[-1])
        s = l.split(';')[0]
        if FORMS.search(s):
            m = OP_SEL.search(s)
            if m and '1' in m.group(1):
                found.append((cur, i, s.strip()))
    return found, len(kernels)


if __name__ == "__main__":
    bad = 0
    for p in sys.argv[1:]:
        found, n = audit(p)
        print("%s: %d kernels, %d packed-fp32 instructions with op_sel" % (p, n, len(found)))
        for k, i, s in found[:20]:
            print("  %s:%d  %s" % (k, i, s))
        bad += len(found)
    sys.exit(1 if bad else 0)
