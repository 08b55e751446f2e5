"""Assembly-level bisect of the round-2 FAILING decoder build (dev container: needs the git history).
tools/ab/make_hist_variants.py perturbs the SOURCE, which lets hipcc re-allocate registers and re-schedule the whole
kernel; here the failing build's device assembly is edited by hand and re-assembled, so a variant differs from the failing
binary in exactly the lines named:
  a0     no edit (round trip through the assembler: must fail like fdprio)
  up     the five MFMAs hipcc sank below the slab-end s_barrier moved back in front of it
  upK    only the first K of them moved
  w0     the three partial s_waitcnt lgkmcnt(n) in front of the faulty prologue moves (section 8) made full waits
  pb     s_barrier + 7 s_nop between the tile prologue and the block-input code (control: ins581x8, the same 32 bytes
         without the barrier)
  bn32   32 wait states between that s_barrier and the five MFMAs
  tn64   64 wait states behind the five MFMAs on BOTH paths (in front of the s_cbranch), none at the loads
-> rfdnet_amd/lib/variants/librfd_fdasm_<name>.so; tools/ab/prio_check.py runs them.
Pipeline: hipcc -S --cuda-device-only -> edit -> clang -x assembler -> lld -> clang-offload-bundler -> host object with
-fcuda-include-gpubinary -> link with the other sources."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from rfdnet_amd import build as B  # noqa: E402

LL = "/opt/rocm/lib/llvm/bin/"
HIPCC = "/opt/rocm/bin/hipcc"
KERNEL = "_ZN12_GLOBAL__N_118occ_decode8_kernelILi3E"


def slab_end(lines):
    """-> (index of the s_barrier, indices of the MFMAs sunk below it, index of the s_cbranch) inside the X3 kernel."""
    st = next(i for i, l in enumerate(lines) if l.startswith(KERNEL))
    en = next(i for i in range(st, len(lines)) if "s_endpgm" in lines[i])
    for i in range(st, en):
        if lines[i].split(";")[0].strip() != "s_barrier":
            continue
        j, sunk = i + 1, []
        while lines[j].split(";")[0].strip().startswith("v_mfma"):
            sunk.append(j)
            j += 1
        if len(sunk) == 5 and lines[j].split(";")[0].strip().startswith("s_cbranch_scc1"):
            return i, sunk, j
    raise SystemExit("slab-end pattern not found")


def kernel_range(lines):
    st = next(i for i, l in enumerate(lines) if l.startswith(KERNEL))
    en = next(i for i in range(st, len(lines)) if "s_endpgm" in lines[i])
    return st, en


def real_instructions(lines):
    """File line numbers of the X3 kernel's instructions (labels, directives, comments skipped)."""
    st, en = kernel_range(lines)
    out = []
    for i in range(st + 1, en + 1):
        t = lines[i].split(";")[0].strip()
        if t and not t.startswith(".") and not t.endswith(":"):
            out.append(i)
    return out


def out_is_branch(line):
    return line.split(";")[0].strip().startswith("s_branch")


def everywhere(lines, match, before=(), after=(), pick=None):
    """Insert instructions before / after every instruction of the X3 kernel whose mnemonic starts with `match`
    (pick: optional predicate on the running site number, for bisecting over sites)."""
    st, en = kernel_range(lines)
    out, site = list(lines[:st]), 0
    for l in lines[st:en]:
        ins = l.split(";")[0].strip()
        hit = ins.startswith(match)
        if hit:
            site += 1
            hit = pick is None or pick(site - 1)
        if hit:
            out += ["\t" + b for b in before]
        out.append(l)
        if hit:
            out += ["\t" + a for a in after]
    return out + list(lines[en:]), site


GLOBAL_EDITS = {
    # name: (mnemonic prefix, before, after)
    "lg0": ("ds_read", (), ("s_waitcnt lgkmcnt(0)",)),          # every LDS read lands before anything else issues
    "vm0": ("global_load_lds", (), ("s_waitcnt vmcnt(0)",)),    # every LDS-DMA piece lands before anything else issues
    "dn1": ("global_load_lds", (), ("s_nop 0",)),               # round 2's accidental remedy, without the re-schedule
    "mn4": ("v_mfma", (), ("s_nop 3",)),                        # 4 wait states behind every MFMA
    "pn8": ("ds_read_b128", ("s_nop 7",), ()),                  # 8 wait states in front of every fragment / table read
    "noprio": ("s_setprio", ("s_nop 0",), ()),                  # handled below: the s_setprio itself is dropped
}


def edit(lines, name):
    if name == "a0":
        return list(lines)
    if name.split("@")[0] in GLOBAL_EDITS:
        # name@lo-hi restricts the edit to sites lo..hi-1 (bisect over sites)
        base, _, rng = name.partition("@")
        m, b, a = GLOBAL_EDITS[base]
        pick = None
        if rng:
            lo, hi = (int(x) for x in rng.split("-"))
            pick = lambda k: lo <= k < hi   # noqa: E731
        if base == "noprio":
            st, en = kernel_range(lines)
            return [l for i, l in enumerate(lines) if not (st <= i < en and l.split(";")[0].strip().startswith("s_setprio"))]
        out, n = everywhere(lines, m, b, a, pick)
        print("  %s: %d sites" % (name, n))
        return out
    if name.startswith("ins"):            # insK[xN]: N s_nop 0 (4 bytes each) in front of the K-th instruction of the kernel
        k, _, n = name[3:].partition("x")
        idx = real_instructions(lines)[int(k)]
        return lines[:idx] + ["\ts_nop 0"] * int(n or 1) + lines[idx:]
    if name.startswith("swap"):           # swapK: exchange instructions K and K+1 (caller checks they are independent)
        r = real_instructions(lines)
        k = int(name[4:])
        out = list(lines)
        out[r[k]], out[r[k + 1]] = lines[r[k + 1]], lines[r[k]]
        return out
    if name.startswith("mv"):             # mvA_B: instruction A moved behind instruction B (caller checks the data flow)
        r = real_instructions(lines)
        a, b = (int(x) for x in name[2:].split("_"))
        out = list(lines)
        moved = out[r[a]]
        out.insert(r[b] + 1, moved)
        del out[r[a] if a < b else r[a] + 1]
        return out
    if name == "w0":                      # the three partial LDS waits in front of the faulty moves -> full waits (same size)
        r = real_instructions(lines)
        out = list(lines)
        for k in (541, 560, 563):
            assert out[r[k]].split(";")[0].strip().startswith("s_waitcnt lgkmcnt("), out[r[k]]
            out[r[k]] = "\ts_waitcnt lgkmcnt(0)"
        return out
    if name == "pb":                      # s_barrier + 7 s_nop (32 bytes) between the tile prologue and the block-input code
        r = real_instructions(lines)
        k = 581
        assert out_is_branch(lines[r[k]]), lines[r[k]]
        return lines[:r[k]] + ["\ts_barrier"] + ["\ts_nop 0"] * 7 + lines[r[k]:]
    if name.startswith("pn_top"):         # N wait states in front of the FIRST LDS-DMA only (same delay, other place)
        n = int(name[6:])
        st, en = kernel_range(lines)
        first = next(i for i in range(st, en) if lines[i].split(";")[0].strip().startswith("global_load_lds"))
        return lines[:first - 2] + ["\ts_nop 0"] * n + lines[first - 2:]
    bar, sunk, br = slab_end(lines)
    out = list(lines)
    if name == "a0":
        pass
    elif name == "up":
        moved = [out[k] for k in sunk]
        out = out[:bar] + moved + [out[bar]] + out[sunk[-1] + 1:]
    elif name.startswith("up"):           # upK: only the first K of the five
        k = int(name[2:])
        moved = [out[i] for i in sunk[:k]]
        out = out[:bar] + moved + [out[bar]] + out[sunk[k - 1] + 1:]
    elif name.startswith("bn"):
        n = int(name[2:])
        out = out[:bar + 1] + ["\ts_nop 15"] * (n // 16) + out[bar + 1:]
    elif name.startswith("tn"):
        n = int(name[2:])
        out = out[:br] + ["\ts_nop 15"] * (n // 16) + out[br:]
    else:
        raise SystemExit("unknown variant " + name)
    return out


def main(names, shipped=False, cur_tag="shasm", cur_flags=("-DDEC8_PRIO=1",)):
    """shipped: the CURRENT occ_decoder8.hip with -DDEC8_PRIO=1 (the shipped structure + the static priority) instead of
    the historic failing source; libraries are then called librfd_shasm_<name>.so."""
    out_dir = os.path.join(B.LIB_DIR, "variants")
    tmp = os.path.join(out_dir, "_hist")
    os.makedirs(tmp, exist_ok=True)
    flags = [f for f in B.HIPCC_FLAGS if f not in ("-shared",)]
    if shipped:
        src = os.environ.get("RFD_ASM_SRC") or os.path.join(ROOT, "rfdnet_amd", "csrc", "occ_decoder8.hip")
        flags = flags + list(cur_flags)
        base_s = os.path.join(tmp, cur_tag + ".dev.s")
    else:
        src = os.path.join(tmp, "occ_decoder8_fdprio.hip")
        if not os.path.exists(src):
            subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "ab", "make_hist_variants.py"), "fdprio"])
        base_s = os.path.join(tmp, "fdprio.dev.s")
    tag = cur_tag if shipped else "fdasm"
    subprocess.check_call([HIPCC] + flags + ["-S", "--cuda-device-only", "-o", base_s, src], stderr=subprocess.DEVNULL)
    lines = open(base_s).read().splitlines()
    # the other translation units as objects, once (hipcc would take an object on a -x hip command line for source)
    from concurrent.futures import ThreadPoolExecutor
    obj_dir = os.path.join(tmp, "others")
    os.makedirs(obj_dir, exist_ok=True)

    def obj(path):
        o = os.path.join(obj_dir, os.path.basename(path) + ".o")
        if not os.path.exists(o) or os.path.getmtime(o) < os.path.getmtime(path):
            subprocess.check_call([HIPCC] + flags + ["-c", path, "-o", o], stderr=subprocess.DEVNULL)
        return o
    with ThreadPoolExecutor(8) as ex:
        others = list(ex.map(obj, [s for s in B.sources() if not s.endswith("occ_decoder8.hip")]))
    for name in names:
        p = os.path.join(tmp, "%s_%s" % (tag, name.replace("@", "_")))
        open(p + ".s", "w").write("\n".join(edit(lines, name)) + "\n")
        subprocess.check_call([LL + "clang", "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c",
                               p + ".s", "-o", p + ".dev.o"])
        subprocess.check_call([LL + "lld", "-flavor", "gnu", "-m", "elf64_amdgpu", "--no-undefined", "-shared",
                               "-o", p + ".hsaco", p + ".dev.o"])
        subprocess.check_call([LL + "clang-offload-bundler", "-type=o", "-bundle-align=4096",
                               "-targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950",
                               "-input=/dev/null", "-input=" + p + ".hsaco", "-output=" + p + ".hipfb"])
        subprocess.check_call([HIPCC] + flags + ["--cuda-host-only", "-c", src, "-Xclang", "-fcuda-include-gpubinary",
                                                 "-Xclang", p + ".hipfb", "-o", p + ".host.o"], stderr=subprocess.DEVNULL)
        subprocess.check_call([HIPCC, "-shared", "-fPIC", "-o", os.path.join(out_dir, "librfd_%s_%s.so" % (tag, name.replace("@", "_")))]
                              + others + [p + ".host.o"], stderr=subprocess.DEVNULL)
        print(tag + "_" + name, "built")


if __name__ == "__main__":
    args = sys.argv[1:]
    if args and args[0] == "--shipped-prio":
        main(args[1:], shipped=True)
    elif args and args[0] == "--current":     # --current TAG "-DFOO=1 -DBAR=0" names...: today's source with these flags
        main(args[3:], shipped=True, cur_tag=args[1], cur_flags=tuple(args[2].split()))
    else:
        main(args or ["a0", "up", "bn32", "tn64"])
