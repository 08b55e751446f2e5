"""Cold-process correctness sweep of decoder builds (the net for the round-2 priority failure):
   python tools/ab/prio_check.py N name1 name2 ...     (names as in tools/ab/dec_ab.py; 'base' = shipped library)
Each of the N processes per build is a fresh GPU context running tools/dbg_map.py (8 launches of 8 x 1024 points,
majority vote per point); a process is BAD if any launch has a point off the majority by more than 1e-5."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
n = int(sys.argv[1])
for name in sys.argv[2:]:
    env = dict(os.environ, GRAFT_REPO_ROOT=ROOT)
    if name != "base":
        env["RFD_HIP_LIB"] = os.path.join(ROOT, "rfdnet_amd", "lib", "variants", "librfd_%s.so" % name)
    bad, detail, err = 0, [], 0
    for i in range(n):
        try:      # a hand-edited build may hang: never let one process eat the GPU box's time limit
            r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "dbg_map.py")], env=env, capture_output=True,
                               text=True, timeout=240)
        except subprocess.TimeoutExpired:
            err += 1
            detail.append("timed out after 240 s")
            break
        lines = [l for l in r.stdout.splitlines() if "BAD" in l]
        if r.returncode != 0:
            err += 1
            detail.append("rc %d: %s" % (r.returncode, (r.stderr or "")[-300:]))
        elif lines:
            bad += 1
            if len(detail) < 3:
                detail.append(lines[0][:300])
    print("PRIOCHECK %-22s %d/%d processes BAD, %d errored" % (name, bad, n, err))
    for d in detail:
        print("    " + d)
    sys.stdout.flush()
