R=$PWD
mkdir -p gpurun_out/c31
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace -d $R/gpurun_out/c31/kt -- python $R/bench.py --no-cpu-baseline --no-latency --steps 6 --warmup 2 > $R/gpurun_out/c31/bench.json 2> $R/gpurun_out/c31/kt.err
cd $R
DB=$(find gpurun_out/c31/kt -name "*.db" | head -1)
python tools/gpu_timeline.py $DB > gpurun_out/c31/timeline.txt 2>&1
python - "$DB" > gpurun_out/c31/copies.txt 2>&1 <<'PY'
import sqlite3, sys
con = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')")]
print([t for t in tabs if 'copy' in t.lower() or 'memory' in t.lower()])
for t in tabs:
    if 'memory_copy' in t.lower():
        cols = [r[1] for r in con.execute("pragma table_info(%s)" % t)]
        print(t, cols)
        rows = con.execute("select * from %s" % t).fetchall()
        print(len(rows))
        si, ei = cols.index('start'), cols.index('end')
        szi = cols.index('size') if 'size' in cols else None
        big = sorted(rows, key=lambda r: -(r[ei] - r[si]))[:15]
        for r in big:
            print("%.3f ms" % ((r[ei] - r[si]) / 1e6), r[szi] if szi is not None else '', [r[i] for i in range(len(cols)) if cols[i] in ('name', 'kind', 'src_agent_id', 'dst_agent_id')])
        break
PY
rm -rf gpurun_out/c31/kt
cat gpurun_out/c31/timeline.txt; head -30 gpurun_out/c31/copies.txt; cut -c1-200 gpurun_out/c31/bench.json
