# grid shapes of the runtime's blit kernels in the steady-state headline (how many CUs does the mesh copy take?)
R=$PWD; O=$PWD/gpurun_out/r6p3; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d $O/kt -o kt -- python $R/bench.py --no-cpu-baseline --no-extras --no-latency --steps 4 --warmup 2 > $O/bench.json 2> $O/bench.err
DB=$(find $O/kt -name "*.db" | head -1)
python - <<P
import sqlite3
con = sqlite3.connect("$DB"); cur = con.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(rocpd_kernel_dispatch)")]
print(cols)
q = ("select s.kernel_name, d.grid_size_x, d.workgroup_size_x, count(*), sum(d.end-d.start)/1e6, min(d.end-d.start)/1e3, max(d.end-d.start)/1e3 "
     "from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id "
     "where s.kernel_name like '%rocclr%' group by 1,2,3 order by 5 desc limit 25")
for r in cur.execute(q): print(r)
P
rm -rf $O/kt
