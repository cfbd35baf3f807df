bash tools/prof_script.sh libk295 tools/lib_kernel_names.py 295424
bash tools/prof_script.sh libk36 tools/lib_kernel_names.py 36928
python - <<'PY'
import sqlite3,glob
for d in ('gpurun_out/prof_libk295/p_results.db','gpurun_out/prof_libk36/p_results.db'):
    db=sqlite3.connect(d)
    with open(d.replace('/p_results.db','_names.txt'),'w') as f:
        for r in db.execute('select name,total_calls,average from top_kernels'):
            f.write('%s | %d | %.1f\n'%r)
PY
rm -rf gpurun_out/prof_libk295 gpurun_out/prof_libk36
python bench.py > gpurun_out/bench_r04_base.json 2> gpurun_out/bench_r04_base.err
python tools/gemm_bench.py 5 36928,295424 > gpurun_out/gemm_bench_r04_base.txt 2>&1
