# A/B of the large-GEMM kernel inside the real pipeline: VITCAP_GEMM_4W="<form for tile_hint 5>,<form for auto>" (-1 = 8-wave kernel)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/ab_4w.txt; : > $OUT
for v in "-1,-1" "-1,2" "-1,1"; do
  for args in "--steps 30 --warmup 3 --pipeline 0" "--steps 8 --warmup 2 --pipeline 0 --batch 512"; do
    echo "== VITCAP_GEMM_4W=$v  $args" >> $OUT
    VITCAP_GEMM_4W=$v python bench.py $args --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
r = d['roofline']
print('   value %.1f img/s  ms/step %.3f  e2e frac %.4f  dominant %s frac %.4f decode_ms %.3f' % (d['value'], d['ms_per_step'], d['end_to_end_frac_of_bf16_peak'], r['kernel'], r['frac'], d['decode_phase_ms_per_batch']))" >> $OUT 2>&1
  done
done
