for b in 64 512; do
echo "tiles batch=$b"
python bench.py --steps 6 --warmup 2 --pipeline 0 --gemm-tiles 1 --batch $b --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['end_to_end_frac_of_bf16_peak'], d['roofline']['frac'])"
done
echo "pipelined 512"
python bench.py --steps 6 --warmup 2 --batch 512 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['end_to_end_frac_of_bf16_peak'], d['roofline']['frac'])"
