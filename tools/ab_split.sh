for b in 16 32 128 512; do
for sp in 0 2; do
echo "batch=$b parts=$sp"
VITCAP_ENCODE_SPLIT=$sp python bench.py --steps 10 --warmup 2 --batch $b --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done
done
