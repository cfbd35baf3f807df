for n in 256 248 240 224 208 192; do
  echo "ENC_CUS=$n: $(VITCAP_ENC_CUS=$n python bench.py --steps 40 --warmup 5 --no-cpu-baseline --isolated 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'])")"
done
