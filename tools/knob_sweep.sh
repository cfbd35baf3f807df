for v in "" "VITCAP_ENCODE_SPLIT=2" "VITCAP_ENCODE_SPLIT=3" "VITCAP_DECODE_PRIORITY=0" "VITCAP_TAG_FORK=0"; do
  echo "== [$v]"
  env $v python bench.py --steps 100 --warmup 5 --no-cpu-baseline --isolated 0 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])"
done
