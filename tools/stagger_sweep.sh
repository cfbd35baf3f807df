for s in "0,0,0" "2,50,0" "4,25,0" "4,20,0" "8,12,0" "2,50,1" "4,25,1" "3,33,0" "4,15,1"; do VITCAP_GEMM_STAGGER=$s python tools/stagger_bench.py 2>&1 | grep STAGGER; done
