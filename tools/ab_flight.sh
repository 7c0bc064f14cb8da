#!/bin/bash
# flight-only A/B: abf.sh outdir variant...
O=$1; shift; mkdir -p $O
for i in 1 2; do for v in default "$@"; do
  if [ $v = default ]; then L=""; else L=flybody_amd/csrc/variants/lib$v.so; fi
  a=$(FLYBODY_ENV_LIB=$L timeout -k 10 120 python bench.py --no-cpu-baseline --steps 300 --warmup 50 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel_ms'])")
  echo "$v flight $a" | tee -a $O/ab.log
done; done
