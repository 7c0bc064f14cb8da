#!/bin/bash
# usage: ab.sh outdir variant... ; runs flight + ball bench for default lib and each variant, twice
O=$1; shift
mkdir -p $O
for i in 1 2; do
for v in default "$@"; do
  if [ $v = default ]; then L=""; else L=flybody_amd/csrc/variants/lib$v.so; fi
  a=$(FLYBODY_ENV_LIB=$L timeout -k 10 120 python bench.py --no-cpu-baseline --no-async-groups --steps 300 --warmup 50 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel_ms'])")
  b=$(FLYBODY_ENV_LIB=$L timeout -k 10 200 python bench.py --no-cpu-baseline --no-async-groups --workload walk_on_ball 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel_ms'])")
  echo "$v flight $a ball $b" | tee -a $O/ab.log
done; done
